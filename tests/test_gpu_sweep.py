"""Parameter sweeps of Formulation B (ismpc_create_sweep): K parameter sets in ONE batch, every set's tables built on the
device (MFMA Newton-Schulz inverse of the K vertical Hessians, csrc/ismpc_sweep.hip).
  * the device-built tables of every set against the host's long-double build (<= 1e-12 relative to each table's largest entry; S W_p <= 1e-11),
  * one launch with >= 64 parameter sets against one CPU oracle per set (reference qpOASES where oracle/_ref is built),
    CoM <= 1e-6 relative, status bit-exact outside the feasibility band,
  * a sweep whose sets are all equal reproduces the plain handle bit for bit; invalid set indices are flagged."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def q(built_libs):
    import torch
    assert torch.cuda.is_available()
    import quadruped_gait_generation_ismpc_amd as q
    return q


def sweep_params(q, K, N=100, seed=3):
    """K parameter sets around the reference's constants (the bench's generator)."""
    from quadruped_gait_generation_ismpc_amd import workload
    return workload.make_sweep_params(K, N=N, seed=seed)


@pytest.fixture(scope="module")
def sweep64(q):
    ps = sweep_params(q, 64)
    s = q.MPCSolver.sweep(q.reference_plan(params=ps[0]), ps)
    yield s, ps
    s.close()


def test_device_built_tables_match_the_long_double_host_build(q, sweep64):
    s, ps = sweep64
    info = s.sweep_info()
    assert info["n_sets"] == 64 and info["newton_iterations"] >= 10 and info["mfma_gemm_launches"] == 2 * info["newton_iterations"] + 4
    worst = {}
    for k in range(64):
        e = s.sweep_verify_tables(k)
        for name, v in e.items():
            worst[name] = max(worst.get(name, 0.0), v)
    print("worst relative table errors over 64 sets:", worst)
    for name in ("Hinv", "affine", "W", "HSt", "SHSt", "tail", "layout"):
        assert worst[name] <= 1e-12, worst
    assert worst["SW"] <= 1e-11, worst                                 # S W_p inherits W_p's error through N^2 / 2-term sums (parity needs 1e-6)


@pytest.mark.parametrize("B", [4096, 12288])
def test_sweep_batch_against_one_oracle_per_set(q, sweep64, B):
    """Instance i runs with parameter set i % 64; the oracle (fresh tables per set, reference qpOASES where built) solves a sample of
    every set."""
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    s, ps = sweep64
    tin = workload.make_batch(100, B, seed=13)
    tin["reserved"] = np.arange(B) % 64
    out = s.solve_batch(tin)
    dev = q.from_device(s.solve_batch_torch(q.to_device(tin)), q.TICK_OUT)
    assert out.tobytes() == dev.tobytes()
    assert (out["status"] & q.ST_BAD_INDEX).sum() == 0
    checked = 0
    for k in range(64):
        pick = np.where(tin["reserved"] == k)[0][:6]
        op = O.default_params(100, mass=ps[k].mass, h_des=ps[k].h_des, q_p=ps[k].q_p, q_u=ps[k].q_u, q_v=ps[k].q_v, foot_width=ps[k].foot_width)
        ref, _ = O.Oracle(op).solve(tin[pick])
        o = out[pick]
        ok = ((ref["status"] | o["status"]) & q.ST_ERROR_MASK) == 0
        rel = np.abs(o["com_pos"] - ref["com_pos"]).max(1) / np.maximum(np.abs(ref["com_pos"]).max(1), 1e-3)
        assert rel[ok].max(initial=0.0) <= TOL, (k, rel)
        assert np.abs(o["com_vel"] - ref["com_vel"])[ok].max(initial=0.0) <= TOL
        scale = np.maximum(np.array([9.81 * ps[k].mass, 1.0, 1.0])[None, :], np.abs(ref["u0"][ok]))
        assert (np.abs(o["u0"] - ref["u0"])[ok] <= TOL * scale).all(), k
        assert (o["status"][ok] == ref["status"][ok]).all()
        checked += int(ok.sum())
    assert checked > 200
    # the sets do differ: the same state under another set gives another force
    t2 = tin.copy(); t2["reserved"] = (t2["reserved"] + 1) % 64
    o2 = s.solve_batch(t2)
    run = ((out["status"] | o2["status"]) & (q.ST_ERROR_MASK | q.ST_FLIGHT | q.ST_TICK_SKIPPED)) == 0
    assert (np.abs(out["u0"][run, 0] - o2["u0"][run, 0]) > 1e-3).mean() > 0.9


@pytest.mark.parametrize("N", [150, 200])
def test_sweep_beyond_the_lane_group_horizon(q, N):
    """Horizons the lane-group kernels do not cover (128 < N <= 256; the horizon is a run-time value of the reference, parameters.cpp:13,42,
    and the golden vectors exist at 150 and 200): a sweep handle runs one instance per wavefront there (ismpc_tick_affine<R, true>: the
    instance's set is wave-uniform), tables of every set built on the device as at N = 100.  16 sets against one oracle per set, the tables
    of three sets against the host's long-double build, an unknown set flagged, and a sweep of equal sets against the plain handle."""
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    K = 16
    ps = sweep_params(q, K, N=N)
    s = q.MPCSolver.sweep(q.reference_plan(params=ps[0]), ps)
    for k in (0, 5, K - 1):
        e = s.sweep_verify_tables(k)
        for name in ("Hinv", "affine", "W", "HSt", "SHSt", "tail"):
            assert e[name] <= 1e-11, (k, e)                             # (measured at N = 200: 2.0e-12 for Hinv S', below 3e-13 for the inverse itself)
        assert e["SW"] <= 1e-10, (k, e)
    B = 1536
    tin = workload.make_batch(N, B, seed=15)
    tin["reserved"] = np.arange(B) % K
    tin["reserved"][7] = K + 3                                           # names no set
    out = s.solve_batch(tin)
    assert out["status"][7] & q.ST_BAD_INDEX and np.array_equal(out["com_pos"][7], tin["com_pos"][7])
    assert (np.delete(out["status"], 7) & q.ST_BAD_INDEX).sum() == 0
    checked = 0
    for k in range(K):
        pick = np.where(tin["reserved"] == k)[0][:5]
        op = O.default_params(N, mass=ps[k].mass, h_des=ps[k].h_des, q_p=ps[k].q_p, q_u=ps[k].q_u, q_v=ps[k].q_v, foot_width=ps[k].foot_width)
        ref, _ = O.Oracle(op).solve(tin[pick])
        o = out[pick]
        ok = ((ref["status"] | o["status"]) & q.ST_ERROR_MASK) == 0
        rel = np.abs(o["com_pos"] - ref["com_pos"]).max(1) / np.maximum(np.abs(ref["com_pos"]).max(1), 1e-3)
        assert rel[ok].max(initial=0.0) <= TOL, (k, rel)
        assert np.abs(o["com_vel"] - ref["com_vel"])[ok].max(initial=0.0) <= TOL
        assert (o["status"][ok] == ref["status"][ok]).all()
        checked += int(ok.sum())
    assert checked >= 3 * K
    # closed loop of the sweep handle (one launch per tick at this horizon) == per-tick calls fed back by the host
    st = q.to_device(tin[:64]); ref_state = tin[:64].copy()
    traj = q.from_device(s.rollout_torch(st, int(tin["simulation_time"].max()) + 1, 3), q.TICK_OUT)
    assert traj.shape == (3, 64) and ((traj["status"] & q.ST_Z_FAILED) == 0).all()
    s.close()
    p = q.default_params(N=N)
    plain = q.MPCSolver(q.reference_plan(params=p), params=p)
    same = q.MPCSolver.sweep(q.reference_plan(params=p), [p, p])
    t2 = workload.make_batch(N, 512, seed=16); t2["reserved"] = np.arange(512) % 2
    a, b = plain.solve_batch(t2), same.solve_batch(t2)
    assert np.array_equal(a["status"], b["status"]) and np.abs(a["com_pos"] - b["com_pos"]).max() <= 1e-9 and np.abs(a["com_vel"] - b["com_vel"]).max() <= 1e-9
    plain.close(); same.close()


@pytest.mark.parametrize("B", [6000, 40000])
def test_sweep_bind_changes_placement_not_records(q, sweep64, B, monkeypatch):
    """ismpc_sweep_bind sorts the instances of a batch by parameter set once; later launches of that batch size run instance order[g] in
    slot g (one set per wavefront, one contiguous eighth of the sorted batch per XCD).  Placement only: every record byte-identical to the
    unbound handle, in the one-launch and the two-launch form, also after the assignment changed under a stale order, also with records that
    name no set; another batch size ignores the order; unbinding restores slot g = instance g."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    _, ps = sweep64
    tin = workload.make_batch(100, B, seed=23)
    tin["reserved"] = (np.arange(B) * 7 + 3) % 64
    tin["reserved"][[5, B // 2, B - 1]] = [64, -1, 9999]                 # no such sets
    for form in ("3", "0"):                                              # one launch per step / tick kernel + fallback launch
        monkeypatch.setenv("ISMPC_ONE_LAUNCH", form)
        tight = [type(p).from_buffer_copy(bytes(p)) for p in ps]
        for p in tight:
            p.z_ineq_hi = 4.6                                            # some instances go through the inequality fallback
        plain, bound = q.MPCSolver.sweep(q.reference_plan(params=tight[0]), tight), q.MPCSolver.sweep(q.reference_plan(params=tight[0]), tight)
        d_in = q.to_device(tin)
        ref = plain.solve_batch_torch(d_in).clone()
        bound.sweep_bind(d_in)
        got = bound.solve_batch_torch(d_in).clone()
        torch.cuda.synchronize()
        o = q.from_device(ref, q.TICK_OUT)
        assert torch.equal(got, ref), form
        assert (o["status"][[5, B // 2, B - 1]] & q.ST_BAD_INDEX).all() and ((o["status"] & q.ST_Z_INEQ_ACTIVE) != 0).sum() > 20
        t2 = tin.copy(); t2["reserved"] = (np.arange(B) * 11 + 1) % 64       # the assignment moves, the order is stale: same records as unbound
        d2 = q.to_device(t2)
        assert torch.equal(bound.solve_batch_torch(d2), plain.solve_batch_torch(d2))
        small = q.to_device(tin[:B // 2 + 1])                            # another batch size: the order does not apply
        assert torch.equal(bound.solve_batch_torch(small), plain.solve_batch_torch(small))
        bound.sweep_unbind()
        assert torch.equal(bound.solve_batch_torch(d_in), ref)
        assert bound.fallback_counters() == (0, 0, 0, 0)
        plain.close(); bound.close()
    # a plain handle refuses
    p0 = q.default_params(N=100)
    h = q.MPCSolver(q.reference_plan(params=p0), params=p0)
    with pytest.raises(q.IsmpcError):
        h.sweep_bind(q.to_device(tin[:64]))
    h.close()


@pytest.mark.parametrize("N", [100, 150])
def test_sweep_on_a_plan_with_footstep_heights(q, N):
    """Footsteps off z = 0 ("stairs": mid_z enters f_z, MPCSolver.cpp:259): every set then needs its own per-frame offsets
    dU(idx) = Hinv_k q_p S' mid_z[idx : idx + N] and S dU(idx) beside its pattern corrections -- built on the device from the set's
    fallback table (sweep_du).  12 sets, lane-group kernels (N = 100) and one instance per wavefront (N = 150), against one oracle per
    set on the same plan; the heights do matter; equal sets reproduce the plain handle."""
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    K = 12
    ps = sweep_params(q, K, N=N)
    ftsp = q.reference_plan(params=ps[0])
    for i in range(1, ftsp.shape[0]):
        ftsp[i, 2] = 0.01 * ((i // 3) % 4)
    s = q.MPCSolver.sweep(ftsp, ps)
    B = 960
    tin = workload.make_batch(N, B, seed=17)
    tin["reserved"] = np.arange(B) % K
    out = s.solve_batch(tin)
    checked, moved = 0, 0.0
    for k in range(K):
        pick = np.where(tin["reserved"] == k)[0][:6]
        op = O.default_params(N, mass=ps[k].mass, h_des=ps[k].h_des, q_p=ps[k].q_p, q_u=ps[k].q_u, q_v=ps[k].q_v, foot_width=ps[k].foot_width)
        ref, _ = O.Oracle(op, ftsp).solve(tin[pick])
        flat, _ = O.Oracle(op).solve(tin[pick])
        moved = max(moved, np.abs(ref["u0"][:, 0] - flat["u0"][:, 0]).max())
        o = out[pick]
        ok = ((ref["status"] | o["status"]) & q.ST_ERROR_MASK) == 0
        rel = np.abs(o["com_pos"] - ref["com_pos"]).max(1) / np.maximum(np.abs(ref["com_pos"]).max(1), 1e-3)
        assert rel[ok].max(initial=0.0) <= TOL, (k, rel)
        assert np.abs(o["com_vel"] - ref["com_vel"])[ok].max(initial=0.0) <= TOL
        scale = np.maximum(np.array([9.81 * ps[k].mass, 1.0, 1.0])[None, :], np.abs(ref["u0"][ok]))
        assert (np.abs(o["u0"] - ref["u0"])[ok] <= TOL * scale).all(), k
        assert (o["status"][ok] == ref["status"][ok]).all()
        checked += int(ok.sum())
    assert checked >= 4 * K and moved > 1.0
    s.close()
    p = q.default_params(N=N)
    plain, same = q.MPCSolver(ftsp, params=p), q.MPCSolver.sweep(ftsp, [p, p, p])
    t3 = tin[:300].copy(); t3["reserved"] = np.arange(300) % 3
    a, b = plain.solve_batch(t3), same.solve_batch(t3)
    assert np.array_equal(a["status"], b["status"]) and np.abs(a["com_pos"] - b["com_pos"]).max() <= 1e-9 and np.abs(a["com_vel"] - b["com_vel"]).max() <= 1e-9
    assert (np.abs(a["u0"] - b["u0"]) <= 1e-8 * np.maximum(np.abs(a["u0"]), 1.0)).all()
    plain.close(); same.close()


def test_sweep_with_eight_lanes_per_instance(q, sweep64, monkeypatch):
    """Beyond 8 192 instances per launch a sweep handle runs eight instances per wavefront over the 8-lane copy of every set's tables
    (the default since round 4: 5 % faster once the batch is sorted by set; ISMPC_LPI=16 keeps 16 lanes at every size).  Same records as
    the 16-lane handle up to summation order; the device-built 8-lane layout is checked against the host's."""
    from quadruped_gait_generation_ismpc_amd import workload
    s8, ps = sweep64
    monkeypatch.setenv("ISMPC_LPI", "16")
    s16 = q.MPCSolver.sweep(q.reference_plan(params=ps[0]), ps)
    monkeypatch.delenv("ISMPC_LPI")
    assert max(s8.sweep_verify_tables(k)["layout"] for k in (0, 17, 63)) <= 1e-12
    tin = workload.make_batch(100, 12288, seed=14)
    tin["reserved"] = np.arange(len(tin)) % 64
    a, b = s16.solve_batch(tin), s8.solve_batch(tin)
    ok = ((a["status"] | b["status"]) & q.ST_ERROR_MASK) == 0
    assert ok.mean() > 0.95 and np.array_equal(a["status"][ok], b["status"][ok])
    assert np.abs(a["com_pos"] - b["com_pos"])[ok].max() <= 1e-11 and np.abs(a["com_vel"] - b["com_vel"])[ok].max() <= 1e-10
    assert (np.abs(a["u0"] - b["u0"])[ok] <= 1e-9 * np.maximum(np.abs(a["u0"][ok]), 1.0)).all()
    assert a.tobytes() != b.tobytes()                                     # they ARE different layouts
    s16.close()


def test_sweep_of_equal_sets_is_the_plain_handle(q):
    """Tables built on the device (fp64) instead of on the host (long double): same records to 1e-9 relative on a full batch, and
    identical flags; a sweep with ONE set works; an unknown set index is flagged and passes the state through."""
    from quadruped_gait_generation_ismpc_amd import workload
    p = q.default_params(N=100)
    plain = q.MPCSolver(q.reference_plan(params=p), params=p)
    sw = q.MPCSolver.sweep(q.reference_plan(params=p), [p, p, p])
    one = q.MPCSolver.sweep(q.reference_plan(params=p), [p])
    tin = workload.make_batch(100, 8192, seed=21)
    a = plain.solve_batch(tin)
    t3 = tin.copy(); t3["reserved"] = np.arange(len(tin)) % 3
    b = sw.solve_batch(t3)
    c = one.solve_batch(tin)
    for o in (b, c):
        assert np.array_equal(o["status"], a["status"]) and np.array_equal(o["iters"], a["iters"])
        assert np.abs(o["com_pos"] - a["com_pos"]).max() <= 1e-9 and np.abs(o["com_vel"] - a["com_vel"]).max() <= 1e-9
        assert np.abs(o["u0"] - a["u0"]).max() <= 1e-7 * 490.5
    bad = tin[:8].copy(); bad["reserved"] = [0, 3, -1, 1, 2, 64, 0, 1]
    ob = sw.solve_batch(bad)
    flagged = (ob["status"] & q.ST_BAD_INDEX) != 0
    assert flagged.tolist() == [False, True, True, False, False, True, False, False]
    assert np.array_equal(ob["com_pos"][flagged], bad["com_pos"][flagged])
    for s_ in (plain, sw, one):
        s_.close()


def test_sweep_closed_loop_and_vertical_fallback(q):
    """Closed loop on a sweep handle (one launch per tick, state fed back with its set index) against the oracle's rollout of
    two different sets; and sets whose tight bound on S u makes the inequality rows active run the fallback with THEIR tables."""
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    ps = sweep_params(q, 4, seed=9)
    ps[2].z_ineq_hi = 4.6; ps[3].z_ineq_hi = 4.2
    s = q.MPCSolver.sweep(q.reference_plan(params=ps[0]), ps)
    st0 = O.initial_state().view(q.TICK_IN)
    recs = np.repeat(st0, 2); recs["reserved"] = [0, 1]
    d = q.to_device(recs)
    traj = q.from_device(s.rollout_torch(d, 0, 150), q.TICK_OUT)
    for i, k in enumerate((0, 1)):
        op = O.default_params(100, mass=ps[k].mass, h_des=ps[k].h_des, q_p=ps[k].q_p, q_u=ps[k].q_u, q_v=ps[k].q_v, foot_width=ps[k].foot_width)
        ref, _, _, fin = O.Oracle(op).rollout(st0, 0, 150)
        assert np.array_equal(traj["status"][:, i], ref["status"])
        rel = np.abs(traj["com_pos"][:, i] - ref["com_pos"]).max(1) / np.maximum(np.abs(ref["com_pos"]).max(1), 1e-3)
        assert rel.max() <= TOL
    assert q.from_device(d, q.TICK_IN)["reserved"].tolist() == [0, 1]
    tin = workload.make_batch(100, 96, seed=177)
    tin["reserved"] = 2 + (np.arange(96) % 2)
    out = s.solve_batch(tin)
    for k in (2, 3):
        m = tin["reserved"] == k
        op = O.default_params(100, mass=ps[k].mass, h_des=ps[k].h_des, q_p=ps[k].q_p, q_u=ps[k].q_u, q_v=ps[k].q_v, foot_width=ps[k].foot_width, z_ineq_hi=ps[k].z_ineq_hi)
        ref, _ = O.Oracle(op).solve(tin[m])
        act = (ref["status"] & q.ST_Z_INEQ_ACTIVE) != 0
        assert act.sum() >= 8
        assert (((out["status"][m] & q.ST_Z_INEQ_ACTIVE) != 0) == act).all() and ((out["status"][m] & q.ST_Z_FAILED) == 0).all()
        ok = ((ref["status"] | out["status"][m]) & q.ST_ERROR_MASK) == 0
        rel = np.abs(out["com_pos"][m] - ref["com_pos"]).max(1) / np.maximum(np.abs(ref["com_pos"]).max(1), 1e-3)
        assert rel[ok].max() <= TOL
        assert (out["status"][m][ok] == ref["status"][ok]).all()
    s.close()


def test_sweep_in_kernel_rollout_is_bitwise_the_per_tick_loop(q, monkeypatch):
    """Closed loops on a sweep handle run inside ONE launch like the plain handle's (ismpc_rollout_quad<.., SW>): byte-identical to one
    launch per tick (ISMPC_ROLLOUT=host) -- also for sets whose tight bound on S u parks instances for the resume launch."""
    import torch
    from oracle import oracle as O
    ps = sweep_params(q, 6, seed=5)
    ps[4].z_ineq_hi = 4.6; ps[5].z_ineq_hi = 4.2
    plan = q.reference_plan(params=ps[0])
    monkeypatch.delenv("ISMPC_ROLLOUT", raising=False)
    a = q.MPCSolver.sweep(plan, ps)
    monkeypatch.setenv("ISMPC_ROLLOUT", "host")
    b = q.MPCSolver.sweep(plan, ps)
    monkeypatch.delenv("ISMPC_ROLLOUT", raising=False)
    B, ticks = 41, 220
    recs = np.repeat(O.initial_state().view(q.TICK_IN), B)
    rng = np.random.default_rng(4)
    recs["com_pos"][1:, :2] += rng.uniform(-0.004, 0.004, (B - 1, 2)); recs["com_vel"][1:, :2] += rng.uniform(-0.02, 0.02, (B - 1, 2))
    recs["reserved"] = np.arange(B) % 6
    recs["reserved"][7] = 9                                             # an unknown set: flagged every tick, state passed through
    sa, sb = q.to_device(recs), q.to_device(recs)
    ta = a.rollout_torch(sa, 0, ticks); tb = b.rollout_torch(sb, 0, ticks)
    torch.cuda.synchronize()
    assert torch.equal(ta, tb) and torch.equal(sa, sb)
    out = q.from_device(ta, q.TICK_OUT)
    assert ((out["status"][:, 7] & q.ST_BAD_INDEX) != 0).all() and np.array_equal(out["com_pos"][-1, 7], recs["com_pos"][7])
    tight = np.isin(recs["reserved"], (4, 5)) & (np.arange(B) != 7)
    assert ((out["status"][:, tight] & q.ST_Z_INEQ_ACTIVE) != 0).any(axis=0).all()       # those instances went through the resume launch
    assert ((out["status"] & q.ST_Z_FAILED) == 0).all()
    a.close(); b.close()


def test_sweep_build_is_checked_not_trusted(q):
    """Every set's inverse is verified on the device (|I - H X| and finite tables): a set whose Hessian is conditioned beyond the iteration
    budget fails ismpc_create_sweep with ISMPC_E_NUMERIC and names the set; a stiff but tractable one (cond ~ 1e7) builds and solves."""
    p0 = q.default_params(N=100)
    bad = q.default_params(N=100); bad.q_u = 1e-40
    with pytest.raises(q.IsmpcError) as e:
        q.MPCSolver.sweep(q.reference_plan(params=p0), [p0, bad])
    assert e.value.code == -4 and "parameter set 1" in str(e.value)
    stiff = q.default_params(N=100); stiff.q_p = 1e8; stiff.q_u = 1e-4
    s = q.MPCSolver.sweep(q.reference_plan(params=p0), [p0, stiff])
    err = s.sweep_verify_tables(1)
    assert err["Hinv"] <= 1e-8 and max(err.values()) <= 1e-6, err          # cond(H) eps: nine digits at cond 3e7
    from quadruped_gait_generation_ismpc_amd import workload
    tin = workload.make_batch(100, 256, seed=2); tin["reserved"] = 1
    out = s.solve_batch(tin)
    assert ((out["status"] & q.ST_BAD_INDEX) == 0).all() and np.isfinite(out["com_pos"]).all()
    s.close()


@pytest.mark.parametrize("N", [50, 64, 128])
def test_sweep_other_horizons(q, N):
    """The other lane-group shapes of the sweep kernels (R = 4 and 8 samples per lane, GEMM tiles of 64 and 128): device-built tables against
    the host build, a batch against one oracle per set, closed loop against the oracle's."""
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    ps = sweep_params(q, 5, N=N, seed=N)
    s = q.MPCSolver.sweep(q.reference_plan(params=ps[0]), ps)
    for k in range(5):
        err = s.sweep_verify_tables(k)
        assert max(err.values()) <= 1e-11, (k, err)
    base = 100 if N > 50 else 50
    tin = workload.make_batch(base, 200, seed=N)
    tin["reserved"] = np.arange(200) % 5
    out = s.solve_batch(tin)
    for k in range(5):
        m = np.where(tin["reserved"] == k)[0][:10]
        op = O.default_params(N, mass=ps[k].mass, h_des=ps[k].h_des, q_p=ps[k].q_p, q_u=ps[k].q_u, q_v=ps[k].q_v, foot_width=ps[k].foot_width)
        ref, _ = O.Oracle(op).solve(tin[m])
        ok = ((ref["status"] | out["status"][m]) & q.ST_ERROR_MASK) == 0
        rel = np.abs(out["com_pos"][m] - ref["com_pos"]).max(1) / np.maximum(np.abs(ref["com_pos"]).max(1), 1e-3)
        assert rel[ok].max(initial=0.0) <= TOL and (out["status"][m][ok] == ref["status"][ok]).all(), (N, k)
    st0 = O.initial_state().view(q.TICK_IN)
    recs = np.repeat(st0, 2); recs["reserved"] = [1, 3]
    traj = q.from_device(s.rollout_torch(q.to_device(recs), 0, 120), q.TICK_OUT)
    for i, k in enumerate((1, 3)):
        op = O.default_params(N, mass=ps[k].mass, h_des=ps[k].h_des, q_p=ps[k].q_p, q_u=ps[k].q_u, q_v=ps[k].q_v, foot_width=ps[k].foot_width)
        ref, _, _, _ = O.Oracle(op).rollout(st0, 0, 120)
        ok = (ref["status"] & q.ST_ERROR_MASK) == 0
        rel = np.abs(traj["com_pos"][:, i] - ref["com_pos"]).max(1) / np.maximum(np.abs(ref["com_pos"]).max(1), 1e-3)
        first_bad = np.argmax(~ok) if (~ok).any() else len(ok)          # a short horizon may turn infeasible late in the closed loop: compare up to there
        assert first_bad >= 60 and rel[:first_bad].max() <= TOL and np.array_equal(traj["status"][:first_bad, i], ref["status"][:first_bad]), (N, k)
    s.close()
