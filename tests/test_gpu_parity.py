"""GPU parity tests proper: the HIP path, called through the C ABI, against (i) the committed golden
vectors produced with the reference's qpOASES, (ii) the CPU oracle on the same seeded inputs, and
(iii) size-independent properties at BASELINE.json's full batch sizes.

Tolerances (fp64): relative CoM error <= 1e-6 as BASELINE.json's north_star states
(||dCoM||inf / max(||CoM||inf, 1e-3)); integer / index quantities bit exact.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HORIZONS = (50, 100, 150, 200)
TOL = 1e-6


@pytest.fixture(scope="module")
def q(built_libs):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import quadruped_gait_generation_ismpc_amd as q
    return q


@pytest.fixture(scope="module")
def O():
    from oracle import oracle as O
    return O


_solvers = {}
# affine tables with 16 lanes per instance (four instances per wavefront; default) / 8 lanes per instance (eight) / 32 (two) / one
# instance per wavefront; per-tick MFMA formulation
PATHS = ("affine", "lpi8", "lpi32", "auto", "wave", "dense")
_ENV = {"affine": {"ISMPC_PATH": "affine", "ISMPC_LPI": "16"}, "lpi8": {"ISMPC_PATH": "affine", "ISMPC_LPI": "8"},
        "lpi32": {"ISMPC_PATH": "affine", "ISMPC_LPI": "32"}, "auto": {"ISMPC_PATH": "affine"},       # auto: 32 up to 2 048 instances, 16 above
        "hostloop32": {"ISMPC_PATH": "affine", "ISMPC_LPI": "32", "ISMPC_ROLLOUT": "host"},
        "wave": {"ISMPC_PATH": "wave"}, "dense": {"ISMPC_PATH": "dense"},
        "hostloop": {"ISMPC_PATH": "affine", "ISMPC_LPI": "16", "ISMPC_ROLLOUT": "host"},
        "hostloop8": {"ISMPC_PATH": "affine", "ISMPC_LPI": "8", "ISMPC_ROLLOUT": "host"}}


def solver_for(q, N, path="affine", plan=None, **over):
    key = (N, path, plan, tuple(sorted(over.items())))
    if key not in _solvers:
        p = q.default_params(N=N, **over)
        ftsp = q.reference_plan(params=p)
        if plan == "stairs":
            for i in range(1, ftsp.shape[0]):
                ftsp[i, 2] = 0.01 * ((i // 3) % 4)
        saved = {k: os.environ.get(k) for k in ("ISMPC_PATH", "ISMPC_LPI", "ISMPC_ROLLOUT")}
        for k in saved:
            os.environ.pop(k, None)
        os.environ.update(_ENV[path])
        try:
            _solvers[key] = q.MPCSolver(ftsp, params=p)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
    return _solvers[key]


def rel_com(a, b):
    return np.abs(a["com_pos"] - b["com_pos"]).max(1) / np.maximum(np.abs(b["com_pos"]).max(1), 1e-3)


def assert_parity(q, out, ref, ok=None, z_fallback=True):
    """z_fallback=False (dense A/B path): instances whose vertical inequality rows are active are only flagged there."""
    if ok is None:
        ok = ((ref["status"] | out["status"]) & q.ST_ERROR_MASK) == 0
    if not z_fallback:
        ok = ok & ((ref["status"] & q.ST_Z_INEQ_ACTIVE) == 0)
    assert ok.any()
    assert rel_com(out, ref)[ok].max() <= TOL
    assert np.abs(out["com_vel"] - ref["com_vel"])[ok].max() <= TOL
    # u0 = (vertical force [N], ZMP x [m], ZMP y [m]): the force is compared on its natural scale m g = 490 N
    # (qpOASES stops at 2.2e-7 relative homotopy length: a force pinned at 0 by a bound comes back as +-1e-6 N)
    scale = np.maximum(np.array([490.5, 1.0, 1.0])[None, :], np.abs(ref["u0"][ok]))
    assert (np.abs(out["u0"] - ref["u0"])[ok] <= TOL * scale).all()
    assert (out["status"][ok] == ref["status"][ok]).all()


def _on_feasibility_boundary(q, N, rec, band):
    """Is one of the two horizontal QPs of this instance within `band` (relative) of infeasibility?  Feasible <=>
    |beq - a'mid| <= h sum|a| (a box around mid and one equality row); evaluated from the decision trajectories of the
    HIP path run with the ZMP box widened by (1 +- band)."""
    res = []
    for scale in (1.0 - band, 1.0 + band):
        s = solver_for(q, N, "affine", foot_width=0.09 * scale, first_step_halfwidth=1.0 * scale)
        o = s.solve_batch(np.array([rec], dtype=q.TICK_IN))
        res.append(int(o["status"][0]) & (q.ST_X_INFEASIBLE | q.ST_Y_INFEASIBLE))
    return res[0] != res[1]


def load_golden(q, name):
    z = np.load(os.path.join(GOLDEN, name))
    return z, z["tick_in"].view(q.TICK_IN).reshape(-1), z["tick_out"].view(q.TICK_OUT).reshape(-1)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("N", HORIZONS)
def test_golden_vectors(q, N, path):
    """Committed inputs + reference-qpOASES outputs (tests/golden/make_golden.py)."""
    z, tin, ref = load_golden(q, f"formB_vectors_N{N}.npz")
    out = solver_for(q, N, path).solve_batch(tin)
    assert (out["status"] == ref["status"]).all()          # incl. which instances are infeasible / in flight
    assert_parity(q, out, ref)


def test_config1_single_tick_through_reference_call_shape(q):
    """BASELINE config 1: N = 50, one tick, through MPCSolver(ftsp).solve(State, WalkState, ftsp)."""
    z, tin, ref = load_golden(q, "formB_kat_config1.npz")
    p = q.default_params(N=50)
    plan = q.reference_plan(params=p)
    solver = q.MPCSolver(plan, params=p)
    for k in range(2):
        cur = q.State(comPos=tin["com_pos"][k].copy(), comVel=tin["com_vel"][k].copy())
        ws = q.WalkState(simulationTime=float(tin["simulation_time"][k]), mpcIter=int(tin["mpc_iter"][k]),
                         controlIter=int(tin["control_iter"][k]), footstepCounter=int(tin["footstep_counter"][k]))
        nxt = solver.solve(cur, ws, plan)
        assert solver.itr == ws.mpcIter and solver.fsCount == ws.footstepCounter      # MPCSolver.cpp:206-207
        err = np.abs(nxt.comPos - ref["com_pos"][k]).max() / max(np.abs(ref["com_pos"][k]).max(), 1e-3)
        assert err <= TOL and np.abs(nxt.comVel - ref["com_vel"][k]).max() <= TOL


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("N,scale", [(100, 1.0), (100, 2.0), (200, 1.0), (64, 1.0), (37, 1.0), (129, 0.5), (256, 0.5)])
def test_against_oracle_seeded(q, O, N, scale, path):
    """Fresh seeded batch (SURVEY.md 8d generator around the N=100/200 nominal gait), oracle run here."""
    from quadruped_gait_generation_ismpc_amd import workload
    base = 200 if N > 150 else (100 if N > 50 else 50)
    tin = workload.make_batch(base, 96, scale=scale, seed=7 + N)
    if N == 256:
        tin = tin[tin["simulation_time"] < 1250]
    orc = O.Oracle(O.default_params(N))
    ref, info = orc.solve(tin)
    out = solver_for(q, N, path).solve_batch(tin)
    ok = (ref["status"] & q.ST_ERROR_MASK) == 0
    # a QP within 1e-9 of the feasibility boundary may be classified either way: exclude from status equality
    assert ((out["status"] != ref["status"]) & ok).sum() == 0
    # a horizontal QP within 1e-9 (relative) of the feasibility boundary may be classified either way by qpOASES' own
    # termination test (homotopy length 2.2e-7, Options.cpp:206); everything outside that band must agree
    diff = (out["status"] & q.ST_ERROR_MASK) != (ref["status"] & q.ST_ERROR_MASK)
    for b in np.where(diff)[0]:
        assert _on_feasibility_boundary(q, N, tin[b], band=1e-9), (b, out["status"][b], ref["status"][b])
    assert_parity(q, out, ref, z_fallback=(path != "dense"))


@pytest.mark.parametrize("lay", ["affine", "lpi8", "lpi32"])
@pytest.mark.parametrize("N,over,dz", [(100, dict(z_ineq_hi=4.6), 0.0), (100, dict(z_ineq_hi=4.2), 0.0), (100, dict(z_ineq_hi=3.2), 0.0),
                                        (50, dict(), 0.12), (37, dict(), 0.10), (150, dict(z_ineq_hi=10.5), 0.0), (100, dict(), 0.25)])
def test_vertical_inequality_rows_active(q, O, N, over, dz, lay):
    """0 <= S_bar_z u <= z_hi (MPCSolver.cpp:158-160) made active -- by a tight upper bound (rows at the end of the
    horizon) or by a CoM far above h_des (negative first forces hit the lower bound): the second launch
    (ismpc_tick_affine_fallback) solves the inequality-constrained vertical QP; parity with the oracle's full QP."""
    from quadruped_gait_generation_ismpc_amd import workload
    base = 100 if N >= 100 else 50
    tin = workload.make_batch(base, 48, seed=77 + N)
    if N == 150:
        tin = workload.make_batch(150, 48, seed=77 + N)
    tin["com_pos"][:, 2] += dz
    tin["com_vel"][:, 2] += 0.2 * np.sign(dz)
    s = solver_for(q, N, lay, **over)
    orc = O.Oracle(O.default_params(N, **over))
    ref, info = orc.solve(tin)
    out = s.solve_batch(tin)
    act = (ref["status"] & q.ST_Z_INEQ_ACTIVE) != 0
    assert act.sum() >= 8, act.sum()                           # the case does exercise the fallback
    assert ((out["status"] & q.ST_Z_FAILED) == 0).all()      # the working set may hold every row of the horizon: nothing is left unsolved
    assert (((out["status"] & q.ST_Z_INEQ_ACTIVE) != 0) == act).all()
    ok = ((ref["status"] | out["status"]) & q.ST_ERROR_MASK) == 0
    assert (out["status"][ok] == ref["status"][ok]).all()
    assert_parity(q, out, ref, ok)
    assert ((out["iters"][act & ok] >> 16) & 255).min() >= 1   # fallback iterations are reported


@pytest.mark.parametrize("N,over,dz", [(100, dict(z_ineq_hi=3.2), 0.0), (50, dict(), 0.12), (100, dict(), 0.25)])
def test_fallback_working_set_moves_to_the_pool(q, N, over, dz, monkeypatch):
    """The fallback keeps its working set (G^-1, multipliers) in the wavefront's LDS window and moves it to a pool slot in HBM when it
    outgrows the window (16 entries).  With the window cut to 2 entries (ISMPC_Z_LDS_Q) every solve with three active rows goes
    through the move: where the working set lives does not change one bit of the result."""
    from quadruped_gait_generation_ismpc_amd import workload
    tin = workload.make_batch(150 if N == 150 else (100 if N >= 100 else 50), 96, seed=77 + N)
    tin["com_pos"][:, 2] += dz
    tin["com_vel"][:, 2] += 0.2 * np.sign(dz)
    p = q.default_params(N=N, **over)
    ftsp = q.reference_plan(params=p)
    ref = q.MPCSolver(ftsp, params=p).solve_batch(tin)
    monkeypatch.setenv("ISMPC_Z_LDS_Q", "2")
    out = q.MPCSolver(ftsp, params=p).solve_batch(tin)
    zits = (ref["iters"] >> 16) & 255
    assert (zits >= 3).sum() >= 8, zits.max()                   # working sets beyond two entries exist in this batch
    assert ((out["status"] & q.ST_Z_FAILED) == 0).all()
    assert out.tobytes() == ref.tobytes()


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("N", [50, 100, 150])
def test_non_flat_plan(q, O, N, path):
    """Footsteps at varying heights: mid_z enters f_z (MPCSolver.cpp:259) -- the dU / W tables of the fast
    path, the suffix sums of the dense one."""
    from quadruped_gait_generation_ismpc_amd import workload
    tin = workload.make_batch(N, 64, seed=21)
    s = solver_for(q, N, path, plan="stairs")
    orc = O.Oracle(O.default_params(N), s.ftsp)
    ref, info = orc.solve(tin)
    flat, _ = O.Oracle(O.default_params(N)).solve(tin)
    assert np.abs(ref["u0"][:, 0] - flat["u0"][:, 0]).max() > 1.0       # the heights do matter
    out = s.solve_batch(tin)
    assert_parity(q, out, ref)


@pytest.mark.parametrize("path", PATHS)
def test_decision_trajectories(q, O, path):
    """Full decisionVariables_z/_x/_y (MPCSolver.cpp:269,395,396) via the device-pointer entry point."""
    import torch
    N = 100
    z, tin, ref = load_golden(q, f"formB_vectors_N{N}.npz")
    s = solver_for(q, N, path)
    d_in = q.to_device(tin)
    traj = torch.zeros((len(tin), 3, N), dtype=torch.float64, device="cuda:0")
    d_out = s.solve_batch_torch(d_in, u_traj=traj)
    torch.cuda.synchronize()
    out = q.from_device(d_out, q.TICK_OUT)
    assert_parity(q, out, ref)
    t = traj.cpu().numpy()
    ok = (ref["status"] & q.ST_ERROR_MASK) == 0
    g = z["u_traj"]
    assert np.abs(t[ok, 0] - g[ok, 0]).max() <= 1e-6 * np.abs(g[ok, 0]).max()
    assert np.abs(t[ok, 1:] - g[ok, 1:]).max() <= 2e-6          # qpOASES stops at 2.2e-7 homotopy length
    assert np.array_equal(t[:, :, 0], out["u0"])


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("batch", [1, 3, 15, 16, 17, 1000])
def test_ragged_batches_and_batch_independence(q, batch, path):
    from quadruped_gait_generation_ismpc_amd import workload
    s = solver_for(q, 100, path)
    tin = workload.make_batch(100, 1024, seed=3)
    full = s.solve_batch(tin)
    part = s.solve_batch(tin[:batch])
    assert part.tobytes() == full[:batch].tobytes()        # bit identical whatever the batch / workgroup tiling
    perm = np.random.default_rng(0).permutation(batch)
    assert s.solve_batch(tin[:batch][perm]).tobytes() == full[:batch][perm].tobytes()


def test_batch_size_classes_agree_to_rounding(q):
    """The lane layout follows the batch size (32 lanes per instance up to 2 048 instances per launch, 16 up to 8 192, 8 beyond): inside a
    class an instance's record is byte-identical whatever the batch (above), across classes the sums are taken in another order and the
    records agree to rounding (a QP within rounding of the feasibility boundary may be flagged either way)."""
    from quadruped_gait_generation_ismpc_amd import workload
    s = solver_for(q, 100, "auto")
    tin = workload.make_batch(100, 9000, seed=4)
    big, mid, small = s.solve_batch(tin), s.solve_batch(tin[:8000]), s.solve_batch(tin[:2000])
    assert mid.tobytes() != big[:8000].tobytes() and small.tobytes() != mid[:2000].tobytes()      # they ARE different layouts
    for a, b in ((big[:8000], mid), (mid[:2000], small), (big[:2000], small)):
        same = a["status"] == b["status"]
        assert same.mean() >= 0.999
        for i in np.flatnonzero(~same):           # every disagreement is a horizontal QP within 1e-9 (relative) of the feasibility boundary, and nothing else differs
            assert ((a["status"][i] ^ b["status"][i]) & ~(q.ST_X_INFEASIBLE | q.ST_Y_INFEASIBLE)) == 0, (i, a["status"][i], b["status"][i])
            assert _on_feasibility_boundary(q, 100, tin[i], band=1e-9), (i, a["status"][i], b["status"][i])
        ok = same & ((a["status"] & q.ST_ERROR_MASK) == 0)
        assert np.abs(a["com_pos"] - b["com_pos"])[ok].max() <= 1e-13 and np.abs(a["com_vel"] - b["com_vel"])[ok].max() <= 1e-12
        assert (np.abs(a["u0"] - b["u0"])[ok] <= 1e-10 * np.maximum(np.abs(a["u0"][ok]), 1.0)).all()
    assert s.solve_batch(tin[:8500]).tobytes() == big[:8500].tobytes()                             # same class: the bytes


def test_empty_batch(q):
    s = solver_for(q, 100)
    assert len(s.solve_batch(np.zeros(0, dtype=q.TICK_IN))) == 0


def test_passthrough_rules(q, O):
    """MPCSolver.cpp:214 gate and the midpoint window guard: state returned unchanged, flagged."""
    s = solver_for(q, 100)
    st = O.initial_state().view(q.TICK_IN); st["footstep_counter"] = 1
    recs = np.repeat(st, 4)
    recs["simulation_time"] = [0.0, 1700.0, -5.0, 3.0]
    recs["mpc_iter"] = [0, 0, 0, -2]
    out = s.solve_batch(recs)
    assert out["status"][0] & q.ST_ERROR_MASK == 0
    for k in (1, 2, 3):
        assert out["status"][k] == q.ST_BAD_INDEX
        assert np.array_equal(out["com_pos"][k], recs["com_pos"][k]) and np.all(out["u0"][k] == 0)
    ref, _ = O.Oracle(O.default_params(100), backend="gi").solve(recs)
    assert np.array_equal(ref["status"], out["status"])
    # dt = 0.05 -> the tick runs only when controlIter % 5 == 0
    p5 = dict(mpc_dt=0.05, N=20, S=7, F=2)
    s5 = solver_for(q, 20, mpc_dt=0.05, S=7, F=2)
    r = np.repeat(st, 2); r["control_iter"] = [3, 5]
    o5 = s5.solve_batch(r)
    assert o5["status"][0] == q.ST_TICK_SKIPPED and np.array_equal(o5["com_pos"][0], r["com_pos"][0])
    assert o5["status"][1] & q.ST_TICK_SKIPPED == 0
    ref5, _ = O.Oracle(O.default_params(20, mpc_dt=0.05, S=7, F=2), O.reference_plan(S=7, F=2, mpc_dt=0.05), backend="gi").solve(r)
    assert np.array_equal(ref5["status"], o5["status"]) and rel_com(o5, ref5).max() <= TOL


def test_first_step_box_and_flight(q, O):
    """footstepCounter <= 1 uses the +-1 m box and no vertical equalities (MPCSolver.cpp:262-263,334-337);
    lambda_0 <= 2 skips Stage 3 (:322)."""
    z, tin, ref = load_golden(q, "preroll_N100.npz")
    sel = np.r_[0:45, 395:405, 440:450]
    out = solver_for(q, 100).solve_batch(tin[sel])
    assert_parity(q, out, ref[sel])
    assert (out["status"][45:] & q.ST_FLIGHT).any() and not (out["status"][:45] & q.ST_FLIGHT).any()
    fl = (out["status"] & q.ST_FLIGHT) != 0
    assert np.all(out["u0"][fl][:, 1:] == 0.0) and np.all(out["u0"][fl][:, 0] == 0.0)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("N,ticks", [(100, 600), (200, 300)])
def test_closed_loop_rollout_on_device(q, N, ticks, path):
    """ismpc_rollout_device = Controller.cpp:297-310,346-348,503-504 around solve(); against the
    committed nominal pre-roll (CPU oracle + reference qpOASES).  Counters bit exact."""
    import torch
    z, tin, ref = load_golden(q, f"preroll_N{N}.npz")
    s = solver_for(q, N, path)
    st0 = tin[:1].copy()
    st0["simulation_time"] = 0.0; st0["footstep_counter"] = 0; st0["mpc_iter"] = 0; st0["control_iter"] = 0
    d_state = q.to_device(np.repeat(st0, 5))
    traj = s.rollout_torch(d_state, 0, ticks)
    torch.cuda.synchronize()
    out = q.from_device(traj, q.TICK_OUT)                     # [ticks, 5]
    fin = q.from_device(d_state, q.TICK_IN)
    for b in range(5):
        assert rel_com(out[:, b], ref[:ticks]).max() <= TOL
        assert np.array_equal(out["status"][:, b], ref["status"][:ticks])
    nxt = tin[ticks]                                          # the oracle's input record of the next frame
    assert fin["control_iter"][0] == nxt["control_iter"] or nxt["control_iter"] == 0
    assert fin["footstep_counter"][0] in (nxt["footstep_counter"], nxt["footstep_counter"] - 1)
    assert fin["simulation_time"][0] == ticks - 1


def test_full_size_properties(q, O):
    """BASELINE config 2/3 sizes (1 024, the 8 192 shard and the full 65 536): properties that need no oracle, plus the oracle on a
    random sample of each batch.
    x/y QPs (MPCSolver.cpp:395-396): equality row satisfied, box respected, and the solution has the
    KKT form u = clip(mid - nu * a) with ONE multiplier nu; z QP: u_i = 0 on the equality samples."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    N = 100
    s = solver_for(q, N)
    mid = s.midpoint()
    for batch in (1024, 8192, 65536):
        tin = workload.make_batch(N, batch, seed=11)
        d_in = q.to_device(tin)
        traj = torch.zeros((batch, 3, N), dtype=torch.float64, device="cuda:0")
        out = q.from_device(s.solve_batch_torch(d_in, u_traj=traj), q.TICK_OUT)
        torch.cuda.synchronize()
        t = traj.cpu().numpy()
        again = q.from_device(s.solve_batch_torch(d_in), q.TICK_OUT)
        assert again.tobytes() == out.tobytes()               # deterministic
        st = out["status"]
        run = ((st & (q.ST_ERROR_MASK | q.ST_FLIGHT)) == 0)
        assert 0.6 < run.mean() < 0.9 and 0.15 < ((st & q.ST_FLIGHT) != 0).mean() < 0.3
        idx = tin["simulation_time"].astype(int)
        win = idx[:, None] + np.arange(N)[None, :]
        half = np.where(tin["footstep_counter"] > 1, 0.045, 1.0)[:, None]
        for ax in (0, 1):
            u = t[:, 1 + ax]; m = mid[win, ax]
            v = u - m
            assert (np.abs(v[run]) <= half[run] + 1e-12).all()
        # vertical equalities
        fc = tin["footstep_counter"]; it = tin["mpc_iter"]
        for b in np.where(((st & q.ST_ERROR_MASK) == 0) & (fc > 1))[0][:512]:
            lo, hi = (35 - it[b], 45 - it[b]) if it[b] < 35 else (0, 45 - it[b])
            assert np.all(t[b, 0, lo:hi] == 0.0)
            assert np.abs(t[b, 0, hi:hi + 20]).min() > 1.0
        # flight instances coast: x' = x + dt * xd
        fl = (st & q.ST_FLIGHT) != 0
        assert np.allclose(out["com_pos"][fl][:, :2], tin["com_pos"][fl][:, :2] + 0.01 * tin["com_vel"][fl][:, :2], rtol=0, atol=1e-15)
        # the equality row of both horizontal QPs holds: a'u = beq with a, beq recomputed by the oracle for a sample, and the
        # whole record against the oracle (reference qpOASES where oracle/_ref is built) for the same sample
        pick = np.random.default_rng(batch).choice(batch, 192, replace=False)
        ref, info = O.Oracle(O.default_params(N)).solve(tin[pick])
        assert_parity(q, out[pick], ref)
        # status equality by the rule of test_against_oracle_seeded: the only admissible difference is a horizontal QP within
        # 1e-9 (relative) of its feasibility boundary, which qpOASES' own termination test may classify either way
        for b in np.where(out["status"][pick] != ref["status"])[0]:
            assert ((out["status"][pick][b] ^ ref["status"][b]) & ~(q.ST_X_INFEASIBLE | q.ST_Y_INFEASIBLE)) == 0
            assert _on_feasibility_boundary(q, N, tin[pick[b]], band=1e-9), (batch, pick[b], out["status"][pick][b], ref["status"][b])


@pytest.mark.parametrize("batch,over", [(65536, dict()), (40001, dict()), (100, dict()), (65536, dict(z_ineq_hi=4.6))])
def test_zero_copy_host_path_is_bitwise_the_device_path(q, batch, over):
    """ismpc_solve_batch with page-locked caller buffers: the kernel reads and writes the caller's records in place over
    PCIe (no staging).  Same records, bit for bit, as the device-pointer entry point and as the staged pageable path --
    ragged batch sizes included, and with the vertical inequality rows active (tight z_ineq_hi: the fallback launch reads
    and rewrites records in host memory too)."""
    from quadruped_gait_generation_ismpc_amd import workload
    s = solver_for(q, 100, "auto", **over)
    tin = workload.make_batch(100, batch, seed=5)
    pin_in, pin_out = q.PinnedRecords(batch, q.TICK_IN), q.PinnedRecords(batch, q.TICK_OUT)
    pin_in.array[:] = tin
    pin_out.array["status"] = -1
    a = s.solve_batch(pin_in.array, out=pin_out.array).copy()
    b = s.solve_batch(tin)
    dev = q.from_device(s.solve_batch_torch(q.to_device(tin)), q.TICK_OUT)
    assert a.tobytes() == dev.tobytes() and b.tobytes() == dev.tobytes()
    assert pin_in.array.tobytes() == tin.tobytes()                                  # inputs untouched
    if over:
        assert ((dev["status"] & q.ST_Z_INEQ_ACTIVE) != 0).sum() > 100
        assert ((dev["status"] & q.ST_Z_FAILED) == 0).all()
    pin_in.free(); pin_out.free()


def test_host_registered_caller_buffers(q):
    """ismpc_host_register pins buffers the caller already owns (plain numpy arrays here): same zero-copy path, same bytes; after
    ismpc_host_unregister the very same arrays take the staged path, same bytes again."""
    import ctypes as C
    from quadruped_gait_generation_ismpc_amd import workload, _lib
    lib = _lib.load()
    s = solver_for(q, 100, "auto")
    B = 20000
    tin = workload.make_batch(100, B, seed=6)
    out = np.zeros(B, dtype=q.TICK_OUT)
    ref = s.solve_batch(tin).copy()
    assert lib.ismpc_host_register(tin.ctypes.data_as(C.c_void_p), tin.nbytes) == 0, _lib.last_error()
    assert lib.ismpc_host_register(out.ctypes.data_as(C.c_void_p), out.nbytes) == 0, _lib.last_error()
    try:
        a = s.solve_batch(tin, out=out).copy()
    finally:
        assert lib.ismpc_host_unregister(tin.ctypes.data_as(C.c_void_p)) == 0 and lib.ismpc_host_unregister(out.ctypes.data_as(C.c_void_p)) == 0
    out[:] = 0
    b = s.solve_batch(tin, out=out).copy()
    assert a.tobytes() == ref.tobytes() and b.tobytes() == ref.tobytes()


@pytest.mark.parametrize("batch", [8192, 65536])
def test_bitwise_reproducible_across_launch_variants_of_one_path(q, batch):
    """Same inputs, same path: byte-identical records run to run (8 192 takes the one-launch kernel, 65 536 the two-launch form)."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    s = solver_for(q, 100, "affine")
    d_in = q.to_device(workload.make_batch(100, batch), "cuda:0")
    o1 = s.solve_batch_torch(d_in).clone(); o2 = s.solve_batch_torch(d_in).clone()
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("sweep", [False, True])
def test_one_launch_and_two_launch_forms_agree_bitwise(q, sweep, monkeypatch):
    """A batch beyond the resident size takes ismpc_tick_quad_one: a wavefront that defers an instance (active vertical inequality rows)
    runs the fallback for it itself, on the instance's own parameter set in a sweep.  ISMPC_ONE_LAUNCH=0 is the two-launch form
    (ismpc_tick_quad + the deferred list + ismpc_tick_affine_fallback).  Same arithmetic: byte-identical records, also through ten
    ticks of the host-driven closed loop (state fed back in place, Controller.cpp:346-348)."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    B = 40000
    tin = workload.make_batch(100, B, seed=55)
    monkeypatch.setenv("ISMPC_ROLLOUT", "host")
    if sweep:
        sets = workload.make_sweep_params(8, N=100)
        for p in sets:
            p.z_ineq_hi = 4.6
        plan = q.reference_plan(params=sets[0])
        tin["reserved"] = np.arange(B) % 8
        make = lambda: q.MPCSolver.sweep(plan, sets)
    else:
        p = q.default_params(N=100, z_ineq_hi=4.6)
        plan = q.reference_plan(params=p)
        make = lambda: q.MPCSolver(plan, params=p)
    monkeypatch.setenv("ISMPC_ONE_LAUNCH", "3")           # (the default switches to two launches while instances are being deferred)
    one = make()
    monkeypatch.setenv("ISMPC_ONE_LAUNCH", "0")
    two = make()
    a, b = one.solve_batch(tin), two.solve_batch(tin)
    act = (a["status"] & q.ST_Z_INEQ_ACTIVE) != 0
    assert act.sum() > 100 and ((a["status"] & q.ST_Z_FAILED) == 0).all()
    assert a.tobytes() == b.tobytes()
    sa, sb = q.to_device(tin), q.to_device(tin)
    ta = one.rollout_torch(sa, int(tin["simulation_time"].max()) + 1, 10); tb = two.rollout_torch(sb, int(tin["simulation_time"].max()) + 1, 10)
    torch.cuda.synchronize()
    assert torch.equal(ta, tb) and torch.equal(sa, sb)


def test_deferred_list_counter_survives_rollouts_and_graph_replays(q, monkeypatch):
    """The deferred list of the two-launch form is counted by a self-resetting counter of the handle (DevConst::zflag): the fallback's last
    workgroup zeroes it.  Round 3 keyed two counters to the parity of the launch id, which a rollout between two ticks (it consumes an id
    without touching the list) or a hipGraph replay of one captured step (same id every time) turned into a count that only grew -- the
    fallback then re-solved instances nobody deferred and the append index ran past the list.  Here: tick, in-kernel rollout, tick on ONE
    handle against fresh handles, then one captured two-launch step replayed eight times; bytes equal and all counters back at zero."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    B = 40000                                                    # beyond the resident size: the list is in use
    p = q.default_params(N=100, z_ineq_hi=4.6)
    plan = q.reference_plan(params=p)
    monkeypatch.setenv("ISMPC_ONE_LAUNCH", "0")                   # the two-launch form on every step
    tin = workload.make_batch(100, B, seed=77)
    frame = int(tin["simulation_time"].max()) + 1
    h = q.MPCSolver(plan, params=p)
    d_in = q.to_device(tin)
    o1 = h.solve_batch_torch(d_in).clone()
    st = d_in.clone(); h.rollout_torch(st, frame, 3, want_traj=False)          # consumes a launch id, parks and resumes instances
    o2 = h.solve_batch_torch(d_in).clone()
    o3 = h.solve_batch_torch(st).clone()
    torch.cuda.synchronize()
    assert h.fallback_counters() == (0, 0, 0, 0)
    f1, f2 = q.MPCSolver(plan, params=p), q.MPCSolver(plan, params=p)
    r1 = f1.solve_batch_torch(d_in); st2 = d_in.clone(); f2.rollout_torch(st2, frame, 3, want_traj=False); r3 = f2.solve_batch_torch(st2)
    torch.cuda.synchronize()
    act = (q.from_device(r1, q.TICK_OUT)["status"] & q.ST_Z_INEQ_ACTIVE) != 0
    assert act.sum() > 100
    assert torch.equal(o1, r1) and torch.equal(o2, r1) and torch.equal(st, st2) and torch.equal(o3, r3)
    # one captured step, replayed: the scratch is sized first (no allocation inside the capture), the handle warmed up before the capture
    g = q.MPCSolver(plan, params=p); g.reserve(B)
    out = torch.empty((B, 80), dtype=torch.uint8, device="cuda:0")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g.solve_batch_torch(d_in, out); side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            g.solve_batch_torch(d_in, out)
    for k in range(8):
        out.zero_(); graph.replay(); torch.cuda.synchronize()
        assert torch.equal(out, r1), k
        c = g.fallback_counters()
        assert c == (0, 0, 0, 0), (k, c)
    for s_ in (h, f1, f2, g):
        s_.close()


@pytest.mark.parametrize("lay,host", [("affine", "hostloop"), ("lpi8", "hostloop8"), ("lpi32", "hostloop32")])
@pytest.mark.parametrize("N,ticks,over", [(100, 300, dict()), (50, 200, dict()), (100, 200, dict(z_ineq_hi=5.0)), (50, 200, dict(z_ineq_hi=1.3))])
def test_in_kernel_rollout_is_bitwise_the_per_tick_loop(q, O, N, ticks, over, lay, host):
    """ismpc_rollout_device keeps the tick loop inside one launch (ismpc_rollout_quad); ISMPC_ROLLOUT=host runs one launch per
    tick.  Same bookkeeping (Controller.cpp:297-310,503-504), same arithmetic: byte-identical trajectories and final states --
    also when the vertical inequality rows become active in the middle of the rollout (tight z_ineq_hi: first active around
    tick 88; the instance is parked by the first launch and resumed, fallback included, by the second) -- and the
    unperturbed instance against the oracle's closed loop."""
    import torch
    B = 37
    st0 = O.initial_state().view(q.TICK_IN)
    recs = np.repeat(st0, B)
    rng = np.random.default_rng(9)
    recs["com_pos"][1:, :2] += rng.uniform(-0.004, 0.004, (B - 1, 2))
    recs["com_vel"][1:, :2] += rng.uniform(-0.02, 0.02, (B - 1, 2))
    a, b = solver_for(q, N, lay, **over), solver_for(q, N, host, **over)
    sa, sb = q.to_device(recs), q.to_device(recs)
    ta = a.rollout_torch(sa, 0, ticks); tb = b.rollout_torch(sb, 0, ticks)
    torch.cuda.synchronize()
    assert torch.equal(ta, tb) and torch.equal(sa, sb)
    out = q.from_device(ta, q.TICK_OUT)                      # [ticks, B]
    ref, _, _, fin = O.Oracle(O.default_params(N, **over)).rollout(st0, 0, ticks)
    assert np.array_equal(out["status"][:, 0], ref["status"])
    assert rel_com(out[:, 0], ref).max() <= TOL
    endst = q.from_device(sa, q.TICK_IN)
    for k in ("mpc_iter", "control_iter", "footstep_counter", "simulation_time"):
        assert endst[k][0] == fin[k][0], k                  # counters bit exact
    if over:
        assert ((out["status"] & q.ST_Z_INEQ_ACTIVE) != 0).any(axis=0).all()      # every instance went through the resume launch
        assert ((out["status"] & q.ST_Z_FAILED) == 0).all()
    # without a trajectory buffer: same final state
    sc = q.to_device(recs)
    a.rollout_torch(sc, 0, ticks, want_traj=False); torch.cuda.synchronize()
    assert torch.equal(sc, sa)


def test_handle_scratch_growth_across_streams_and_independent_handles(q):
    """A handle's per-launch scratch outlives the call that allocated it: growing it on another stream than the previous launch's drains that
    stream first (round 2 freed it while kernels of the other stream could still use it).  A long closed loop on stream A, then at once a
    larger batch on stream B through the SAME handle, with the inequality fallback active in both; and two handles on two streams at once."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    p = q.default_params(N=100, z_ineq_hi=4.6)
    plan = q.reference_plan(params=p)
    h1, h2 = q.MPCSolver(plan, params=p), q.MPCSolver(plan, params=p)
    small, big = workload.make_batch(100, 3000, seed=31), workload.make_batch(100, 40000, seed=32)
    ref_small, ref_big = h2.solve_batch(small), h2.solve_batch(big)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    d_small, d_big = q.to_device(small), q.to_device(big)
    torch.cuda.synchronize()
    outs = []
    with torch.cuda.stream(sa):
        for _ in range(200):
            o_small = h1.solve_batch_torch(d_small)                       # 200 launches queued on stream A (scratch sized for 3 000)
    with torch.cuda.stream(sb):
        o_big = h1.solve_batch_torch(d_big)                               # growth to 40 000 on stream B while A may still be running
        for _ in range(20):
            outs.append(h2.solve_batch_torch(d_small))                    # another handle, concurrently
    torch.cuda.synchronize()
    assert q.from_device(o_small, q.TICK_OUT).tobytes() == ref_small.tobytes()
    assert q.from_device(o_big, q.TICK_OUT).tobytes() == ref_big.tobytes()
    assert q.from_device(outs[-1], q.TICK_OUT).tobytes() == ref_small.tobytes()
    assert ((ref_big["status"] & q.ST_Z_INEQ_ACTIVE) != 0).sum() > 100
    h1.close(); h2.close()
