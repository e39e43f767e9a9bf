"""GPU parity for Formulation A (the MATLAB ISMPC generators): the HIP range-space active-set kernel,
through the C ABI of include/ismpc_a.h, against
  (i)  the reference's checked-in MATLAB trajectories (tests/golden/formA_matlab_*.npz) -- whole files,
  (ii) the CPU oracle (oracle/ismpc_oracle_a.c) tick by tick, nominal and with impulsive pushes.
Tolerances: CoM vs the oracle <= 1e-6 relative (north star); vs the MATLAB files the print / quadprog
limits of SURVEY.md A.3 (trot 3e-6 m, walk 5e-5 m, velocity 1e-4 m/s); counters bit exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLDEN, "formA_matlab_meta.json")))
TOL_COM = {"trot": 3e-6, "walk": 5e-5}
# Oracle QP backends: "gi" = the oracle's own dense Goldfarb-Idnani (exact to rounding), "ref" = the reference's vendored
# qpOASES 3.2 (oracle/_ref, setToMPC).  qpOASES stops at a relative homotopy length of 1e9 EPS = 2.2e-7 (Options.cpp:206):
# its first ZMP velocity u0 sits up to 2e-5 m/s off the exact minimiser (measured gi-vs-ref on these very rollouts), which
# moves the CoM by < 3e-8 m and the velocity by < 2e-7 m/s per tick.  CoM / velocity / footstep tolerances are the same for
# both backends (the north star's 1e-6); only u0 is compared at the reference solver's own accuracy.
BACKENDS = ("gi", "ref")
TOL_U0 = {"gi": 1e-6, "ref": 6e-5}


def need_backend(backend):
    from oracle import oracle as O
    if backend == "ref" and not O.have_ref():
        pytest.skip("oracle/_ref (the reference's qpOASES) is not built here")


@pytest.fixture(scope="module")
def FA(built_libs):
    import torch
    assert torch.cuda.is_available()
    from quadruped_gait_generation_ismpc_amd import formulation_a as FA
    return FA


def _oracle_pool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("oracle_a_pool", os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "oracle_a_pool.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _record_maxima(key, values):
    """Measured parity maxima, printed (pytest -s) and appended to gpurun_out/parity_maxima.jsonl when that directory exists: the fp32
    tolerances of these tests are set from them (2x measured), not guessed."""
    line = json.dumps({"case": key, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in values.items()}})
    print("PARITY", line)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_maxima.jsonl"), "a") as f:
            f.write(line + "\n")


def q_to_dev(a):
    import quadruped_gait_generation_ismpc_amd as q
    return q.to_device(a)


def q_from_dev(t, dt):
    import quadruped_gait_generation_ismpc_amd as q
    return q.from_device(t, dt)


def make_gen(FA, name):
    m = META[name]
    kind = FA.WALK if m["gait"] == "walk" else FA.TROT
    g = FA.default_gait(kind, m["phi"], m["disp_A"])
    fp, ce = FA.plan(g)
    return FA.GaitGenerator(FA.default_params(kind), ce), g, m


@pytest.mark.parametrize("name", sorted(META))
def test_matlab_fixture_whole_file_on_device(FA, name):
    import torch
    gen, g, m = make_gen(FA, name)
    z = np.load(os.path.join(GOLDEN, f"formA_matlab_{name}.npz"))
    # trotting/phipi4/15cm holds 3 200 rows: the script opens its files 'a+' (quad_as_bip_no_plots.m:105-114), the file is a
    # 1 200-tick run followed by a fresh 2 000-tick run -- both are checked against ONE 2 000-tick rollout below
    ticks = 2000
    st = q_to_dev(gen.initial_state(g.disp_C, batch=3))
    traj = gen.rollout_torch(st, ticks)
    torch.cuda.synchronize()
    out = q_from_dev(traj, FA.OUT_A)                       # [ticks, 3]
    assert (out["status"] == 0).all()
    for b in range(3):
        assert out[:, b].tobytes() == out[:, 0].tobytes()  # identical instances -> identical bits
    runs = [(slice(0, 1200), 1200), (slice(1200, 3200), 2000)] if name == "trot_phipi4_15" else [(slice(0, 2000), 2000)]
    assert z["com"].shape[0] == runs[-1][0].stop                           # every row of the file is checked
    for rows, n in runs:
        com = z["com"][rows]
        err = np.abs(out["com_before"][:n, 0] - com[:, :2])
        assert err[:20].max() <= 6e-8 * max(1.0, np.abs(com[:20, :2]).max())
        assert err.max() <= TOL_COM[m["gait"]], err.max()
        if m["has_velocity"]:
            assert np.abs(out["vel_after"][:n, 0] - z["vel"][rows, :2]).max() <= 1e-4
    fin = q_from_dev(st, FA.STATE_A)
    step = gen.params.step
    assert fin["j"][0] == ticks + 1 and fin["fc"][0] == ticks // step + 1 and fin["rebuilt"][0] == 1


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,ticks", [("walk_phipi4", 400), ("trot_phipi4", 250), ("walk_phi0", 200), ("trot_phi0", 200)])
def test_rollout_against_oracle(FA, name, ticks, backend):
    import torch
    from oracle import oracle_a as A
    need_backend(backend)
    gen, g, m = make_gen(FA, name)
    kind = A.WALK if m["gait"] == "walk" else A.TROT
    ref = A.SimA(A.gait(kind, m["phi"], m["disp_A"]), A.params(kind), backend=backend).run(ticks)
    st = q_to_dev(gen.initial_state(g.disp_C, batch=1))
    out = q_from_dev(gen.rollout_torch(st, ticks), FA.OUT_A)[:, 0]
    torch.cuda.synchronize()
    assert (out["status"] == 0).all() and (ref["rv"] == 0).all()
    rel = np.abs(out["com_before"] - ref["com_before"]).max(1) / np.maximum(np.abs(ref["com_before"]).max(1), 1e-3)
    assert rel.max() <= 1e-6
    assert np.abs(out["vel_after"] - ref["vel_after"]).max() <= 1e-6
    assert np.abs(out["u0"] - ref["u0"]).max() <= TOL_U0[backend] and np.abs(out["f0"] - ref["f0"]).max() <= 1e-7


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("kind_name", ["walk", "trot"])
def test_pushed_ticks_against_oracle(FA, kind_name, backend):
    """Config-4 style instances: nominal state at a random tick + impulsive velocity push, ONE tick, batched."""
    import torch
    from oracle import oracle_a as A
    kind = A.WALK if kind_name == "walk" else A.TROT
    phi, dA = np.pi / 4, 0.1
    g = FA.default_gait(kind, phi, dA)
    fp, ce = FA.plan(g)
    need_backend(backend)
    gen = FA.GaitGenerator(FA.default_params(kind), ce)
    sim = A.SimA(A.gait(kind, phi, dA), A.params(kind), backend=backend)
    rng = np.random.default_rng(3)
    states, pushes, refs = [], [], []
    nticks = 420 if kind == A.WALK else 300
    for t in range(nticks):
        take = rng.random() < 0.12
        push = (rng.uniform(-0.03, 0.03), rng.uniform(-0.05, 0.05)) if (take and rng.random() < 0.7) else (0.0, 0.0)
        if take:
            o = sim.state
            fsx, fsy, _, _ = sim.get_plan()
            s = np.zeros(1, dtype=FA.STATE_A)
            for k in ("x", "xd", "xz", "y", "yd", "yz", "cur_x", "cur_y", "fc", "j"):
                s[k] = o[k]
            s["off_x"] = fsx[0] - ce[0, 0]; s["off_y"] = fsy[0] - ce[0, 1]; s["rebuilt"] = int(o["fc"] >= 2)
            states.append(s[0]); pushes.append(push)
        ref = sim.tick(push)
        if take:
            refs.append((ref.copy(), sim.state.copy()))
        assert ref["rv"][0] == 0 and ref["rv"][1] == 0
    states = np.array(states, dtype=FA.STATE_A); pushes = np.array(pushes)
    assert len(states) > 20
    d_st = q_to_dev(states)
    d_push = torch.from_numpy(pushes.copy()).to("cuda:0")
    out = q_from_dev(gen.tick_torch(d_st, d_push), FA.OUT_A)
    torch.cuda.synchronize()
    new = q_from_dev(d_st, FA.STATE_A)
    assert (out["status"] == 0).all()
    for i, (r, s_after) in enumerate(refs):
        assert np.abs(out["u0"][i] - r["u0"]).max() <= TOL_U0[backend] * max(1.0, np.abs(r["u0"]).max()), (i, out["u0"][i], r["u0"])
        assert np.abs(out["f0"][i] - r["f0"]).max() <= 1e-7
        assert np.abs(out["vel_after"][i] - r["vel_after"]).max() <= (1e-7 if backend == "gi" else 1e-6)
        for k in ("x", "xd", "xz", "y", "yd", "yz", "cur_x", "cur_y"):
            assert abs(new[k][i] - s_after[k]) <= 1e-7 * max(1.0, abs(s_after[k])), (i, k)
        assert new["fc"][i] == s_after["fc"] and new["j"][i] == s_after["j"]       # counters bit exact
    assert (out["active"] & 0xffff).max() > 40                                     # the pushes do load the working set


def test_overflow_and_bad_index_flags(FA):
    import torch
    g = FA.default_gait(FA.WALK, 0.0, 0.1)
    fp, ce = FA.plan(g)
    gen = FA.GaitGenerator(FA.default_params(FA.WALK, C=150, P=300), ce)       # F = 3 is too small for C = 150
    st = gen.initial_state(g.disp_C, batch=2); st["j"] = [1, 20]
    d = q_to_dev(st)
    out = q_from_dev(gen.tick_torch(d), FA.OUT_A)
    assert out["status"][0] == 0 and out["status"][1] == FA.ST_OVERFLOW
    assert q_from_dev(d, FA.STATE_A)["j"][1] == 20                             # untouched
    gen = FA.GaitGenerator(FA.default_params(FA.WALK), ce)
    st = gen.initial_state(g.disp_C, batch=2); st["j"] = [1, 4900]; st["fc"] = [1, 99]
    out = q_from_dev(gen.tick_torch(q_to_dev(st)), FA.OUT_A)
    assert out["status"][0] == 0 and out["status"][1] == FA.ST_BAD_INDEX


@pytest.mark.parametrize("name", sorted(META))
def test_foot_files_from_device_rollout(FA, name, tmp_path):
    """SURVEY.md 8f2 + 8f3 end to end on the device: 2000 ticks with the swing-foot QP after every tick, the foot
    plan read back, the four foot files generated -- against the checked-in MATLAB foot_*.txt, against the oracle,
    and written / re-read in the text wire format the DART controller parses (Controller.cpp:147-281)."""
    import torch
    from oracle import oracle_a as A
    gen, g, m = make_gen(FA, name)
    fp0, ce = FA.plan(g)
    ticks = 2000
    st = q_to_dev(gen.initial_state(g.disp_C, batch=2))
    feet = gen.feet_init_torch(g, fp0, batch=2)
    traj = gen.rollout_feet_torch(st, feet, ticks)
    torch.cuda.synchronize()
    out = q_from_dev(traj, FA.OUT_A)
    assert (out["status"] == 0).all()
    fpl = feet.cpu().numpy()
    assert np.array_equal(fpl[0], fpl[1])
    files = FA.foot_trajectories(g, gen.params.step, fpl[0], ticks)
    z = np.load(os.path.join(GOLDEN, f"formA_matlab_{name}.npz"))
    tol = 3e-6 if m["gait"] == "trot" else 5e-5
    have = [ft for ft in ("fl", "fr", "rl", "rr") if f"foot_{ft}" in z.files]      # trotting/phi0 holds two of the four files
    assert len(have) == (2 if name == "trot_phi0" else 4)
    for k, ft in enumerate(("fl", "fr", "rl", "rr")):
        if ft in have:
            assert np.abs(files[k] - z[f"foot_{ft}"]).max() <= tol, (ft, np.abs(files[k] - z[f"foot_{ft}"]).max())
    kind = A.WALK if m["gait"] == "walk" else A.TROT
    sim = A.SimA(A.gait(kind, m["phi"], m["disp_A"]), A.params(kind), backend="gi")
    sim.enable_feet(); sim.run(ticks)
    assert np.abs(files - sim.foot_trajectories(ticks)).max() <= 1e-7
    # wire format round trip: what the controller's sscanf("%f %f %f") would read
    k0 = ("fl", "fr", "rl", "rr").index(have[0])
    p = tmp_path / f"foot_{have[0]}_{name}.txt"
    FA.write_trajectory_txt(str(p), files[k0])
    back = np.loadtxt(p)
    assert back.shape == (ticks, 3) and np.abs(back - z[f"foot_{have[0]}"]).max() <= tol + 1e-6


@pytest.mark.parametrize("kind_name,over", [("walk", dict(C=150, P=300, F=4)),                 # BASELINE config 4 shape (RL=3, F=4)
                                            ("walk", dict(C=200, P=400, F=5)),                 # config 5 shape, step 50 (RL=4, F=5)
                                            ("trot", dict(C=200, P=400, F=6, step=40, ds=24)),  # config 5 shape, step 40 (RL=4, F=6)
                                            ("walk", dict(C=60, P=120, F=3))])                 # short horizon (RL=2 with idle lanes)
@pytest.mark.parametrize("backend", BACKENDS)
def test_other_horizons_against_oracle(FA, kind_name, over, backend):
    """F_A = ceil(C/step)+1 footsteps for long horizons (SURVEY.md 'Index limits'); every (rows-per-lane, F) kernel
    instantiation that the BASELINE configs reach, nominal closed loop + pushed single ticks, against the oracle."""
    import torch
    from oracle import oracle_a as A
    kind = A.WALK if kind_name == "walk" else A.TROT
    phi, dA = np.pi / 4, 0.1
    g = FA.default_gait(kind, phi, dA)
    fp, ce = FA.plan(g)
    gen = FA.GaitGenerator(FA.default_params(kind, **over), ce)
    need_backend(backend)
    okw = {("C_" if k == "C" else k): v for k, v in over.items()}
    sim = A.SimA(A.gait(kind, phi, dA), A.params(kind, **okw), backend=backend)
    ticks = 130 if backend == "gi" else 70
    st = q_to_dev(gen.initial_state(g.disp_C, batch=1))
    out = q_from_dev(gen.rollout_torch(st, ticks), FA.OUT_A)[:, 0]
    torch.cuda.synchronize()
    ref = sim.run(ticks)
    assert (out["status"] == 0).all() and (ref["rv"] == 0).all()
    assert np.abs(out["com_before"] - ref["com_before"]).max() <= 1e-6 * max(1.0, np.abs(ref["com_before"]).max())
    assert np.abs(out["u0"] - ref["u0"]).max() <= TOL_U0[backend] and np.abs(out["f0"] - ref["f0"]).max() <= 1e-7
    # pushed ticks from the end state
    rng = np.random.default_rng(1)
    fin = q_from_dev(st, FA.STATE_A)
    pushes = np.stack([rng.uniform(-0.03, 0.03, 12), rng.uniform(-0.05, 0.05, 12)], 1)
    d_st = q_to_dev(np.repeat(fin, 12)); d_push = torch.from_numpy(pushes.copy()).to("cuda:0")
    o2 = q_from_dev(gen.tick_torch(d_st, d_push), FA.OUT_A)
    base = sim.state.copy(); plan = sim.get_plan()
    for i in range(12):
        sim.state = base; sim.set_plan(*plan)
        r = sim.tick(tuple(pushes[i]))
        assert r["rv"][0] == 0 and r["rv"][1] == 0
        assert np.abs(o2["u0"][i] - r["u0"]).max() <= TOL_U0[backend] * max(1.0, np.abs(r["u0"]).max()), (i, o2["u0"][i], r["u0"])
        assert np.abs(o2["f0"][i] - r["f0"]).max() <= 1e-7
    assert (o2["status"] == 0).all()


def _mc_instances(FA, A, n, seed=5, C_=200):
    """BASELINE configs[4] draw: trot / walk by instance parity, height ~ U(0.50, 0.62), step ~ U{40..100},
    ds = round(0.6 step), F = ceil(C / step) + 1."""
    rng = np.random.default_rng(seed)
    inst = np.zeros(n, dtype=FA.INST_A)
    for i in range(n):
        kind = A.TROT if i % 2 == 0 else A.WALK
        step = int(rng.integers(40, 101))
        inst[i] = (rng.uniform(0.50, 0.62), 1e7 if kind == A.TROT else 1e9, step, int(round(0.6 * step)), -(-C_ // step) + 1, 0 if kind == A.TROT else 1)
    return inst


@pytest.mark.parametrize("name", ["walk_C150", "trot_C160"])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_handle_parameters_given_per_instance_equal_the_plain_tick(FA, name, precision):
    """Two independent set-ups of the same QP: the plain tick (handle constants prepared on the host, the anticipative tail from the
    host-built table per tick index) and the per-instance tick (PiPre record of the tick prologue, prefix sums of the stability row
    in closed form, the tail summed with the centreline in closed form) -- given the handle's OWN parameters per instance they must
    return the same tick, on a pushed bench batch."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    n = 4096
    w = workload.make_batch_a(name, n, stream=3)
    g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
    par = FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"])
    gen = FA.GaitGenerator(par, ce, precision=precision)
    inst = np.zeros(n, dtype=FA.INST_A)
    inst["height"] = par.height; inst["Qf"] = par.Qf; inst["step"] = par.step; inst["ds"] = par.ds; inst["F"] = par.F; inst["plan"] = 0
    dp = torch.from_numpy(w["push"].copy()).to("cuda:0")
    s1, s2 = q_to_dev(w["state"]), q_to_dev(w["state"])
    o1 = q_from_dev(gen.tick_torch(s1, dp), FA.OUT_A)
    o2 = q_from_dev(gen.tick_inst_torch(s2, q_to_dev(inst), dp), FA.OUT_A)
    torch.cuda.synchronize()
    f1, f2 = q_from_dev(s1, FA.STATE_A), q_from_dev(s2, FA.STATE_A)
    assert (o1["status"] == o2["status"]).all() and (o1["status"] == 0).mean() > 0.99
    assert (f1["j"] == f2["j"]).all() and (f1["fc"] == f2["fc"]).all() and (f1["rebuilt"] == f2["rebuilt"]).all()
    ok = o1["status"] == 0
    # measured (parity_maxima.jsonl): fp64 3.7e-10 / 3.1e-11 / 3.7e-12 (rounding of the two set-ups through the QP's conditioning); the fp32
    # solves come out bitwise equal (state 4e-16: cosh / sinh of the LIP update from two formulas), bounded here by the fp32-vs-oracle limits
    tol_u, tol_f, tol_s = (2e-9, 2e-10, 2e-11) if precision == "f64" else (6e-5, 6e-8, 6e-8)
    du = np.abs(o1["u0"] - o2["u0"])[ok].max(); df = np.abs(o1["f0"] - o2["f0"])[ok].max()
    ds = max(np.abs(f1[k] - f2[k])[ok].max() for k in ("x", "xd", "xz", "y", "yd", "yz"))
    _record_maxima("inst_equals_plain", dict(name=name, precision=precision, du=float(du), df=float(df), ds=float(ds)))
    assert du <= tol_u and df <= tol_f and ds <= tol_s, (du, df, ds)


@pytest.mark.parametrize("backend", BACKENDS)
def test_per_instance_parameters_against_oracle(FA, backend):
    """Monte-Carlo batch (BASELINE configs[4]): every instance has its own CoM height, step timing, footstep count, Qf and
    gait; one launch on the device against one oracle run per instance (closed loop, then one pushed tick each)."""
    import torch
    from oracle import oracle_a as A
    need_backend(backend)
    Cn, Pn, n, ticks = 200, 400, (10 if backend == "gi" else 4), (110 if backend == "gi" else 45)
    phi, dA = np.pi / 4, 0.1
    inst = _mc_instances(FA, A, n)
    assert inst["F"].max() <= 6 and inst["F"].min() >= 3
    plans = [FA.plan(FA.default_gait(k, phi, dA))[1] for k in (A.TROT, A.WALK)]
    gen = FA.GaitGenerator(FA.default_params(A.TROT, C=Cn, P=Pn, F=6), plans[0])
    assert gen.add_plan(plans[1]) == 1
    d_inst = q_to_dev(inst)
    st = q_to_dev(gen.initial_state(0.88, batch=n))
    out = q_from_dev(gen.rollout_inst_torch(st, d_inst, ticks), FA.OUT_A)
    torch.cuda.synchronize()
    fin = q_from_dev(st, FA.STATE_A)
    rng = np.random.default_rng(2)
    pushes = np.stack([rng.uniform(-0.03, 0.03, n), rng.uniform(-0.05, 0.05, n)], 1)
    o2 = q_from_dev(gen.tick_inst_torch(st, d_inst, torch.from_numpy(pushes.copy()).to("cuda:0")), FA.OUT_A)
    assert (out["status"] == 0).all() and (o2["status"] == 0).all()
    for i in range(n):
        kind = A.TROT if inst["plan"][i] == 0 else A.WALK
        p = A.params(kind, C_=Cn, P=Pn, F=int(inst["F"][i]), step=int(inst["step"][i]), ds=int(inst["ds"][i]), Qf=float(inst["Qf"][i]))
        p.height = float(inst["height"][i])
        sim = A.SimA(A.gait(kind, phi, dA), p, backend=backend)
        ref = sim.run(ticks)
        assert (ref["rv"] == 0).all()
        o = out[:, i]
        assert np.abs(o["com_before"] - ref["com_before"]).max() <= 1e-6 * max(1.0, np.abs(ref["com_before"]).max()), i
        assert np.abs(o["vel_after"] - ref["vel_after"]).max() <= 1e-6, i
        assert np.abs(o["u0"] - ref["u0"]).max() <= TOL_U0[backend] and np.abs(o["f0"] - ref["f0"]).max() <= 1e-7, i
        s_end = sim.state
        assert int(s_end["fc"]) == int(fin["fc"][i]) and int(s_end["j"]) == int(fin["j"][i])        # counters bit-exact
        r = sim.tick(tuple(pushes[i]))
        assert r["rv"][0] == 0 and r["rv"][1] == 0
        assert np.abs(o2["u0"][i] - r["u0"]).max() <= TOL_U0[backend] * max(1.0, np.abs(r["u0"]).max()), (i, o2["u0"][i], r["u0"])
        assert np.abs(o2["f0"][i] - r["f0"]).max() <= 1e-7, i


def test_per_instance_swing_foot_qps_against_oracle(FA, tmp_path):
    """SURVEY 8f2 for Monte-Carlo batches (BASELINE configs[4]): every instance carries its own gait (trot / walk by base plan),
    CoM height, step timing and footstep count, and its own foot plan re-placed by the swing-foot QP after every tick
    (trotting/quad_as_bip_no_plots.m:332-426 + compute_two_feet1.m, walking/quad_walk_no_plots.m:336-504 +
    compute_one_feet_walk.m:84-140).  One device rollout against one oracle run with enable_feet() per instance: foot plans,
    and the four foot files wherever the script's writer applies (the trot writer hard-codes 50 swing rows per step)."""
    import torch
    from oracle import oracle_a as A
    Cn, Pn, n, ticks = 200, 400, 8, 420
    phi, dA = np.pi / 4, 0.1
    inst = _mc_instances(FA, A, n, seed=9)
    gaits = [FA.default_gait(k, phi, dA) for k in (A.TROT, A.WALK)]
    fplans, plans = zip(*[FA.plan(g) for g in gaits])
    gen = FA.GaitGenerator(FA.default_params(A.TROT, C=Cn, P=Pn, F=6), plans[0]); gen.add_plan(plans[1])
    d_inst = q_to_dev(inst)
    st = q_to_dev(gen.initial_state(0.88, batch=n))
    feet = gen.feet_init_inst_torch(gaits, fplans, d_inst)
    out = q_from_dev(gen.rollout_feet_inst_torch(st, d_inst, feet, ticks), FA.OUT_A)
    torch.cuda.synchronize()
    assert (out["status"] == 0).all()
    fpl = feet.cpu().numpy()
    moved = 0
    for i in range(n):
        kind = A.TROT if inst["plan"][i] == 0 else A.WALK
        p = A.params(kind, C_=Cn, P=Pn, F=int(inst["F"][i]), step=int(inst["step"][i]), ds=int(inst["ds"][i]), Qf=float(inst["Qf"][i]))
        p.height = float(inst["height"][i])
        sim = A.SimA(A.gait(kind, phi, dA), p, backend="gi")
        sim.enable_feet()
        ref = sim.run(ticks)
        assert (ref["rv"] == 0).all()
        assert np.abs(out["com_before"][:, i] - ref["com_before"]).max() <= 1e-6
        fo = sim.foot_plan()
        assert np.abs(fpl[i][:fo.shape[0]] - fo).max() <= 1e-7, (i, np.abs(fpl[i][:fo.shape[0]] - fo).max())
        moved += int(np.abs(fo[:len(fplans[inst["plan"][i]])] - fplans[inst["plan"][i]][:fo.shape[0]]).max() > 1e-9)
        step = int(inst["step"][i])
        if kind == A.WALK or step > 50:
            files = FA.foot_trajectories(gaits[inst["plan"][i]], step, fpl[i], ticks)
            assert np.abs(files - sim.foot_trajectories(ticks)).max() <= 1e-7, i
            if i < 2:
                pth = tmp_path / f"foot_fl_{i}.txt"
                FA.write_trajectory_txt(str(pth), files[0])
                assert np.abs(np.loadtxt(pth) - files[0]).max() <= 1e-6
    assert moved >= n // 2                                     # the QPs did re-place feet
    # handle-wide entry points still refuse a per-instance batch that was never initialised
    gen2 = FA.GaitGenerator(FA.default_params(A.TROT, C=Cn, P=Pn, F=6), plans[0]); gen2.add_plan(plans[1])
    with pytest.raises(FA.IsmpcAError):
        gen2.rollout_feet_inst_torch(q_to_dev(gen2.initial_state(0.88, batch=n)), d_inst, feet, 1)


def test_per_instance_invalid_records_are_flagged(FA):
    import torch
    from oracle import oracle_a as A
    g = FA.default_gait(A.WALK, 0.0, 0.1)
    _, ce = FA.plan(g)
    gen = FA.GaitGenerator(FA.default_params(A.WALK), ce)
    inst = np.zeros(4, dtype=FA.INST_A)
    inst[0] = (0.56, 1e9, 50, 30, 3, 0)          # the handle's own values
    inst[1] = (0.56, 1e9, 50, 50, 3, 0)          # ds >= step
    inst[2] = (0.56, 1e9, 50, 30, 4, 0)          # F beyond the handle's
    inst[3] = (0.56, 1e9, 50, 30, 3, 1)          # plan that was never added
    st0 = gen.initial_state(g.disp_C, batch=4)
    st = q_to_dev(st0)
    o = q_from_dev(gen.tick_inst_torch(st, q_to_dev(inst)), FA.OUT_A)
    after = q_from_dev(st, FA.STATE_A)
    assert o["status"][0] == 0 and after["j"][0] == 2
    assert (o["status"][1:] & FA.ST_BAD_INDEX).all()
    assert (after["j"][1:] == 1).all() and (after["x"][1:] == st0["x"][1:]).all()
    # record 0 equals the handle-wide path bit for bit in its counters and to rounding in the solution
    st_b = q_to_dev(st0[:1]); ob = q_from_dev(gen.tick_torch(st_b), FA.OUT_A)
    assert np.abs(ob["u0"][0] - o["u0"][0]).max() <= 1e-9 and np.abs(ob["f0"][0] - o["f0"][0]).max() <= 1e-10


@pytest.mark.parametrize("route", ["default", "passes_only", "no_peel"])
@pytest.mark.parametrize("name", ["walk_C100", "walk_C150", "trot_C160"])
def test_block_warm_start_equals_cold_start(FA, name, route, monkeypatch):
    """The block warm start (a few exact Goldfarb-Idnani steps, primal-dual passes, rounds of both) only changes the route,
    never the optimum: 4 096 perturbed instances per workload, pushes from mild to far beyond what the ZMP band can absorb
    (infeasible QPs included), warm-started handle against a handle created with ISMPC_A_WARM=0.  Routes: the default; the
    passes alone from the equality-only point (no exact steps first, no rounds); negative run ends leave alone."""
    import torch
    z = np.load(os.path.join(GOLDEN, f"prerollA_{name}.npz"))
    tab = z["state"].view(FA.STATE_A).reshape(-1)
    kind = int(z["gait"]); g = FA.default_gait(kind, float(z["phi"]), float(z["disp_A"]))
    _, ce = FA.plan(g)
    p = FA.default_params(kind, C=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
    monkeypatch.delenv("ISMPC_A_WARM", raising=False)
    if route == "passes_only": monkeypatch.setenv("ISMPC_A_WARM", "4,6,0,6,0,1,0,8")
    if route == "no_peel": monkeypatch.setenv("ISMPC_A_WARM", "6,12,0,6,3,0,2,8")
    warm = FA.GaitGenerator(p, ce)
    monkeypatch.setenv("ISMPC_A_WARM", "0")
    cold = FA.GaitGenerator(p, ce)
    rng = np.random.default_rng(11)
    B = 4096
    st0 = tab[rng.integers(0, len(tab), B)].copy()
    scale = rng.choice([1.0, 3.0, 10.0, 30.0], B, p=[0.55, 0.25, 0.15, 0.05])
    push = np.stack([rng.uniform(-0.03, 0.03, B), rng.uniform(-0.05, 0.05, B)], 1) * scale[:, None]
    d_push = torch.from_numpy(push.copy()).to("cuda:0")
    sw, sc = q_to_dev(st0), q_to_dev(st0)
    ow = q_from_dev(warm.tick_torch(sw, d_push), FA.OUT_A)
    oc = q_from_dev(cold.tick_torch(sc, d_push), FA.OUT_A)
    torch.cuda.synchronize()
    inf = FA.ST_X_INFEASIBLE | FA.ST_Y_INFEASIBLE
    assert ((ow["status"] & ~(inf | FA.ST_UNVERIFIED)) == 0).all() and ((oc["status"] & ~(inf | FA.ST_UNVERIFIED)) == 0).all()
    # the same QPs are reported infeasible; the only admissible difference is a degenerate vertex (every variable pinned)
    # that one route verifies and the other reports as UNVERIFIED
    diff = (ow["status"] & inf) != (oc["status"] & inf)
    assert diff.sum() <= B // 1000
    assert (((ow["status"] | oc["status"])[diff] & FA.ST_UNVERIFIED) != 0).all()
    ok = (ow["status"] == 0) & (oc["status"] == 0)
    assert ok.sum() > B // 2 and (~ok).sum() > 0                                   # both kinds are present
    assert np.abs(ow["u0"][ok] - oc["u0"][ok]).max() <= 1e-7 * max(1.0, np.abs(oc["u0"][ok]).max())
    assert np.abs(ow["f0"][ok] - oc["f0"][ok]).max() <= 1e-8
    assert (ow["active"][ok] == oc["active"][ok]).mean() > 0.99                    # same working-set sizes (ties aside)
    a, b = q_from_dev(sw, FA.STATE_A), q_from_dev(sc, FA.STATE_A)
    assert (a["fc"] == b["fc"]).all() and (a["j"] == b["j"]).all()
    assert ow["iters_x"][ok].mean() < (0.4 if route == "default" else 0.6) * oc["iters_x"][ok].mean()   # and it does shorten the route


def test_pushed_rollout_with_history_against_oracle(FA):
    """Closed loop with a disturbance in the middle: 40 nominal ticks, ONE pushed tick (caller-driven, working-set history
    switched on so that the stale guess is used and has to be repaired), 60 more ticks -- per instance against the oracle."""
    import torch
    from oracle import oracle_a as A
    phi, dA = np.pi / 4, 0.1
    g = FA.default_gait(A.WALK, phi, dA)
    _, ce = FA.plan(g)
    gen = FA.GaitGenerator(FA.default_params(A.WALK), ce)
    gen.set_warm_history(True)
    pushes = np.array([[0.03, -0.05], [-0.025, 0.04], [0.0, 0.05], [0.02, 0.0]])
    n = len(pushes)
    st = q_to_dev(gen.initial_state(g.disp_C, batch=n))
    outs = []
    for t in range(40):
        outs.append(q_from_dev(gen.tick_torch(st), FA.OUT_A))
    outs.append(q_from_dev(gen.tick_torch(st, torch.from_numpy(pushes.copy()).to("cuda:0")), FA.OUT_A))
    for t in range(60):
        outs.append(q_from_dev(gen.tick_torch(st), FA.OUT_A))
    torch.cuda.synchronize()
    out = np.stack(outs)                                                            # [101, n]
    assert (out["status"] == 0).all()
    for i in range(n):
        sim = A.SimA(A.gait(A.WALK, phi, dA), A.params(A.WALK), backend="gi")
        ref = list(sim.run(40)) + [sim.tick(tuple(pushes[i]))] + list(sim.run(60))
        rc = np.array([r["com_before"] for r in ref]); ru = np.array([r["u0"] for r in ref]); rf = np.array([r["f0"] for r in ref])
        assert np.abs(out["com_before"][:, i] - rc).max() <= 1e-6 * max(1.0, np.abs(rc).max()), i
        assert np.abs(out["u0"][:, i] - ru).max() <= 1e-6 * max(1.0, np.abs(ru).max()), i
        assert np.abs(out["f0"][:, i] - rf).max() <= 1e-7, i
    # the history does its job: the typical QP of the loop costs a few passes (step changes and the push cost more)
    both = np.concatenate([out["iters_x"].ravel(), out["iters_y"].ravel()])
    assert np.median(both) <= 4


def test_full_batch_is_bitwise_reproducible(FA):
    """Work is claimed dynamically (any wavefront may solve any QP, in any order), results must not depend on it:
    16 384 pushed instances twice, byte-identical outputs and states."""
    import torch
    z = np.load(os.path.join(GOLDEN, "prerollA_walk_C150.npz"))
    tab = z["state"].view(FA.STATE_A).reshape(-1)
    kind = int(z["gait"]); g = FA.default_gait(kind, float(z["phi"]), float(z["disp_A"]))
    _, ce = FA.plan(g)
    gen = FA.GaitGenerator(FA.default_params(kind, C=int(z["C"]), P=int(z["P"]), F=int(z["F"])), ce)
    rng = np.random.default_rng(4)
    B = 16384
    st0 = tab[rng.integers(0, len(tab), B)].copy()
    push = torch.from_numpy(np.stack([rng.uniform(-0.03, 0.03, B), rng.uniform(-0.05, 0.05, B)], 1)).to("cuda:0")
    s1, s2 = q_to_dev(st0), q_to_dev(st0)
    o1 = gen.tick_torch(s1, push); o2 = gen.tick_torch(s2, push)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(s1, s2)
    o = q_from_dev(o1, FA.OUT_A)
    assert (o["status"] == 0).all()


def _lip_matrices(eta, dt):
    ch, sh = np.cosh(eta * dt), np.sinh(eta * dt)
    return np.array([[ch, sh / eta, 1 - ch], [eta * sh, ch, -eta * sh], [0, 0, 1]]), np.array([dt - sh / eta, 1 - ch, dt])


# BASELINE configs[3]: walk, C=150, 16 384 instances, one GPU, in both arithmetic types (the config names fp32).
# BASELINE configs[4]: Monte-Carlo, C=200, GLOBAL batch 131 072 = eight shards of 16 384, rank r draws make_inst_mc(stream=r)
# exactly as bench.py --gpus 8 does: every shard in fp32 (the dtype the config names), shard 0 in fp64 too.
FULL_BATCH_CASES = [("walk_C150", "f64", 0), ("walk_C150", "f32", 0), ("mc_C200", "f64", 0)] + [("mc_C200", "f32", r) for r in range(8)]


@pytest.mark.parametrize("workload_name,precision,stream", FULL_BATCH_CASES)
def test_full_batch_properties_and_oracle_sample(FA, workload_name, precision, stream):
    """The exact bench workloads at BASELINE's sizes and dtypes.  Size-independent properties on EVERY instance -- the LIP
    update of the state from the returned u0 (quad_walk_no_plots.m:297-322), the footstep bookkeeping (:522-556), the record
    echoing its input -- and the oracle (reference qpOASES where built) on a random sample of the same shard."""
    import torch
    from oracle import oracle_a as A
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    B = 16384
    backend = "ref" if O.have_ref() else "gi"
    f32 = precision == "f32"
    # fp32 solve: 2x the maxima measured on these very samples (gpurun_out/parity_maxima.jsonl, round 4: 128 / 64 oracle instances per case,
    # eleven cases: u0 <= 2.2e-5 relative -- the size of qpOASES's own distance from the exact minimiser, TOL_U0["ref"] --, f0 <= 2.6e-8 m,
    # velocity <= 1.9e-8 m/s, next CoM <= 2.4e-8 relative); rounds 2-3 held them at 2e-3 / 2e-5 / 5e-6
    tol_u0 = 6e-5 if f32 else TOL_U0[backend]
    tol_f0, tol_v = (6e-8, 4e-8) if f32 else (1e-7, 1e-6)
    phi, dA = np.pi / 4, 0.1
    if workload_name == "mc_C200":
        Cn, Pn = 200, 400
        inst, push = workload.make_inst_mc(B, stream=stream)
        plans = [FA.plan(FA.default_gait(k, phi, dA))[1] for k in (0, 1)]
        gen = FA.GaitGenerator(FA.default_params(0, C=Cn, P=Pn, F=6), plans[0], precision=precision); gen.add_plan(plans[1])
        d_inst = q_to_dev(inst)
        d = q_to_dev(gen.initial_state(0.88, batch=B))
        if f32:                                             # the nominal pre-roll that spreads the gait phases is data preparation: fp64
            prep = FA.GaitGenerator(FA.default_params(0, C=Cn, P=Pn, F=6), plans[0]); prep.add_plan(plans[1])
            prep.rollout_inst_torch(d, d_inst, 60); torch.cuda.synchronize(); prep.close()
        else:
            gen.rollout_inst_torch(d, d_inst, 60)
        st0 = q_from_dev(d, FA.STATE_A).copy()
        out = q_from_dev(gen.tick_inst_torch(d, d_inst, torch.from_numpy(push.copy()).to("cuda:0")), FA.OUT_A)
        eta = np.sqrt(9.8 / inst["height"]); step = inst["step"]
        plan_of = lambda i: plans[inst["plan"][i]]
    else:
        w = workload.make_batch_a(workload_name, B, stream=stream)
        Cn, Pn = w["C"], w["P"]
        g = FA.default_gait(w["kind"], w["phi"], w["disp_A"])
        _, ce = FA.plan(g)
        gen = FA.GaitGenerator(FA.default_params(w["kind"], C=Cn, P=Pn, F=w["F"]), ce, precision=precision)
        st0, push = w["state"], w["push"]
        d = q_to_dev(st0)
        out = q_from_dev(gen.tick_torch(d, torch.from_numpy(push.copy()).to("cuda:0")), FA.OUT_A)
        eta = np.full(B, np.sqrt(9.8 / 0.56)); step = np.full(B, gen.params.step)
        plan_of = lambda i: ce
    torch.cuda.synchronize()
    new = q_from_dev(d, FA.STATE_A)
    assert (out["status"] == 0).all()
    # the record echoes its input; the state is the LIP update with the returned u0; counters advance as the script's
    assert np.array_equal(out["com_before"][:, 0], st0["x"]) and np.array_equal(out["com_before"][:, 1], st0["y"])
    for i in np.random.default_rng(0).choice(B, 2048, replace=False):
        Au, Bu = _lip_matrices(eta[i], 0.01)
        for ax, (p_, v_, z_) in enumerate((("x", "xd", "xz"), ("y", "yd", "yz"))):
            s_ = np.array([st0[p_][i], st0[v_][i] + push[i, ax], st0[z_][i]])
            nxt = Au @ s_ + Bu * out["u0"][i, ax]
            assert np.abs(nxt - np.array([new[p_][i], new[v_][i], new[z_][i]])).max() <= 1e-12, (i, ax)
            assert abs(out["vel_after"][i, ax] - nxt[1]) <= 1e-12
    stepped = st0["j"] + 1 >= step * st0["fc"]
    assert np.array_equal(new["j"], st0["j"] + 1) and np.array_equal(new["fc"], st0["fc"] + stepped)
    assert np.array_equal(new["cur_x"][stepped], out["f0"][stepped, 0]) and np.array_equal(new["cur_x"][~stepped], st0["cur_x"][~stepped])
    # oracle on a sample of the same batch: 128 instances (64 of the Monte-Carlo shards, whose oracle also runs the 60-tick pre-roll), solved
    # in worker processes on the host cores (tests/helpers/oracle_a_pool.py; they never touch the GPU)
    pool = _oracle_pool()
    pick = np.random.default_rng(1 + stream).choice(B, 128 if workload_name != "mc_C200" else 64, replace=False)
    items = []
    for i in pick:
        if workload_name == "mc_C200":
            items.append(dict(kind=(A.TROT if inst["plan"][i] == 0 else A.WALK), phi=phi, dA=dA, C=Cn, P=Pn, F=int(inst["F"][i]), preroll=60,
                              mc=dict(step=int(inst["step"][i]), ds=int(inst["ds"][i]), Qf=float(inst["Qf"][i]), height=float(inst["height"][i])),
                              state=st0[i:i + 1].tobytes(), push=(float(push[i, 0]), float(push[i, 1]))))
        else:
            items.append(dict(kind=w["kind"], phi=w["phi"], dA=w["disp_A"], C=Cn, P=Pn, F=w["F"], state=st0[i:i + 1].tobytes(), push=(float(push[i, 0]), float(push[i, 1]))))
    worst = dict(u0=0.0, f0=0.0, vel=0.0, com_rel=0.0)
    for i, r in zip(pick, pool.run(items, backend)):
        if workload_name == "mc_C200":
            assert r["pre_rv_max"] == 0
            for k in ("x", "xd", "xz", "y", "yd", "yz", "cur_x", "cur_y"):
                assert abs(r["pre"][k] - st0[k][i]) <= 1e-6 * max(1.0, abs(st0[k][i])), (i, k)     # the device pre-roll landed where the oracle's does
            assert r["pre"]["fc"] == int(st0["fc"][i]) and r["pre"]["j"] == int(st0["j"][i])
        assert r["rv"] == [0, 0]
        e_u0 = np.abs(out["u0"][i] - r["u0"]).max() / max(1.0, np.abs(r["u0"]).max())
        e_f0 = np.abs(out["f0"][i] - r["f0"]).max(); e_v = np.abs(out["vel_after"][i] - r["vel_after"]).max()
        e_c = max(abs(new[k][i] - r["after"][k]) / max(abs(r["after"][k]), 1e-3) for k in ("x", "y"))
        worst = dict(u0=max(worst["u0"], e_u0), f0=max(worst["f0"], e_f0), vel=max(worst["vel"], e_v), com_rel=max(worst["com_rel"], e_c))
        assert e_u0 <= tol_u0, (i, out["u0"][i], r["u0"])
        assert e_f0 <= tol_f0 and e_v <= tol_v, (i, e_f0, e_v)
        assert e_c <= (6e-8 if f32 else 1e-6), (i, e_c)       # the north star's figure (1e-6): next CoM, relative; fp32 held at 2x its measured 2.4e-8
    _record_maxima(f"full_batch:{workload_name}:{precision}:stream{stream}:{backend}", dict(worst, sample=len(pick), tol_u0=tol_u0, tol_f0=tol_f0, tol_v=tol_v))
    gen.close()


@pytest.mark.parametrize("workload_name", ["walk_C150", "mc_C200", "trot_C160"])
def test_fp32_solve_against_fp64_and_oracle(FA, workload_name):
    """BASELINE configs[3] / [4] name fp32: GaitGenerator(precision="f32") solves the QPs in fp32 (right-hand sides formed in
    fp64, LIP update in fp64).  Same pushed batches as the bench, 4 096 instances: every status equal to the fp64 solve's,
    next CoM within 2e-9 relative (the north star's tolerance is 1e-6), velocities within 3e-7 m/s, footsteps within 7e-8 m, first ZMP
    velocity within 3e-4 m/s -- twice the measured maxima (u0 moves the CoM by B_upd u0, |B_upd| = 3e-6 .. 9e-4, which is why the CoM is
    so much tighter than u0); and the fp32 result against the fp64 oracle (reference qpOASES where built) on a sample."""
    import torch
    from oracle import oracle_a as A
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    B = 4096
    phi, dA = np.pi / 4, 0.1
    outs, states = {}, {}
    for prec in ("f64", "f32"):
        if workload_name == "mc_C200":
            inst, push = workload.make_inst_mc(B)
            plans = [FA.plan(FA.default_gait(k, phi, dA))[1] for k in (0, 1)]
            gen = FA.GaitGenerator(FA.default_params(0, C=200, P=400, F=6), plans[0], precision=prec); gen.add_plan(plans[1])
            d_inst = q_to_dev(inst)
            if prec == "f64":
                d = q_to_dev(gen.initial_state(0.88, batch=B)); gen.rollout_inst_torch(d, d_inst, 60); st0 = q_from_dev(d, FA.STATE_A).copy()
            d = q_to_dev(st0)
            o = gen.tick_inst_torch(d, d_inst, torch.from_numpy(push.copy()).to("cuda:0"))
        else:
            w = workload.make_batch_a(workload_name, B)
            g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
            gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=prec)
            st0, push = w["state"], w["push"]
            d = q_to_dev(st0)
            o = gen.tick_torch(d, torch.from_numpy(push.copy()).to("cuda:0"))
        torch.cuda.synchronize()
        outs[prec] = q_from_dev(o, FA.OUT_A); states[prec] = q_from_dev(d, FA.STATE_A)
    o64, o32, s64, s32 = outs["f64"], outs["f32"], states["f64"], states["f32"]
    assert (o64["status"] == 0).all() and (o32["status"] == 0).all()
    com64 = np.stack([s64["x"], s64["y"]], 1); com32 = np.stack([s32["x"], s32["y"]], 1)
    rel = np.abs(com64 - com32).max(1) / np.maximum(np.abs(com64).max(1), 1e-3)
    # tolerances = 2x the maxima measured on these 4 096 instances per workload (gpurun_out/parity_maxima.jsonl, round 4: next CoM 9.9e-10
    # relative, velocity 1.4e-7 m/s, footstep 3.3e-8 m, u0 1.5e-4 m/s); rounds 2-3 held them at 1e-6 / 5e-6 / 2e-5 / 2e-3
    assert rel.max() <= 2e-9, rel.max()
    assert np.abs(np.stack([s64["xd"], s64["yd"]], 1) - np.stack([s32["xd"], s32["yd"]], 1)).max() <= 3e-7
    _record_maxima(f"fp32_vs_fp64:{workload_name}", dict(com_rel=rel.max(), vel=np.abs(np.stack([s64["xd"], s64["yd"]], 1) - np.stack([s32["xd"], s32["yd"]], 1)).max(),
                                                        f0=np.abs(o64["f0"] - o32["f0"]).max(), u0=np.abs(o64["u0"] - o32["u0"]).max(), sample=B))
    assert np.abs(o64["f0"] - o32["f0"]).max() <= 7e-8 and np.abs(o64["u0"] - o32["u0"]).max() <= 3e-4
    assert np.array_equal(s64["fc"], s32["fc"]) and np.array_equal(s64["j"], s32["j"])               # counters bit exact
    # the fp32 result against the oracle
    backend = "ref" if O.have_ref() else "gi"
    for i in np.random.default_rng(3).choice(B, 16 if workload_name != "mc_C200" else 6, replace=False):
        if workload_name == "mc_C200":
            kind = A.TROT if inst["plan"][i] == 0 else A.WALK
            p = A.params(kind, C_=200, P=400, F=int(inst["F"][i]), step=int(inst["step"][i]), ds=int(inst["ds"][i]), Qf=float(inst["Qf"][i]))
            p.height = float(inst["height"][i])
            sim = A.SimA(A.gait(kind, phi, dA), p, backend=backend)
        else:
            sim = A.SimA(A.gait(w["kind"], w["phi"], w["disp_A"]), A.params(w["kind"], C_=w["C"], P=w["P"], F=w["F"]), backend=backend)
        sim.load_product_state(st0[i])
        r = sim.tick(tuple(push[i])); after = sim.state
        assert r["rv"][0] == 0 and r["rv"][1] == 0
        for k in ("x", "y"):
            assert abs(s32[k][i] - after[k]) <= 1e-6 * max(abs(after[k]), 1e-3), (i, k)
        for k in ("xd", "yd"):
            assert abs(s32[k][i] - after[k]) <= 3e-7, (i, k)
        assert np.abs(o32["f0"][i] - r["f0"]).max() <= 7e-8, i


@pytest.mark.parametrize("name", sorted(META))
def test_fp32_closed_loop_reproduces_matlab_fixture(FA, name):
    """2 000 ticks of closed loop with the fp32 solve: still inside the print / quadprog limits of every checked-in MATLAB
    trajectory file (both appended runs of trotting/phipi4/15cm) and within 2e-6 m of the fp64 closed loop
    (measured 6e-7)."""
    import torch
    gen64, g, m = make_gen(FA, name)
    kind = FA.WALK if m["gait"] == "walk" else FA.TROT
    _, ce = FA.plan(g)
    gen32 = FA.GaitGenerator(FA.default_params(kind), ce, precision="f32")
    z = np.load(os.path.join(GOLDEN, f"formA_matlab_{name}.npz"))
    tr = {}
    for key, gen in (("f64", gen64), ("f32", gen32)):
        st = q_to_dev(gen.initial_state(g.disp_C, batch=2))
        tr[key] = q_from_dev(gen.rollout_torch(st, 2000), FA.OUT_A)[:, 0]
    torch.cuda.synchronize()
    assert (tr["f32"]["status"] == 0).all()
    runs = [(slice(0, 1200), 1200), (slice(1200, 3200), 2000)] if name == "trot_phipi4_15" else [(slice(0, 2000), 2000)]
    for rows, n in runs:
        assert np.abs(tr["f32"]["com_before"][:n] - z["com"][rows, :2]).max() <= TOL_COM[m["gait"]]
    assert np.abs(tr["f32"]["com_before"] - tr["f64"]["com_before"]).max() <= 2e-6
    assert np.abs(tr["f32"]["vel_after"] - tr["f64"]["vel_after"]).max() <= 5e-6


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_heavily_pushed_flags_match_reference_solver(FA, precision):
    """1 500 instances pushed 3x .. 30x harder than the bench workload: many QPs have every ZMP row active (the horizon is
    saturated: a degenerate vertex for the dual method, where a'a - G_EE cancels completely) and many are infeasible.
    Feasible / infeasible must be what the reference's qpOASES says (oracle/_ref; the oracle's own solver otherwise), QP by
    QP, and no feasible QP may come back flagged UNVERIFIED; solutions of the feasible ones within tolerance."""
    import torch
    from oracle import oracle_a as A
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    B = 1500
    w = workload.make_batch_a("walk_C100", B, seed=77)
    rng = np.random.default_rng(5)
    push = w["push"] * rng.choice([3.0, 10.0, 30.0], B, p=[0.4, 0.4, 0.2])[:, None]
    g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
    gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=precision)
    d = q_to_dev(w["state"])
    out = q_from_dev(gen.tick_torch(d, torch.from_numpy(push.copy()).to("cuda:0")), FA.OUT_A)
    torch.cuda.synchronize()
    backend = "ref" if O.have_ref() else "gi"
    sim = A.SimA(A.gait(w["kind"], w["phi"], w["disp_A"]), A.params(w["kind"], C_=w["C"], P=w["P"], F=w["F"]), backend=backend)
    inf_gpu = np.stack([(out["status"] & FA.ST_X_INFEASIBLE) != 0, (out["status"] & FA.ST_Y_INFEASIBLE) != 0], 1)
    inf_ref = np.zeros((B, 2), dtype=bool); u_ref = np.zeros((B, 2)); f_ref = np.zeros((B, 2))
    for i in range(B):
        sim.load_product_state(w["state"][i])
        r = sim.tick(tuple(push[i]))
        inf_ref[i] = r["rv"] != 0; u_ref[i] = r["u0"]; f_ref[i] = r["f0"]
    assert inf_ref.any() and (~inf_ref).sum() > B                         # both kinds are well represented
    assert (out["status"] & ~(FA.ST_X_INFEASIBLE | FA.ST_Y_INFEASIBLE | FA.ST_UNVERIFIED)) .max() == 0
    assert np.array_equal(inf_gpu, inf_ref), np.argwhere(inf_gpu != inf_ref)[:10]
    ok = ~inf_ref
    tol_u = (TOL_U0[backend] if precision == "f64" else 2e-3)
    _record_maxima(f"heavily_pushed:{precision}:{backend}", dict(u0=np.abs(out["u0"] - u_ref)[ok].max() / max(1.0, np.abs(u_ref[ok]).max()),
                                                                 f0=np.abs(out["f0"] - f_ref)[ok].max(), vel=0.0, com_rel=0.0, sample=int(ok.sum())))
    assert np.abs(out["u0"] - u_ref)[ok].max() <= tol_u * max(1.0, np.abs(u_ref[ok]).max())
    assert np.abs(out["f0"] - f_ref)[ok].max() <= (1e-7 if precision == "f64" else 2e-5)
    assert ((out["active"] & 0xffff) >= w["C"]).any() or ((out["active"] >> 16) >= w["C"]).any()     # saturated horizons are in the set


def test_fp32_pinned_horizon_is_resolved_in_fp64(FA, monkeypatch):
    """A working set that pins (nearly) the whole horizon is beyond the fp32 block solve: its check fails.  Such a QP is not
    started cold (100-150 one-row steps: one of them makes a launch of 16 384 instances 2-3x longer) but handed to the fp64
    instantiation in a small launch (up to 64 workgroups) behind the fp32 one.  Workload: the bench generator at 1.5x its push, stream 1
    (holds one such QP in 32 768); checked: same flags and next state as the fp64 solve, and the worst QP stays short."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    w = workload.make_batch_a("walk_C150", 16384, stream=1, push_scale=1.5)
    g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
    p = FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"])
    d_push = torch.from_numpy(w["push"].copy()).to("cuda:0")
    monkeypatch.delenv("ISMPC_A_F32_RESOLVE", raising=False)
    res = {}
    for key, prec, env in (("f64", "f64", None), ("f32", "f32", None), ("f32_cold", "f32", "0")):
        if env is not None: monkeypatch.setenv("ISMPC_A_F32_RESOLVE", env)
        gen = FA.GaitGenerator(p, ce, precision=prec)
        st = q_to_dev(w["state"])
        res[key] = (q_from_dev(gen.tick_torch(st, d_push), FA.OUT_A), q_from_dev(st, FA.STATE_A))
        gen.close()
    o64, s64 = res["f64"]; o32, s32 = res["f32"]; oc, _ = res["f32_cold"]
    assert (o32["status"] == o64["status"]).all() and (o64["status"] == 0).all()
    worst = lambda o: int(np.maximum(o["iters_x"], o["iters_y"]).max())
    assert worst(oc) > 100                                   # the QP is in this batch, and cold it is this expensive
    assert worst(o32) <= 40 and worst(o64) <= 40
    for k in ("x", "y"):
        assert np.abs(s32[k] - s64[k]).max() <= 1e-7 * np.abs(s64[k]).max()
        assert np.abs(s32[k + "d"] - s64[k + "d"]).max() <= 5e-6
    assert (s32["fc"] == s64["fc"]).all() and (s32["j"] == s64["j"]).all()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("name,stream,scale", [("walk_C150", 1, 1.5), ("trot_C160", 2, 2.0), ("walk_C100", 3, 2.5)])
def test_other_streams_and_harder_pushes_against_oracle(FA, name, stream, scale, precision):
    """The route of the block warm start was tuned on the bench workloads (stream 0, push scale 1).  Other random streams and
    1.5-2.5x the push: a sample of each batch against the oracle (reference qpOASES where built), QP by QP -- the same QPs are
    infeasible, and the feasible ones return the oracle's u0 / f0 / next velocity."""
    import torch
    from oracle import oracle_a as A
    from oracle import oracle as O
    from quadruped_gait_generation_ismpc_amd import workload
    B = 4096
    backend = "ref" if O.have_ref() else "gi"
    w = workload.make_batch_a(name, B, stream=stream, push_scale=scale)
    g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
    gen = FA.GaitGenerator(FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"]), ce, precision=precision)
    d = q_to_dev(w["state"])
    out = q_from_dev(gen.tick_torch(d, torch.from_numpy(w["push"].copy()).to("cuda:0")), FA.OUT_A)
    inf = FA.ST_X_INFEASIBLE | FA.ST_Y_INFEASIBLE
    assert ((out["status"] & ~inf) == 0).all()                                    # nothing unverified, nothing else flagged
    tol_u0 = TOL_U0[backend] if precision == "f64" else 2e-3
    # the hardest-pushed instances of the batch and a random sample
    hard = np.argsort(-np.abs(w["push"]).max(1))[:16]
    pick = np.unique(np.concatenate([hard, np.random.default_rng(5).choice(B, 32, replace=False)]))
    for i in pick:
        sim = A.SimA(A.gait(w["kind"], w["phi"], w["disp_A"]), A.params(w["kind"], C_=w["C"], P=w["P"], F=w["F"]), backend=backend)
        sim.load_product_state(w["state"][i])
        r = sim.tick(tuple(w["push"][i]))
        for ax, bit in ((0, FA.ST_X_INFEASIBLE), (1, FA.ST_Y_INFEASIBLE)):
            assert (r["rv"][ax] != 0) == bool(out["status"][i] & bit), (i, ax, r["rv"], out["status"][i])
            if r["rv"][ax] == 0:
                assert abs(out["u0"][i, ax] - r["u0"][ax]) <= tol_u0 * max(1.0, abs(r["u0"][ax])), (i, ax, out["u0"][i], r["u0"])
                assert abs(out["f0"][i, ax] - r["f0"][ax]) <= (1e-7 if precision == "f64" else 2e-5), (i, ax)
                assert abs(out["vel_after"][i, ax] - r["vel_after"][ax]) <= (1e-6 if precision == "f64" else 5e-6), (i, ax)
    gen.close()


def test_entry_points_run_on_the_handles_device_whatever_is_current(FA):
    """Every entry point runs on the handle's device and leaves the caller's current device as it found it -- also the swing-foot
    launches behind a tick (round 2 ran those on whatever device was current).  Needs two GPUs: skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the pool leases one)")
    g = FA.default_gait(FA.WALK, np.pi / 4, 0.1)
    fp0, ce = FA.plan(g)
    torch.cuda.set_device(1)
    gen = FA.GaitGenerator(FA.default_params(FA.WALK), ce, device=0)          # created while device 1 is current
    assert torch.cuda.current_device() == 1
    st = q_to_dev(gen.initial_state(g.disp_C, batch=2))                      # cuda:0
    feet = gen.feet_init_torch(g, fp0, batch=2, device="cuda:0")
    with torch.cuda.device(0):
        stream0 = torch.cuda.current_stream().cuda_stream
    traj = torch.empty((30, 2, 80), dtype=torch.uint8, device="cuda:0")
    import ctypes as C
    rc = FA._l().ismpc_a_rollout_feet_device(gen._h, 2, C.c_void_p(st.data_ptr()), 30, C.c_void_p(traj.data_ptr()), C.c_void_p(feet.data_ptr()),
                                            C.c_void_p(stream0) if stream0 else None)
    assert rc == 0 and torch.cuda.current_device() == 1
    torch.cuda.synchronize(0)
    out = q_from_dev(traj, FA.OUT_A)
    assert (out["status"] == 0).all()
    torch.cuda.set_device(0)
    ref = FA.GaitGenerator(FA.default_params(FA.WALK), ce, device=0)
    st2 = q_to_dev(ref.initial_state(g.disp_C, batch=2)); feet2 = ref.feet_init_torch(g, fp0, batch=2)
    t2 = ref.rollout_feet_torch(st2, feet2, 30); torch.cuda.synchronize()
    assert torch.equal(t2, traj) and torch.equal(feet2, feet)


def test_scratch_growth_across_streams(FA):
    """As for Formulation B: the handle's copy of the previous state and the work lists grow inside an asynchronous entry point; when the
    call's stream is not the previous call's, that stream is drained before the old block is released."""
    import torch
    from quadruped_gait_generation_ismpc_amd import workload
    w = workload.make_batch_a("walk_C100", 20000, seed=3)
    g = FA.default_gait(w["kind"], w["phi"], w["disp_A"]); _, ce = FA.plan(g)
    p = FA.default_params(w["kind"], C=w["C"], P=w["P"], F=w["F"])
    gen, ref = FA.GaitGenerator(p, ce, precision="f32"), FA.GaitGenerator(p, ce, precision="f32")
    push = torch.from_numpy(w["push"].copy()).to("cuda:0")
    r_small = q_from_dev(ref.tick_torch(q_to_dev(w["state"][:1500]), push[:1500].contiguous()), FA.OUT_A)
    r_big = q_from_dev(ref.tick_torch(q_to_dev(w["state"]), push), FA.OUT_A)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    st_small = [q_to_dev(w["state"][:1500]) for _ in range(40)]; st_big = q_to_dev(w["state"]); ps = push[:1500].contiguous()
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        for st in st_small:
            o_small = gen.tick_torch(st, ps)
    with torch.cuda.stream(sb):
        o_big = gen.tick_torch(st_big, push)
    torch.cuda.synchronize()
    a, b = q_from_dev(o_small, FA.OUT_A), q_from_dev(o_big, FA.OUT_A)
    for k in ("status", "u0", "f0", "vel_after", "com_before"):
        assert np.array_equal(a[k], r_small[k]) and np.array_equal(b[k], r_big[k]), k
    gen.close(); ref.close()
