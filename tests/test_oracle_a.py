"""CPU-only: the Formulation-A oracle (oracle/ismpc_oracle_a.c = the MATLAB generators restated in C)
against the reference's OWN known-answer data: the checked-in MATLAB trajectories
AMR_code_DART/MATLAB_trajectories/**/ComTrajectory_*.txt / ComVelocity_*.txt (committed as
tests/golden/formA_matlab_*.npz by tests/golden/make_golden_a.py).  This is the pin of the oracle.

Tolerances: the files hold 7 significant digits; MATLAB quadprog (interior point) is
less exact than an active-set solve, most visibly with Q_f = 1e9 (walk): trot 3e-6 m, walk 5e-5 m on
CoM, 1e-4 m/s on velocity; the first 20 ticks (no inequality active) agree to print rounding.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import oracle_a as A

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLDEN, "formA_matlab_meta.json")))
TOL_COM = {"trot": 3e-6, "walk": 5e-5}
# ticks replayed per fixture in test_matlab_fixture_replay (the whole files: test_trot_15cm_file_is_two_appended_runs and
# test_foot_files_replay below, and every file on the device in tests/test_gpu_formulation_a.py)
TICKS = {"walk_phi0": 700, "walk_phipi4": 700, "walk_phipi2": 700, "trot_phi0": 330, "trot_phipi4": 330,
         "trot_phipi4_15": 330, "trot_phipi2": 330}


def replay(name, ticks, backend="gi"):
    m = META[name]
    kind = A.WALK if m["gait"] == "walk" else A.TROT
    sim = A.SimA(A.gait(kind, m["phi"], m["disp_A"]), A.params(kind), backend=backend)
    outs = sim.run(ticks)
    z = np.load(os.path.join(GOLDEN, f"formA_matlab_{name}.npz"))
    return outs, z, m


@pytest.mark.parametrize("name", sorted(META))
def test_matlab_fixture_replay(name, built_libs):
    ticks = TICKS[name]
    outs, z, m = replay(name, ticks)
    assert (outs["rv"] == 0).all()
    com = z["com"][:ticks]
    assert np.all(com[:, 2] == 0.56)                                   # third column = height
    err = np.abs(outs["com_before"] - com[:, :2])
    assert err[:20].max() <= 6e-8 * max(1.0, np.abs(com[:20, :2]).max())   # print rounding of %e with 7 digits
    assert err.max() <= TOL_COM[m["gait"]], err.max()
    if m["has_velocity"]:
        verr = np.abs(outs["vel_after"] - z["vel"][:ticks, :2])
        assert verr.max() <= 1e-4, verr.max()
    # the gait actually steps: footstep counter advances every `step` ticks (quad_walk_no_plots.m:522)
    step = 50 if m["gait"] == "walk" else 80
    assert np.array_equal(np.where(outs["stepped"] == 1)[0] + 2, np.arange(1, ticks // step + 1) * step)


def test_trot_15cm_file_is_two_appended_runs(built_libs):
    """trotting/phipi4/15cm holds 3 200 rows because the script opens every output file in append mode
    (trotting/quad_as_bip_no_plots.m:105-114, fopen(..., 'a+')): it is a 1 200-tick run followed by a fresh 2 000-tick run from
    the same initial state.  Both runs are known answers for ONE 2 000-tick replay: rows 0..1199 against its first 1 200
    ticks, rows 1200..3199 against all of it."""
    outs, z, m = replay("trot_phipi4_15", 2000)
    com, vel = z["com"], z["vel"]
    assert com.shape == (3200, 3) and np.array_equal(com[1200], com[0]) and np.array_equal(com[1200:2400], com[:1200])
    assert (outs["rv"] == 0).all()
    for rows, n in ((slice(0, 1200), 1200), (slice(1200, 3200), 2000)):
        assert np.abs(outs["com_before"][:n] - com[rows, :2]).max() <= TOL_COM["trot"]
        assert np.abs(outs["vel_after"][:n] - vel[rows, :2]).max() <= 1e-4


@pytest.mark.parametrize("name", ["walk_phipi4", "trot_phipi4"])
def test_matlab_fixture_with_reference_qpoases(name, built_libs):
    """Same replay with every QP solved by the reference's vendored qpOASES (oracle/_ref)."""
    if not O.have_ref():
        pytest.skip("oracle/_ref not built in this environment")
    ticks = 160
    outs, z, m = replay(name, ticks, backend="ref")
    gi, _, _ = replay(name, ticks, backend="gi")
    assert (outs["rv"] == 0).all()
    assert np.abs(outs["com_before"] - z["com"][:ticks, :2]).max() <= TOL_COM[m["gait"]]
    assert np.abs(outs["com_before"] - gi["com_before"]).max() <= 1e-7      # two solvers, one minimiser
    assert np.abs(outs["u0"] - gi["u0"]).max() <= 1e-4                      # qpOASES stops at 2.2e-7 homotopy length; Q_f = 1e7 scales it up


def test_plan_generators(built_libs):
    """init_quadruped.m / init_quadruped2.m: structural facts of the plans."""
    fp, ce = A.plan(A.gait(A.WALK, np.pi / 4, 0.1))
    assert fp.shape == (101, 8) and ce.shape == (100, 2)              # the walk loop writes row 101
    assert np.all(ce[96:] == 0.0)                                      # rows 97..100 stay (0,0)
    assert np.allclose(ce[0], [0.44, 0.0]) and np.array_equal(ce[1], ce[0])
    sx, sy = 0.1 * np.cos(np.pi / 4), 0.1 * np.sin(np.pi / 4)
    # 8-row cycle from row 6: FR moves at +1, BL at +3, FL at +5, BR at +7
    j = 6 - 1
    assert np.allclose(fp[j + 1, 4:6] - fp[j, 4:6], [sx, sy]) and np.allclose(fp[j + 3, 0:2] - fp[j + 2, 0:2], [sx, sy])
    assert np.allclose(fp[j + 5, 6:8] - fp[j + 4, 6:8], [sx, sy]) and np.allclose(fp[j + 7, 2:4] - fp[j + 6, 2:4], [sx, sy])
    fp, ce = A.plan(A.gait(A.TROT, 0.0, 0.15))
    assert fp.shape == (100, 8)
    assert np.allclose(ce[0], [0.44, 0.0])
    # diagonal pairs alternate; first step is half length
    assert np.isclose(fp[1, 0], 0.075) and np.isclose(fp[1, 4], 0.88 + 0.075)
    assert np.isclose(fp[3, 0] - fp[2, 0], 0.15) and np.isclose(fp[2, 2] - fp[1, 2], 0.15)
    # centre = intersection of the diagonals: lies on both
    k = 10
    d1 = np.cross(np.r_[fp[k, 4:6] - fp[k, 0:2], 0], np.r_[ce[k] - fp[k, 0:2], 0])[2]
    d2 = np.cross(np.r_[fp[k, 6:8] - fp[k, 2:4], 0], np.r_[ce[k] - fp[k, 2:4], 0])[2]
    assert abs(d1) < 1e-12 and abs(d2) < 1e-12


def test_mapping_overflow_is_reported(built_libs):
    """C=150, step=50, F=3: the horizon spans more than F-1 step boundaries; the .m file dies on a
    dimension mismatch there (SURVEY.md 'Index limits'); the oracle reports it instead."""
    sim = A.SimA(A.gait(A.WALK, 0.0, 0.1), A.params(A.WALK, C_=150, P=300), backend="gi")
    rvs = [int(sim.tick()["rv"][0]) for _ in range(25)]
    assert rvs[:19] == [0] * 19 and rvs[19] == -2      # tick 20: sample 150 falls in the 4th double support
    sim = A.SimA(A.gait(A.WALK, 0.0, 0.1), A.params(A.WALK, C_=150, P=300, F=4), backend="gi")
    assert all(int(sim.tick()["rv"][0]) == 0 for _ in range(60))


FOOT_TOL = {"trot": 3e-6, "walk": 5e-5}


@pytest.mark.parametrize("name", sorted(META))
def test_foot_files_replay(name, built_libs):
    """the swing-foot re-placement QPs + foot trajectory writer of the scripts, against every checked-in
    foot_{fl,fr,rl,rr}_*.txt (2000 rows each; trotting/phi0 holds two of the four files, trotting/phipi4/15cm the feet of
    its 2 000-tick run)."""
    m = META[name]
    kind = A.WALK if m["gait"] == "walk" else A.TROT
    sim = A.SimA(A.gait(kind, m["phi"], m["disp_A"]), A.params(kind), backend="gi")
    sim.enable_feet()
    outs = sim.run(2000)
    assert (outs["rv"] == 0).all()
    tr = sim.foot_trajectories(2000)
    z = np.load(os.path.join(GOLDEN, f"formA_matlab_{name}.npz"))
    have = [ft for ft in ("fl", "fr", "rl", "rr") if f"foot_{ft}" in z.files]
    assert len(have) == (2 if name == "trot_phi0" else 4)
    for k, ft in enumerate(("fl", "fr", "rl", "rr")):
        if ft not in have:
            continue
        fix = z[f"foot_{ft}"]
        assert fix.shape == (2000, 3)
        assert np.abs(tr[k] - fix).max() <= FOOT_TOL[m["gait"]], (ft, np.abs(tr[k] - fix).max())
        assert np.abs(tr[k][:, 2] - fix[:, 2]).max() <= 1e-9          # swing height parabola -3.2e-5 k^2 + 1.6e-3 k


def test_product_foot_writer_and_wire_format(built_libs, tmp_path):
    """Host side of the product (no GPU needed): ismpc_a_foot_trajectories == the oracle's writer on the same
    foot_plan, and ismpc_a_write_trajectory_txt prints what MATLAB's fprintf('%d %d %d\\n') prints."""
    from quadruped_gait_generation_ismpc_amd import formulation_a as FA
    for kind, phi, dA, step in ((A.TROT, np.pi / 4, 0.1, 80), (A.WALK, np.pi / 4, 0.1, 50)):
        sim = A.SimA(A.gait(kind, phi, dA), A.params(kind), backend="gi")
        sim.enable_feet(); sim.run(400)
        fp = sim.foot_plan()
        mine = FA.foot_trajectories(FA.default_gait(kind, phi, dA), step, fp, 400)
        assert np.array_equal(mine, sim.foot_trajectories(400))
    rows = np.array([[0.88, 0.259394, 0.0], [7.070705e-04, 2.601011e-01, 1.568e-03], [0.44, 0.0, 0.56], [-3.0, 1e-7, 2.0]])
    path = tmp_path / "t.txt"
    FA.write_trajectory_txt(str(path), rows)
    assert path.read_text().splitlines() == ["8.800000e-01 2.593940e-01 0", "7.070705e-04 2.601011e-01 1.568000e-03",
                                             "4.400000e-01 0 5.600000e-01", "-3 1.000000e-07 2"]
    back = np.loadtxt(path)
    assert np.abs(back - rows).max() <= 5e-7 * np.abs(rows).max()
