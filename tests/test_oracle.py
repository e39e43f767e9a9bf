"""CPU-only: the oracle against the committed golden vectors, the reference's qpOASES build
(when oracle/_ref is present) and first-principles QP optimality conditions."""
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HORIZONS = (50, 100, 150, 200)


def _load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return z, z["tick_in"].view(O.TICK_IN).reshape(-1), z["tick_out"].view(O.TICK_OUT).reshape(-1)


def rel_com(a, b):
    return np.abs(a["com_pos"] - b["com_pos"]).max(1) / np.maximum(np.abs(b["com_pos"]).max(1), 1e-3)


@pytest.mark.parametrize("N", HORIZONS)
@pytest.mark.parametrize("backend", ["gi", "ref"])
def test_oracle_reproduces_golden_vectors(N, backend, built_libs):
    if backend == "ref" and not O.have_ref():
        pytest.skip("oracle/_ref not built in this environment")
    z, tin, ref = _load(f"formB_vectors_N{N}.npz")
    out, info, traj = O.Oracle(O.default_params(N), backend=backend).solve(tin, want_traj=True)
    ok = (ref["status"] & O.ST_ERROR_MASK) == 0
    # the golden vectors were produced with the reference's qpOASES: the restatement + either QP
    # backend must land on the same unique minimiser (qpOASES stops at 2.2e-7 homotopy length)
    assert rel_com(out, ref)[ok].max() <= 1e-6
    assert np.abs(out["com_vel"] - ref["com_vel"])[ok].max() <= 1e-6
    assert (np.abs(out["u0"] - ref["u0"])[ok] <= 1e-6 * np.maximum(1.0, np.abs(ref["u0"][ok]))).all()
    assert (out["status"][ok] == ref["status"][ok]).all()
    assert np.abs(traj - z["u_traj"])[ok].max() <= 2e-5


def test_preroll_known_answers(built_libs):
    """Numbers SURVEY.md (3.6, A.4) lists for the nominal closed loop at N = 100."""
    z, tin, out = _load("preroll_N100.npz")
    assert (z["rv"] <= 0).all() and z["nwsr"].max() == 0        # nominal QPs never touch an inequality
    np.testing.assert_allclose(out["u0"][0], [490.50, -0.0187, -0.0237], atol=5e-4)
    assert tin["footstep_counter"][0] == 1 and tin["footstep_counter"][45] == 2 and tin["mpc_iter"][45] == 0
    np.testing.assert_allclose(out["u0"][45][0], 467.55, atol=5e-3)
    assert tin["footstep_counter"][100] == 3 and tin["mpc_iter"][100] == 10
    np.testing.assert_allclose(out["com_pos"][100], [0.18551, -0.02163, 0.67906], atol=5e-6)
    np.testing.assert_allclose(out["u0"][100][1], 0.2000, atol=5e-5)
    assert tin["footstep_counter"][399] == 9 and tin["mpc_iter"][399] == 39
    assert out["status"][399] & O.ST_FLIGHT and np.all(out["u0"][399] == 0.0)
    zc = out["com_pos"][:700, 2]
    assert 0.67 < zc.min() and zc.max() < 0.72
    flight = (out["status"][90:] & O.ST_FLIGHT) != 0
    assert abs(flight.mean() - 10 / 45) < 0.02                  # F of every S+F ticks coast


def test_config1_kat(built_libs):
    z, tin, ref = _load("formB_kat_config1.npz")
    np.testing.assert_allclose(ref["u0"][0], [490.50, -0.0070, -0.0262], atol=5e-4)
    out, info = O.Oracle(O.default_params(50), backend="gi").solve(tin)
    assert rel_com(out, ref).max() <= 1e-6 and (out["status"] == ref["status"]).all()


def _random_qp(rng, n, ne, ni):
    M = rng.standard_normal((n, n))
    H = M @ M.T + n * np.eye(n) * rng.uniform(0.01, 1.0)
    g = rng.standard_normal(n) * 3
    x0 = rng.standard_normal(n)
    A = rng.standard_normal((ne + ni, n))
    lb = np.empty(ne + ni); ub = np.empty(ne + ni)
    lb[:ne] = ub[:ne] = A[:ne] @ x0
    c = A[ne:] @ x0
    lb[ne:] = c - rng.uniform(0.0, 0.5, ni); ub[ne:] = c + rng.uniform(0.0, 0.5, ni)
    return H, g, A, lb, ub


@pytest.mark.parametrize("seed", range(6))
def test_gi_solver_kkt(seed, built_libs):
    rng = np.random.default_rng(seed)
    n, ne, ni = 24, 3, 40
    H, g, A, lb, ub = _random_qp(rng, n, ne, ni)
    x, rv, it = O.solve_qp(H, g, A, lb, ub, backend="gi")
    assert rv == 0
    r = A @ x
    assert (r >= lb - 1e-9).all() and (r <= ub + 1e-9).all()
    # stationarity with multipliers supported on the active rows, right signs on inequalities
    act = np.where((np.abs(r - lb) < 1e-8) | (np.abs(r - ub) < 1e-8))[0]
    grad = H @ x + g
    mu, *_ = np.linalg.lstsq(A[act].T, -grad, rcond=None)
    assert np.abs(A[act].T @ mu + grad).max() < 1e-7
    for k, m in zip(act, mu):
        if k < ne:
            continue
        if abs(r[k] - lb[k]) < 1e-8 and abs(r[k] - ub[k]) > 1e-8:
            assert m <= 1e-8           # -grad = A' mu, lower bound active -> mu <= 0 in this sign convention
        if abs(r[k] - ub[k]) < 1e-8 and abs(r[k] - lb[k]) > 1e-8:
            assert m >= -1e-8


@pytest.mark.parametrize("seed", range(4))
def test_gi_matches_reference_qpoases(seed, built_libs):
    if not O.have_ref():
        pytest.skip("oracle/_ref not built in this environment")
    rng = np.random.default_rng(100 + seed)
    H, g, A, lb, ub = _random_qp(rng, 30, 2, 45)
    x1, rv1, _ = O.solve_qp(H, g, A, lb, ub, backend="gi")
    x2, rv2, _ = O.solve_qp(H, g, A, lb, ub, backend="ref")
    assert rv1 == 0 and rv2 == 0
    assert np.abs(x1 - x2).max() <= 1e-6 * max(1.0, np.abs(x2).max())


def test_infeasible_detected_by_both(built_libs):
    n = 5
    H = np.eye(n); g = np.zeros(n)
    A = np.vstack([np.ones((1, n)), np.eye(n)])
    lb = np.r_[10.0, -np.ones(n)]; ub = np.r_[10.0, np.ones(n)]
    x, rv, _ = O.solve_qp(H, g, A, lb, ub, backend="gi")
    assert rv == 37
    if O.have_ref():
        assert O.solve_qp(H, g, A, lb, ub, backend="ref")[1] == 37


def test_passthrough_rules(built_libs):
    orc = O.Oracle(O.default_params(100), backend="gi")
    st = O.initial_state(); st["footstep_counter"] = 1
    bad = st.copy(); bad["simulation_time"] = 1700.0         # window [idx, idx+2N) leaves the plan
    out, _ = orc.solve(bad)
    assert out["status"][0] == O.ST_BAD_INDEX and np.all(out["com_pos"] == bad["com_pos"])
    neg = st.copy(); neg["mpc_iter"] = -1
    assert orc.solve(neg)[0]["status"][0] == O.ST_BAD_INDEX


def test_vertical_equalities_zero_force(built_libs):
    """u_i = 0 on the flight samples of MPCSolver.cpp:223-243 once footstepCounter > 1."""
    orc = O.Oracle(O.default_params(100), backend="gi")
    st = O.initial_state(); st["footstep_counter"] = 3; st["simulation_time"] = 100
    for it, rng_ in ((10, range(25, 35)), (40, range(0, 5))):
        st["mpc_iter"] = it; st["control_iter"] = it
        out, info, traj = orc.solve(st, want_traj=True)
        assert info["ne_z"][0] == len(rng_)
        assert np.abs(traj[0, 0, list(rng_)]).max() < 1e-9
        others = np.setdiff1d(np.arange(100), list(rng_))
        assert np.abs(traj[0, 0, others]).min() > 1.0
