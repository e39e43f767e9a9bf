#!/usr/bin/env python3
"""Numpy check of the CLOSED FORM of the second Goldfarb-Idnani step of the Formulation-A solver (working set = stability row + ONE ZMP
row j; a second ZMP row `row` is about to enter; no kinematic row) against the general structured solve of proto_structured.py
(Solver.solve_new, itself checked against a dense solve).  The kernel's fast path (csrc/ismpc_a_wave.hpp, ISMPC_A_SECOND_STEP) evaluates
exactly these expressions:

    v = V_j = [M~_j, dt PA_j],  g = 1 / (j dt^2),  K = diag(I_F, -a'a) + g v v'        (Sherman-Morrison)
    kappa = sg w + g (M~_j . mt),  w = row / j (row < j) or 1 (row > j),  mt = sg M~_row
    beta = |M~_j|^2 - vE^2 / a'a,   eta = kappa beta + vE dX / a'a,   alpha = kappa - g eta / (1 + g beta),   dX = sg dt PA_row
    c[:F] = alpha M~_j,   cE = (dX - alpha vE) / a'a
    rho_j = (M~_j . mt - alpha beta - vE dX / a'a) / (j dt^2) + sg w
    d.r = sg (dt^2 min(row, j) + M_row . M_j / Qf) rho_j + dX cE
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proto_structured import Solver, A


def closed_form(S, j, row, sg):
    dt, Qf, aa, F = S.dt, S.Qf, S.aa, S.F
    v1 = S.Mt(j); vE = dt * S.PA[j]
    g = 1.0 / (j * dt * dt)
    mt = sg * S.Mt(row)
    w = row / j if row < j else 1.0
    dX = sg * dt * S.PA[row]
    kappa = sg * w + g * (v1 @ mt)
    beta = v1 @ v1 - vE * vE / aa
    eta = kappa * beta + vE * dX / aa
    alpha = kappa - g * eta / (1.0 + g * beta)
    c1 = alpha * v1
    cE = (dX - alpha * vE) / aa
    rho = (v1 @ mt - alpha * beta - vE * dX / aa) * g + sg * w
    dZ = sg * (dt * dt * min(row, j) + (S.M[row - 1] @ S.M[j - 1]) / Qf)
    return rho, cE, dZ * rho + dX * cE, c1


if __name__ == "__main__":
    worst = 0.0; n = 0
    for kind, Cn, Fn in ((A.WALK, 100, 3), (A.TROT, 160, 3), (A.WALK, 150, 4)):
        sim = A.SimA(A.gait(kind, np.pi / 4, 0.1), A.params(kind, C_=Cn, P=2 * Cn, F=Fn), backend="gi"); p = sim.p
        sim.run(37)
        rng = np.random.default_rng(5)
        for axis in (0, 1):
            D = sim.axis_data(axis)
            S = Solver(D, p.dt, p.Qf)
            for _ in range(400):
                j, row = rng.choice(np.arange(1, S.C + 1), 2, replace=False)
                sj, sg = rng.choice([-1.0, 1.0], 2)
                rho, rX, dd, c1, dZ, dX = S.solve_new(np.array([j]), [sj], [(0, 1.0)], int(row), sg)
                r2, cE2, dd2, c12 = closed_form(S, int(j), int(row), sg)
                e = max(abs(rho[0] - r2) / max(1e-12, abs(rho[0])), abs(rX[0] - cE2) / max(1e-12, abs(rX[0])), abs(dd - dd2) / max(1e-12, abs(dd)),
                        np.abs(c1 - c12).max() / max(1e-12, np.abs(c1).max()))
                worst = max(worst, e); n += 1
    print("cases", n, "worst relative difference to the general structured solve %.2e" % worst)
    assert worst < 1e-8
