#!/usr/bin/env python3
"""numpy check of the STRUCTURED block solve the wave kernel's warm start uses: given a working set of ZMP rows (with
signs), the minimiser and its multipliers from (a) one pass over the active rows accumulating G = V'K^-1 V and
g = V'K^-1 c (K^-1 tridiagonal: only gaps between consecutive active rows), (b) an (F+1)x(F+1) solve, (c) a tridiagonal
apply and one suffix sum.  Compared with a dense KKT solve (proto_pdas.solve_on)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle_a as A
import proto_pdas as PP


def block_solve(D, dt, Qf, W):
    """W: dict ZMP row (1-based) -> +1 / -1.  Returns u, f, lam (signed, per active row), lamE."""
    a, b = D["a"], D["b"]; C = len(a); F = len(D["pref"])
    M = D["M"][:, 1:]; pref = D["pref"]; sq = np.sqrt(Qf); isq = 1 / sq
    PA = np.concatenate([[0.0], np.cumsum(a)])           # PA[i] = sum_{k<=i} a_k  (1-based i)
    aa = a @ a
    m = F + 1                                            # unknowns: w (F), lamE   (kinematic rows inactive here)
    rows = sorted(W)
    def V(i): return np.concatenate([M[i - 1] * isq, [dt * PA[i]]])
    def cval(i): return (D["zlo"][i - 1] if W[i] > 0 else D["zhi"][i - 1]) + M[i - 1] @ pref
    G = np.zeros((m, m)); g = np.zeros(m)
    vprev = np.zeros(m); cprev = 0.0; iprev = 0
    for i in rows:
        v = V(i); c = cval(i); gap = i - iprev
        d = v - vprev
        G += np.outer(d, d) / (gap * dt * dt); g += d * (c - cprev) / (gap * dt * dt)
        vprev, cprev, iprev = v, c, i
    T = G.copy()
    T[:F, :F] += np.eye(F); T[F, F] -= aa
    rhs = g.copy(); rhs[F] -= b
    y = np.linalg.solve(T, rhs)
    w, lamE = y[:F], y[F]
    s = {i: cval(i) - V(i) @ y for i in rows}
    lam = {}
    for k, i in enumerate(rows):
        ip = rows[k - 1] if k > 0 else 0; sp = s[ip] if k > 0 else 0.0
        r = (s[i] - sp) / (i - ip)
        if k + 1 < len(rows): r -= (s[rows[k + 1]] - s[i]) / (rows[k + 1] - i)
        lam[i] = r / (dt * dt)
    lv = np.zeros(C)
    for i in rows: lv[i - 1] = lam[i]
    u = dt * np.cumsum(lv[::-1])[::-1] + a * lamE
    f = pref - w * isq
    return u, f, lam, lamE


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "walk_C100"
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", f"prerollA_{name}.npz"))
    kind = int(z["gait"]); p = A.params(kind, C_=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
    sim = A.SimA(A.gait(kind, float(z["phi"]), float(z["disp_A"])), p, backend="gi")
    rng = np.random.default_rng(1)
    sim.run(40)
    worst = 0.0
    for t in range(10):
        sim.run(int(rng.integers(1, 12)))
        for axis in (0, 1):
            D = sim.axis_data(axis)
            Q = PP.build(D, p.dt, p.Qf)
            C = p.C
            nact = int(rng.integers(1, 60))
            W0 = {int(r): int(rng.choice([-1, 1])) for r in rng.choice(C, nact, replace=False)}          # 0-based dense rows
            x, mu = PP.solve_on(*Q, W0)
            u, f, lam, lamE = block_solve(D, p.dt, p.Qf, {r + 1: s for r, s in W0.items()})
            err = max(np.abs(u - x[:C]).max(), np.abs(f - x[C:]).max())
            # multipliers: dense mu (>= 0 convention) vs sign * lam
            em = max(abs(mu[r] - W0[r] * lam[r + 1]) / max(1.0, abs(mu[r])) for r in W0)
            worst = max(worst, err, em)
            print(f"t={t} axis={axis} active={nact} |x err|={err:.2e} mu rel err={em:.2e}")
    print("worst", worst)
