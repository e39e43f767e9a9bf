#!/usr/bin/env python3
"""Prototype (dense numpy): how many primal-dual active-set passes ("add every violated row at once, drop every row
whose multiplier went negative") does the Formulation-A QP need before the Goldfarb-Idnani loop only has a few rows left
to fix?  Decides whether a block warm start is worth building into the wave kernel."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_a as A

def build(D, dt, Qf):
    a, b = D["a"], D["b"]; C = len(a); F = len(D["pref"]); n = C + F
    H = np.concatenate([np.ones(C), Qf * np.ones(F)])
    g = np.concatenate([np.zeros(C), -Qf * D["pref"]])
    E = np.concatenate([a, np.zeros(F)])
    N = np.zeros((C + F, n))
    N[:C, :C] = dt * np.tril(np.ones((C, C)))
    N[:C, C:] = -D["M"][:, 1:]
    for r in range(F):
        N[C + r, C + r] = 1.0
        if r >= 1: N[C + r, C + r - 1] = -1.0
    lo = np.concatenate([D["zlo"], D["klo"]]); hi = np.concatenate([D["zhi"], D["khi"]])
    return H, g, E, b, N, lo, hi

def solve_on(H, g, E, b, N, lo, hi, W):
    """W: dict row -> +1 (lower active) / -1 (upper).  Returns x, mu (>=0 means correctly signed)."""
    rows = sorted(W)
    A_ = np.vstack([E] + [N[r] for r in rows]); rhs = np.array([b] + [lo[r] if W[r] > 0 else hi[r] for r in rows])
    Hi = 1.0 / H
    S = (A_ * Hi) @ A_.T
    x0 = -Hi * g
    lam = np.linalg.solve(S, A_ @ x0 - rhs)
    x = x0 - Hi * (A_.T @ lam)
    mu = {r: (-lam[k + 1] if W[r] > 0 else lam[k + 1]) for k, r in enumerate(rows)}     # lower: N x >= lo -> multiplier -lam
    return x, mu

def pdas(H, g, E, b, N, lo, hi, passes):
    W = {}; x, mu = solve_on(H, g, E, b, N, lo, hi, W)
    hist = []
    for p in range(passes):
        cv = N @ x
        tol = 1e-11 * (np.abs(cv) + np.maximum(np.abs(lo), np.abs(hi))) + 1e-13
        new = {r: s for r, s in W.items() if mu[r] > 0}
        for r in np.nonzero(cv < lo - tol)[0]: new.setdefault(int(r), +1)
        for r in np.nonzero(cv > hi + tol)[0]: new.setdefault(int(r), -1)
        if new == W: hist.append(len(W)); break
        W = new; x, mu = solve_on(H, g, E, b, N, lo, hi, W); hist.append(len(W))
    cv = N @ x
    tol = 1e-9
    nviol = int(((cv < lo - tol) | (cv > hi + tol)).sum()); nneg = sum(1 for r in W if mu[r] < -1e-12)
    return x, W, mu, hist, nviol, nneg

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "walk_C100"
    ntest = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"prerollA_{name}.npz"))
    kind = int(z["gait"]); p = A.params(kind, C_=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
    sim = A.SimA(A.gait(kind, float(z["phi"]), float(z["disp_A"])), p, backend="gi")
    rng = np.random.default_rng(0)
    tot = []
    pre = sim.run(int(rng.integers(5, 150)))
    for t in range(ntest):
        sim.run(int(rng.integers(1, 12)))
        st = sim.state.copy()
        push = (rng.uniform(-0.03, 0.03), rng.uniform(-0.05, 0.05))
        st2 = st.copy(); st2["xd"] += push[0]; st2["yd"] += push[1]
        sim.state = st2
        for axis in (0, 1):
            D = sim.axis_data(axis)
            H, g, E, b, N, lo, hi = build(D, p.dt, p.Qf)
            for passes in (12,):
                x, W, mu, hist, nviol, nneg = pdas(H, g, E, b, N, lo, hi, passes)
                tot.append((len(hist), len(W), nviol, nneg))
                print(f"t={t} axis={axis} sets={hist} viol={nviol} neg={nneg}")
        sim.state = st
    tot = np.array(tot)
    print("mean passes", tot[:, 0].mean(), "max", tot[:, 0].max(), "unconverged", int(((tot[:, 2] > 0) | (tot[:, 3] > 0)).sum()), "of", len(tot))
