#!/usr/bin/env python3
"""Prototype (dense numpy): how many primal-dual active-set passes ("add every violated row at once, drop every row
whose multiplier went negative") does the Formulation-A QP need before the Goldfarb-Idnani loop only has a few rows left
to fix?  Decides whether a block warm start is worth building into the wave kernel."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", f"prerollA_{name}.npz"))
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


def gi_from(H, g, E, b, N, lo, hi, W, x, mu, max_steps=1000):
    """Dense Goldfarb-Idnani continuation from the S-pair (x, W, mu >= 0).  Returns (x, W, steps)."""
    Hi = 1.0 / H
    W = dict(W); mu = dict(mu); steps = 0
    nrm = np.sqrt(((N * N) * Hi).sum(1))
    while True:
        cv = N @ x
        tol = 1e-11 * (np.abs(cv) + np.maximum(np.abs(lo), np.abs(hi))) + 1e-13
        vl = cv - lo; vh = hi - cv
        cand = np.zeros(len(cv)); sgn = np.zeros(len(cv))
        for r in range(len(cv)):
            if r in W: continue
            if vl[r] < -tol[r] and vl[r] / nrm[r] < cand[r]: cand[r] = vl[r] / nrm[r]; sgn[r] = +1
            if vh[r] < -tol[r] and vh[r] / nrm[r] < cand[r]: cand[r] = vh[r] / nrm[r]; sgn[r] = -1
        p = int(np.argmin(cand))
        if cand[p] >= 0: return x, W, steps
        sg = sgn[p]; n = sg * N[p]                      # constraint n.x >= sg*bound
        viol = (vl[p] if sg > 0 else vh[p])
        mu_p = 0.0
        while True:
            steps += 1
            if steps > max_steps: raise RuntimeError("GI stuck")
            rows = sorted(W)
            A_ = np.vstack([E] + [W[r] * N[r] for r in rows])
            S = (A_ * Hi) @ A_.T
            d = A_ @ (Hi * n)
            r_ = np.linalg.solve(S, d)
            z = Hi * n - Hi * (A_.T @ r_)
            zn = z @ n
            t1 = np.inf; lrow = None
            for k, r in enumerate(rows):
                if r_[k + 1] > 0:
                    tt = mu[r] / r_[k + 1]
                    if tt < t1: t1 = tt; lrow = r
            t2 = -viol / zn if zn > 1e-12 * (n @ (Hi * n)) else np.inf
            t = min(t1, t2)
            if not np.isfinite(t): raise RuntimeError("infeasible")
            if np.isfinite(t2): x = x + t * z
            for k, r in enumerate(rows): mu[r] -= t * r_[k + 1]
            mu_p += t
            if np.isfinite(t2) and t == t2:
                W[p] = int(sg); mu[p] = mu_p; break
            del W[lrow]; del mu[lrow]
            cvp = N[p] @ x
            viol = (cvp - lo[p]) if sg > 0 else (hi[p] - cvp)


def hybrid(H, g, E, b, N, lo, hi, passes, cleanup=6):
    """PDAS passes, then drop-negative clean-up, then GI.  Returns (work in GI-step equivalents, detail)."""
    W = {}; x, mu = solve_on(H, g, E, b, N, lo, hi, W)
    npass = 0
    for p in range(passes):
        cv = N @ x
        tol = 1e-11 * (np.abs(cv) + np.maximum(np.abs(lo), np.abs(hi))) + 1e-13
        new = {r: s for r, s in W.items() if mu[r] > 0}
        for r in np.nonzero(cv < lo - tol)[0]: new.setdefault(int(r), +1)
        for r in np.nonzero(cv > hi + tol)[0]: new.setdefault(int(r), -1)
        if new == W: break
        W = new; x, mu = solve_on(H, g, E, b, N, lo, hi, W); npass += 1
    ncl = 0
    while any(v < 0 for v in mu.values()):
        if ncl >= cleanup:                                   # give up: cold start
            W = {}; x, mu = solve_on(H, g, E, b, N, lo, hi, W); ncl += 1; break
        W = {r: s for r, s in W.items() if mu[r] >= 0}
        x, mu = solve_on(H, g, E, b, N, lo, hi, W); ncl += 1
    x, W2, steps = gi_from(H, g, E, b, N, lo, hi, W, x, mu)
    return npass, ncl, steps, x, W2


def study(name, ntest, passes_list=(0, 1, 2, 3, 4, 6)):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", f"prerollA_{name}.npz"))
    kind = int(z["gait"]); p = A.params(kind, C_=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
    sim = A.SimA(A.gait(kind, float(z["phi"]), float(z["disp_A"])), p, backend="gi")
    rng = np.random.default_rng(0)
    sim.run(int(rng.integers(5, 150)))
    res = {P: [] for P in passes_list}
    for t in range(ntest):
        sim.run(int(rng.integers(1, 12)))
        st = sim.state.copy()
        st2 = st.copy(); st2["xd"] += rng.uniform(-0.03, 0.03); st2["yd"] += rng.uniform(-0.05, 0.05)
        sim.state = st2
        for axis in (0, 1):
            D = sim.axis_data(axis)
            Q = build(D, p.dt, p.Qf)
            xs = None
            for P in passes_list:
                npass, ncl, steps, x, W = hybrid(*Q, P)
                if xs is None: xs = x
                assert np.abs(x - xs).max() < 1e-7 * max(1, np.abs(xs).max()), (P, np.abs(x - xs).max())
                res[P].append((npass, ncl, steps))
        sim.state = st
    for P in passes_list:
        r = np.array(res[P])
        print(f"{name} passes<={P}: pdas {r[:,0].mean():.2f} cleanup {r[:,1].mean():.2f} GI steps {r[:,2].mean():.1f}  "
              f"work(2/pass) {(2 * (r[:,0] + r[:,1]) + r[:,2]).mean():.1f}   max GI {r[:,2].max()}")

if __name__ == "__main__" and os.environ.get("STUDY"):
    study(sys.argv[1] if len(sys.argv) > 1 else "walk_C100", int(sys.argv[2]) if len(sys.argv) > 2 else 20)


def runs_of(W, C):
    """contiguous same-sign runs of active ZMP rows (0-based dense rows < C): dict row -> (lo, hi)."""
    out = {}
    rows = sorted(r for r in W if r < C)
    k = 0
    while k < len(rows):
        j = k
        while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1 and W[rows[j + 1]] == W[rows[k]]: j += 1
        for t in range(k, j + 1): out[rows[t]] = (rows[k], rows[j])
        k = j + 1
    return out


def hybrid2(H, g, E, b, N, lo, hi, C, max_add, max_drop, grow=2, cap=64, strict_ends=False, first='all'):
    """PDAS passes with geometric peeling at run ends; drop-only passes at the end; then GI.  (passes, cold, GI steps, x)"""
    W = {}; x, mu = solve_on(H, g, E, b, N, lo, hi, W)
    D = 1; npass = 0; cold = 0
    for p in range(max_add + max_drop + 1):
        adding = p < max_add
        neg = {r for r in W if (mu[r] <= 0 if adding else mu[r] < 0)}
        drop = set(neg)
        rn = runs_of(W, C)
        ends = True
        for r in neg:
            if r not in rn:
                if strict_ends: ends = False
                continue
            l, h = rn[r]
            if r == l and r != h: drop |= set(range(l, min(h, l + D - 1) + 1))
            elif r == h and r != l: drop |= set(range(max(l, h - D + 1), h + 1))
            elif l == h: pass
            elif strict_ends: ends = False
        new = {r: s for r, s in W.items() if r not in drop}
        if adding:
            cv = N @ x
            tol = 1e-11 * (np.abs(cv) + np.maximum(np.abs(lo), np.abs(hi))) + 1e-13
            vl = np.maximum(lo - cv, 0.0)[:C]; vh = np.maximum(cv - hi, 0.0)[:C]
            viol = np.maximum(vl, vh)
            thr = 0.0
            if first == 'half' and p == 0: thr = 0.5 * viol.max()
            if first == 'q' and p == 0: thr = 0.25 * viol.max()
            if first == 'half2' and p <= 1: thr = 0.5 * viol.max()
            for r in np.nonzero((cv[:C] < (lo - tol)[:C]) & (viol >= thr))[0]: new.setdefault(int(r), +1)
            for r in np.nonzero((cv[:C] > (hi + tol)[:C]) & (viol >= thr))[0]: new.setdefault(int(r), -1)
        if new == W: break
        if p >= max_add + max_drop:
            W = {}; x, mu = solve_on(H, g, E, b, N, lo, hi, W); cold = 1; break
        D = min(D * grow, cap) if (neg and ends) else 1
        W = new; x, mu = solve_on(H, g, E, b, N, lo, hi, W); npass += 1
    x, W2, steps = gi_from(H, g, E, b, N, lo, hi, W, x, mu)
    return npass, cold, steps, x


def study2(name, ntest, cfgs=((4, 6, 2, 'all'), (4, 6, 2, 'half'), (4, 6, 2, 'q'), (4, 6, 2, 'half2'), (6, 6, 2, 'half2'))):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", f"prerollA_{name}.npz"))
    kind = int(z["gait"]); p = A.params(kind, C_=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
    sim = A.SimA(A.gait(kind, float(z["phi"]), float(z["disp_A"])), p, backend="gi")
    rng = np.random.default_rng(0)
    sim.run(int(rng.integers(5, 150)))
    res = {c: [] for c in cfgs}
    for t in range(ntest):
        sim.run(int(rng.integers(1, 12)))
        st = sim.state.copy()
        st2 = st.copy(); st2["xd"] += rng.uniform(-0.03, 0.03); st2["yd"] += rng.uniform(-0.05, 0.05)
        sim.state = st2
        for axis in (0, 1):
            D = sim.axis_data(axis)
            Q = build(D, p.dt, p.Qf)
            xs = None
            for c in cfgs:
                npass, cold, steps, x = hybrid2(*Q, p.C, c[0], c[1], grow=c[2], first=(c[3] if len(c) > 3 else 'all'))
                if xs is None: xs = x
                assert np.abs(x - xs).max() < 1e-7 * max(1, np.abs(xs).max()), (c, np.abs(x - xs).max())
                res[c].append((npass, cold, steps))
        sim.state = st
    for c in cfgs:
        r = np.array(res[c])
        print(f"{name} add<={c[0]} drop<={c[1]} grow={c[2]} first={c[3] if len(c) > 3 else 'all'}: passes {r[:,0].mean():.2f} cold {r[:,1].mean():.2f} GI steps {r[:,2].mean():.1f}  "
              f"work(1.2/pass) {(1.2 * r[:,0] + r[:,2]).mean():.1f}   max GI {r[:,2].max()}")

if __name__ == "__main__" and os.environ.get("STUDY2"):
    study2(sys.argv[1] if len(sys.argv) > 1 else "walk_C100", int(sys.argv[2]) if len(sys.argv) > 2 else 20)
