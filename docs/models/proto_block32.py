#!/usr/bin/env python3
"""numpy model of the wave kernel's structured block solve (ismpc_a_wave.hpp: block_solve, kinematic rows pinned) in a chosen
dtype, on real pushed QPs, at the exact optimal working set: how accurate is the fp32 solve?  (row residuals on the active
rows, worst violation of the inactive ones, multiplier signs, distance to the fp64 optimum)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle_a as A
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proto_pdas import build, hybrid

def block_solve(D, cur, W, dt, Qf, eta, dtype):
    T = dtype
    a = D["a"].astype(T); C = len(a); F = len(D["pref"])
    M = D["M"]                                   # C x (F+1), exact weights
    k1 = np.array([int(np.nonzero(M[i])[0][0]) for i in range(C)])
    w1 = np.array([M[i, k1[i]] for i in range(C)]).astype(T)
    PA = np.concatenate([[0.0], np.cumsum(D["a"])]).astype(T)
    zlo = T(D["zlo"][0] - M[0, 0] * cur + cur * 0 if False else 0)  # placeholder
    # uniform band relative to the current footstep: lo_i - M_i0 cur + cur  (see ismpc_a_wave.hpp header)
    lo = (D["zlo"] - M[:, 0] * cur) ; hi = (D["zhi"] - M[:, 0] * cur)
    zlo = T(lo[0] ); zhi = T(hi[0])
    assert np.abs(lo - lo[0]).max() < 1e-12
    # shift: bounds relative -> subtract nothing more: rows are  lo0 <= dt cs - M'(f - cur) - ... careful: lo_i here = lo0 (absolute zmp based)
    pf = np.concatenate([[0.0], D["pref"] - cur, [0.0]]).astype(T)
    zlo = T(lo[0] + cur); zhi = T(hi[0] + cur)                         # -(zmp - cur) -+ w/2
    beq = T(D["b"]); aa = T((D["a"] ** 2).sum())
    dtT, idt, idt2 = T(dt), T(1.0 / dt), T(1.0 / (dt * dt))
    isq = T(1.0 / np.sqrt(Qf))
    rows = sorted(r + 1 for r in W if r < C)                          # 1-based active ZMP rows
    sgn = {r + 1: W[r] for r in W if r < C}
    def theta(i):                                                     # weights over columns 1..F
        th = np.zeros(F, dtype=T); kk = k1[i - 1]; w = w1[i - 1]
        if kk >= 1: th[kk - 1] = w
        if kk + 1 <= F: th[kk] = T(1) - w
        return th
    c = {i: (zlo if sgn[i] > 0 else zhi) + (w1[i - 1] * pf[k1[i - 1]] + (T(1) - w1[i - 1]) * pf[k1[i - 1] + 1]) for i in rows}
    Th = np.zeros((F, F), dtype=T); psi = np.zeros(F, dtype=T); gam = np.zeros(F, dtype=T); sig = T(0); gE = T(0)
    prev = 0; thp = np.zeros(F, dtype=T); pap = T(0); cp = T(0)
    for i in rows:
        om = idt2 / T(i - prev)
        dth = theta(i) - thp; dE = dtT * (PA[i] - pap); dc = c[i] - cp
        Th += om * np.outer(dth, dth); psi += om * dth * dE; gam += om * dth * dc; sig += om * dE * dE; gE += om * dE * dc
        prev = i; thp = theta(i); pap = PA[i]; cp = c[i]
    m = F + 1
    Amat = np.zeros((m, m), dtype=T); rhs = np.zeros(m, dtype=T)
    Amat[:F, :F] = np.eye(F, dtype=T) + Th * isq * isq
    Amat[:F, F] = psi * isq; Amat[F, :F] = psi * isq; Amat[F, F] = sig - aa
    if os.environ.get("EXACT_D"):
        # aa - sig as a sum of per-gap terms, in fp64 from the fp64 prefix sums of a and a^2 (a gap of one row contributes exactly 0)
        PAd = np.concatenate([[0.0], np.cumsum(D["a"])]); PA2d = np.concatenate([[0.0], np.cumsum(D["a"] ** 2)])
        Dd = 0.0; pv = 0
        for i in rows:
            Dd += (PA2d[i] - PA2d[pv]) - (PAd[i] - PAd[pv]) ** 2 / (i - pv); pv = i
        Dd += PA2d[C] - PA2d[pv]
        Amat[F, F] = T(-Dd)
    rhs[:F] = gam * isq; rhs[F] = gE - beq
    # Gauss-Jordan without pivoting in dtype
    Aug = np.concatenate([Amat, rhs[:, None]], 1).astype(T)
    for kk in range(m):
        ipv = T(1) / Aug[kk, kk]
        for i in range(m):
            if i != kk: Aug[i, kk + 1:] = Aug[i, kk + 1:] - (Aug[i, kk] * ipv) * Aug[kk, kk + 1:]
    cc = np.array([Aug[i, m] / Aug[i, i] for i in range(m)], dtype=T)
    cE = cc[F]
    comb = np.concatenate([[0.0], cc[:F] * isq, [0.0]]).astype(T)
    s = {0: T(0)}
    split = bool(os.environ.get("EXACT_D"))
    PAd = np.concatenate([[0.0], np.cumsum(D["a"])])
    for i in rows:
        s[i] = c[i] - (w1[i - 1] * comb[k1[i - 1]] + (T(1) - w1[i - 1]) * comb[k1[i - 1] + 1])
        if not split: s[i] = s[i] - dtT * PA[i] * cE
    u = np.zeros(C, dtype=T); mu = {}
    chain = [0] + rows
    for idx, i in enumerate(rows):
        p = chain[idx]
        d1 = (s[i] - s[p]) / T(i - p)
        abar = T((PAd[i] - PAd[p]) / (i - p))                       # mean of a over the segment, from the fp64 prefix sums
        for r in range(p + 1, i + 1):
            u[r - 1] = d1 * idt + ((cE * (a[r - 1] - abar)) if split else cE * a[r - 1])
        lam = d1
        if split: lam = lam - dtT * cE * abar
        if idx + 1 < len(rows):
            n = rows[idx + 1]; lam = lam - (s[n] - s[i]) / T(n - i)
            if split: lam = lam + dtT * cE * T((PAd[n] - PAd[i]) / (n - i))
        lam = lam * idt2
        mu[i] = lam if sgn[i] > 0 else -lam
    last = rows[-1] if rows else 0
    for r in range(last + 1, C + 1): u[r - 1] = cE * a[r - 1]
    f = (pf[1:F + 1] - comb[1:F + 1]).astype(T)
    # row values in dtype
    cs = np.cumsum(u, dtype=T)
    fl = np.concatenate([[0.0], f, [0.0]]).astype(T)
    v = dtT * cs - (w1 * fl[k1] + (T(1) - w1) * fl[k1 + 1])
    return u, f, v, mu, zlo, zhi, rows, sgn

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "walk_C100"
    ntest = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", f"prerollA_{name}.npz"))
    kind = int(z["gait"]); p = A.params(kind, C_=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
    sim = A.SimA(A.gait(kind, float(z["phi"]), float(z["disp_A"])), p, backend="gi")
    rng = np.random.default_rng(0)
    sim.run(int(rng.integers(5, 150)))
    eta = np.sqrt(p.grav / p.height)
    for t in range(ntest):
        sim.run(int(rng.integers(1, 12)))
        st = sim.state.copy()
        st2 = st.copy(); st2["xd"] += rng.uniform(-0.03, 0.03); st2["yd"] += rng.uniform(-0.05, 0.05)
        sim.state = st2
        for axis in (0, 1):
            D = sim.axis_data(axis)
            cur = float(st["cur_x"] if axis == 0 else st["cur_y"])
            Q = build(D, p.dt, p.Qf)
            npass, ncl, steps, x, W = hybrid(*Q, 4)
            Wz = {r: s for r, s in W.items() if r < p.C}
            if any(r >= p.C for r in W): continue                      # kinematic rows active: not the block phase
            out = {}
            for T in (np.float64, np.float32):
                u, f, v, mu, zlo, zhi, rows, sgn = block_solve(D, cur, Wz, p.dt, p.Qf, eta, T)
                act = np.array([i - 1 for i in rows]); ina = np.array([i for i in range(p.C) if i + 1 not in sgn])
                res = max(abs(float(v[i - 1]) - float(zlo if sgn[i] > 0 else zhi)) for i in rows) if rows else 0.0
                viol = max(0.0, float((zlo - v[ina]).max()), float((v[ina] - zhi).max())) if len(ina) else 0.0
                out[T] = (res, viol, min(mu.values()) if mu else 0.0, float(np.abs(u - x[:p.C]).max()), float(np.abs(f - (x[p.C:] - cur)).max()))
            print(f"t={t} ax={axis} |W|={len(Wz)}  f64: res {out[np.float64][0]:.1e} viol {out[np.float64][1]:.1e} du {out[np.float64][3]:.1e} | "
                  f"f32: res {out[np.float32][0]:.1e} viol {out[np.float32][1]:.1e} minmu {out[np.float32][2]:.2e} du {out[np.float32][3]:.1e} df {out[np.float32][4]:.1e}")
        sim.state = st


def first_pass_study(name, ntest=40):
    """What does the FIRST adding pass of the block warm start see?  From the equality-only point u = (b / a'a) a every violated
    row enters at once; is the structured solve on that set well posed (active rows back on their bounds, stability row met)?"""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", f"prerollA_{name}.npz"))
    kind = int(z["gait"]); p = A.params(kind, C_=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
    sim = A.SimA(A.gait(kind, float(z["phi"]), float(z["disp_A"])), p, backend="gi")
    rng = np.random.default_rng(1)
    sim.run(int(rng.integers(5, 150)))
    eta = np.sqrt(p.grav / p.height)
    os.environ["EXACT_D"] = "1"
    bad = 0; tot = 0
    for t in range(ntest):
        sim.run(int(rng.integers(1, 12)))
        st = sim.state.copy()
        st2 = st.copy(); st2["xd"] += rng.uniform(-0.03, 0.03); st2["yd"] += rng.uniform(-0.05, 0.05)
        sim.state = st2
        for axis in (0, 1):
            D = sim.axis_data(axis); cur = float(st["cur_x"] if axis == 0 else st["cur_y"])
            a = D["a"]; C = len(a); M = D["M"]
            u = (D["b"] / (a @ a)) * a
            f = D["pref"]
            v = p.dt * np.cumsum(u) - M[:, 1:] @ f
            W = {}
            for i in range(C):
                if v[i] < D["zlo"][i] - 1e-12: W[i] = +1
                elif v[i] > D["zhi"][i] + 1e-12: W[i] = -1
            if not W: continue
            tot += 1
            uu, ff, vv, mu, zlo, zhi, rows, sgn = block_solve(D, cur, W, p.dt, p.Qf, eta, np.float64)
            res = max(abs(float(vv[i - 1]) - float(zlo if sgn[i] > 0 else zhi)) for i in rows)
            eqr = abs(float(a @ uu) - D["b"])
            nneg = sum(1 for m_ in mu.values() if m_ <= 0)
            lo_n = sum(1 for s_ in sgn.values() if s_ > 0); hi_n = len(sgn) - lo_n
            flag = res > 1e-8 * 0.4 + 1e-10 or eqr > 1e-8 * (1 + abs(D["b"]))
            bad += flag
            print(f"t={t} ax={axis} |W1|={len(W)} (lo {lo_n} hi {hi_n}) res {res:.1e} eqr {eqr:.1e} neg-mult {nneg} max|u| {np.abs(uu).max():.2e} {'<-- CHECK FAILS' if flag else ''}")
        sim.state = st
    print("first pass check fails:", bad, "of", tot)

if __name__ == "__main__" and os.environ.get("FIRST"):
    first_pass_study(sys.argv[1] if len(sys.argv) > 1 else "walk_C150", int(sys.argv[2]) if len(sys.argv) > 2 else 30)
