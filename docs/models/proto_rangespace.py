#!/usr/bin/env python3
"""Prototype (numpy) of the range-space dual active-set solver planned for the Formulation-A HIP kernel:
diagonal H, structured rows, explicit inverse of S = N' H^-1 N with rank-1 border / Schur updates.
Validated here against the oracle's dense Goldfarb-Idnani on pushed closed-loop rollouts."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle_a as A


def solve_axis(D, dt, Qf, maxit=2000, refine=2):
    a, b, zlo, zhi, M, klo, khi, pref = D["a"], D["b"], D["zlo"], D["zhi"], D["M"][:, 1:], D["klo"], D["khi"], D["pref"]
    C, F = len(a), len(pref)
    PA = np.concatenate([[0.0], np.cumsum(a)])          # PA[i] = sum_{k<i} a_k
    aa = a @ a
    u = np.zeros(C); f = pref.copy()
    # row ids: 0 = E, 1..C = Z_i, C+1..C+F = K_r
    def ip(r1, r2):
        if r1 > r2: r1, r2 = r2, r1
        if r1 == 0:
            if r2 == 0: return aa
            if r2 <= C: return dt * PA[r2]
            return 0.0
        if r2 <= C:
            return dt * dt * min(r1, r2) + (M[r1 - 1] @ M[r2 - 1]) / Qf
        if r1 <= C:                                       # Z_i with K_r
            r = r2 - C
            return (-M[r1 - 1][r - 1] + (M[r1 - 1][r - 2] if r >= 2 else 0.0)) / Qf
        ra, rb = r1 - C, r2 - C
        if ra == rb: return (1.0 + (1.0 if ra >= 2 else 0.0)) / Qf
        return -1.0 / Qf if rb - ra == 1 else 0.0
    def rowvals():
        zeta = dt * np.cumsum(u) - M @ f
        kin = f - np.concatenate([[0.0], f[:-1]])
        return np.concatenate([zeta, kin])
    lo = np.concatenate([zlo, klo]); hi = np.concatenate([zhi, khi])
    norms = np.sqrt(np.array([ip(r, r) for r in range(1, C + F + 1)]))
    # equality
    s = a @ u - b
    t = -s / aa
    u = u + t * a
    act = [(0, 1.0)]; mu = [t]; Sinv = np.array([[1.0 / aa]])
    state = np.zeros(C + F, dtype=int)                   # 0 free, +1 lower active, -1 upper active
    iters = 0
    while True:
        iters += 1
        if iters > maxit: return None, iters, len(act)
        v = rowvals()
        vl = (v - lo); vh = (hi - v)
        tol = 1e-11 * (np.abs(v) + np.maximum(np.abs(lo), np.abs(hi))) + 1e-13
        cand_l = np.where((state == 0) & (vl < -tol), vl / norms, 0.0)
        cand_h = np.where((state == 0) & (vh < -tol), vh / norms, 0.0)
        il, ih = np.argmin(cand_l), np.argmin(cand_h)
        if cand_l[il] >= 0 and cand_h[ih] >= 0: break
        if cand_l[il] <= cand_h[ih]: row, sg, sviol = il + 1, 1.0, vl[il]
        else: row, sg, sviol = ih + 1, -1.0, vh[ih]
        mu_p = 0.0
        while True:
            q = len(act)
            dprime = np.array([sg * sj * ip(row, rj) for (rj, sj) in act])
            r = Sinv @ dprime
            npn = ip(row, row)
            gamma = npn - dprime @ r
            # z = Hinv (n+ - N r)
            zu = np.zeros(C); zf = np.zeros(F)
            def addrow(rr, coef):
                if rr == 0: zu[:] += coef * a
                elif rr <= C: zu[:rr] += coef * dt; zf[:] += coef * (-M[rr - 1]) / Qf
                else:
                    k = rr - C
                    zf[k - 1] += coef / Qf
                    if k >= 2: zf[k - 2] -= coef / Qf
            addrow(row, sg)
            for (rj, sj), rjv in zip(act, r): addrow(rj, -sj * rjv)
            t1 = np.inf; l = -1
            for j in range(1, q):
                if r[j] > 0 and mu[j] / r[j] < t1: t1 = mu[j] / r[j]; l = j
            t2 = -sviol / gamma if gamma > 1e-12 * npn else np.inf
            t = min(t1, t2)
            if not np.isfinite(t): return "infeasible", iters, q
            if np.isfinite(t2):
                u += t * zu; f += t * zf
            mu = [m - t * rr for m, rr in zip(mu, r)]; mu_p += t
            if np.isfinite(t2) and t == t2:
                # border
                n = q + 1
                Sn = np.empty((n, n))
                Sn[:q, :q] = Sinv + np.outer(r, r) / gamma
                Sn[:q, q] = -r / gamma; Sn[q, :q] = -r / gamma; Sn[q, q] = 1.0 / gamma
                Sinv = Sn; act.append((row, sg)); mu.append(mu_p); state[row - 1] = int(sg)
                break
            # drop l
            rl, sl = act[l]
            state[rl - 1] = 0
            keep = [k for k in range(q) if k != l]
            col = Sinv[keep, l]
            Sinv = Sinv[np.ix_(keep, keep)] - np.outer(col, col) / Sinv[l, l]
            act.pop(l); mu.pop(l)
            v = rowvals()
            sviol = (v[row - 1] - lo[row - 1]) if sg > 0 else (hi[row - 1] - v[row - 1])
    # refinement on the final active set: N' x = rhs exactly
    for _ in range(refine):
        v = rowvals()
        res = []
        for (rj, sj) in act:
            if rj == 0: res.append(b - a @ u)
            else: res.append((lo[rj - 1] if sj > 0 else hi[rj - 1]) - v[rj - 1])
        res = np.array(res) * np.array([sj for (_, sj) in act])       # sigma n.x = sigma bound
        dm = Sinv @ res
        for (rj, sj), c in zip(act, dm):
            coef = sj * c
            if rj == 0: u += coef * a
            elif rj <= C: u[:rj] += coef * dt; f += coef * (-M[rj - 1]) / Qf
            else:
                k = rj - C
                f[k - 1] += coef / Qf
                if k >= 2: f[k - 2] -= coef / Qf
    return np.concatenate([u, f]), iters, len(act)


if __name__ == "__main__":
    for kind, phi, dA, name in [(A.WALK, np.pi / 4, 0.1, "walk"), (A.TROT, np.pi / 4, 0.1, "trot")]:
        sim = A.SimA(A.gait(kind, phi, dA), A.params(kind), backend="gi")
        p = sim.p
        rng = np.random.default_rng(0)
        worst = 0.0; worst_u0 = 0.0; its = []; qs = []
        T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
        for t in range(T):
            push = (0, 0)
            if t % 97 == 50: push = (rng.uniform(-0.03, 0.03), rng.uniform(-0.05, 0.05))
            st = sim.state.copy()
            # the QP data of this tick as the oracle sees it (push is applied to the state before assembly)
            st2 = st.copy(); st2["xd"] += push[0]; st2["yd"] += push[1]
            sim.state = st2
            Dx, Dy = sim.axis_data(0), sim.axis_data(1)
            sim.state = st
            out, sx, sy = sim.tick(push, want_solution=True)
            for D, ref in ((Dx, sx), (Dy, sy)):
                sol, it, q = solve_axis(D, p.dt, p.Qf, refine=int(os.environ.get("REFINE","2")))
                assert sol is not None and not isinstance(sol, str), (t, sol)
                e = np.abs(sol - ref)
                worst = max(worst, e[:p.C].max() / max(1e-3, np.abs(ref[:p.C]).max()), e[p.C:].max())
                worst_u0 = max(worst_u0, e[0])
                its.append(it); qs.append(q)
        print(name, "ticks", T, "max rel err u %.3e" % worst, "u0 abs %.3e" % worst_u0, "iters mean %.1f max %d" % (np.mean(its), max(its)), "active mean %.1f max %d" % (np.mean(qs), max(qs)))
