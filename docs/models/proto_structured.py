#!/usr/bin/env python3
"""Prototype (numpy) of the STRUCTURED range-space dual active-set solver for the Formulation-A QP:
no matrix over the working set at all.  For the active ZMP rows (sorted by index) the Gram block is
dt^2 * min(i_j, i_k) -- the covariance of a random walk -- whose inverse is tridiagonal, i.e. K^-1 y only needs
the previous / next active row; everything else (footstep coupling M~, stability row, kinematic rows) is a
rank <= 2F+1 border handled by an m x m system (m = F + 1 + #active kinematic rows).
Checked here against a dense solve of S r = d at every step and against the oracle's solutions."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle_a as A

CHECK = int(os.environ.get("CHECK", "1"))


class Solver:
    def __init__(self, D, dt, Qf):
        self.a, self.b = D["a"], D["b"]
        self.zlo, self.zhi = D["zlo"], D["zhi"]
        self.M = D["M"][:, 1:]
        self.klo, self.khi, self.pref = D["klo"], D["khi"], D["pref"]
        self.C, self.F = len(self.a), len(self.pref)
        self.dt, self.Qf = dt, Qf
        self.PA = np.concatenate([[0.0], np.cumsum(self.a)])
        self.aa = self.a @ self.a
        self.sq = np.sqrt(Qf)

    # ---- closed forms ----
    def Mt(self, i):                       # M~ row of ZMP row i (F)
        return self.M[i - 1] / self.sq

    def kvec(self, r):                     # footstep-space vector of kinematic row r: e_r - e_{r-1}
        e = np.zeros(self.F); e[r - 1] = 1.0
        if r >= 2: e[r - 2] = -1.0
        return e

    def V(self, i, X):                     # border row of active ZMP row i: [M~_i, B_i(X)]  (unsigned)
        C = self.C
        out = list(self.Mt(i))
        for (row, sg) in X:
            if row == 0: out.append(self.dt * self.PA[i])
            else: out.append(sg * (-(self.M[i - 1] @ self.kvec(row - C))) / self.Qf)
        return np.array(out)

    def ipX(self, r1, s1, r2, s2):
        C = self.C
        if r1 == 0 and r2 == 0: return self.aa
        if r1 == 0 or r2 == 0: return 0.0
        return s1 * s2 * (self.kvec(r1 - C) @ self.kvec(r2 - C)) / self.Qf

    def ip(self, r1, r2):                  # unsigned H^-1 inner product of any two rows
        C, dt, Qf = self.C, self.dt, self.Qf
        if r1 > r2: r1, r2 = r2, r1
        if r1 == 0:
            if r2 == 0: return self.aa
            if r2 <= C: return dt * self.PA[r2]
            return 0.0
        if r2 <= C: return dt * dt * r1 + (self.M[r1 - 1] @ self.M[r2 - 1]) / Qf
        if r1 <= C: return -(self.M[r1 - 1] @ self.kvec(r2 - C)) / Qf
        return (self.kvec(r1 - C) @ self.kvec(r2 - C)) / Qf

    # ---- structured solve of S r = d for the new row (row, sg) ----
    def solve_new(self, Zs, sig, X, row, sg):
        """Zs: sorted active ZMP indices; sig: their signs; X: list of (row, sign) with row 0 first.
        Returns rho (unsigned: r_Z = sig*rho), r_X, dd = d.r, c1 = M~' rho."""
        C, F, dt = self.C, self.F, self.dt
        qz, mx = len(Zs), len(X); m = F + mx
        gaps = np.diff(np.concatenate([[0], Zs])) if qz else np.zeros(0)
        Vr = np.array([self.V(i, X) for i in Zs]) if qz else np.zeros((0, m))
        # G = V' K^-1 V / dt^2 : sum over gaps of outer(dV, dV) / (g dt^2)
        dV = np.diff(np.vstack([np.zeros((1, m)), Vr]), axis=0) if qz else np.zeros((0, m))
        G = (dV.T * (1.0 / (gaps * dt * dt))) @ dV if qz else np.zeros((m, m))
        # new row data
        if row <= C and row >= 1:
            mt = sg * self.Mt(row)                               # delta_Z = sg dt^2 k_i + M~ mt
            # interpolation of V at i+ (K^-1 k_i is 2-sparse)
            if qz:
                pos = np.searchsorted(Zs, row)
                if pos == 0: vint = Vr[0] * (row / Zs[0])
                elif pos == qz: vint = Vr[-1]
                else:
                    a_, b_ = Zs[pos - 1], Zs[pos]; th = (row - a_) / (b_ - a_)
                    vint = Vr[pos - 1] + th * (Vr[pos] - Vr[pos - 1])
            else: vint = np.zeros(m)
            h = sg * vint + G[:, :F] @ mt
        else:
            mt = sg * (-self.kvec(row - C)) / self.sq            # delta_Z = M~ mt
            vint = None
            h = G[:, :F] @ mt
        dX = np.array([sg * sx * self.ip(row, rx) if rx != 0 else sg * self.ip(row, 0) for (rx, sx) in X])
        SXX = np.array([[self.ipX(r1, s1, r2, s2) for (r2, s2) in X] for (r1, s1) in X])
        # small system  [[I + G11, G12], [G21, G22 - SXX]] c = [h1 ; h2 - dX]
        K = G.copy(); K[:F, :F] += np.eye(F); K[F:, F:] -= SXX
        rhs = h.copy(); rhs[F:] -= dX
        # scale rows/cols of the X block (entries down to 1/Qf)
        sc = np.ones(m); sc[F:] = 1.0 / np.sqrt(np.abs(np.diag(SXX)))
        c = sc * np.linalg.solve((K * sc[:, None]) * sc[None, :], rhs * sc)
        # rho = K^-1 (delta_Z - V c)/dt^2 = sg*w_interp + TV (mt - c)  with TV_j from neighbours
        y = mt_full = np.concatenate([mt, np.zeros(mx)]) - c       # coefficient on V columns
        rho = np.zeros(qz)
        if qz:
            Vy = Vr @ y
            e = np.diff(np.concatenate([[0.0], Vy])) / (gaps * dt * dt)
            rho = e - np.concatenate([e[1:], [0.0]])
            if vint is not None:                                     # sg * K^-1 k_i
                pos = np.searchsorted(Zs, row)
                if pos == 0: rho[0] += sg * row / Zs[0]
                elif pos == qz: rho[-1] += sg
                else:
                    a_, b_ = Zs[pos - 1], Zs[pos]; th = (row - a_) / (b_ - a_)
                    rho[pos - 1] += sg * (1 - th); rho[pos] += sg * th
        rX = c[F:]
        # d.r
        if qz:
            if row <= C and row >= 1: dZ = np.array([sg * (dt * dt * min(row, i) + (self.M[row - 1] @ self.M[i - 1]) / self.Qf) for i in Zs])
            else: dZ = np.array([sg * (-(self.M[i - 1] @ self.kvec(row - C))) / self.Qf for i in Zs])
            dd = dZ @ rho
        else: dZ = np.zeros(0); dd = 0.0
        dd += dX @ rX
        return rho, rX, dd, c[:F], dZ, dX

    def dense_check(self, Zs, sig, X, row, sg, rho, rX):
        rows = [(i, s) for i, s in zip(Zs, sig)] + list(X)
        S = np.array([[s1 * s2 * self.ip(r1, r2) for (r2, s2) in rows] for (r1, s1) in rows])
        d = np.array([sg * s1 * self.ip(row, r1) for (r1, s1) in rows])
        r = np.linalg.solve(S, d)
        mine = np.concatenate([np.array(sig) * rho, rX])
        err = np.abs(mine - r).max() / max(1e-300, np.abs(r).max())
        assert err < 1e-6, (err, len(Zs), len(X))

    def solve(self, maxit=5000):
        C, F, dt, Qf = self.C, self.F, self.dt, self.Qf
        a, M = self.a, self.M
        u = np.zeros(C); f = self.pref.copy()
        t0 = self.b / self.aa; u = t0 * a
        mu_e = t0
        Zs = []; sig = []; muZ = []            # sorted active ZMP rows
        X = [(0, 1.0)]; muX = [mu_e]
        lo = np.concatenate([self.zlo, self.klo]); hi = np.concatenate([self.zhi, self.khi])
        norms = np.sqrt(np.array([self.ip(r, r) for r in range(1, C + F + 1)]))
        state = np.zeros(C + F, dtype=int)
        iters = 0
        def rowvals():
            return np.concatenate([dt * np.cumsum(u) - M @ f, f - np.concatenate([[0.0], f[:-1]])])
        while True:
            v = rowvals()
            vl = v - lo; vh = hi - v
            tol = 1e-11 * (np.abs(v) + np.maximum(np.abs(lo), np.abs(hi))) + 1e-13
            cl = np.where((state == 0) & (vl < -tol), vl / norms, 0.0); chh = np.where((state == 0) & (vh < -tol), vh / norms, 0.0)
            il, ih = np.argmin(cl), np.argmin(chh)
            if cl[il] >= 0 and chh[ih] >= 0: break
            if cl[il] <= chh[ih]: row, sg, sviol = il + 1, 1.0, vl[il]
            else: row, sg, sviol = ih + 1, -1.0, vh[ih]
            mu_p = 0.0
            while True:
                iters += 1
                if iters > maxit: return None, iters
                rho, rX, dd, c1, dZ, dX = self.solve_new(np.array(Zs, dtype=int), sig, X, row, sg)
                if CHECK: self.dense_check(Zs, sig, X, row, sg, rho, rX)
                npn = self.ip(row, row)
                gamma = npn - dd
                rZ = np.array(sig) * rho if Zs else np.zeros(0)
                # step lengths
                t1 = np.inf; l = None
                for j in range(len(Zs)):
                    if rZ[j] > 0 and muZ[j] / rZ[j] < t1: t1 = muZ[j] / rZ[j]; l = ("Z", j)
                for j in range(1, len(X)):
                    if rX[j] > 0 and muX[j] / rX[j] < t1: t1 = muX[j] / rX[j]; l = ("X", j)
                t2 = -sviol / gamma if gamma > 1e-12 * npn else np.inf
                t = min(t1, t2)
                if not np.isfinite(t): return "infeasible", iters
                if np.isfinite(t2):
                    # z_u = sg dt [k < i+] - dt sum_j rho_j [k < i_j] - r_E a ;  z_f from c1 and the X / new-row footstep parts
                    imp = np.zeros(C)
                    if 1 <= row <= C: imp[row - 1] += sg * dt
                    for i, rj in zip(Zs, rho): imp[i - 1] -= dt * rj
                    zu = np.cumsum(imp[::-1])[::-1] - rX[0] * a
                    nf = np.zeros(F)
                    if 1 <= row <= C: nf += sg * (-M[row - 1])
                    else: nf += sg * self.kvec(row - C)
                    nf += self.sq * c1                                   # - sum_j r_j n_jf = + M' rho = sqrt(Qf) c1
                    for (rx, sx), rj in zip(X[1:], rX[1:]): nf -= rj * sx * self.kvec(rx - C)
                    zf = nf / Qf
                    u += t * zu; f += t * zf
                muZ = [m_ - t * r_ for m_, r_ in zip(muZ, rZ)]; muX = [m_ - t * r_ for m_, r_ in zip(muX, rX)]; mu_p += t
                if np.isfinite(t2) and t == t2:
                    if row <= C:
                        pos = int(np.searchsorted(Zs, row)); Zs.insert(pos, row); sig.insert(pos, sg); muZ.insert(pos, mu_p)
                    else: X.append((row, sg)); muX.append(mu_p)
                    state[row - 1] = int(sg)
                    break
                kind, j = l
                if kind == "Z": state[Zs[j] - 1] = 0; Zs.pop(j); sig.pop(j); muZ.pop(j)
                else: state[X[j][0] - 1] = 0; X.pop(j); muX.pop(j)
                v = rowvals(); sviol = (v[row - 1] - lo[row - 1]) if sg > 0 else (hi[row - 1] - v[row - 1])
        return np.concatenate([u, f]), iters


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    for kind, phi, dA, name in [(A.WALK, np.pi / 4, 0.1, "walk"), (A.TROT, np.pi / 4, 0.1, "trot")]:
        sim = A.SimA(A.gait(kind, phi, dA), A.params(kind), backend="gi"); p = sim.p
        rng = np.random.default_rng(0); worst = 0.0; its = []
        for t in range(T):
            push = (0, 0)
            if t % 97 == 50: push = (rng.uniform(-0.03, 0.03), rng.uniform(-0.05, 0.05))
            st = sim.state.copy(); st2 = st.copy(); st2["xd"] += push[0]; st2["yd"] += push[1]; sim.state = st2
            Dx, Dy = sim.axis_data(0), sim.axis_data(1); sim.state = st
            out, sx, sy = sim.tick(push, want_solution=True)
            for D, ref in ((Dx, sx), (Dy, sy)):
                sol, it = Solver(D, p.dt, p.Qf).solve()
                assert sol is not None and not isinstance(sol, str), (t, sol)
                e = np.abs(sol - ref)
                worst = max(worst, e[:p.C].max() / max(1e-3, np.abs(ref[:p.C]).max()), e[p.C:].max()); its.append(it)
        print(name, "ticks", T, "max rel err %.3e" % worst, "iters mean %.1f max %d" % (np.mean(its), max(its)))
