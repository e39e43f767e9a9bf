#!/usr/bin/env python3
"""Prototype (dense numpy, CPU): the block warm start of the Formulation A wave kernel, pass by pass, on the bench workloads
(workload.make_batch_a + the oracle's per-axis QP data).  Emulates the kernel's rules -- adding passes (drop mu <= 0 with
geometric peeling of run ends, add every violated row), drop-only passes, then Goldfarb-Idnani from the valid pair -- and
variants of them; prints the work per QP and, with SHOW=n, the evolution of the working set of the n worst QPs.
usage: python docs/models/proto_passes.py walk_C150 [nqp] [variant ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_a as A
from quadruped_gait_generation_ismpc_amd import workload
sys.path.insert(0, os.path.join(ROOT, "docs", "models"))
from proto_pdas import build, solve_on, gi_from, runs_of


def gi_some(Q, W, x, mu, max_adds):
    """Goldfarb-Idnani from the valid pair (x, W, mu >= 0) until feasible or until max_adds rows have entered.
    Returns (x, W, mu, steps, done)."""
    H, g, E, b, N, lo, hi = Q
    Hi = 1.0 / H
    W = dict(W); mu = dict(mu); steps = 0; adds = 0
    nrm = np.sqrt(((N * N) * Hi).sum(1))
    while True:
        cv = N @ x
        tol = 1e-11 * (np.abs(cv) + np.maximum(np.abs(lo), np.abs(hi))) + 1e-13
        vl = (cv - lo); vh = (hi - cv)
        cl = np.where(vl < -tol, vl / nrm, 0.0); ch = np.where(vh < -tol, vh / nrm, 0.0)
        for r in W: cl[r] = 0.0; ch[r] = 0.0
        p = int(np.argmin(np.minimum(cl, ch)))
        if min(cl[p], ch[p]) >= 0: return x, W, mu, steps, True
        if adds >= max_adds: return x, W, mu, steps, False
        sg = 1.0 if cl[p] <= ch[p] else -1.0
        n = sg * N[p]
        viol = vl[p] if sg > 0 else vh[p]
        mu_p = 0.0
        while True:
            steps += 1
            rows = sorted(W)
            A_ = np.vstack([E] + [W[r] * N[r] for r in rows])
            S = (A_ * Hi) @ A_.T
            r_ = np.linalg.solve(S, A_ @ (Hi * n))
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
                W[p] = int(sg); mu[p] = mu_p; adds += 1; break
            del W[lrow]; del mu[lrow]
            cvp = N[p] @ x
            viol = (cvp - lo[p]) if sg > 0 else (hi[p] - cvp)


def picture(W, C):
    s = ["."] * C
    for r, sg in W.items():
        if r < C: s[r] = "L" if sg > 0 else "U"
    return "".join(s)


def passes(Q, C, add=4, drop=6, peel_drop=True, peel_add=True, readd=0, min_viol=6, log=None, damp=None, perend=False, cap=64, gi_first=0, thin=False, rounds=0, round_adds=12, guard=False, keep1=False, lump=False, skipdrop=False):
    """Returns (block solves, GI steps, |W| at GI start, x).  Variants:
    peel_drop / peel_add: geometric peeling in drop-only / adding passes; readd: re-entries into an adding pass when a valid pair
    still has >= min_viol violated rows; damp: None | 'ends' (an adding pass adds, of each run of violated rows, ...)"""
    H, g, E, b, N, lo, hi = Q
    W = {}; x, mu = solve_on(H, g, E, b, N, lo, hi, W)
    gi0 = 0; ns_tot = 0
    if gi_first > 0:
        x, W, mu, gi0, done = gi_some(Q, W, x, mu, gi_first)
        if log is not None: log.append(f"G  |W|={len(W):3d} steps={gi0:2d}  " + picture(W, C))
        if done: return 0, gi0, len(W), x
    for rnd in range(rounds):
        ns, W, x, mu = block_passes(Q, C, W, x, mu, add, drop, peel_drop, peel_add, readd, min_viol, log, perend, cap, thin, guard, keep1, lump, skipdrop)
        ns_tot += ns
        x, W, mu, st, done = gi_some(Q, W, x, mu, round_adds)
        gi0 += st
        if log is not None: log.append(f"G  |W|={len(W):3d} steps={st:2d}  " + picture(W, C))
        if done: return ns_tot, gi0, len(W), x
    ns, W, x, mu = block_passes(Q, C, W, x, mu, add, drop, peel_drop, peel_add, readd, min_viol, log, perend, cap, thin, guard, keep1, lump, skipdrop)
    ns_tot += ns
    q0 = len(W)
    x, W2, steps = gi_from(H, g, E, b, N, lo, hi, W, x, mu)
    if log is not None: log.append(f"GI {steps:3d} steps        " + " " * 8 + picture(W2, C))
    return ns_tot, steps + gi0, q0, x


def block_passes(Q, C, W, x, mu, add, drop, peel_drop, peel_add, readd, min_viol, log, perend, cap, thin, guard, keep1, lump, skipdrop):
    H, g, E, b, N, lo, hi = Q
    peel = 1; nsolve = 0; force_add = False; extra = readd; pc = {}
    newrows = set(); stop_adding = False
    budget = add + drop + readd * (1 + drop)
    p = 0
    while True:
        adding = (p < add and not stop_adding) or force_add
        force_add = False
        cv = N @ x
        tol = 1e-11 * (np.abs(cv) + np.maximum(np.abs(lo), np.abs(hi))) + 1e-13
        neg = {r for r in W if r < C and (mu[r] <= 0 if adding else mu[r] < 0)}
        dropset = set(neg)
        collapse = skipdrop and nsolve > 0 and 2 * len(neg) > len(W)
        if collapse:
            # most multipliers negative at once: the last cut took a pin away.  Drop nothing, let the violated rows back in
            dropset = set(); adding = True
        elif lump:
            # a negative run end: the rows it takes with it are those whose multipliers, summed from that end, stay <= 0
            # (a multiplier at row i acts on u_j, j <= i, exactly like one at any later row: lumping the cut rows' multipliers
            # onto the new end row leaves the earlier part of the horizon as it is, and the new end must come out positive)
            rn = runs_of(W, C)
            for (l, h) in set(rn.values()):
                if l == h: continue
                if h in neg:
                    c = 0.0
                    for r in range(h, l, -1):
                        c += mu[r]
                        if c <= 0: dropset.add(r)
                if l in neg:
                    c = 0.0
                    for r in range(l, h):
                        c += mu[r]
                        if c <= 0: dropset.add(r)
        elif perend:
            # every run end keeps its own peel length: an end that is negative again one pass after it was cut doubles its cut
            pcn = {}
            rn = runs_of(W, C)
            for (l, h) in set(rn.values()):
                if l == h: continue
                if l in neg:
                    d = pc.get(l, 1)
                    for r in range(l, min(h - 1 if keep1 else h, l + d - 1) + 1): dropset.add(r)
                    if l + d <= h: pcn[l + d] = min(2 * d, cap)
                if h in neg:
                    d = pc.get(h, 1)
                    for r in range(max(l + 1 if keep1 else l, h - d + 1), h + 1): dropset.add(r)
                    if h - d >= l: pcn[h - d] = min(2 * d, cap)
            pc = pcn
        elif peel > 1 and neg and (peel_add if adding else peel_drop):
            rn = runs_of(W, C)
            for r in list(W):
                if r >= C or r not in rn: continue
                l, h = rn[r]
                if l == h: continue
                if r - l < peel and l in neg: dropset.add(r)
                if h - r < peel and h in neg: dropset.add(r)
        peel = min(2 * peel, 64) if neg else 1
        new = {r: s for r, s in W.items() if r not in dropset}
        if guard and nsolve > 0 and 2 * len(neg) > len(W) and newrows:
            # most multipliers negative at once: the rows that entered last over-constrained the problem (the stability multiplier
            # changed sign).  Take exactly those out again and stop adding: drop-only passes, then Goldfarb-Idnani
            new = {r: s for r, s in W.items() if r not in newrows}
            stop_adding = True; adding = False; pc = {}
        if adding and thin:
            # a stretch of violated rows between two rows that are active on the same bound rides that bound: all of it enters;
            # an open stretch usually ends as a touching point: only its most violated row enters
            vs = np.zeros(C, dtype=int)
            vs[(cv[:C] < (lo - tol)[:C])] = +1; vs[(cv[:C] > (hi + tol)[:C])] = -1
            for r in W:
                if r < C: vs[r] = 0
            nrm_ = np.sqrt(((N[:C] * N[:C]) / H).sum(1))
            r = 0
            while r < C:
                if vs[r] == 0: r += 1; continue
                l = r
                while r + 1 < C and vs[r + 1] == vs[l]: r += 1
                h = r; r += 1
                sg = int(vs[l])
                closed = (l - 1 in W and W[l - 1] == sg and l - 1 not in dropset) and (h + 1 in W and W[h + 1] == sg and h + 1 not in dropset)
                if closed:
                    for q in range(l, h + 1): new[q] = sg
                else:
                    viol = (lo[l:h + 1] - cv[l:h + 1]) if sg > 0 else (cv[l:h + 1] - hi[l:h + 1])
                    new[l + int(np.argmax(viol / nrm_[l:h + 1]))] = sg
        elif adding:
            for r in np.nonzero(cv[:C] < (lo - tol)[:C])[0]:
                if int(r) not in W: new[int(r)] = +1
            for r in np.nonzero(cv[:C] > (hi + tol)[:C])[0]:
                if int(r) not in W: new[int(r)] = -1
        if collapse and new == W:                     # nothing came back: the negatives have to go after all
            new = {r: s for r, s in W.items() if r not in neg}
        newrows = set(new) - set(W)
        if log is not None: log.append(("A" if adding else "D") + f"{p:2d} |W|={len(W):3d} neg={len(neg):3d} " + picture(W, C))
        if new == W:
            if not adding and extra > 0:
                nv = int(((cv[:C] < (lo - tol)[:C]) | (cv[:C] > (hi + tol)[:C])).sum())
                if nv >= min_viol: extra -= 1; force_add = True; p += 1; continue
            break
        if nsolve >= budget:
            W = {}; x, mu = solve_on(H, g, E, b, N, lo, hi, W); break
        W = new; x, mu = solve_on(H, g, E, b, N, lo, hi, W); nsolve += 1; p += 1
    return nsolve, W, x, mu


VARIANTS = {
    "kernel": dict(),
    "nopeel_drop": dict(peel_drop=False),
    "nopeel": dict(peel_drop=False, peel_add=False),
    "add6": dict(add=6, drop=6),
    "add8": dict(add=8, drop=6),
    "readd1": dict(readd=1),
    "readd2": dict(readd=2),
    "readd2_nopeel_drop": dict(readd=2, peel_drop=False),
    "perend": dict(perend=True),
    "gi2_perend": dict(perend=True, gi_first=2),
    "gi3_perend": dict(perend=True, gi_first=3),
    "gi4_perend": dict(perend=True, gi_first=4),
    "gi6_perend": dict(perend=True, gi_first=6),
    "gi3_perend_add6": dict(perend=True, gi_first=3, add=6),
    "gi3_kernel": dict(gi_first=3),
    "new": dict(perend=True, gi_first=3, add=6, drop=12),
    "new_r1": dict(perend=True, gi_first=3, add=6, drop=12, rounds=1, round_adds=8),
    "new_r2": dict(perend=True, gi_first=3, add=6, drop=12, rounds=2, round_adds=8),
    "new_r2_4": dict(perend=True, gi_first=3, add=6, drop=12, rounds=2, round_adds=4),
    "new_r3_4": dict(perend=True, gi_first=3, add=4, drop=12, rounds=3, round_adds=4),
    "keep1": dict(perend=True, gi_first=3, add=6, drop=12, rounds=2, round_adds=8, keep1=True),
    "base": dict(perend=True, gi_first=3, add=6, drop=12, rounds=2, round_adds=8),
    "lump": dict(perend=True, gi_first=3, add=6, drop=12, rounds=2, round_adds=8, lump=True),
    "lump_gi0": dict(perend=True, gi_first=0, add=6, drop=12, rounds=2, round_adds=8, lump=True),
    "lump_gi1": dict(perend=True, gi_first=1, add=6, drop=12, rounds=2, round_adds=8, lump=True),
    "lump_gi2": dict(perend=True, gi_first=2, add=6, drop=12, rounds=2, round_adds=8, lump=True),
    "lump_gi4": dict(perend=True, gi_first=4, add=6, drop=12, rounds=2, round_adds=8, lump=True),
    "lump_gi2_r4": dict(perend=True, gi_first=2, add=6, drop=12, rounds=2, round_adds=4, lump=True),
    "lump_sd": dict(perend=True, gi_first=2, add=6, drop=12, rounds=2, round_adds=8, lump=True, skipdrop=True),
    "lump_g2": dict(perend=True, gi_first=2, add=6, drop=12, rounds=2, round_adds=8, lump=True),
    "lump_43": dict(perend=True, gi_first=3, add=4, drop=6, rounds=2, round_adds=8, lump=True),
    "guard": dict(perend=True, gi_first=3, add=6, drop=12, guard=True),
    "thin": dict(perend=True, gi_first=3, add=6, drop=12, thin=True),
    "thin_gi1": dict(perend=True, gi_first=1, add=6, drop=12, thin=True),
    "thin_gi0": dict(perend=True, gi_first=0, add=6, drop=12, thin=True),
    "thin_add8": dict(perend=True, gi_first=3, add=8, drop=12, thin=True),
    "thin_gi2_add8": dict(perend=True, gi_first=2, add=8, drop=12, thin=True),
    "perend_add6": dict(perend=True, add=6, drop=6),
    "perend_add8": dict(perend=True, add=8, drop=6),
    "perend_add10": dict(perend=True, add=10, drop=6),
    "perend_readd1": dict(perend=True, readd=1),
    "perend_add8_readd1": dict(perend=True, add=8, readd=1),
}

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "walk_C150"
    nqp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    variants = sys.argv[3:] or ["kernel"]
    w = workload.make_batch_a(name, 16384)
    p = A.params(w["kind"], C_=w["C"], P=w["P"], F=w["F"])
    sim = A.SimA(A.gait(w["kind"], w["phi"], w["disp_A"]), p, backend="gi")
    show = int(os.environ.get("SHOW", "0"))
    res = {v: [] for v in variants}
    logs = []
    for k in range(int(os.environ.get('START', '0')), int(os.environ.get('START', '0')) + nqp // 2):
        rec = w["state"][k].copy()
        rec["xd"] += w["push"][k, 0]; rec["yd"] += w["push"][k, 1]
        sim.load_product_state(rec)
        for axis in (0, 1):
            Q = build(sim.axis_data(axis), p.dt, p.Qf)
            xs = None
            for v in variants:
                lg = [] if (show and v == variants[0]) else None
                ns, st, q0, x = passes(Q, p.C, log=lg, **VARIANTS[v])
                if xs is None: xs = x
                assert np.abs(x - xs).max() < 1e-6 * max(1, np.abs(xs).max()), (v, np.abs(x - xs).max())
                res[v].append((ns, st, q0))
                if lg is not None: logs.append((ns + st, k, axis, lg))
    for v in variants:
        r = np.array(res[v]); work = r[:, 0] + r[:, 1]
        big = work >= 32
        print(f"{name} {v:22s}: block solves {r[:,0].mean():5.2f}  GI steps {r[:,1].mean():6.2f}  work {work.mean():6.2f}  "
              f">=32: {big.mean():.3f} of QPs, {work[big].sum() / work.sum():.3f} of work   max {work.max()}")
    logs.sort(key=lambda t: -t[0])
    for wk, k, axis, lg in logs[:show]:
        print(f"--- instance {k} axis {axis}: work {wk}")
        for l in lg: print("   " + l)
