#!/usr/bin/env python3
"""Prototype (dense numpy, CPU): one closed-loop tick of one instance of a bench batch with the previous tick's working set as
the first guess -- the situation of the slowest QPs of a device rollout (scripts/bench_rollout.py with a -DISMPC_A_DIAG build
names them).  usage: python docs/models/proto_closed.py walk_C150 <instance> <axis> <tick>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "docs", "models"))
from oracle import oracle_a as A
from quadruped_gait_generation_ismpc_amd import workload
from proto_pdas import build, solve_on
from proto_passes import block_passes, gi_some, picture

name, inst, axis, tick = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = workload.make_batch_a(name, 16384)
p = A.params(w["kind"], C_=w["C"], P=w["P"], F=w["F"])
sim = A.SimA(A.gait(w["kind"], w["phi"], w["disp_A"]), p, backend="gi")
sim.load_product_state(w["state"][inst])
sim.tick(push=tuple(w["push"][inst]))                     # the pushed tick (bench_rollout: one tick_torch, then the rollout)
C = p.C


def optimum(Q):
    H, g, E, b, N, lo, hi = Q
    W = {}; x, mu = solve_on(H, g, E, b, N, lo, hi, W)
    x, W, mu, st, done = gi_some(Q, W, x, mu, 10 ** 6)
    return W


prevW = None
for t in range(tick + 1):
    Q = build(sim.axis_data(axis), p.dt, p.Qf)
    Wopt = optimum(Q)
    if t == tick:
        guess = {r - 1: s for r, s in prevW.items() if 1 <= r < C}          # moved down by one row
        print("previous optimum (shifted) ", picture(guess, C))
        print("this tick's optimum        ", picture(Wopt, C), " kinematic rows active:", sorted(r - C for r in Wopt if r >= C))
        H, g, E, b, N, lo, hi = Q
        x, mu = solve_on(H, g, E, b, N, lo, hi, guess)
        log = []
        for rnd in range(3):
            ns, W, x, mu = block_passes(Q, C, dict(guess) if rnd == 0 else W, x, mu, 6, 12, True, True, 0, 6, log, True, 64, False, False, bool(int(os.environ.get('KEEP1', '0'))), bool(int(os.environ.get('LUMP', '0'))), bool(int(os.environ.get('SKIPDROP', '0'))))
            x, W, mu, st, done = gi_some(Q, W, x, mu, 8 if rnd < 2 else 10 ** 6)
            log.append(f"G  |W|={len(W):3d} steps={st:2d}  " + picture(W, C) + ("  done" if done else ""))
            if done: break
        for l in log: print("   " + l)
    prevW = Wopt
    sim.tick()
