"""Synthetic batches for the ISMPC tick: nominal closed-loop pre-roll
+ per-instance perturbation.  Data generation only -- no part of the tick is computed here.

The pre-roll tables (state and WalkState the solver saw at every frame of the nominal closed
loop, Controller.cpp:297-310,346-348,503-504) are committed fixtures under tests/golden/,
produced once by tests/golden/make_golden.py with the CPU oracle.
"""
import os

import numpy as np

from ._lib import TICK_IN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 20261003

# perturbation half-widths of SURVEY.md 8d config 2
PERTURB = dict(pos_xy=0.015, vel_xy=0.10, pos_z=0.005, vel_z=0.03)


def load_preroll(N):
    path = os.path.join(GOLDEN, f"preroll_N{N}.npz")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: run tests/golden/make_golden.py (needs the CPU oracle)")
    z = np.load(path)
    return z["tick_in"].view(TICK_IN).reshape(-1), z["frame_lo"].item(), z["frame_hi"].item()


def make_batch(N, batch, scale=1.0, seed=SEED, first_instance=0):
    """`batch` perturbed instances around the nominal gait at horizon N.
    Instance i depends only on (seed, first_instance + i): shards of a larger batch are
    reproducible on every rank without communication."""
    table, lo, hi = load_preroll(N)
    ids = np.arange(first_instance, first_instance + batch, dtype=np.uint64)
    # counter-based: one Philox stream per instance id, 7 draws each
    out = np.zeros(batch, dtype=TICK_IN)
    u = np.empty((batch, 7))
    # Philox is counter based; jump by instance id in blocks to stay vectorised
    blk = 4096
    for s in range(0, batch, blk):
        e = min(s + blk, batch)
        for j in range(s, e):
            g = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, int(ids[j])]))
            u[j] = g.random(7)
    frame = lo + np.minimum((u[:, 0] * (hi - lo + 1)).astype(np.int64), hi - lo)
    out[:] = table[frame]
    pm = lambda col: (2.0 * u[:, col] - 1.0)
    out["com_pos"][:, 0] += scale * PERTURB["pos_xy"] * pm(1)
    out["com_pos"][:, 1] += scale * PERTURB["pos_xy"] * pm(2)
    out["com_vel"][:, 0] += scale * PERTURB["vel_xy"] * pm(3)
    out["com_vel"][:, 1] += scale * PERTURB["vel_xy"] * pm(4)
    out["com_pos"][:, 2] += scale * PERTURB["pos_z"] * pm(5)
    out["com_vel"][:, 2] += scale * PERTURB["vel_z"] * pm(6)
    return out


# ---- Formulation A (the MATLAB generators' tick): BASELINE configs[3], [4] -------------------------------------------
PUSH_A = (0.03, 0.05)          # xd += U(+-0.03), yd += U(+-0.05)


def make_batch_a(name, batch, seed=SEED, stream=0, push_scale=1.0):
    """SURVEY.md 8d config 4 generator: nominal state at a random tick of the committed closed-loop table
    tests/golden/prerollA_<name>.npz (walk_C100 | walk_C150 | trot_C160) + an impulsive velocity push.
    `stream` separates ranks (each rank draws its own instances, no communication).
    Returns dict(kind, phi, disp_A, C, P, F, state [batch] STATE_A records, push [batch, 2])."""
    from .formulation_a import STATE_A
    path = os.path.join(GOLDEN, f"prerollA_{name}.npz")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: run tests/golden/make_golden_a.py:make_prerolls (needs the CPU oracle)")
    z = np.load(path)
    tab = z["state"].view(STATE_A).reshape(-1)
    rng = np.random.Generator(np.random.Philox(key=seed + stream))
    jj = rng.integers(0, len(tab), batch)
    push = np.stack([rng.uniform(-PUSH_A[0], PUSH_A[0], batch), rng.uniform(-PUSH_A[1], PUSH_A[1], batch)], 1) * push_scale
    return dict(kind=int(z["gait"]), phi=float(z["phi"]), disp_A=float(z["disp_A"]), C=int(z["C"]), P=int(z["P"]), F=int(z["F"]),
                state=tab[jj].copy(), push=np.ascontiguousarray(push))


def make_inst_mc(batch, seed=SEED, stream=0, C=200):
    """SURVEY.md 8d config 5 (BASELINE configs[4]) per-instance draw: trot / walk by instance parity, CoM height ~
    U(0.50, 0.62), step ~ U{40..100}, ds = round(0.6 step), F = ceil(C / step) + 1, Qf = 1e7 (trot) / 1e9 (walk).
    Returns (INST_A records [batch], push [batch, 2]); plan 0 = trot plan, plan 1 = walk plan (phi = pi/4, disp_A = 0.1)."""
    from .formulation_a import INST_A
    rng = np.random.Generator(np.random.Philox(key=seed + stream))
    inst = np.zeros(batch, dtype=INST_A)
    step = rng.integers(40, 101, batch)
    trot = (np.arange(batch) % 2) == 0
    inst["height"] = rng.uniform(0.50, 0.62, batch); inst["Qf"] = np.where(trot, 1e7, 1e9); inst["step"] = step
    inst["ds"] = np.round(0.6 * step).astype(np.int32); inst["F"] = -(-C // step) + 1; inst["plan"] = np.where(trot, 0, 1)
    push = np.stack([rng.uniform(-PUSH_A[0], PUSH_A[0], batch), rng.uniform(-PUSH_A[1], PUSH_A[1], batch)], 1)
    return inst, np.ascontiguousarray(push)


def make_sweep_params(K, N=100, seed=SEED):
    """K parameter sets for a Formulation B sweep (ismpc_create_sweep): set 0 = the reference's constants (parameters.cpp:9-45,
    MPCSolver.cpp:253-255); the others draw mass, CoM height, the three vertical-QP weights and the foot width around them."""
    from .solver import default_params
    rng = np.random.Generator(np.random.Philox(key=seed + 7919))
    out = []
    for k in range(K):
        p = default_params(N=N)
        if k > 0:
            p.mass = float(rng.uniform(35.0, 70.0)); p.h_des = float(rng.uniform(0.66, 0.72))
            p.q_p = float(1005000.0 * rng.uniform(0.3, 3.0)); p.q_v = float(100.0 * rng.uniform(0.3, 3.0)); p.q_u = float(0.01 * rng.uniform(0.5, 2.0))
            p.foot_width = float(rng.uniform(0.07, 0.11))
        out.append(p)
    return out
