#!/usr/bin/env python3
"""Formulation A throughput (BASELINE configs 4-5 family): one tick over a batch of perturbed nominal
instances.  Not the headline bench (bench.py); numbers go to DESIGN.md.
usage: python scripts/bench_a.py [walk_C150|walk_C100|trot_C160|mc_C200] [batch per GPU] [steps] [--cpu] [--rollout T]
       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_a.py mc_C200 16384 10
mc_C200 = BASELINE configs[4]: trot / walk by instance parity, per-instance CoM height, step timing and footstep count
(131 072 instances over 8 GPUs = 16 384 per GPU).  With more than one rank every rank draws its own instances (Philox key
+ rank), and a step ends with ONE all-gather of the 80-byte output records, as in bench.py."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quadruped_gait_generation_ismpc_amd as q
from quadruped_gait_generation_ismpc_amd import formulation_a as FA

name = sys.argv[1] if len(sys.argv) > 1 else "walk_C150"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
rehearse = world > 1 and os.environ.get("ISMPC_BENCH_REHEARSE") == "1"      # all ranks on cuda:0, gloo: control flow only
dev_index = 0 if rehearse else local_rank
torch.cuda.set_device(dev_index)
DEV = f"cuda:{dev_index}"
dist = None
if world > 1:
    import torch.distributed as dist
    from quadruped_gait_generation_ismpc_amd.distributed import gather_records
    if rehearse: dist.init_process_group("gloo", rank=rank, world_size=world)
    else: dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(DEV))
    all_out = torch.empty((world * batch, 80), dtype=torch.uint8, device=("cpu" if rehearse else DEV))


def end_of_step(out_u8):
    """the one exchange of the path: all-gather of the output records"""
    if world > 1:
        gather_records(out_u8.cpu() if rehearse else out_u8, world, out=all_out, counts=[batch] * world)


def timed(fn, n):
    """n calls of fn between barriers; wall time = max over ranks"""
    if world > 1: dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=("cpu" if rehearse else DEV)); dist.all_reduce(t, op=dist.ReduceOp.MAX); el = float(t.item())
    return el


def report(res):
    res.update({"n_gpus": world, "batch_per_gpu": batch})
    if rehearse: res["note"] = "REHEARSAL: all ranks on one GPU, gloo; not a measurement"
    if rank == 0: print(json.dumps(res), flush=True)
    if world > 1: dist.destroy_process_group()
    sys.exit(0)


if name == "mc_C200":
    rng = np.random.Generator(np.random.Philox(key=20261003 + rank))
    Cn, Pn = 200, 400
    inst = np.zeros(batch, dtype=FA.INST_A)
    step = rng.integers(40, 101, batch)
    trot = (np.arange(batch) % 2) == 0
    inst["height"] = rng.uniform(0.50, 0.62, batch); inst["Qf"] = np.where(trot, 1e7, 1e9); inst["step"] = step
    inst["ds"] = np.round(0.6 * step).astype(np.int32); inst["F"] = -(-Cn // step) + 1; inst["plan"] = np.where(trot, 0, 1)
    plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
    gen = FA.GaitGenerator(FA.default_params(0, C=Cn, P=Pn, F=6), plans[0], device=dev_index); gen.add_plan(plans[1])
    d_inst = q.to_device(inst, DEV)
    d0 = q.to_device(gen.initial_state(0.88, batch=batch), DEV)
    gen.rollout_inst_torch(d0, d_inst, 60)                                   # nominal closed loop to spread the gait phases
    push = np.stack([rng.uniform(-0.03, 0.03, batch), rng.uniform(-0.05, 0.05, batch)], 1)
    dpush = torch.from_numpy(push.copy()).to(DEV); d = d0.clone()
    out = gen.tick_inst_torch(d, d_inst, dpush); torch.cuda.synchronize()
    o = q.from_device(out, FA.OUT_A)
    def one():
        d.copy_(d0); end_of_step(gen.tick_inst_torch(d, d_inst, dpush))
    one()
    el = timed(one, steps)
    report({"workload": name, "batch": world * batch, "steps": steps, "ticks_per_s": world * batch * steps / el, "ms_per_step": 1e3 * el / steps,
            "status_nonzero": int((o["status"] != 0).sum()),
            "iters_mean": float((o["iters_x"] + o["iters_y"]).mean() / 2), "iters_max": int(max(o["iters_x"].max(), o["iters_y"].max())),
            "active_mean": float(((o["active"] & 0xffff) + (o["active"] >> 16)).mean() / 2)})
z = np.load(os.path.join(ROOT, "tests", "golden", f"prerollA_{name}.npz"))
tab = z["state"].view(FA.STATE_A).reshape(-1)
kind = int(z["gait"]); g = FA.default_gait(kind, float(z["phi"]), float(z["disp_A"]))
fp, ce = FA.plan(g)
p = FA.default_params(kind, C=int(z["C"]), P=int(z["P"]), F=int(z["F"]))
gen = FA.GaitGenerator(p, ce, device=dev_index)
rng = np.random.Generator(np.random.Philox(key=20261003 + rank))
jj = rng.integers(0, len(tab), batch)
st0 = tab[jj].copy()
push = np.stack([rng.uniform(-0.03, 0.03, batch), rng.uniform(-0.05, 0.05, batch)], 1)
d0 = q.to_device(st0, DEV); d = d0.clone(); dpush = torch.from_numpy(push.copy()).to(DEV)
out = gen.tick_torch(d, dpush); torch.cuda.synchronize()
o = q.from_device(out, FA.OUT_A)
if "--rollout" in sys.argv:
    # closed loop: one pushed tick, then T ticks on the device (Monte-Carlo robustness use: the previous working set is the first guess)
    T = int(sys.argv[sys.argv.index("--rollout") + 1])
    gen.rollout_torch(d, 3); torch.cuda.synchronize()
    d.copy_(d0); gen.tick_torch(d, dpush)
    t0 = time.perf_counter(); traj = gen.rollout_torch(d, T); torch.cuda.synchronize(); el = time.perf_counter() - t0
    tr = q.from_device(traj, FA.OUT_A)
    report({"workload": name + " rollout", "batch": batch, "ticks": T, "ticks_per_s": batch * T / el, "ms_per_tick": 1e3 * el / T,
            "status_nonzero": int((tr["status"] != 0).sum()), "iters_mean": float((tr["iters_x"] + tr["iters_y"]).mean() / 2),
            "active_mean": float(((tr["active"] & 0xffff) + (tr["active"] >> 16)).mean() / 2)})
def one():
    d.copy_(d0); end_of_step(gen.tick_torch(d, dpush))
one()
el = timed(one, steps)
res = {"workload": name, "batch": world * batch, "steps": steps, "ticks_per_s": world * batch * steps / el, "ms_per_step": 1e3 * el / steps,
       "status_nonzero": int((o["status"] != 0).sum()),
       "iters_mean": float((o["iters_x"] + o["iters_y"]).mean() / 2), "iters_max": int(max(o["iters_x"].max(), o["iters_y"].max())),
       "active_mean": float(((o["active"] & 0xffff) + (o["active"] >> 16)).mean() / 2), "active_max": int(max((o["active"] & 0xffff).max(), (o["active"] >> 16).max()))}
if "--cpu" in sys.argv:
    from oracle import oracle_a as A
    n = 200
    sim = A.SimA(A.gait(kind, float(z["phi"]), float(z["disp_A"])), A.params(kind, C_=int(z["C"]), P=int(z["P"]), F=int(z["F"])), backend="ref" if A.O.have_ref() else "gi")
    # replay the nominal loop: same per-tick QPs as the table rows (cold start each tick, like the reference)
    t0 = time.perf_counter(); sim.run(n); el = time.perf_counter() - t0
    res["cpu_ticks_per_s"] = n / el; res["cpu_backend"] = sim.backend
report(res)
