"""Test infrastructure: the Formulation A oracle (oracle/oracle_a.py, reference qpOASES where built) on many instances at once, in
worker PROCESSES that never import torch or touch the GPU (the calling test process has: it must not fork).  Used by
tests/test_gpu_formulation_a.py to put 64-128 oracle instances beside every full-size batch in seconds.

    results = run(items, backend)      items: list of dicts
        {"kind", "phi", "dA", "C", "P", "F", "state": STATE_A record bytes, "push": (px, py)}                       one pushed tick
        + {"mc": {"step", "ds", "Qf", "height"}, "preroll": n}      per-instance parameters; the oracle also runs the n-tick nominal pre-roll
    each result: {"rv", "u0", "f0", "vel_after", "after": {x, y}, "pre": state after the pre-roll (mc only), "pre_rv_max"}
"""
import os
import pickle
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _solve(items, backend):
    import numpy as np
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle_a as A
    from quadruped_gait_generation_ismpc_amd.formulation_a import STATE_A     # a numpy dtype; importing it loads no native code
    out = []
    for it in items:
        kind = it["kind"]
        if "mc" in it:
            m = it["mc"]
            p = A.params(kind, C_=it["C"], P=it["P"], F=it["F"], step=m["step"], ds=m["ds"], Qf=m["Qf"])
            p.height = m["height"]
        else:
            p = A.params(kind, C_=it["C"], P=it["P"], F=it["F"])
        sim = A.SimA(A.gait(kind, it["phi"], it["dA"]), p, backend=backend)
        res = {}
        if it.get("preroll"):
            pre = sim.run(it["preroll"])
            res["pre_rv_max"] = int(np.abs(pre["rv"]).max())
            res["pre"] = {k: float(sim.state[k]) for k in ("x", "xd", "xz", "y", "yd", "yz", "cur_x", "cur_y")}
            res["pre"].update(fc=int(sim.state["fc"]), j=int(sim.state["j"]))
        st = np.frombuffer(it["state"], dtype=STATE_A)[0]
        sim.load_product_state(st)
        r = sim.tick(tuple(it["push"]))
        res.update(rv=[int(r["rv"][0]), int(r["rv"][1])], u0=[float(v) for v in r["u0"]], f0=[float(v) for v in r["f0"]],
                   vel_after=[float(v) for v in r["vel_after"]], after={"x": float(sim.state["x"]), "y": float(sim.state["y"])})
        out.append(res)
    return out


def run(items, backend, nproc=None):
    nproc = max(1, min(nproc or 16, os.cpu_count() or 1, len(items)))
    chunks = [items[k::nproc] for k in range(nproc)]
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for k, ch in enumerate(chunks):
            with open(os.path.join(td, f"in{k}.pkl"), "wb") as f:
                pickle.dump({"items": ch, "backend": backend}, f)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), os.path.join(td, f"in{k}.pkl"), os.path.join(td, f"out{k}.pkl")],
                                          cwd=ROOT, stderr=subprocess.PIPE, text=True))
        results = [None] * len(items)
        for k, pr in enumerate(procs):
            _, err = pr.communicate(timeout=1500)
            if pr.returncode != 0:
                raise RuntimeError("oracle worker failed: " + (err or "")[-800:])
            with open(os.path.join(td, f"out{k}.pkl"), "rb") as f:
                for j, r in enumerate(pickle.load(f)):
                    results[k + j * nproc] = r
    return results


if __name__ == "__main__":
    with open(sys.argv[1], "rb") as f:
        job = pickle.load(f)
    with open(sys.argv[2], "wb") as f:
        pickle.dump(_solve(job["items"], job["backend"]), f)
