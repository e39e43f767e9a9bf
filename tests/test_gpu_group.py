"""GPU: the multi-GPU C ABI (include/ismpc_group.h) on the ONE GPU of the test box -- groups of one device, where RCCL really runs
(communicator from ncclCommInitAll and from a unique id, the in-place all-gather and the all-gather-v form) and every record must equal
the plain handle's bytes.  tests/cpp/test_group.cpp is the C++ caller; the Python layer (group.py) is checked beside it."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ragged", [False, True])
def test_cpp_group_of_one_equals_the_plain_handle(built_libs, tmp_path, ragged):
    from quadruped_gait_generation_ismpc_amd import workload
    build = os.path.join(ROOT, "tests", "_build"); os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "test_group")
    pkg = os.path.join(ROOT, "quadruped_gait_generation_ismpc_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_group.cpp"), "-o", exe, "-L", pkg, "-lismpc_hip", f"-Wl,-rpath,{pkg}"])
    B, BA = 20000, 3000
    workload.make_batch(100, B, seed=91).tofile(tmp_path / "tick_in.bin")
    w = workload.make_batch_a("walk_C100", BA)
    assert (w["kind"], w["C"], w["F"]) == (1, 100, 3) and abs(w["phi"] - np.pi / 4) < 1e-15 and w["disp_A"] == 0.1      # what the program's defaults build
    w["state"].tofile(tmp_path / "a_state.bin"); w["push"].tofile(tmp_path / "a_push.bin")
    env = dict(os.environ)
    if ragged:
        env["ISMPC_GROUP_FORCE_RAGGED"] = "1"
    res = subprocess.run([exe, str(tmp_path / "tick_in.bin"), str(B), str(tmp_path / "a_state.bin"), str(tmp_path / "a_push.bin"), str(BA)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-3000:])
    last = res.stdout.strip().splitlines()[-1]
    assert last.startswith("OK world=1 rccl=") and int(last.split("rccl=")[1].split()[0]) > 20000 and ("forced" in last) == ragged


def test_python_group_pipeline_and_formulation_a(built_libs):
    """group.py over the same entry points inside a torch process (RCCL = the copy torch already mapped): host entry point, six
    double-buffered device steps read back one step behind, the unique-id form, and a Monte-Carlo Formulation A tick with per-instance
    records in fp32."""
    import torch
    import quadruped_gait_generation_ismpc_amd as q
    from quadruped_gait_generation_ismpc_amd import group as G, workload, formulation_a as FA
    p = q.default_params(N=100)
    plan = q.reference_plan(params=p)
    B = 12000
    tin = workload.make_batch(100, B, seed=92)
    plain = q.MPCSolver(plan, params=p)
    ref = plain.solve_batch(tin)
    assert G.rccl_version() > 20000
    g = G.Group(plan, p, devices=[0])
    assert (g.world, g.n_local, g.rank(0)) == (1, 1, 0) and g.shard(B) == (0, B)
    assert g.solve_batch(tin).tobytes() == ref.tobytes()
    d_in = q.to_device(np.concatenate([tin, tin]))
    g.reserve(B)
    for k in range(6):
        g.step_device(B, [d_in.data_ptr() + 72 * k], k & 1)
        if k >= 1:
            got = q.from_device(g.result_torch(B, 0, (k - 1) & 1).clone(), q.TICK_OUT)
            assert got.tobytes() == np.roll(ref, -(k - 1)).tobytes(), k
    g.sync(); g.close()
    gr = G.Group.from_rank(plan, p, 0, G.unique_id(), 0, 1)
    assert gr.world == 1 and gr.solve_batch(tin[:777]).tobytes() == plain.solve_batch(tin[:777]).tobytes()      # (777 instances: another lane layout than 12 000)
    gr.close(); plain.close()
    # Formulation A, per-instance gait parameters, fp32 solve
    BA = 4096
    inst, push = workload.make_inst_mc(BA)
    plans = [FA.plan(FA.default_gait(k, np.pi / 4, 0.1))[1] for k in (0, 1)]
    pa = FA.default_params(0, C=200, P=400, F=6)
    gen = FA.GaitGenerator(pa, plans[0], precision="f32"); gen.add_plan(plans[1])
    st0 = gen.initial_state(0.88, batch=BA)
    d_st = q.to_device(st0); d_inst = q.to_device(inst); d_push = torch.from_numpy(push.copy()).to("cuda:0")
    o_ref = q.from_device(gen.tick_inst_torch(d_st, d_inst, d_push), FA.OUT_A); s_ref = q.from_device(d_st, FA.STATE_A)
    ga = G.GroupA(pa, plans[0], devices=[0]); assert ga.add_plan(plans[1]) == 1
    ga.set_precision(True)
    st = st0.copy()
    o = ga.tick_batch(st, inst=inst, push=push)
    assert ga.world == 1 and o.tobytes() == o_ref.tobytes() and st.tobytes() == s_ref.tobytes()
    ga.close(); gen.close()
