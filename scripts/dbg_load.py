import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l))
import quadruped_gait_generation_ismpc_amd as q
if order == "torch_first":
    import torch; print("cuda", torch.cuda.is_available(), torch.version.hip); print(maps())
try:
    s = q.MPCSolver(q.reference_plan()); print("create ok")
except Exception as e: print("ERR", e)
print(maps())
if order != "torch_first":
    import torch; print("cuda", torch.cuda.is_available()); print(maps())
    x = torch.ones(4, device="cuda"); print(x.sum().item())
