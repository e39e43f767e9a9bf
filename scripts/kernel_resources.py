#!/usr/bin/env python3
"""Register / scratch / LDS budget of every gfx950 kernel in the library, from hipcc's own resource analysis
(-Rpass-analysis=kernel-resource-usage; the same numbers llvm-readelf --notes shows in the code object's metadata:
.vgpr_count, .vgpr_spill_count, .private_segment_fixed_size, .group_segment_fixed_size).  No GPU needed.
usage: python scripts/kernel_resources.py [--md profiles/r03/kernel_resources.md] [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quadruped_gait_generation_ismpc_amd", "csrc")
args = sys.argv[1:]
md = None
if "--md" in args:
    md = args[args.index("--md") + 1]; del args[args.index("--md"):args.index("--md") + 2]
rows = []
for src in ("ismpc_hip.hip", "ismpc_a_hip.hip", "ismpc_a_wave_rl2.hip", "ismpc_a_wave_rl3.hip", "ismpc_a_wave_rl4.hip"):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.path.join(td, "x.o")]
                           + os.environ.get("ISMPC_HIPCC_FLAGS", "").split(), capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (?:\S+ )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::|ismpc_a::", "", name); name = re.sub(r"\(.*$", "", name); name = re.sub(r"^void ", "", name)
            cur = {"kernel": name, "file": src}; rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = int(v)
rows = [r for r in rows if not args or any(a in r["kernel"] for a in args)]
hdr = ["kernel", "VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize", "TotalSGPRs", "SGPRs Spill", "LDS Size", "Occupancy"]
out = ["| " + " | ".join(["kernel", "VGPR", "AGPR", "VGPR spill", "scratch B/lane", "SGPR", "SGPR spill", "LDS B/WG", "waves/SIMD"]) + " |", "|" + "---|" * len(hdr)]
for r in rows:
    out.append("| `" + r["kernel"] + "` | " + " | ".join(str(r.get(h, "")) for h in hdr[1:]) + " |")
txt = "\n".join(out)
print(txt)
if md:
    with open(md, "w") as f:
        f.write("# Kernel resources (hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize, -Rpass-analysis=kernel-resource-usage)\n\n"
                "Regenerate: `python scripts/kernel_resources.py --md " + os.path.relpath(md, ROOT) + "`\n\n" + txt + "\n")
