#!/bin/bash
# rebuilds the library with different workgroup sizes for the Formulation B fast kernel and times each
set -e
mkdir -p gpurun_out; : > gpurun_out/wg_sweep.log
for w in 1 2 4 8 16; do
  ISMPC_HIPCC_FLAGS="-DISMPC_AFF_WAVES=$w" python -c "from quadruped_gait_generation_ismpc_amd import build; build.build(force=True)" 2>/dev/null
  for b in 8192 65536; do
    echo "waves=$w batch=$b $(timeout -k 10 300 python bench.py --no-cpu-baseline --batch-per-gpu $b | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")" | tee -a gpurun_out/wg_sweep.log
  done
done
