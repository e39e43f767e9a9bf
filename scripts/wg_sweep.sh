#!/bin/bash
# workgroup sizes for the Formulation B kernels: every variant is built into build/variants/ and loaded with ISMPC_LIB;
# the in-tree default library is never touched
set -e
mkdir -p build/variants; : > gpurun_out/wg_sweep.log
for w in 1 2 4 8 16; do
  lib=$PWD/build/variants/libismpc_wg$w.so
  python -c "from quadruped_gait_generation_ismpc_amd import build; build.build(out='$lib', flags='-DISMPC_QUAD_WAVES=$w')"
  for leg in shard_b8192 headline; do
    echo "waves=$w $leg $(ISMPC_LIB=$lib timeout -k 10 300 python bench.py --only $leg --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])")" | tee -a gpurun_out/wg_sweep.log
  done
done
