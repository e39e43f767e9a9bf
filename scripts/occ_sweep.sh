#!/bin/bash
# residency targets for the Formulation A wave kernel: variants go to build/variants/ (ISMPC_LIB), never in-tree
set -e
mkdir -p build/variants; : > gpurun_out/occ_sweep.log
for occ in 1 2 3 4; do
  lib=$PWD/build/variants/libismpc_occ$occ.so
  python -c "from quadruped_gait_generation_ismpc_amd import build; build.build(out='$lib', flags='-DISMPC_A_WAVE_MINBLOCKS=$occ')"
  for leg in a_walk_C100 config3_walk_C150 a_trot_C160 config4_mc_C200; do
    echo "occ=$occ $leg $(ISMPC_LIB=$lib timeout -k 10 300 python bench.py --only $leg --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e' % d['value'], d['roofline']['kernel_ms'])")" | tee -a gpurun_out/occ_sweep.log
  done
done
