#!/bin/bash
# rebuilds the library with different residency targets for the Formulation A wave kernel and times each
set -e
mkdir -p gpurun_out
for occ in 1 2 3 4; do
  ISMPC_HIPCC_FLAGS="-DISMPC_A_WAVE_MINBLOCKS=$occ" python -c "from quadruped_gait_generation_ismpc_amd import build; build.build(force=True)" 2>/dev/null
  for w in walk_C100 walk_C150 trot_C160 mc_C200; do
    echo "occ=$occ $(timeout -k 10 300 python scripts/bench_a.py $w 16384 5)" | tee -a gpurun_out/occ_sweep.log
  done
done
