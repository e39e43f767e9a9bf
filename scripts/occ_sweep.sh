#!/bin/bash
# residency targets for the Formulation A wave kernel: variants go to build/variants/ (ISMPC_LIB), never in-tree.
# usage: scripts/occ_sweep.sh build   (here, no GPU)   |   scripts/occ_sweep.sh run   (GPU box)
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p build/variants gpurun_out
VARIANTS=("a:3:3:3:3:3" "b:3:3:3:2:2" "c:4:3:3:3:2")   # f32 RL2 : f32 RL3 : f32 RL4 : f64 RL2 : f64 RL3 (RL4: 2)
if [ "$1" = build ]; then
  for v in "${VARIANTS[@]}"; do IFS=: read n a b b4 c d <<< "$v"
    python -c "from quadruped_gait_generation_ismpc_amd import build; build.build(out='build/variants/libismpc_occ_$n.so', flags='-DISMPC_A_OCC_F32_RL2=$a -DISMPC_A_OCC_F32_RL3=$b -DISMPC_A_OCC_F32_RL4=$b4 -DISMPC_A_OCC_F64_RL2=$c -DISMPC_A_OCC_F64_RL3=$d -DISMPC_A_OCC_F64_RL4=2')" &
  done; wait; ls -la build/variants; exit 0
fi
: > gpurun_out/occ_sweep.log
for v in "${VARIANTS[@]}"; do IFS=: read n a b b4 c d <<< "$v"
  lib=$PWD/build/variants/libismpc_occ_$n.so
  for leg in a_walk_C100 config3_walk_C150 a_trot_C160 config4_mc_C200; do for dt in f32 f64; do
    echo "occ f32 $a/$b/$b4 f64 $c/$d $leg $dt $(ISMPC_LIB=$lib timeout -k 10 120 python bench.py --only $leg --dtype $dt --no-cpu-baseline --min-region-ms 10 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e' % d['value'], d['roofline']['kernel_ms'])")" | tee -a gpurun_out/occ_sweep.log
  done; done
done
