#!/bin/bash
# Knapsack-loop scheduling variants of the Formulation B lane-group kernels (ISMPC_KF_*: see tick_group_core), side by side.
# build (CPU box):  python -c "from quadruped_gait_generation_ismpc_amd import build as b; b.build(out='build/variants/libismpc_kf0.so', flags='-DISMPC_KF_INLINE=0'); b.build(out='build/variants/libismpc_kfr.so', flags='-DISMPC_KF_ROLLOUT=1'); b.build(out='build/variants/libismpc_kfm.so', flags='-DISMPC_KF_MAIN=1')"
# run (GPU box):    scripts/kf_sweep.sh
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
one() { python bench.py --only $1 --no-cpu-baseline --no-extras --steps 50 --warmup 10 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', d['roofline']['kernel'], 'kernel_us %.2f' % (1e3*d['roofline']['kernel_ms']), 'value %.3e' % d['value'])"; }
for leg in config1_b1024 shard_b8192 headline; do
  one $leg default
  ISMPC_LIB=build/variants/libismpc_kf0.so one $leg inline_kf0
  ISMPC_LIB=build/variants/libismpc_kfm.so one $leg main_kf1
done
for v in "" build/variants/libismpc_kfr.so; do
  ISMPC_LIB=$v python scripts/bench_rollout.py b8192 b65536 | sed "s|^|rollout lib=${v:-default} |"
done
