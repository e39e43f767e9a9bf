#!/bin/bash
# A/B of two builds of the library on the bench legs: scripts/ab_lib.sh <variant.so> [leg[:dtype] ...]   (GPU box)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
V=$1; shift
LEGS=${@:-headline shard_b8192 config1_b1024 a_walk_C100:f32 a_walk_C100:f64 config3_walk_C150:f32 config3_walk_C150:f64 a_trot_C160:f32 a_trot_C160:f64 config4_mc_C200:f32 config4_mc_C200:f64}
for spec in $LEGS; do leg=${spec%%:*}; dt=f64; [[ $spec == *:* ]] && dt=${spec##*:}
  for lib in "" "$V"; do
    echo "${lib:-default} $leg $dt $(ISMPC_LIB=$lib timeout -k 10 120 python bench.py --only $leg --dtype $dt --no-cpu-baseline --no-extras --min-region-ms 20 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])")"
  done
done
