#!/bin/bash
# How the Formulation A wave kernel hands out QPs (run-time knobs): ISMPC_A_CLAIM = QPs per work-counter atomic,
# ISMPC_A_STATIC = sixteenths of a launch dealt out statically first.  usage: scripts/claim_sweep.sh "claim:static" ...
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for cfg in ${@:-0:0 0:8 0:12 1:8 1:12 2:12}; do ch=${cfg%%:*}; sq=${cfg##*:}
  for leg in a_walk_C100 config3_walk_C150 a_trot_C160 config4_mc_C200; do for dt in f32 f64; do
    echo "claim=$ch static=$sq $leg $dt $(ISMPC_A_CLAIM=$ch ISMPC_A_STATIC=$sq timeout -k 10 120 python bench.py --only $leg --dtype $dt --no-cpu-baseline --min-region-ms 10 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])")"
  done; done
  ISMPC_A_CLAIM=$ch ISMPC_A_STATIC=$sq python scripts/bench_rollout.py walk_C150 | cut -c1-140 | sed "s/^/claim=$ch static=$sq /"
done
