#!/bin/bash
# QPs per work-counter atomic (ISMPC_A_CLAIM, run-time knob) for the Formulation A wave kernel
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for ch in ${@:-1 2 4 8 16}; do
  for leg in a_walk_C100 config3_walk_C150 config4_mc_C200; do for dt in f32 f64; do
    echo "claim=$ch $leg $dt $(ISMPC_A_CLAIM=$ch timeout -k 10 120 python bench.py --only $leg --dtype $dt --no-cpu-baseline --min-region-ms 10 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])")"
  done; done
  ISMPC_A_CLAIM=$ch python scripts/bench_rollout.py walk_C150 | cut -c1-140 | sed "s/^/claim=$ch /"
done
