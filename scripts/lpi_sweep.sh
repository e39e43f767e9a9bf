#!/bin/bash
# lanes per instance of the Formulation B lane-group kernels (ISMPC_LPI = 8 | 16 | 32, run-time knob), per batch size
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for lpi in 16 32 8; do
  for leg in config1_b1024 shard_b8192 headline; do
    echo "lpi=$lpi $leg $(ISMPC_LPI=$lpi python bench.py --only $leg --no-cpu-baseline --no-extras --steps 50 --warmup 10 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel'], 'kernel_us %.2f' % (1e3*d['roofline']['kernel_ms']), 'value %.3e' % d['value'])")"
  done
  ISMPC_LPI=$lpi python scripts/bench_rollout.py b8192 b65536 | cut -c1-170 | sed "s/^/lpi=$lpi /"
done
