#!/bin/bash
# The parameter-sweep bench leg with and without ismpc_sweep_bind (instances sorted by set, one eighth of the sorted batch per XCD): GPU box.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for b in 1 0 1 0; do
  ISMPC_BENCH_SWEEP_BIND=$b python bench.py --only sweep_k64_b65536 --no-cpu-baseline --no-extras --full-line --min-region-ms 20 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bind=$b', '%.4e ticks/s' % d['value'], 'step_ms %.4f' % d['ms_per_step'], r['kernel'], 'kernel_ms %.4f train %.4f' % (r['kernel_ms'], r['kernel_ms_train']), 'deferred fraction', d['config']['z_inequality_active_fraction'])"
done
