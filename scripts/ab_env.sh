#!/bin/bash
# A/B of an environment knob of the default library on bench legs: scripts/ab_env.sh VAR=value [leg[:dtype] ...]   (GPU box)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
KV=$1; shift
LEGS=${@:-headline shard_b8192 config1_b1024}
for spec in $LEGS; do leg=${spec%%:*}; dt=f64; [[ $spec == *:* ]] && dt=${spec##*:}
  for kv in "_ISMPC_NONE=1" "$KV"; do
    echo "$kv $leg $dt $(env "$kv" timeout -k 10 120 python bench.py --only $leg --dtype $dt --no-cpu-baseline --no-extras --min-region-ms 20 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.3e' % d['value'], 'kernel_ms %.4f' % r['kernel_ms'], 'step_interval_ms', r.get('step_interval_ms'))")"
  done
done
