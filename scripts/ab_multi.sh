#!/bin/bash
# A/B/C... of several builds of the library on bench legs (GPU box): scripts/ab_multi.sh "<lib1.so> <lib2.so> ..." [leg[:dtype] ...]
# ("default" names the in-tree library).  One line per (leg, library): ticks/s and the dominant kernel's time.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
LIBS=$1; shift
LEGS=${@:-config3_walk_C150:f32 config3_walk_C150:f64 config4_mc_C200:f32 config4_mc_C200:f64}
for spec in $LEGS; do leg=${spec%%:*}; dt=f64; [[ $spec == *:* ]] && dt=${spec##*:}
  for lib in $LIBS; do
    L=$lib; [[ $lib == default ]] && L=""
    echo "$leg $dt $(basename $lib) $(ISMPC_LIB=$L timeout -k 10 120 python bench.py --only $leg --dtype $dt --no-cpu-baseline --no-extras --min-region-ms 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])")"
  done
done
