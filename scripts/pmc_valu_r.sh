#!/bin/bash
# VALU instructions per wavefront of ismpc_tick_quad<R> for several horizons (per-sample vs fixed cost)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_r; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for N in 50 100; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/n$N -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --batch-per-gpu 8192 --horizon $N > $OUT/n$N.json 2> $OUT/n$N.err
  python3 $R/scripts/pmc_summary.py $OUT/n$N "ismpc_tick_quad" $OUT/n$N.sum.json | grep -E "valu_insts|salu_insts" | tr -d '\n'; echo " N=$N"
done
