#!/bin/bash
# rocprofv3 kernel-trace summaries for the committed profiles (run on the GPU box): bench.py default, 1 024 and 65 536
# instances, and the Formulation A workloads.  PMC passes are separate (scripts/pmc.sh, scripts/pmc_a.sh).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 8192 1024 65536; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/affine_b$b -- python3 $R/bench.py --no-cpu-baseline --batch-per-gpu $b > $OUT/affine_b$b.json 2> $OUT/affine_b$b.err || exit 1
done
for w in walk_C100 walk_C150 trot_C160 mc_C200; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/formA_$w -- python3 $R/scripts/bench_a.py $w 16384 10 > $OUT/formA_$w.json 2> $OUT/formA_$w.err || exit 1
done
find $OUT -name "*kernel_stats.csv" | head -20
