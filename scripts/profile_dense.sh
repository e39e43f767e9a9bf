#!/bin/bash
# rocprofv3 summaries of the dense MFMA path of Formulation B (ISMPC_PATH=dense, kept for A/B against the fast path)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_dense; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ISMPC_PATH=dense
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/dense_b8192.json 2> $OUT/stats.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS --output-format csv -d $OUT/pmc -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/pmc.json 2> $OUT/pmc.err || exit 1
ls $OUT/*/*/
