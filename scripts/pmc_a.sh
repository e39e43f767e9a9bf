#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=${1:-walk_C100}
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/scripts/bench_a.py $W 16384 3 > $OUT/$name.json 2> $OUT/$name.err; }
run valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM && \
run busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
cat $OUT/valu.json
