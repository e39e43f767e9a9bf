#!/bin/bash
# PMC passes for the bench kernel (separate passes, --pmc only with --kernel-trace; see MI355X_MICROARCH.md "rocprofv3 PMC slots")
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=${1:-8192}
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --batch-per-gpu $B > $OUT/$name.json 2> $OUT/$name.err; }
run valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM && \
run busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_F64 && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
ls $OUT/*/*/* | head -30
