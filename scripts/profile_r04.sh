#!/bin/bash
# (ISMPC_GIT_HEAD=<commit> in the environment names the commit in every pmc_*.json: the GPU box has no .git)
# Regenerates EXACTLY the files committed under profiles/r04/ (run on the GPU box, then copy gpurun_out/profiles_r04/* there):
#   <leg>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `python3 bench.py --only <leg>` (one kernel shape per process)
#   <leg>_bench_line.json     the JSON line that same profiled run printed (its roofline.kernel_ms must agree with the CSV)
#   pmc_<key>.json            PMC passes of the same command, separate runs (--pmc only with --kernel-trace), summarised by
#                             scripts/pmc_summary.py; <key> is what bench.py's executed_work() looks up
# usage: scripts/profile_r04.sh [leg ...]        default: every leg bench.py reports
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out/profiles_r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ALL=0; [ $# -eq 0 ] && ALL=1
ONLY_BUILD=0; [[ "$*" == "sweep_build" ]] && { ONLY_BUILD=1; set -- none; }      # `profile_r04.sh sweep_build`: only the sweep-build block at the end
ONLY_FINAL=0; [[ "$*" == "final" ]] && { ONLY_FINAL=1; set -- none; }            # `profile_r04.sh final`: only the driver-command block (a gpurun call is at most 20 minutes: the legs go in several calls)
LEGS=${@:-headline shard_b8192 shard_b16384 shard_b32768 config1_b1024 sweep_k64_b65536 config3_walk_C150 config3_walk_C150:f32 config4_mc_C200 config4_mc_C200:f32}
for spec in $LEGS; do
  [[ $spec == none ]] && continue
  leg=${spec%%:*}; dt=f64; [[ $spec == *:* ]] && dt=${spec##*:}
  case $leg in
    headline)      key=headline_b65536; kern='ismpc_tick_quad_one<'; batch=65536; steps=40 ;;
    shard_b8192)   key=shard_b8192;     kern='ismpc_tick_quad_inline<'; batch=8192;  steps=40 ;;
    shard_b16384)  key=shard_b16384;    kern='ismpc_tick_quad_inline<'; batch=16384; steps=40 ;;
    shard_b32768)  key=shard_b32768;    kern='ismpc_tick_quad_one<'; batch=32768; steps=40 ;;
    config1_b1024) key=config1_b1024;   kern='ismpc_tick_quad_inline<'; batch=1024;  steps=40 ;;
    sweep_k64_b65536) key=sweep_k64_b65536; kern='ismpc_tick_quad<'; batch=65536; steps=40 ;;      # + the MFMA table build: pmc_sweep_gemm.json below
    *)             key=$leg;            kern='ismpc_a_tick_wave<double'; [[ $dt == f32 ]] && kern='ismpc_a_tick_wave<float'; batch=16384; steps=5 ;;   # the Monte-Carlo pre-roll runs in the OTHER precision
  esac
  [[ $dt != f64 ]] && key=${key}_$dt
  CMD="python3 $R/bench.py --only $leg --dtype $dt --no-cpu-baseline --no-extras --full-line --steps $steps --warmup 3 --min-region-ms 5"
  export ISMPC_PROFILES_DIR=$OUT          # the bench line of the profiled run reads the PMC summary written by THIS session, once it exists
  echo "== $key"
  rm -rf $OUT/tmp_$key; mkdir -p $OUT/tmp_$key
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$key/stats -- $CMD > $OUT/${key}_bench_line.json 2> $OUT/tmp_$key/stats.err || { tail -5 $OUT/tmp_$key/stats.err; exit 1; }
  cp $(find $OUT/tmp_$key/stats -name "*kernel_stats.csv" | head -1) $OUT/${key}_kernel_stats.csv
  run() { name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/tmp_$key/$name -- $CMD > $OUT/tmp_$key/$name.json 2> $OUT/tmp_$key/$name.err || { tail -5 $OUT/tmp_$key/$name.err; exit 1; }; }
  # executed floating-point work (roofline.frac): wave-instructions by type; flops = 64 lanes x (ADD + MUL + TRANS + 2 FMA)
  run fp64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F64
  run fp32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU
  run valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
  run busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_F64
  run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU
  run mem TA_TA_BUSY_sum TA_FLAT_LOAD_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  [[ $leg == sweep_k64_b65536 ]] && run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU
  per_step=1; [[ $leg == config4_mc_C200 && "$ISMPC_A_BUCKET" == 1 ]] && per_step=4      # opt-in: one kernel per footstep count 3..6
  python3 $R/scripts/pmc_summary.py $OUT/tmp_$key "$kern" $OUT/pmc_$key.json batch=$batch leg=\"$key\" launches_per_step=$per_step > /dev/null
  grep -E "valu_insts_per_wave|hbm_bytes_per_launch|wave_cycles_per_wave" $OUT/pmc_$key.json | tr -d '\n'; echo
  # the sweep's table build: the batched MFMA product (sweep_gemm), same passes, its own summary
  [[ $leg == sweep_k64_b65536 ]] && python3 $R/scripts/pmc_summary.py $OUT/tmp_$key "sweep_gemm" $OUT/pmc_sweep_gemm.json sets=64 NG=128 > /dev/null
  head -3 $OUT/${key}_kernel_stats.csv | cut -c1-160
  # the un-profiled line of the same leg, now that its PMC summary exists (roofline.achieved = executed flops / isolated launch time)
  timeout -k 10 400 $CMD > $OUT/${key}_bench_line_unprofiled.json 2> $OUT/tmp_$key/unprof.err || { tail -5 $OUT/tmp_$key/unprof.err; exit 1; }
  rm -rf $OUT/tmp_$key/*/*/*agent_info.csv
done

# the sweep's table build at a size that fills the chip (512 sets): kernel stats + the MFMA counters of sweep_gemm
if [ $ALL -eq 1 ] || [ $ONLY_BUILD -eq 1 ]; then
  SB="python3 $R/scripts/sweep_build_probe.py 512"
  rm -rf $OUT/tmp_sweep_build; mkdir -p $OUT/tmp_sweep_build
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_sweep_build/stats -- $SB > $OUT/sweep_build_k512.json 2> $OUT/tmp_sweep_build/stats.err || { tail -5 $OUT/tmp_sweep_build/stats.err; exit 1; }
  cp $(find $OUT/tmp_sweep_build/stats -name "*kernel_stats.csv" | head -1) $OUT/sweep_build_k512_kernel_stats.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/tmp_sweep_build/mfma -- $SB > /dev/null 2> $OUT/tmp_sweep_build/mfma.err || { tail -5 $OUT/tmp_sweep_build/mfma.err; exit 1; }
  python3 $R/scripts/pmc_summary.py $OUT/tmp_sweep_build "sweep_gemm" $OUT/pmc_sweep_gemm_k512.json sets=512 NG=128 > /dev/null
  grep -E "mfma_busy_frac|mfma_f64_flops" $OUT/pmc_sweep_gemm_k512.json | tr -d '\n'; echo; grep sweep_gemm $OUT/sweep_build_k512_kernel_stats.csv | cut -c1-40,200-260
  rm -rf $OUT/tmp_sweep_build/*/*/*agent_info.csv
fi
# the complete line (every leg, host-path extras, cpu_baseline) and the box's VALU rates, when the whole set was regenerated
if [ $ALL -eq 1 ] || [ $ONLY_FINAL -eq 1 ]; then
  export ISMPC_PROFILES_DIR=$OUT
  [ -d $R/profiles/r04 ] && [ $ONLY_FINAL -eq 1 ] && export ISMPC_PROFILES_DIR=$R/profiles/r04      # the summaries of the earlier calls, already copied in place
  # the EXACT driver command, once under rocprofv3 --kernel-trace --stats (its kernel_stats.csv: "dominant kernel time per step <= driver
  # ms_per_step" can be checked from the files) and once plain (the line the driver would record + bench_detail.json)
  rm -rf $OUT/tmp_driver; mkdir -p $OUT/tmp_driver
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_driver/stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_bench_line_profiled.json 2> $OUT/tmp_driver/stats.err || { tail -5 $OUT/tmp_driver/stats.err; exit 1; }
  cp $(find $OUT/tmp_driver/stats -name "*kernel_stats.csv" | head -1) $OUT/driver_cmd_kernel_stats.csv
  rm -rf $OUT/tmp_driver
  timeout -k 10 900 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_bench_line.json 2> $OUT/bench_full.err || { tail -5 $OUT/bench_full.err; exit 1; }
  cp $R/bench_detail.json $OUT/bench_full_line.json
  [ -x $R/build/valu_peak ] && $R/build/valu_peak > $OUT/valu_peak.json
  [ -x $R/build/launch_floor ] && $R/build/launch_floor > $OUT/launch_floor.json      # what a launch costs before it does any work (scripts/micro/launch_floor.hip)
fi
