#!/bin/bash
# Instruction-cache behaviour of the Formulation A wave kernels (their code is 60-90 KB; the instruction cache is shared by two CUs): GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | head -40 > $OUT/avail.txt
for spec in config3_walk_C150:f32 config4_mc_C200:f64; do
  leg=${spec%%:*}; dt=${spec##*:}
  CMD="python3 $R/bench.py --only $leg --dtype $dt --no-cpu-baseline --no-extras --full-line --steps 5 --warmup 2 --min-region-ms 5"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $OUT/${leg}_$dt -- $CMD > /dev/null 2> $OUT/${leg}_$dt.err || tail -3 $OUT/${leg}_$dt.err
  python3 - <<PY
import csv,glob,collections
s=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("$OUT/${leg}_$dt/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ismpc_a_tick_wave" in r["Kernel_Name"]:
            s[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("$leg $dt", {k: round(s[k]/n[k]) for k in sorted(s)})
PY
done
cat $OUT/avail.txt | head -20
