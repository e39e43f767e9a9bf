#!/bin/bash
# ISMPC_A_WARM=add,drop sweep for the Formulation A wave kernel
mkdir -p gpurun_out; : > gpurun_out/warm_sweep.log
for cfg in 4,6,0 4,6,1 4,6,2 4,6,4 3,6,2 2,6,3 3,4,3; do
  for w in walk_C100 walk_C150 trot_C160 mc_C200; do
    echo "warm=$cfg $(ISMPC_A_WARM=$cfg timeout -k 10 300 python scripts/bench_a.py $w 16384 5 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['workload'], '%.3e' % d['ticks_per_s'], 'iters', '%.1f' % d['iters_mean'], d['iters_max'])")" | tee -a gpurun_out/warm_sweep.log
  done
done
