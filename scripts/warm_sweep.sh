#!/bin/bash
# ISMPC_A_WARM=add,drop,extra sweep for the Formulation A wave kernel (run-time knob: no rebuild)
mkdir -p gpurun_out; : > gpurun_out/warm_sweep.log
for cfg in 4,6,0 4,6,1 4,6,2 3,6,2 2,6,3 3,4,3; do
  for leg in a_walk_C100 config3_walk_C150 a_trot_C160 config4_mc_C200; do
    echo "warm=$cfg $leg $(ISMPC_A_WARM=$cfg timeout -k 10 300 python bench.py --only $leg --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e' % d['value'], 'iters', '%.1f' % d['config']['iterations_per_qp_mean'], d['config']['iterations_per_qp_max'])")" | tee -a gpurun_out/warm_sweep.log
  done
done
