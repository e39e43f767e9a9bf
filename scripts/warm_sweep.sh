#!/bin/bash
# ISMPC_A_WARM=add,drop,extra sweep for the Formulation A wave kernel (run-time knob: no rebuild)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out; : > gpurun_out/warm_sweep.log
for cfg in ${@:-4,6,0 4,12,0 4,24,0 6,24,0 3,24,0 4,24,1 4,32,0 8,32,0}; do
  for leg in config3_walk_C150 config4_mc_C200 a_trot_C160; do for dt in f32 f64; do
    echo "warm=$cfg $leg $dt $(ISMPC_A_WARM=$cfg timeout -k 10 120 python bench.py --only $leg --dtype $dt --no-cpu-baseline --min-region-ms 10 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e' % d['value'], 'iters', '%.1f' % d['config']['iterations_per_qp_mean'], d['config']['iterations_per_qp_max'], 'bad', d['config']['status_nonzero'])")" | tee -a gpurun_out/warm_sweep.log
  done; done
done
