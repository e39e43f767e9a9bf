#!/bin/bash
# usage: scripts/sweep.sh  -- kernel time vs batch for the fast (affine) path and the dense MFMA path
for path in affine dense; do for b in 1024 8192 65536; do
  echo -n "path=$path batch=$b: "
  ISMPC_PATH=$path python bench.py --no-cpu-baseline --steps 100 --warmup 10 --batch-per-gpu $b 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ticks/s %.3e  ms/step %.4f kernel_ms %.4f frac %.3f'%(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac']))"
done; done
