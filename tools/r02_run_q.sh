#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02_sweep2.log
: > $O
for rep in 1 2 3; do for cfg in "64 5" "72 5" "56 5" "72 3" "72 8" "88 5"; do set -- $cfg
  v=$(MRB_VIT_RESERVE=$1 MRB_VIT_TAIL=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['exclusive']['avg_us'])")
  echo "reserve=$1 tail=$2 ms_per_step,fc1_in_step_us,fc1_excl_us: $v" >> $O
done; done
sort $O
