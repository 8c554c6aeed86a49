#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab3_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run ct_auto X=1
run ct1 MRB_LORA_TN_CT=1
run ct_auto2 X=1
run ct1_2 MRB_LORA_TN_CT=1
} | tee $O/r06_ab3.txt
