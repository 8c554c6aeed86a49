#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab9_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run default X=1
run vit_qkv_cfg14 MRB_VIT_CFG=14,0,0,0
run default2 X=1
run vit_qkv_cfg14_2 MRB_VIT_CFG=14,0,0,0
run vit_qkv_fc1_cfg14 MRB_VIT_CFG=14,0,0,14
} | tee $O/r06_ab9.txt
