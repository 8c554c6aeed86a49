#!/bin/bash
# CU reserve per ViT GEMM (qkv, proj, fc2, fc1): fewer rounds of 256x256 tiles for qkv / fc1 at the price of fewer CUs for the trained clip
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab10_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run default X=1
run q48_f40 MRB_VIT_RESERVE_BY_GEMM=48,-1,-1,40
run q48 MRB_VIT_RESERVE_BY_GEMM=48,-1,-1,-1
run f40 MRB_VIT_RESERVE_BY_GEMM=-1,-1,-1,40
run q48_p72_f40 MRB_VIT_RESERVE_BY_GEMM=48,72,72,40
run uniform40 MRB_VIT_RESERVE=40
run default2 X=1
} | tee $O/r06_ab10.txt
