#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab5_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run fused_cast X=1
run two_launches MRB_QF_LN_BWD_CAST=0
run fused_cast2 X=1
run two_launches2 MRB_QF_LN_BWD_CAST=0
} | tee $O/r06_ab5.txt
