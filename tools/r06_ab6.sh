#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab6_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run all_on X=1
run all_off MRB_QF_LN_BWD_CAST=0 MRB_QF_GELU_BWD_FUSED=0 MRB_QF_KT_FWD=0
run all_on2 X=1
run all_off2 MRB_QF_LN_BWD_CAST=0 MRB_QF_GELU_BWD_FUSED=0 MRB_QF_KT_FWD=0
run all_on3 X=1
} | tee $O/r06_ab6.txt
