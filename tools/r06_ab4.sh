#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab4_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run ks_auto X=1
run ks_max2 MRB_ENC_BWD_MAX_KS=2
run ks_max3 MRB_ENC_BWD_MAX_KS=3
run ks_auto2 X=1
run ks_max2_2 MRB_ENC_BWD_MAX_KS=2
} | tee $O/r06_ab4.txt
