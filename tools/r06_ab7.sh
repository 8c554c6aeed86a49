#!/bin/bash
# per-block CU reserve of the look-ahead's first leg: 64 beside the decoder's few-CU chain, less beside the encoder backward's full-chip kernels
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab7_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run default X=1
run sched14:64,39:32 MRB_VIT_RESERVE_SCHED=14:64,39:32
run sched14:64,39:0 MRB_VIT_RESERVE_SCHED=14:64,39:0
run sched16:64,39:32 MRB_VIT_RESERVE_SCHED=16:64,39:32
run sched12:64,39:32 MRB_VIT_RESERVE_SCHED=12:64,39:32
run sched14:64,39:48 MRB_VIT_RESERVE_SCHED=14:64,39:48
run default2 X=1
} | tee $O/r06_ab7.txt
