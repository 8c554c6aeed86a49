#!/bin/bash
# same-box A/B of the round-6 switches (headline bench, no CPU baseline): one line per variant -> gpurun_out/r06_ab.txt
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab_err.log | python tools/bench_brief.py | sed "s/^/$name: /"
}
{
run default X=1
run wi_w4=0 MRB_ENC_WI_W4=0
run fuse_g=0 MRB_ENC_FUSE_G=0
run kv_merge=0 MRB_QF_KV_BWD_MERGE=0
run default2 X=1
} | tee $O/r06_ab.txt
