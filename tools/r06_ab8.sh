#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab8_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run default X=1
run thin_rows8 MRB_LORA_THIN_ROWS=8
run qkv_t3 MRB_ENC_QKV_T3=1
run enc_bwd_prefetch32 MRB_ENC_BWD_PREFETCH=32
run grads_at_wi MRB_ENC_GRADS_AT_WI=1
run vit_cfg14_fc2 MRB_VIT_CFG=0,0,14,0
run default2 X=1
} | tee $O/r06_ab8.txt
