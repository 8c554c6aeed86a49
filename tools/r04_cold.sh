#!/bin/bash
# warm vs cold weights for the encoder-forward GEMMs, tile-walk group size
ulimit -c 0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for gm in 8 4 16 32 2; do
echo "== cold, GROUP_M=$gm"; MRB_GROUP_M=$gm CFGS=2,4,8 COLD=24 timeout 300 python tools/enc_fwd_gemm_bench.py
done
echo "== warm, GROUP_M=32"; MRB_GROUP_M=32 CFGS=2,4,8 timeout 300 python tools/enc_fwd_gemm_bench.py
} > gpurun_out/r04_cold_gm.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_cold_gm.log
