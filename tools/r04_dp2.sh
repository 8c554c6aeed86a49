#!/bin/bash
# dec_proj streaming kernel: bit-exact test, stand-alone bench per grid, decoder loop, determinism stress, step A/B
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "dec_proj" 2>&1 | tail -5 | tee $O/r04b_dp2_tests.log
for g in 0 128 64; do echo "== MRB_DEC_GRID=$g"; MRB_DEC_GRID=$g timeout 300 python tools/dec_proj_bench.py 2>&1 | grep -v amdgpu.ids; done | tee $O/r04b_dec_proj_bench.txt
echo "== v1"; MRB_DEC_PROJ_V2=0 timeout 300 python tools/dec_proj_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/r04b_dec_proj_bench.txt
timeout 600 python tools/determinism_check.py 150 8 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/r04b_determinism.txt
bash tools/gpu_job.sh ab "v1:MRB_DEC_PROJ_V2=0" "v2_g0:MRB_DEC_GRID=0" "v2_g64:MRB_DEC_GRID=64" "v2_g96:MRB_DEC_GRID=96" "v2_g128:MRB_DEC_GRID=128"
cp $O/ab.log $O/r04b_ab.log
