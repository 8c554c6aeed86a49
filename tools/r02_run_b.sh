#!/bin/bash
# round-2 GPU pass B: tile-phase stamps of the ViT GEMM, the tests changed since pass A, the bench line with the timed CPU baseline
cd "$(dirname "$0")/.."
O=gpurun_out
MRBLIP_LIB=exp_libs/lib_stamps.so timeout 300 python tools/w4_stamps.py > $O/r02_w4_stamps.log 2>&1
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_dp_gpu.py tests/test_train_entry_gpu.py -m gpu -q 2>&1 | tail -25 > $O/r02_gputest_b.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/r02_bench_qvh_b.json 2> $O/r02_bench_qvh_b.err
cat $O/r02_w4_stamps.log; tail -8 $O/r02_gputest_b.log; cut -c1-300 $O/r02_bench_qvh_b.json; grep -v amdgpu.ids $O/r02_bench_qvh_b.err | tail -5
