#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "deferred" 2>&1 | tail -15 > $O/r02_gputest_j.log
timeout 300 python tools/w4d_bench.py > $O/r02_w4d_bench.log 2>&1
tail -12 $O/r02_gputest_j.log; grep -v amdgpu $O/r02_w4d_bench.log
