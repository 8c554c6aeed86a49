#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/r02_gputest_f.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_f.json 2> $O/r02_bench_qvh_f.err
tail -8 $O/r02_gputest_f.log; cut -c1-300 $O/r02_bench_qvh_f.json; grep -v amdgpu.ids $O/r02_bench_qvh_f.err | tail -5
