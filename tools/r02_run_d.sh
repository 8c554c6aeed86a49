#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
bash tools/dp_debug.sh > $O/r02_dp_debug.log 2>&1
timeout 300 python tools/lora_rows_bench.py > $O/r02_lora_rows_bench2.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "lora or rmsnorm" 2>&1 | tail -15 > $O/r02_gputest_d.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_d.json 2> $O/r02_bench_qvh_d.err
cat $O/r02_dp_debug.log; grep -v amdgpu $O/r02_lora_rows_bench2.log; tail -6 $O/r02_gputest_d.log; cut -c1-300 $O/r02_bench_qvh_d.json; grep -v amdgpu.ids $O/r02_bench_qvh_d.err | tail -5
