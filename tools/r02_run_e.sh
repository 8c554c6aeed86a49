#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 300 python tools/lora_rows_bench.py > $O/r02_lora_rows_bench3.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/r02_gputest_e.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_e.json 2> $O/r02_bench_qvh_e.err
MRB_FUSE_NORM_LORA=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_e_nofuse.json 2>> $O/r02_bench_qvh_e.err
grep -v amdgpu $O/r02_lora_rows_bench3.log; tail -8 $O/r02_gputest_e.log; cut -c1-300 $O/r02_bench_qvh_e.json; cut -c1-300 $O/r02_bench_qvh_e_nofuse.json; grep -v amdgpu.ids $O/r02_bench_qvh_e.err | tail -5
