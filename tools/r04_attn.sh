#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_variable_length_gpu.py -q -x -k "attention or attn or variable" 2>&1 | tail -3 | tee $O/r04_attn_tests.log
ATTN_ONLY=t5enc,t5enc_masked python tools/attn_bench.py 2>/dev/null | tee $O/r04_attn_bench.log
for rep in 1 2; do echo "bench | $(timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"; done | tee -a $O/r04_attn_bench.log
