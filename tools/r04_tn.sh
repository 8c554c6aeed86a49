#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "lora or grad or step" 2>&1 | tail -3
python tools/lora_grads_bench.py 2>/dev/null | tail -5 | tee gpurun_out/r04_lora_tn_tr.log
for rep in 1 2 3; do echo "tr reads | $(timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"; done | tee -a gpurun_out/r04_lora_tn_tr.log
