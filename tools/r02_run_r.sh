#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02_tall16.log
: > $O
for v in 0 1; do echo "== MRB_SKINNY_TALL16=$v" >> $O; MRB_SKINNY_TALL16=$v timeout 200 python tools/lora_rows_bench.py 2>&1 | grep '^{' | cut -c1-130 >> $O; done
MRB_SKINNY_TALL16=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -3 >> $O
for v in 0 1; do MRB_SKINNY_TALL16=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels 2>/dev/null | cut -c1-200 >> $O; done
for v in 0 1; do MRB_SKINNY_TALL16=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-lookahead 2>/dev/null | cut -c1-200 >> $O; done
cat $O
