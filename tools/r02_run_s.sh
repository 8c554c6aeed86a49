#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02_rowv.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "row_major_v or attention" 2>&1 | tail -4 > $O
timeout 100 python tools/attn_vit_bench.py 2>&1 | grep "S=" >> $O
for v in 1 0 1 0; do MRB_VIT_ROWV=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-kernels 2>/dev/null | cut -c1-210 >> $O; done
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_verify_fp32_gpu.py -m gpu -q -x 2>&1 | tail -3 >> $O
cat $O
