#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "ordered or norms or three_w" 2>&1 | grep -E "^E|passed|failed" | head -12 | tee $O/r04_det_tests.log
timeout 900 python tools/determinism_check.py 100 8 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/r04_determinism2.txt
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_train_entry_gpu.py tests/test_dp_gpu.py tests/test_frame_shard_gpu.py -q -x 2>&1 | tail -3 | tee -a $O/r04_det_tests.log
for rep in 1 2; do for spec in "ordered:" "atomic:MRB_NORM_DW_ATOMIC=1"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 24 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee $O/r04_det_ab.log
