#!/bin/bash
# A/B: the encoder prefetch also covers the LoRA K-extension operands
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_round4_paths_gpu.py tests/test_kernels_gpu.py -k "not full_size" -q -x 2>&1 | tail -3 | tee $O/r04_pf_tests.log
for rep in 1 2 3; do for spec in "ext:" "no ext:MRB_ENC_PREFETCH_EXT=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 $wl 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee $O/r04_pf_ab5.log
