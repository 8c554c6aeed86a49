#!/bin/bash
# A/B: the LoRA "down" products of the encoder forward inside the GEMMs that consume them
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_round4_paths_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -3 | tee $O/r04_thin_tests.log
for rep in 1 2 3; do for spec in "thin in gemm:" "own launch:MRB_GEMM_THIN=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  for wl in "" "--workload charades"; do
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 $wl 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
  done
done; done | tee $O/r04_thin_ab.log
