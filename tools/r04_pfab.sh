#!/bin/bash
# prefetch workgroups in the last, partly empty round of a multi-round GEMM (wi)
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
MRB_GEMM_PF_TAIL=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "thin_role or prefetch_workgroups" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "thin_role or prefetch_workgroups" 2>&1 | tail -2
for rep in 1 2 3; do for spec in "front 32:" "tail 32:MRB_GEMM_PF_TAIL=1" "tail 128:MRB_GEMM_PF_TAIL=1 MRB_ENC_PREFETCH=32,128,128" "tail 64:MRB_GEMM_PF_TAIL=1 MRB_ENC_PREFETCH=32,128,64"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee $O/r04_pf_tail.log
