#!/bin/bash
# A/B: role workgroups behind the tiles when the tiles leave CUs idle
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "thin_role or prefetch_workgroups" 2>&1 | tail -3 | tee $O/r04_roles_tests.log
for rep in 1 2 3; do for spec in "roles last:" "roles first:MRB_GEMM_ROLES_LAST=0" "own launch:MRB_GEMM_THIN=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 $wl 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee $O/r04_roles_ab.log
