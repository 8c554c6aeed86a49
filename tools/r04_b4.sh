#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
for rep in 1 2; do for spec in "default:" "thin own launch:MRB_GEMM_THIN=0" "no prefetch:MRB_ENC_PREFETCH=0" "neither:MRB_GEMM_THIN=0 MRB_ENC_PREFETCH=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --batch-per-gpu 4 --steps 8 --warmup 3 --no-cpu-baseline --no-hbm-kernels 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
  echo "$label | $(env $envs timeout 600 python bench.py --workload anet --steps 12 --warmup 4 --no-cpu-baseline --no-hbm-kernels 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee gpurun_out/r04_b4_ab.log
