#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
for rep in 1 2; do for spec in "tail 5:" "tail 7:MRB_VIT_TAIL=7"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --workload charades --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
  echo "$label | $(env $envs timeout 600 python bench.py --workload anet --no-cpu-baseline --no-hbm-kernels --steps 12 --warmup 4 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
  echo "$label | $(env $envs timeout 600 python bench.py --batch-per-gpu 4 --no-cpu-baseline --no-hbm-kernels --steps 8 --warmup 3 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee gpurun_out/r04_tail2.log
