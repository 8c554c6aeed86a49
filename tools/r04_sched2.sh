#!/bin/bash
# look-ahead schedule knobs re-checked after the encoder forward got shorter
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
for rep in 1 2; do for spec in "default:" "tail 3:MRB_VIT_TAIL=3" "tail 7:MRB_VIT_TAIL=7" "tail 9:MRB_VIT_TAIL=9" "reserve 56:MRB_VIT_RESERVE=56" "reserve 72:MRB_VIT_RESERVE=72" "early:MRB_VIT_EARLY=1"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee $O/r04_sched2.log
