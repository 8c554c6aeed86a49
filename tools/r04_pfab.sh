#!/bin/bash
# which launch carries the next layer's qkv weights: wi (plan 1) or wo (plan 2), re-checked with the roles behind wo's tiles
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
for rep in 1 2 3; do for spec in "plan 1:" "plan 2:MRB_ENC_PF_PLAN=2" "plan 2, 32/128/32 + wo 64:MRB_ENC_PF_PLAN=2 MRB_ENC_PREFETCH=32,128,64"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee $O/r04_pf_plan2.log
