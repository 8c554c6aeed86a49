#!/bin/bash
# prefetch workgroup counts per carrier (qkv, o, wi) re-checked with the thin roles aboard
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
for rep in 1 2 3; do for spec in "32,128,32:" "32,128,64:MRB_ENC_PREFETCH=32,128,64" "16,128,32:MRB_ENC_PREFETCH=16,128,32" "32,192,48:MRB_ENC_PREFETCH=32,192,48" "32,96,32:MRB_ENC_PREFETCH=32,96,32" "off:MRB_ENC_PREFETCH=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee $O/r04_pf_counts.log
