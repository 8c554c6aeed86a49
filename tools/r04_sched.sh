#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
B="timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 24 --warmup 8"
for rep in 1 2; do for spec in "base:" "s20_32_96:MRB_VIT_RESERVE_SCHED=20:32,1000:96;MRB_DEC_GRID=32" "s20_48_80:MRB_VIT_RESERVE_SCHED=20:48,1000:80;MRB_DEC_GRID=48" "s16_40_88:MRB_VIT_RESERVE_SCHED=16:40,1000:88;MRB_DEC_GRID=40" "s24_64_96:MRB_VIT_RESERVE_SCHED=24:64,1000:96" "tail8:MRB_VIT_TAIL=8" "tail3:MRB_VIT_TAIL=3"; do
  label=${spec%%:*}; envs=${spec#*:}
  line=$(env $(echo $envs | tr ';' ' ') $B 2>$O/ab_err.log | python tools/bench_brief.py)
  [ -z "$line" ] && line="FAILED: $(grep -v amdgpu.ids $O/ab_err.log | tail -3 | tr '\n' ' ' | cut -c1-300)"
  echo "$label | $line" | cut -c1-150
done; done | tee $O/r04_sched_ab.log
