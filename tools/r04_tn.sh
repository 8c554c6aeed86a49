#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for spec in "pair-shared hashes:MRBLIP_LIB=$PWD/exp_libs/lib_new.so" "before:MRBLIP_LIB=$PWD/exp_libs/lib_old.so"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee gpurun_out/r04_lora_tn_ab.log
