#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for spec in "pair hashes:MRBLIP_LIB=$PWD/exp_libs/lib_new.so" "before:MRBLIP_LIB=$PWD/exp_libs/lib_old.so"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee gpurun_out/r04_pairhash_ab.log
for spec in "pair hashes:MRBLIP_LIB=$PWD/exp_libs/lib_new.so" "before:MRBLIP_LIB=$PWD/exp_libs/lib_old.so"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs python tools/phase_times2.py 2>/dev/null | head -8 | awk '{print $NF, $(NF-1)}' | tr '\n' ' ')"
done | tee -a gpurun_out/r04_pairhash_ab.log
