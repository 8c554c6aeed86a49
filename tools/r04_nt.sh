#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
for rep in 1 2; do
echo "== plain stores"; python tools/vit_gemm_bench.py 2>/dev/null
echo "== nt stores"; MRBLIP_LIB=$PWD/exp_libs/lib_nt.so python tools/vit_gemm_bench.py 2>/dev/null
done | tee gpurun_out/r04_nt_stores.log
for rep in 1 2 3; do for spec in "plain:" "nt stores:MRBLIP_LIB=$PWD/exp_libs/lib_nt.so"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee -a gpurun_out/r04_nt_stores.log
