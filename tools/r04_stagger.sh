#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
for rep in 1 2; do
echo "== baseline"; python tools/vit_gemm_bench.py 2>/dev/null
echo "== stagger 1 us x group"; MRBLIP_LIB=$PWD/exp_libs/lib_stagger100.so python tools/vit_gemm_bench.py 2>/dev/null
echo "== stagger 3 us x group"; MRBLIP_LIB=$PWD/exp_libs/lib_stagger300.so python tools/vit_gemm_bench.py 2>/dev/null
done | tee gpurun_out/r04_stagger.log
for spec in "baseline:" "stagger100:MRBLIP_LIB=$PWD/exp_libs/lib_stagger100.so" "stagger300:MRBLIP_LIB=$PWD/exp_libs/lib_stagger300.so" "baseline:"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done | tee -a gpurun_out/r04_stagger.log
