#!/bin/bash
# round-2 GPU pass C: GELU / stagger variants of the ViT GEMM under the tile-phase stamps, the new LoRA row kernels, tests, bench
cd "$(dirname "$0")/.."
O=gpurun_out
for v in stamps stag100 stag300 stag700; do
  echo "== $v" >> $O/r02_w4_stamps_c.log
  MRBLIP_LIB=exp_libs/lib_$v.so timeout 300 python tools/w4_stamps.py >> $O/r02_w4_stamps_c.log 2>&1
done
timeout 300 python tools/lora_rows_bench.py > $O/r02_lora_rows_bench.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/r02_gputest_c.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/r02_bench_qvh_c.json 2> $O/r02_bench_qvh_c.err
MRB_FUSE_NORM_LORA=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_c_nofuse.json 2>> $O/r02_bench_qvh_c.err
grep -v amdgpu.ids $O/r02_w4_stamps_c.log | cut -c1-420; cat $O/r02_lora_rows_bench.log | grep -v amdgpu; tail -8 $O/r02_gputest_c.log; cut -c1-300 $O/r02_bench_qvh_c.json; cut -c1-300 $O/r02_bench_qvh_c_nofuse.json; grep -v amdgpu.ids $O/r02_bench_qvh_c.err | tail -5
