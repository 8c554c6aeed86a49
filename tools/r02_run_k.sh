#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "deferred" 2>&1 | tail -8 > $O/r02_gputest_k.log
rm -f $O/r02_w4d_bench_k.log
for v in default d_nostore d_late d_late_ns d_noshift; do
  echo "== $v" >> $O/r02_w4d_bench_k.log
  if [ $v == default ]; then timeout 300 python tools/w4d_bench.py >> $O/r02_w4d_bench_k.log 2>&1; else MRBLIP_LIB=exp_libs/lib_$v.so timeout 300 python tools/w4d_bench.py >> $O/r02_w4d_bench_k.log 2>&1; fi
done
tail -6 $O/r02_gputest_k.log; grep -v amdgpu $O/r02_w4d_bench_k.log | grep -v sq8192
