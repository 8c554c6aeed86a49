#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
rm -f $O/r02_groupm.log
for v in default gm2 gm4 gm16 gm61; do
  echo "== $v" >> $O/r02_groupm.log
  if [ $v == default ]; then timeout 300 python tools/w4d_bench.py >> $O/r02_groupm.log 2>&1; else MRBLIP_LIB=exp_libs/lib_$v.so timeout 300 python tools/w4d_bench.py >> $O/r02_groupm.log 2>&1; fi
done
grep -v amdgpu $O/r02_groupm.log
