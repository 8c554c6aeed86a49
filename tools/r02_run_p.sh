#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02_w4_dist.log
: > $O
for rep in 1 2; do
for t in default d4444 d6550 d4660 d5443 d6640 d2554; do
  if [ $t == default ]; then L=""; else L="exp_libs/lib_$t.so"; fi
  MRBLIP_LIB=$L timeout 120 python tools/w4_dist_bench.py $t 2>&1 | grep '^{' >> $O
done
done
cat $O
