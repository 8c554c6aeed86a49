#!/bin/bash
# encoder-forward layer timeline with / without the in-GEMM weight prefetch
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
for spec in "64,128,64:MRB_ENC_PREFETCH=64,128,64" "64,96,32:MRB_ENC_PREFETCH=64,96,32" "32,128,32:MRB_ENC_PREFETCH=32,128,32" "64,192,64:MRB_ENC_PREFETCH=64,192,64" "64,256,48:MRB_ENC_PREFETCH=64,256,48" "128,128,128:MRB_ENC_PREFETCH=128,128,128"; do
  label=${spec%%:*}; envs=${spec#*:}
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof
  env MRB_ENC_PF_PLAN=1 $envs timeout 900 rocprofv3 --kernel-trace -d $O/prof -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-kernels > $O/r04_pfprof_bench.log 2>&1
  cd $R
  DB=$(find gpurun_out/prof -name "*.db" | head -1)
  echo "=== $label"; python tools/prof_layer.py $DB 12 3 | grep -A14 "T5 encoder forward: layer"
  rm -rf gpurun_out/prof
done > $O/r04_pf_layer3.txt 2>&1
grep "===\|gemm_tile\|main-stream" $O/r04_pf_layer3.txt | cut -c1-120
