#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace -d $O/prof -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-kernels > $O/r04_qfprof_bench.log 2>&1
cd $R
DB=$(find gpurun_out/prof -name "*.db" | head -1)
PROF_ORDERED=1 python tools/prof_layer.py $DB 12 3 > $O/r04_qf_ordered.txt
rm -rf gpurun_out/prof
grep -n "step start" $O/r04_qf_ordered.txt | head -2
