#!/bin/bash
# where the step stands now: phase times + one encoder / decoder layer's launches
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=$R
python tools/phase_times2.py > $O/r04_now_phase.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace -d $O/prof -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r04_now_bench.log 2>&1
cd $R
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_layer.py $DB 12 3 > $O/r04_now_layer.txt
python tools/prof_summary.py $DB $O/r04_now_kernel_stats.txt 0.0 > /dev/null
rm -rf gpurun_out/prof
cat $O/r04_now_phase.txt
