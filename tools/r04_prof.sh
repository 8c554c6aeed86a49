#!/bin/bash
# rocprofv3 kernel trace of the headline command -> by-kernel / by-grid summaries
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels "$@" > $O/r04_prof_bench.log 2>&1
cd $R
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/r04_kernel_stats.txt 0.0 > /dev/null
python tools/prof_summary.py $DB $O/r04_kernel_stats_by_grid.txt 0.0 grid > /dev/null
python tools/prof_gaps.py $DB 0.5 > $O/r04_gpu_busy.txt
python tools/prof_layer.py $DB 12 3 > $O/r04_prof_layer.txt
grep '"metric"' $O/r04_prof_bench.log > $O/r04_bench_under_rocprof.json
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_rocprofv3_kernel_stats.csv
rm -rf gpurun_out/prof
head -45 $O/r04_kernel_stats_by_grid.txt | cut -c1-200
