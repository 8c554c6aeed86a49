# usage: bash tools/prof_final.sh <tag>   (on the GPU box): bench line + rocprofv3 kernel summary of the same command
TAG=$1
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/kernel_stats_$TAG.txt 0.0 > /dev/null
python tools/prof_summary.py $DB gpurun_out/kernel_stats_grid_$TAG.txt 0.0 grid > /dev/null
python tools/prof_gaps.py $DB 0.5 > gpurun_out/gaps_$TAG.txt
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/bench_under_rocprof_$TAG.json
rm -rf gpurun_out/prof
