cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/prof_summary.txt 0.0 > /dev/null
python tools/prof_summary.py $DB gpurun_out/prof_summary_grid.txt 0.0 grid > /dev/null
python tools/prof_gaps.py $DB 0.5 > gpurun_out/prof_gaps.txt
rm -rf gpurun_out/prof
