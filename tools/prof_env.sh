# by-grid kernel summary of a short bench run under the caller's environment:  bash tools/prof_env.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1 < /dev/null
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | cut -c1-120
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/prof_${tag}_grid.txt 0.0 grid > /dev/null < /dev/null
rm -rf gpurun_out/prof
