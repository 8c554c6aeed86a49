# rocprofv3 kernel trace of any python tool:  bash tools/prof_any.sh <tag> <script.py> [args...]  -> gpurun_out/prof_<tag>_grid.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/prof_summary.py $DB gpurun_out/prof_${tag}_grid.txt 0.0 grid > /dev/null < /dev/null; fi
rm -rf gpurun_out/prof
