cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/profd
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/profd -- python $GRAFT_REPO_ROOT/tools/dec_prof.py 10 > $GRAFT_REPO_ROOT/gpurun_out/profd_run.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/profd_run.log
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/profd -name "*.db" | head -1)
python tools/prof_summary.py $DB gpurun_out/profd_summary_grid.txt 0.6 grid > /dev/null
rm -rf gpurun_out/profd
