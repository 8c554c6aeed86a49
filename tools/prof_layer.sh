#!/bin/bash
# one encoder layer's launch timeline inside the train step -> gpurun_out/prof_layer.txt   (args: extra bench flags)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-kernels "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log | cut -c1-200
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_layer.py $DB 12 3 > gpurun_out/prof_layer.txt
python tools/prof_summary.py $DB gpurun_out/prof_summary_grid.txt 0.0 grid > /dev/null
rm -rf gpurun_out/prof
