#!/bin/bash
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out
timeout 200 python tools/graph_gap_probe.py > $O/r02_graph_gap.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
MRB_KSPLIT=${MRB_KSPLIT:-1} timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-kernels > $O/prof_bench_h.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python tools/prof_streams.py $DB > $O/r02_streams_h.txt 2>&1
python tools/prof_summary.py $DB $O/r02_kernel_stats_grid_h.txt 0.0 grid > /dev/null
rm -rf gpurun_out/prof
grep -v amdgpu $O/r02_graph_gap.log | tail -3; cat $O/r02_streams_h.txt | cut -c1-330
