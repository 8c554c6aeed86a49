#!/bin/bash
# per-kernel EXCLUSIVE time of the step: rocprofv3 kernel trace of bench.py --no-lookahead (every kernel has the chip to itself or shares it only
# with the gradient side stream) -> gpurun_out/r06_exclusive_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/profx
timeout 600 rocprofv3 --kernel-trace -d $O/profx -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-lookahead > $O/profx_bench.log 2>&1
cd $R
DB=$(find gpurun_out/profx -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/r06_exclusive_kernel_stats.txt 0.0 grid > /dev/null
python tools/prof_gaps.py $DB 0.5 > $O/r06_exclusive_gpu_busy.txt
rm -rf gpurun_out/profx
head -60 $O/r06_exclusive_kernel_stats.txt | cut -c1-64,112-170
tail -6 $O/r06_exclusive_gpu_busy.txt
