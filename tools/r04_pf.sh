#!/bin/bash
ulimit -c 0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
QF=1 timeout 600 python tools/prefetch_bench.py > gpurun_out/r04_prefetch_bench_qf.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_prefetch_bench_qf.log
