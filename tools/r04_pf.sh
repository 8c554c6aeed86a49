#!/bin/bash
ulimit -c 0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ echo "== roles behind the tiles where they leave CUs idle"; timeout 600 python tools/thin_bench.py
echo "== roles always in front"; MRB_GEMM_ROLES_LAST=0 timeout 600 python tools/thin_bench.py; } > gpurun_out/r04_thin_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_thin_bench.log
