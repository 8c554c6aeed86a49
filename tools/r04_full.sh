#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/r04g_tests.log
B="timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels"
for rep in 1 2; do for spec in "old:MRB_GEMM_TOUT=0;MRB_CKV_BATCH=0;MRB_ATTN_XS=0;MRB_DEC_PROJ_V2=0" "new:" "new-ckv:MRB_CKV_BATCH=0" "new-tout:MRB_GEMM_TOUT=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  for wl in "" "--workload charades"; do
    line=$(env $(echo $envs | tr ';' ' ') $B --steps 24 --warmup 8 $wl 2>$O/ab_err.log | python tools/bench_brief.py)
    [ -z "$line" ] && line="FAILED: $(grep -v amdgpu.ids $O/ab_err.log | tail -3 | tr '\n' ' ' | cut -c1-400)"
    echo "$label | $line" | cut -c1-220
  done
done; done | tee $O/r04g_ab.log
