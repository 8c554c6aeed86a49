#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_round4_paths_gpu.py tests/test_verify_fp32_gpu.py tests/test_dp_shard_4rank_gpu.py tests/test_input_delivery_gpu.py "tests/test_fullsize_gpu.py::test_c3_batch4_step_equals_four_accumulated_single_clip_steps" -q --tb=short 2>&1 | grep -E "^E |passed|failed|^tests/|Error|^mr-blip" | head -60 | tee $O/r04h_dbg.log
B="timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels"
for rep in 1 2; do for spec in "old:MRB_GEMM_TOUT=0;MRB_CKV_BATCH=0;MRB_ATTN_XS=0;MRB_DEC_PROJ_V2=0" "new:" "new-ckv:MRB_CKV_BATCH=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  for wl in "" "--workload charades"; do
    line=$(env $(echo $envs | tr ';' ' ') $B --steps 24 --warmup 8 $wl 2>$O/ab_err.log | python tools/bench_brief.py)
    [ -z "$line" ] && line="FAILED: $(grep -v amdgpu.ids $O/ab_err.log | tail -3 | tr '\n' ' ' | cut -c1-400)"
    echo "$label | $line" | cut -c1-220
  done
done; done | tee $O/r04h_ab.log
