#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "head_transposed or grouped_k or cross_block or gemm" 2>&1 | tail -5 | tee $O/r04e_tests.log
timeout 900 python -m pytest tests/test_round4_paths_gpu.py -q -x 2>&1 | tail -15 | tee -a $O/r04e_tests.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_generate_gpu.py tests/test_variable_length_gpu.py tests/test_frame_shard_gpu.py -q -x 2>&1 | tail -6 | tee -a $O/r04e_tests.log
B="timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels"
for rep in 1 2; do for spec in "old:MRB_GEMM_TOUT=0;MRB_CKV_BATCH=0;MRB_ATTN_XS=0;MRB_DEC_PROJ_V2=0" "new:" "new-ckv:MRB_CKV_BATCH=0" "new-tout:MRB_GEMM_TOUT=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  for wl in "" "--workload charades"; do
    line=$(env $(echo $envs | tr ';' ' ') $B --steps 24 --warmup 8 $wl 2>$O/ab_err.log | python tools/bench_brief.py)
    [ -z "$line" ] && line="FAILED: $(tail -3 $O/ab_err.log | tr '\n' ' ' | cut -c1-400)"
    echo "$label | $line" | cut -c1-220
  done
done; done | tee $O/r04e_ab.log
