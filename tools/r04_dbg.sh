#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "cross_block or head_transposed or grouped_k" 2>&1 | grep -E "^E|passed|failed" | head -20 | tee $O/r04f_dbg.log
timeout 900 python -m pytest tests/test_round4_paths_gpu.py -q -x 2>&1 | grep -E "^E|passed|failed|Error|fault" | head -30 | tee -a $O/r04f_dbg.log
B="timeout 300 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 3 --warmup 2"
for spec in "sites3:MRB_TOUT_SITES=3;MRB_CKV_BATCH=0" "all:"; do
  label=${spec%%:*}; envs=${spec#*:}
  line=$(env $(echo $envs | tr ';' ' ') $B 2>$O/ab_err.log | python tools/bench_brief.py)
  [ -z "$line" ] && line="FAILED: $(grep -v amdgpu.ids $O/ab_err.log | tail -3 | tr '\n' ' ' | cut -c1-300)"
  echo "$label | $line" | cut -c1-200
done | tee -a $O/r04f_dbg.log
