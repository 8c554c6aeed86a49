#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "three_w_stages or four_wave" 2>&1 | grep -E "^E|passed|failed" | head -12 | tee $O/r04_w3_tests.log
timeout 600 python tools/w3_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/r04_w3_bench.txt
B="timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 24 --warmup 8"
for rep in 1 2; do for spec in "w2:" "w3:MRB_W4_W3=1"; do
  label=${spec%%:*}; envs=${spec#*:}
  line=$(env $envs $B 2>$O/ab_err.log | python tools/bench_brief.py)
  [ -z "$line" ] && line="FAILED: $(grep -v amdgpu.ids $O/ab_err.log | tail -3 | tr '\n' ' ' | cut -c1-300)"
  echo "$label | $line" | cut -c1-170
done; done | tee $O/r04_w3_ab.log
