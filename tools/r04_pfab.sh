#!/bin/bash
# A/B: weight prefetch riding in the Q-Former's query-chain GEMMs
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_round4_paths_gpu.py -q -x 2>&1 | tail -3 | tee $O/r04_pf_tests.log
for rep in 1 2 3; do for spec in "qf prefetch:" "off:MRB_QF_PREFETCH=0"; do
  label=${spec%%:*}; envs=${spec#*:}
  for wl in "" "--workload charades"; do
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 $wl 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
  done
done; done | tee $O/r04_pf_ab4.log
