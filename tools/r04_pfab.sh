#!/bin/bash
# A/B: weight prefetch riding in the encoder BACKWARD's dX GEMMs (beside the look-ahead ViT)
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "prefetch" 2>&1 | tail -3 | tee $O/r04_pf_tests.log
for rep in 1 2 3; do for spec in "fwd only:" "bwd32:MRB_ENC_BWD_PREFETCH=32" "bwd64:MRB_ENC_BWD_PREFETCH=64" "bwd128:MRB_ENC_BWD_PREFETCH=128"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "$label | $(env $envs timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 30 --warmup 8 2>/dev/null | python tools/bench_brief.py | cut -c1-110)"
done; done | tee $O/r04_pf_ab3.log
