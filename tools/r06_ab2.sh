#!/bin/bash
# same-box A/B: fused Q-Former forward (csrc/qformer.hip) in the step, with the look-ahead's head leg sized for its longer, narrower phase
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_ab2_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run chain X=1 MRB_QF_FUSED=0
run fused_head2:128 MRB_QF_FUSED=1
run fused_head4:64 MRB_QF_FUSED=1 MRB_VIT_HEAD=4:64
run fused_head5:64 MRB_QF_FUSED=1 MRB_VIT_HEAD=5:64
run fused_head6:64 MRB_QF_FUSED=1 MRB_VIT_HEAD=6:64
run chain2 MRB_QF_FUSED=0
} | tee $O/r06_ab2.txt
