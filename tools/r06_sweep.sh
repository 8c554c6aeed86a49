#!/bin/bash
# look-ahead knobs re-swept on the round-6 tree (same box, one line per point) -> gpurun_out/r06_sweep.txt
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
run() { local name=$1; shift
  env "$@" python bench.py --steps 25 --warmup 8 --no-cpu-baseline --no-hbm-kernels 2>>$O/r06_sweep_err.log | python tools/bench_brief.py | sed "s/^/$name: /"; }
{
run default X=1
run head3:128 MRB_VIT_HEAD=3:128
run head2:96 MRB_VIT_HEAD=2:96
run head1:128 MRB_VIT_HEAD=1:128
run tail4 MRB_VIT_TAIL=4
run tail6 MRB_VIT_TAIL=6
run tail7 MRB_VIT_TAIL=7
run reserve56 MRB_VIT_RESERVE=56
run reserve72 MRB_VIT_RESERVE=72
run qkvfuse MRB_ENC_QKV_FUSE_NORM=1
run default2 X=1
} | tee $O/r06_sweep.txt
