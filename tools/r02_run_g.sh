#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "k_split or lora" 2>&1 | tail -12 > $O/r02_gputest_g.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py -m gpu -q 2>&1 | tail -12 >> $O/r02_gputest_g.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_g.json 2> $O/r02_bench_qvh_g.err
MRB_KSPLIT=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_g_nosplit.json 2>> $O/r02_bench_qvh_g.err
MRB_VIT_EARLY=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_g_early.json 2>> $O/r02_bench_qvh_g.err
timeout 300 python tools/dec_prof.py 10 > $O/r02_dec_loop.log 2>&1
cat $O/r02_gputest_g.log | tail -20; for f in g g_nosplit g_early; do cut -c1-260 $O/r02_bench_qvh_$f.json; done; grep -v amdgpu $O/r02_dec_loop.log | tail -2; grep -v amdgpu.ids $O/r02_bench_qvh_g.err | tail -5
