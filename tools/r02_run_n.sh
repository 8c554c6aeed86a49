#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "lora" 2>&1 | tail -8 > $O/r02_gputest_n.log
echo "== VALU" > $O/r02_lora_tn.log; timeout 200 python tools/lora_grads_bench.py 2>&1 | grep -v amdgpu >> $O/r02_lora_tn.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_n.json 2> $O/r02_bench_qvh_n.err
timeout 300 python tools/phase_times2.py 2>&1 | grep -v amdgpu > $O/r02_phases_n.log
tail -6 $O/r02_gputest_n.log; cat $O/r02_lora_tn.log; cut -c1-250 $O/r02_bench_qvh_n.json; cut -c1-250 $O/r02_bench_qvh_n_mfma.json; cat $O/r02_phases_n.log
