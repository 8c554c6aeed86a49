#!/bin/bash
# round-2 GPU pass A: the whole -m gpu suite + the bench lines of every BASELINE workload (artefacts under gpurun_out/, copied to profiles/)
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r02_gputest_a.log
python bench.py --steps 10 --warmup 3 > $O/r02_bench_qvh_a.json 2> $O/r02_bench_qvh_a.err
python bench.py --steps 10 --warmup 3 --workload charades --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_charades_a.json 2>> $O/r02_bench_qvh_a.err
python bench.py --steps 6 --warmup 2 --workload anet --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_anet_a.json 2>> $O/r02_bench_qvh_a.err
python bench.py --steps 6 --warmup 2 --batch-per-gpu 4 --no-cpu-baseline --no-hbm-kernels > $O/r02_bench_qvh_b4_a.json 2>> $O/r02_bench_qvh_a.err
tail -5 $O/r02_gputest_a.log; cat $O/r02_bench_qvh_a.json | cut -c1-600; cat $O/r02_bench_charades_a.json | cut -c1-300; cat $O/r02_bench_anet_a.json | cut -c1-300; cat $O/r02_bench_qvh_b4_a.json | cut -c1-300; tail -3 $O/r02_bench_qvh_a.err
