#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/r02_gputest_o.log
timeout 400 python tools/gen_bench.py 2>&1 | grep -v amdgpu > $O/r02_gen_bench.log
tail -5 $O/r02_gputest_o.log; cat $O/r02_gen_bench.log
