#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_verify_fp32_gpu.py -m gpu -q -x 2>&1 | tail -30 > $O/r02_gputest_i.log
cat $O/r02_gputest_i.log
