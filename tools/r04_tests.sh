#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -E "^E |passed|failed|^FAILED|^tests/.*(Error|assert)" | head -60 | tee $O/r04j_tests.log
