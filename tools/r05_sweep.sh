#!/bin/bash
# same-box sweeps of step-level knobs: bash tools/r05_sweep.sh "ENV=VAL ENV2=VAL" "..." -> gpurun_out/sweep.log (one bench line each, default first and last)
cd "$(dirname "$0")/.."
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])" >> gpurun_out/sweep.log 2>&1; }
run A=1
for cfg in "$@"; do run "$cfg"; done
run A=1
