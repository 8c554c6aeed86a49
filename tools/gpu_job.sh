#!/bin/bash
# ONE parameterised entry for the round's gpurun calls:  gpurun -- 'bash tools/gpu_job.sh <job> [args]'   (logs under gpurun_out/<tag>*)
cd "$(dirname "$0")/.."
job=$1; shift
mkdir -p gpurun_out
B="timeout 600 python bench.py --no-cpu-baseline --no-hbm-kernels"
case "$job" in
  tests)        # the whole -m gpu suite (+ the measured-error log) ; args: extra pytest args
    timeout 1500 python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -15 | tee gpurun_out/tests.log ;;
  bench)        # headline line exactly as the driver runs it
    timeout 900 python bench.py --steps 20 --warmup 5 "$@" 2>/dev/null | tee gpurun_out/bench.json | cut -c1-700 ;;
  quick)        # fixed prompt vs --vary-text (+ any extra flags), short lines
    for extra in "" "--vary-text"; do $B --steps 24 --warmup 8 $extra "$@" 2>/dev/null | python tools/bench_brief.py; done | tee gpurun_out/quick.log ;;
  workloads)    # Charades / ANet / B=4 lines
    for w in "--workload charades" "--workload anet" "--batch-per-gpu 4"; do $B --steps 10 --warmup 3 $w "$@" 2>/dev/null | python tools/bench_brief.py; done | tee gpurun_out/workloads.log ;;
  ab)           # same-box A/B: each argument is "label:ENV=VAL;ENV=VAL" (MRBLIP_LIB=exp_libs/x.so selects a library); the list is run twice
    for rep in 1 2; do for spec in "$@"; do
      label=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
      line=$(env $(echo $envs | tr ';' ' ') $B --steps 24 --warmup 8 2>gpurun_out/ab_err.log | python tools/bench_brief.py)
      [ -z "$line" ] && line="FAILED: $(tail -2 gpurun_out/ab_err.log | tr '\n' ' ' | cut -c1-300)"
      echo "$label | $line"
    done; done | tee gpurun_out/ab.log ;;
  attn)         # standalone attention kernels at the hot-path shapes, per library:  attn exp_libs/base.so "" ...
    for lib in "$@"; do
      echo "== lib: ${lib:-default}"; env ${lib:+MRBLIP_LIB=$lib} ATTN_ONLY=t5enc,t5enc_masked,vit timeout 300 python tools/attn_bench.py 2>&1 | tail -4
      env ${lib:+MRBLIP_LIB=$lib} timeout 200 python tools/attn_vit_bench.py 2>&1 | grep "S=" | grep -v "rel err"
    done | tee gpurun_out/attn.log ;;
  *) echo "unknown job $job"; exit 2 ;;
esac
