#!/bin/bash
# The multi-GPU scaling runs exactly as the round driver launches them (one process per GPU over RCCL/xGMI, weak scaling):
#   bash tools/run_scale.sh [N ...]            default N = 1 2 4 8
#   BATCH_PER_GPU=4 bash tools/run_scale.sh 8  (BASELINE.json configs[2]: QVH, global batch 32 on 8 GPUs)
# Every line carries "collective_selftest" (start-up all-reduce check on the "nccl" = RCCL backend) and "rccl_ranks" (must equal N).
# NCCL_DEBUG=INFO output of rank 0 goes to gpurun_out/scale_N<N>.rccl.log so the ring / channel set-up over xGMI can be read off.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC (the host driver supports nothing else): RCCL's peer mappings fail without it
# Round-6 defaults are the library's own on a one-rank-per-GPU node — listed so that a scaling run cannot inherit a test hook from the shell:
#   MRB_GEMM_THIN_TICKET unset (ticket mode is for ranks SHARING a GPU), MRB_BENCH_SHARE_GPU unset, MRB_GRAPH=auto (captured T5 part only for
#   encoders of <= 512 rows: Charades-STA yes, QVH / ActivityNet no), MRB_THIN_FALLBACK=1 (a thin-role timeout falls back instead of aborting).
unset MRB_GEMM_THIN_TICKET MRB_BENCH_SHARE_GPU MRB_GEMM_THIN MRB_GRAPH
WORKLOAD=${WORKLOAD:-qvh}            # WORKLOAD=charades: BASELINE.json configs[3] (20 frames, mean-pooled frame tokens, captured step)
NS=${@:-1 2 4 8}
B=${BATCH_PER_GPU:-1}
for N in $NS; do
  PORT=$((29500 + N))
  if [ "$N" = "1" ]; then
    python bench.py --gpus 1 --steps 20 --warmup 5 --batch-per-gpu $B --workload $WORKLOAD --no-cpu-baseline --no-hbm-kernels 2>gpurun_out/scale_N1.err | tee gpurun_out/scale_N1.json | python tools/bench_brief.py
  else
    NCCL_DEBUG=INFO NCCL_DEBUG_FILE=gpurun_out/scale_N${N}.rccl.%h.%p.log \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps 20 --warmup 5 --batch-per-gpu $B --workload $WORKLOAD 2>gpurun_out/scale_N${N}.err | tee gpurun_out/scale_N${N}.json | python tools/bench_brief.py
    python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/scale_N${N}.json") if l.startswith("{")][-1])
assert d["n_gpus"] == $N and d.get("rccl_ranks") == $N and d["collective_selftest"]["ok"], d.get("collective_selftest")
assert d.get("thin_role_timeouts", 0) == 0, "a thin-role wait ran out on a GPU this rank owns: see DESIGN section 4.2"
print("N=$N: rccl_ranks", d["rccl_ranks"], "self-test all-reduce", d["collective_selftest"]["allreduce_ms"], "ms for", d["collective_selftest"]["bytes"], "B")
PY
  fi
done
