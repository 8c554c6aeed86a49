#!/bin/bash
# debug of tests/test_dp_gpu.py overlap=1: rank losses at several points
cd "$(dirname "$0")/.."
for ov in 1 0 1; do
MRB_DP_DEBUG=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2965$ov tests/dp_worker.py /tmp/dp_$ov.pt $ov 2>&1 | grep -E "DPDBG|Error|error" 
done
