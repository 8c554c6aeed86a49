#!/bin/bash
# look-ahead tuning sweep on the QVH step (CU reserve x held-back ViT blocks x K-split decoder GEMMs)
cd "$(dirname "$0")/.."
O=gpurun_out
rm -f $O/r02_sweep.log
for ks in 0 1; do for res in 48 64 80; do for tail in 3 5 8; do
  v=$(MRB_KSPLIT=$ks MRB_VIT_RESERVE=$res MRB_VIT_TAIL=$tail timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-hbm-kernels 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['exclusive']['avg_us'])")
  echo "ksplit=$ks reserve=$res tail=$tail ms_per_step,fc1_in_step_us,fc1_excl_us: $v" >> $O/r02_sweep.log
done; done; done
cat $O/r02_sweep.log
