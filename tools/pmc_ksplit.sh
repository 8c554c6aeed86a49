# SQ counters of the T5 encoder's 4-wave-kernel products (and the generic tile they replaced): bash tools/pmc_ksplit.sh -> gpurun_out/pmc_ksplit.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/pmc_ksplit.txt
for W in wi_bwd qkv_bwd wo_bwd qkv_fwd generic_wi_bwd; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU"; do
    i=$((i+1)); rm -rf $R/gpurun_out/pmck_$i
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmck_$i -- python $R/tools/ksplit_one.py $W 8 > $R/gpurun_out/pmck_$i.log 2>&1
  done
  cd $R
  W=$W python - >> gpurun_out/pmc_ksplit.txt <<'PY'
import csv, glob, collections, os
print("==", os.environ["W"])
for i in (1, 2):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmck_{i}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_w4_kernel" in row.get("Kernel_Name", "") or "gemm_tile_kernel" in row.get("Kernel_Name", ""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:34s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
    if not agg:
        print(open(f"gpurun_out/pmck_{i}.log").read()[-800:])
PY
  cd /tmp
done
rm -rf $R/gpurun_out/pmck_1 $R/gpurun_out/pmck_2
