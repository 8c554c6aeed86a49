# SQ counters of the T5-encoder / ViT attention kernels (tools/attn_bench.py shapes): bash tools/pmc_attn.sh  -> gpurun_out/pmc_attn.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmca_$i
  ATTN_ONLY=${ATTN_ONLY:-t5enc,vit} timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmca_$i -- python $R/tools/attn_bench.py > $R/gpurun_out/pmca_$i.log 2>&1
done
cd $R
python - <<'PY' > gpurun_out/pmc_attn.txt
import csv, glob, collections
for i in (1, 2):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmca_{i}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n = row.get("Kernel_Name", "")
            if "attn_" in n:
                agg[n.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kname, d in agg.items():
        print("==", kname)
        for k, v in d.items():
            print(f"   {k:30s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
    if not agg:
        print(open(f"gpurun_out/pmca_{i}.log").read()[-1500:])
PY
cat gpurun_out/pmc_attn.txt
rm -rf gpurun_out/pmca_1 gpurun_out/pmca_2
