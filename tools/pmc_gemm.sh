# SQ counters of the tile GEMM at one shape: bash tools/pmc_gemm.sh M N K cfg
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmcg_$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcg_$i -- python $R/tools/gemm_one.py $1 $2 $3 $4 6 > $R/gpurun_out/pmcg_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for i in (1, 2):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmcg_{i}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_tile_kernel" in row.get("Kernel_Name", ""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:34s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
    if not agg:
        import subprocess; print(open(f"gpurun_out/pmcg_{i}.log").read()[-1500:])
PY
rm -rf gpurun_out/pmcg_1 gpurun_out/pmcg_2
