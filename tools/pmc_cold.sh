# L2 memory-side read bytes (FETCH_SIZE) and wave stall cycles of the wo GEMM [2012 x 2048 x 5120] on a re-used weight set vs a rotation of
# sets larger than the Infinity Cache:  bash tools/pmc_cold.sh   -> gpurun_out/r04_pmc_cold.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in warm cold; do
  for C in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
    tag=${mode}_$(echo $C | cut -c1-8 | tr ' ' '_')
    rm -rf $R/gpurun_out/pmcc_$tag
    if [ $mode = cold ]; then export COLD=24; else unset COLD; fi
    CFGS=4 ONLY_WO=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcc_$tag -- python $R/tools/enc_fwd_gemm_bench.py > $R/gpurun_out/pmcc_$tag.log 2>&1
  done
done
cd $R
python - <<'PY' | tee gpurun_out/r04_pmc_cold.txt
import csv, glob, collections
for mode in ("warm", "cold"):
    for tag in glob.glob(f"gpurun_out/pmcc_{mode}_*/"):
        agg = collections.defaultdict(list)
        for f in glob.glob(tag + "**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "gemm_tile_kernel<64, 128" in row.get("Kernel_Name", ""):
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            v = v[len(v) // 4:]       # skip the warm-up launches
            print(f"{mode:5s} {k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
rm -rf gpurun_out/pmcc_*
