# HBM traffic of the dominant kernel (ViT fc1: gemm_w4_kernel<false, 1, false, 4>) from the L2 memory-side counters: separate --pmc passes (no trace domains
# besides --kernel-trace), as MI355X_MICROARCH.md prescribes.  Writes gpurun_out/pmc_fc1.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lookahead > $R/gpurun_out/pmc_$C.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    vals = []
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            # ViT fc1 = the bias+GELU, bf16-out instance of the 4-wave GEMM (qkv: no GELU, fc2: fp32 out + residual)
            if "gemm_w4_kernel<false, 1, false" in name.replace("(bool)0", "false").replace("(bool)1", "true") and row.get("Counter_Name") == c:
                vals.append(float(row["Counter_Value"]))
    out[c] = dict(n=len(vals), mean=(sum(vals) / len(vals) if vals else None), files=len(files))
json.dump(out, open("gpurun_out/pmc_fc1.json", "w"), indent=1)
print(out)
PY
for C in FETCH_SIZE WRITE_SIZE; do find gpurun_out/pmc_$C -name "*.csv" | head -3; rm -rf gpurun_out/pmc_$C; done
