# HBM traffic of the dominant kernel (ViT fc1: gemm_w4_kernel<false, 1, false, 4>) from the L2 memory-side counters: separate --pmc passes (no trace domains
# besides --kernel-trace), as MI355X_MICROARCH.md prescribes.  Writes gpurun_out/pmc_fc1.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-kernels --no-lookahead > $R/gpurun_out/pmc_$C.log 2>&1
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
# gfx950 corrections (MI355X_MICROARCH.md, HBM section; confirmed here by a calibration copy of known size, tools/pmc_side.sh:
# FETCH_SIZE x 1.999, WRITE_SIZE x 1.000): counter unit KiB; wide streaming reads are counted at half their bytes -> FETCH x 2
M, N, K = 15420, 6144, 1408
fetch = out["FETCH_SIZE"]["mean"] * 1024 * 2 if out["FETCH_SIZE"]["mean"] else None
write = out["WRITE_SIZE"]["mean"] * 1024 if out["WRITE_SIZE"]["mean"] else None
rep = {"kernel": "gemm_w4_kernel<false, 1, false, 4> (256x256x64 tile, 4 waves of 128x128, persistent) ViT fc1 %dx%dx%d, bias + GELU, bf16 out" % (M, N, K),
       "launches_sampled": out["FETCH_SIZE"]["n"], "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": (fetch + write) if fetch and write else None,
       "algorithmic_bytes_per_launch": 2 * (M * K + N * K + M * N) + 4 * N,
       "raw_counters_KiB": out,
       "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_fc1.sh) on bench.py --steps 2 --warmup 1 "
              "--no-lookahead (all 256 CUs); counter unit KiB; gfx950: FETCH_SIZE x 2 (wide streaming reads counted at half their bytes; "
              "calibrated 1.999 with a copy of known size), WRITE_SIZE x 1.  FETCH_SIZE counts L2 misses served by the fabric "
              "(Infinity-Cache hits included), so it is an UPPER bound on HBM reads: the 17.3 MB weight panel and the 43.4 MB activation "
              "panel are re-read by each of the 8 XCD L2s that work on them"}
json.dump(rep, open("gpurun_out/pmc_fc1.json", "w"), indent=1)
print(rep)
PY
for C in FETCH_SIZE WRITE_SIZE; do find gpurun_out/pmc_$C -name "*.csv" | head -3; rm -rf gpurun_out/pmc_$C; done
