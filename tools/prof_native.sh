# which PyTorch-native kernels run inside the step (full names):  bash tools/prof_native.sh
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-hbm-kernels > $GRAFT_REPO_ROOT/gpurun_out/native_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/native_bench.err < /dev/null
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, grid_x, start, end from kernels order by start").fetchall() if "grid_x" in [r[1] for r in con.execute("pragma table_info(kernels)")] else []
if not rows:
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    g = next(c for c in ("grid_size_x", "grid_size", "grid_x") if c in cols)
    rows = con.execute(f"select name, {g}, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][2], rows[-1][3]
cut = t1 - 3.2 * 80e6   # the last ~3 steps only (80 ms each; skips model construction and the first-step workspace fills)
agg = collections.defaultdict(lambda: [0, 0.0])
for n, g, s, e in rows:
    if s < cut or ("at::" not in n and "rocclr" not in n):
        continue
    k = (n[:170], g)
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{c:5d} {t:9.1f} us  grid={g}  {n}")
PY
rm -rf gpurun_out/prof
