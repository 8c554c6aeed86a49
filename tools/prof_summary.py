"""rocprofv3 rocpd .db -> per-kernel summary (calls, total, avg, min, max, % of GPU kernel time), like `--stats`."""
import re
import sqlite3
import sys
import os

NAMELEN = int(os.environ.get("NAMELEN", "90"))


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:NAMELEN + 20]


def main(db, out=None, skip_first_frac=0.0, by_grid=False):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    gcol = next((c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols), None)
    if by_grid and gcol:  # one row per (kernel, grid): separates the GEMM shapes
        rows = cur.execute(f"select name, {gcol}, start, end from kernels order by start").fetchall()
        rows = [(f"{short(n)[:NAMELEN]} grid={g}", s, e) for n, g, s, e in rows]
    else:
        if by_grid:
            print("# no grid column among", cols)
        rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t0 + (t1 - t0) * skip_first_frac
    agg = {}
    for name, s, e in rows:
        if s < cut:
            continue
        a = agg.setdefault(name if (by_grid and gcol) else short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = [f"# {db}: {sum(a[0] for a in agg.values())} kernel dispatches, {tot/1e6:.3f} ms GPU kernel time, wall span {(t1-cut)/1e6:.3f} ms",
             f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:110s} {a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:9.1f} {a[2]/1e3:9.1f} {a[3]/1e3:9.1f} {100*a[1]/tot:6.2f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0,
         by_grid=len(sys.argv) > 4 and sys.argv[4] == "grid")
