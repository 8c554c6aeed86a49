"""rocprofv3 rocpd .db -> GPU busy / idle accounting: how much of the wall span has no kernel running, split by the
length of the kernel that follows the gap (launch-bound phases show up as many small gaps in front of short kernels)."""
import sqlite3
import sys


def main(db, skip_first_frac=0.5):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + (t1 - t0) * skip_first_frac
    rows = [r for r in rows if r[0] >= cut]
    busy = sum(e - s for s, e in rows)
    span = rows[-1][1] - rows[0][0]
    buckets = {}
    last_end = rows[0][1]
    for s, e in rows[1:]:
        gap = max(0, s - last_end)
        d = e - s
        k = "<8us" if d < 8000 else "<32us" if d < 32000 else "<128us" if d < 128000 else ">=128us"
        b = buckets.setdefault(k, [0, 0, 0])
        b[0] += 1; b[1] += d; b[2] += gap
        last_end = max(last_end, e)
    print(f"span {span/1e6:.2f} ms, kernel busy {busy/1e6:.2f} ms ({100*busy/span:.1f}%), idle {100*(1-busy/span):.1f}%")
    for k in ("<8us", "<32us", "<128us", ">=128us"):
        if k in buckets:
            n, d, g = buckets[k]
            print(f"kernels {k:8s}: {n:6d} launches, run {d/1e6:8.2f} ms, idle gap in front {g/1e6:8.2f} ms (avg {g/n/1e3:.2f} us)")


def per_step(db):
    """busy vs wall per optimizer step (steps are delimited by the seed_bump kernel that opens each forward)"""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "seed_bump" in r[0]]
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a:b]
        busy = sum(e - s for _, s, e in seg)
        span = rows[b][1] - seg[0][1]
        small = sum(1 for _, s, e in seg if e - s < 8000)
        print(f"step: {len(seg)} launches ({small} under 8 us), wall {span/1e6:.2f} ms, kernel busy {busy/1e6:.2f} ms ({100*busy/span:.1f}%)")


if __name__ == "__main__":
    per_step(sys.argv[1])
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
