"""rocprofv3 rocpd .db of tools/dec_prof.py (decoder-only loop) -> ordered launches of ONE decoder layer, forward and backward, on the
main queue, with durations and gaps; then the per-kernel sums of the whole last iteration.   usage: prof_dec_layer.py <db> [layer=12]"""
import sqlite3
import sys
from collections import defaultdict


def main(db, layer=12):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((c for c in ("stream_id", "queue_id", "queue", "stream") if c in cols), None)
    rows = cur.execute(f"select name, start, end, {qcol or '0'}, grid_x from kernels order by start").fetchall()
    ce = [i for i, r in enumerate(rows) if "ce_kernel" in r[0]]
    # last full iteration: from the launch after the previous-but-one ce_kernel's backward ... use [ce[-2], ce[-1]) shifted to start at the row_copy before
    seg = rows[ce[-2]:ce[-1]]
    byq = defaultdict(list)
    for r in seg:
        byq[r[3]].append(r)
    mq = max(byq, key=lambda q: len(byq[q]))
    mainq = byq[mq]
    print(f"# iteration window (ce_kernel to ce_kernel): {(seg[-1][2] - seg[0][1]) / 1e3:.1f} us wall, {len(seg)} launches ({len(mainq)} on the main queue)")

    def window(pred, title):
        idx = [i for i, r in enumerate(mainq) if pred(r[0])]
        if len(idx) <= layer + 1:
            print("# no such layer for", title, len(idx))
            return
        i0, i1 = idx[layer], idx[layer + 1]
        t0, t1 = mainq[i0][1], mainq[i1][1]
        print(f"## {title}: layer window {layer}, {i1 - i0} launches, {(t1 - t0) / 1e3:.1f} us wall")
        prev_end = mainq[i0 - 1][2] if i0 > 0 else t0
        busy = 0
        for n, s, e, q, g in mainq[i0:i1]:
            print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {max(0, s - prev_end) / 1e3:5.1f}  grid {g:8d}  {n[:110]}")
            prev_end = e
            busy += e - s
        print(f"  main-queue kernels {busy / 1e3:.1f} us, gaps {(t1 - t0 - busy) / 1e3:.1f} us")

    window(lambda n: "attn_fwd_kernel<64, 13>" in n, "T5 decoder forward (stand-alone loop)")
    window(lambda n: "attn_bwd_dkv_kernel<64, 13>" in n, "T5 decoder backward (stand-alone loop)")
    for q, ks in byq.items():
        agg = defaultdict(lambda: [0, 0])
        for n, s, e, _, g in ks:
            agg[(n[:90], g)][0] += 1
            agg[(n[:90], g)][1] += e - s
        tot = sum(v[1] for v in agg.values())
        print(f"## queue {q}: {len(ks)} launches, {tot / 1e3:.0f} us of kernels")
        for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
            print(f"  {c:4d} x {t / c / 1e3:7.1f} us = {t / 1e3:8.1f} us  grid {g:8d}  {n}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
