"""rocprofv3 rocpd .db -> the ordered launches of ONE T5 encoder layer of the last profiled train step, forward and backward, on the main
stream (the queue with most launches): start offset, duration, gap to the previous launch of that queue, grid, kernel name — and what ran
on the other queues in the same window (summed per kernel).  Layer k forward = from the k-th encoder attention launch to the next one;
backward = between consecutive attn_bwd_dkv_lds launches.   usage: prof_layer.py <db> [layer=12] [back=2: the step that starts at the back-th last seed_bump]"""
import sqlite3
import sys
from collections import defaultdict


def main(db, layer=12, back=2):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((c for c in ("stream_id", "queue_id", "queue", "stream") if c in cols), None)
    rows = cur.execute(f"select name, start, end, {qcol or '0'}, grid_x from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "seed_bump" in r[0]]
    seg = rows[marks[-back]:marks[-back + 1]] if back > 1 else rows[marks[-1]:]
    byq = defaultdict(list)
    for r in seg:
        byq[r[3]].append(r)
    mq = max(byq, key=lambda q: len(byq[q]))
    mainq = byq[mq]

    def window(pred, title, small_only=False):
        idx = [i for i, r in enumerate(mainq) if pred(r[0]) and (not small_only or r[4] <= 512)]
        if len(idx) <= layer + 1:
            print("# no such layer for", title)
            return
        i0, i1 = idx[layer], idx[layer + 1]
        t0, t1 = mainq[i0][1], mainq[i1][1]
        print(f"## {title}: layer {layer}, {i1 - i0} launches, {(t1 - t0) / 1e3:.1f} us wall")
        prev_end = mainq[i0 - 1][2] if i0 > 0 else t0
        busy = 0
        for n, s, e, q, g in mainq[i0:i1]:
            print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {max(0, s - prev_end) / 1e3:5.1f}  grid {g:8d}  {n[:100]}")
            prev_end = e
            busy += e - s
        print(f"  main-stream kernels {busy / 1e3:.1f} us, gaps {(t1 - t0 - busy) / 1e3:.1f} us")
        other = defaultdict(lambda: [0, 0])
        for q, ks in byq.items():
            if q == mq:
                continue
            for n, s, e, _, g in ks:
                ov = min(e, t1) - max(s, t0)
                if ov > 0:
                    other[(q, n[:80])][0] += 1
                    other[(q, n[:80])][1] += ov
        for (q, n), (c, t) in sorted(other.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"  beside it, queue {q}: {c:3d} x {n}  {t / 1e3:.1f} us overlapping")

    def phase(first_pred, title, until_pred=None):
        """all main-stream launches from the first match to the end of the step (or the first until_pred match), summed per kernel"""
        i0 = next((i for i, r in enumerate(mainq) if first_pred(r[0])), None)
        if i0 is None:
            return
        i1 = next((i for i in range(i0 + 1, len(mainq)) if until_pred and until_pred(mainq[i][0])), len(mainq))
        agg = defaultdict(lambda: [0, 0])
        import os
        if os.environ.get("PROF_ORDERED"):   # every launch of the phase in order (offset, duration, gap)
            t0, prev = mainq[i0][1], mainq[i0][1]
            for n, s, e, q, g in mainq[i0:i1]:
                print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {max(0, s - prev) / 1e3:5.1f}  grid {g:8d}  {n[:90]}")
                prev = e
        for n, s, e, q, g in mainq[i0:i1]:
            agg[(n[:70], g)][0] += 1
            agg[(n[:70], g)][1] += e - s
        tot = sum(v[1] for v in agg.values())
        print(f"## {title}: {i1 - i0} launches, {tot / 1e3:.0f} us of kernels, wall {(mainq[i1 - 1][2] - mainq[i0][1]) / 1e3:.0f} us")
        for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"  {c:4d} x {t / c / 1e3:7.1f} us = {t / 1e3:8.1f} us  grid {g:8d}  {n}")

    phase(lambda n: "colsum" in n, "t5_proj + Q-Former backward (from colsum to the end of the step)")
    phase(lambda n: True, "step start: frames forward (up to the first encoder attention)", lambda n: "attn_fwd_lds_kernel<64" in n)
    window(lambda n: "attn_fwd_lds_kernel<64" in n, "T5 encoder forward")
    window(lambda n: "attn_bwd_dkv_lds" in n, "T5 encoder backward")
    # decoder layers: forward = between consecutive causal self-attention launches (attn_fwd_kernel<64, 13>: LUT | CAUSAL | DROP), backward
    # = between consecutive causal self-attention dK/dV launches
    window(lambda n: "attn_fwd_kernel<64, 13>" in n, "T5 decoder forward")
    window(lambda n: "attn_bwd_dkv_kernel<64, 13>" in n, "T5 decoder backward")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12, int(sys.argv[3]) if len(sys.argv) > 3 else 2)
