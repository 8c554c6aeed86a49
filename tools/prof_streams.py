"""rocprofv3 rocpd .db -> per-queue timeline of ONE train step (between the last two seed_bump kernels): for every HIP stream (queue) the
launches, summed kernel time, summed idle gaps between consecutive kernels of that queue, and — for the busiest queue (the main
stream) — the same split per phase (phases are cut at marker kernels: ce_kernel = end of the decoder forward, first attn_bwd_dkv_lds =
start of the encoder backward, ...).  Shows whether a phase is bound by its kernels or by the gaps between them."""
import sqlite3
import sys
from collections import defaultdict


def main(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((c for c in ("stream_id", "queue_id", "queue", "stream") if c in cols), None)
    print("# columns:", cols, "-> queue column:", qcol)
    rows = cur.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "seed_bump" in r[0]]
    a, b = marks[-2], marks[-1]
    seg = rows[a:b]
    t0 = seg[0][1]
    print(f"# step: {len(seg)} launches, wall {(rows[b][1] - t0) / 1e6:.2f} ms")
    byq = defaultdict(list)
    for n, s, e, q in seg:
        byq[q].append((n, s, e))
    for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for _, s, e in ks)
        gaps = sum(max(0, ks[i][1] - ks[i - 1][2]) for i in range(1, len(ks)))
        print(f"queue {q}: {len(ks):5d} launches, first at {(ks[0][1] - t0) / 1e6:7.2f} ms, last ends {(ks[-1][2] - t0) / 1e6:7.2f} ms, kernel time {busy / 1e6:7.2f} ms, gaps {gaps / 1e6:7.2f} ms")
    mainq = max(byq.items(), key=lambda kv: len(kv[1]))[1]
    # phases of the main stream
    cuts = [("start", 0)]
    def first(pred, frm=0):
        for i in range(frm, len(mainq)):
            if pred(mainq[i][0]):
                return i
        return None
    i_enc = first(lambda n: "attn_fwd_lds_kernel<64" in n)
    i_ce = first(lambda n: "ce_kernel" in n)
    i_eb = first(lambda n: "attn_bwd_dkv_lds" in n)
    i_qb = first(lambda n: "colsum" in n)
    for nm, i in (("T5 encoder forward (from its first attention)", i_enc), ("decoder backward (after CE)", i_ce), ("encoder backward (from its first attention bwd)", i_eb),
                  ("t5_proj + Q-Former backward (from colsum)", i_qb)):
        if i is not None:
            cuts.append((nm, i))
    cuts.sort(key=lambda c: c[1])
    cuts.append(("end", len(mainq)))
    print("# main stream by phase (phase = from its marker kernel to the next marker)")
    for (nm, i0), (_, i1) in zip(cuts[:-1], cuts[1:]):
        ks = mainq[i0:i1]
        if not ks:
            continue
        busy = sum(e - s for _, s, e in ks)
        gaps = sum(max(0, ks[i][1] - ks[i - 1][2]) for i in range(1, len(ks)))
        small = sum(1 for _, s, e in ks if e - s < 8000)
        big_gaps = sorted(((ks[i][1] - ks[i - 1][2], ks[i][0][:50]) for i in range(1, len(ks))), reverse=True)[:3]
        print(f"{nm:60s} {len(ks):5d} launches ({small} < 8 us), wall {(ks[-1][2] - ks[0][1]) / 1e6:6.2f} ms = kernels {busy / 1e6:6.2f} + gaps {gaps / 1e6:6.2f} ms"
              f" (avg gap {gaps / max(len(ks) - 1, 1) / 1e3:.2f} us; largest: {[(round(g / 1e3, 1), n) for g, n in big_gaps]})")


if __name__ == "__main__":
    main(sys.argv[1])
