"""VERDICT r3 item 7, host side: what the enqueue loop of one rank costs when the node's cores are shared the way an 8-GPU job shares them.
N bench.py processes (default 4: 4 x ~25 GB of engine state on the one GPU of a test box) run CONCURRENTLY, each pinned with taskset to
1/8 of the box's cores (a rank's share of an 8-rank node whose other cores feed 8 x 8 loader workers), workload Charades-STA (the
configuration whose step is shortest against its launch count: 29 ms, ~1600 launches).  Reported per process: host_enqueue_ms (time to enqueue
one step, from bench.py's own probe) and the ms/step the process saw — the latter is GPU-shared by N and means nothing by itself; the former
must stay below the single-rank step time.   usage: host_stress.py [N=4] [workload=charades]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
WL = sys.argv[2] if len(sys.argv) > 2 else "charades"
GRAPH = sys.argv[3] if len(sys.argv) > 3 else "auto"      # bench.py --graph: 0 eager, 1 / auto: the captured T5 part (round 5)
ncpu = os.cpu_count()
share = max(1, ncpu // 8)
procs = []
for i in range(N):
    cores = f"{i * share}-{(i + 1) * share - 1}"
    cmd = ["taskset", "-c", cores, sys.executable, os.path.join(ROOT, "bench.py"), "--workload", WL, "--steps", "12", "--warmup", "4", "--no-cpu-baseline",
           "--no-hbm-kernels", "--graph", GRAPH]
    procs.append((cores, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)))
print(f"# {N} concurrent bench.py --workload {WL} --graph {GRAPH} processes on one GPU, each pinned to {share} of {ncpu} cores (1/8 of the box)")
worst = 0.0
for cores, p in procs:
    out, _ = p.communicate(timeout=1500)
    line = next((l for l in out.splitlines() if l.startswith("{")), None)
    if line is None:
        print(f"cores {cores}: FAILED (rc {p.returncode})")
        continue
    d = json.loads(line)
    worst = max(worst, d["host_enqueue_ms"])
    print(f"cores {cores}: host_enqueue_ms {d['host_enqueue_ms']:.2f}  launches/step {d['launches_per_step']:.0f}  ms/step seen (GPU shared by {N}) {d['ms_per_step']:.1f}")
print(f"worst host_enqueue_ms under contention: {worst:.2f}")
