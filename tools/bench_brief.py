"""stdin: bench.py's JSON line -> one short line (ms/step, clips/s, MFU, dominant kernel, launches, host enqueue)"""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    ex = r.get("exclusive") or {}
    print(f"{d['config']['workload'][:24]:24s} B={d['config']['batch_per_gpu']} vary={bool(d['config'].get('vary_text'))!s:5s} {d['ms_per_step']:8.2f} ms/step "
          f"{d['value']:7.3f} clips/s mfu {d['step_mfu']:.4f} fc1 {r.get('avg_us')} us ({r.get('frac')}) excl {ex.get('avg_us')} us ({ex.get('frac')}) "
          f"launches {d.get('launches_per_step')} host {d.get('host_enqueue_ms')} ms allocs {d.get('workspace_allocations_in_timed_region')} loss {d['loss']}")
