"""Per-phase GPU time of the QVH train step measured the way bench.py runs it: no host synchronisation between steps (the host stays a step
ahead of the GPU), HIP events at the engine's phase boundaries, read once at the end.  `--no-lookahead`: the main chain alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource  # noqa: E402
from mrblip import prompt as P  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402

dev = torch.device("cuda:0")
LOOKAHEAD = "--no-lookahead" not in sys.argv
WL = next((a.split("=")[1] for a in sys.argv if a.startswith("--workload=")), "qvh")   # qvh | charades | anet
wl = bench.WORKLOADS[WL]
cfg = EngineConfig.flan_t5_xl_qvh(mean_pool=bool(wl.get("mean_pool", False)))
eng = MrBlipEngine(cfg, RandomSource(dev, seed=1234), dev, lora_init=bench.lora_init_nonzero, seed=42)
eng.training = True
tok = FixtureTokenizer()
repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
samples = bench.synthetic_samples(1, wl["T"], wl["duration"], dev, 1234)
layout = P.build_layout(tok, samples, repl, 1 if cfg.mean_pool else cfg.num_query, T=wl["T"])
video = samples["video"]
st = torch.cuda.Stream(device=dev, priority=-1)
per_step = []
with torch.cuda.stream(st):
    for it in range(9):
        eng.phase_events = [] if it >= 5 else None
        eng.zero_grad()
        eng.forward_backward(video, layout, backward=True, next_video=video if LOOKAHEAD else None)
        eng.optimizer_step(lr=3e-4, weight_decay=0.05)
        if eng.phase_events is not None:
            eng._mark("AdamW + operand re-pack")
            per_step.append(eng.phase_events)
torch.cuda.synchronize()
names = [n for n, _ in per_step[0]][1:]
tot = 0.0
for i, n in enumerate(names):
    t = sum(ev[i][1].elapsed_time(ev[i + 1][1]) for ev in per_step) / len(per_step)
    tot += t
    print(f"{n:62s} {t:8.3f} ms")
print(f"{'sum of phases (= step, main stream)':62s} {tot:8.3f} ms   lookahead={LOOKAHEAD} ksplit={os.environ.get('MRB_KSPLIT', '0')}")
