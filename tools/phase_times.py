"""Per-phase GPU time of one QVH train step (HIP events at the engine's phase boundaries)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
from mrblip import prompt as P
from mrblip.tokenizer import FixtureTokenizer

dev = torch.device("cuda:0")
LOOKAHEAD = "--lookahead" in sys.argv
wl = bench.WORKLOADS["qvh"]
cfg = EngineConfig.flan_t5_xl_qvh(mean_pool=False)
eng = MrBlipEngine(cfg, RandomSource(dev, seed=1234), dev, lora_init=bench.lora_init_nonzero, seed=42)
eng.training = True
tok = FixtureTokenizer()
repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
samples = bench.synthetic_samples(1, wl["T"], wl["duration"], dev, 1234)
layout = P.build_layout(tok, samples, repl, cfg.num_query, T=wl["T"])
for it in range(5):
    eng.phase_events = [] if it >= 2 else None
    eng.zero_grad()
    eng.forward_backward(samples["video"], layout, backward=True, next_video=samples["video"] if LOOKAHEAD else None)
    vit_done = None
    if LOOKAHEAD and eng.phase_events is not None:  # everything of the look-ahead is enqueued by now: an event on its stream marks its end
        vit_done = torch.cuda.Event(enable_timing=True)
        vit_done.record(eng._vit_stream)
    eng._mark("fwd/bwd done") if eng.phase_events is not None else None
    eng.optimizer_step(lr=3e-4, weight_decay=0.05)
    if eng.phase_events is not None:
        eng._mark("optimizer_step")
        torch.cuda.synchronize()
        ev = eng.phase_events
        rows = [(ev[i][0], ev[i - 1][1].elapsed_time(ev[i][1])) for i in range(1, len(ev))]
        if it == 4:
            tot = sum(t for _, t in rows)
            for n, t in rows:
                print(f"{n:60s} {t:8.3f} ms  {100 * t / tot:5.1f}%")
            print(f"{'total':60s} {tot:8.3f} ms")
            if vit_done is not None:
                print(f"{'look-ahead ViT finished at':60s} {ev[0][1].elapsed_time(vit_done):8.3f} ms after the step's first mark")
