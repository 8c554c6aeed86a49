"""Host-side enqueue time of one train step (no synchronisation inside) vs its GPU time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
from mrblip import prompt as P
from mrblip.tokenizer import FixtureTokenizer

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["qvh"]
cfg = EngineConfig.flan_t5_xl_qvh(mean_pool=False)
eng = MrBlipEngine(cfg, RandomSource(dev, seed=1234), dev, lora_init=bench.lora_init_nonzero, seed=42)
tok = FixtureTokenizer()
repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
samples = bench.synthetic_samples(1, wl["T"], wl["duration"], dev, 1234)
layout = P.build_layout(tok, samples, repl, cfg.num_query, T=wl["T"])
v = samples["video"]
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.zero_grad()
    eng.forward_backward(v, layout, backward=True, next_video=v)
    tfb = time.perf_counter()
    eng.optimizer_step(lr=3e-4, weight_decay=0.05)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if it >= 2:
        print(f"host: forward_backward enqueued after {1e3 * (tfb - t0):6.1f} ms, optimizer_step returned after {1e3 * (t1 - t0):6.1f} ms; GPU idle after {1e3 * (t2 - t0):6.1f} ms")
