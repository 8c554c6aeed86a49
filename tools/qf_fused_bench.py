"""Q-Former forward at the QVH shape (60 frames x 257 image tokens, width 1408): one fused launch per layer (csrc/qformer.hip) against the
launch chain, chip to itself, dropout on.   python tools/qf_fused_bench.py [frames]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
sys.path.insert(0, ROOT)
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource  # noqa: E402
from mrblip import ops  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
cfg = EngineConfig(vit_depth=1, d_model=256, d_kv=64, t5_heads=4, d_ff=512, t5_layers=1, t5_dec_layers=1)
eng = MrBlipEngine(cfg, RandomSource(dev, seed=5), dev, seed=11)
eng.training = True
Tv = 257
img = torch.zeros(F * Tv, 1408, dtype=torch.bfloat16, device=dev)
img.copy_(torch.randn(F * Tv, 1408, device=dev).bfloat16())
for fused in (False, True, False, True):
    eng.qf_fused = fused
    for _ in range(3):
        eng.qformer_forward(img, F)
    torch.cuda.synchronize()
    n0 = ops.launch_count
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(10):
        eng.qformer_forward(img, F)
    ev[1].record()
    host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    print(f"frames {F}  fused={int(fused)}  {ev[0].elapsed_time(ev[1]) / 10 * 1e3:8.1f} us per forward   {(ops.launch_count - n0) // 10} launches   host {host * 1e6:.0f} us")
