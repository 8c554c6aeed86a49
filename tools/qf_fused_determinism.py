"""the fused Q-Former forward twice on the same inputs: every saved tensor must come out bit-identical (dropout on)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_qformer_fused_gpu as T
eng = T._engine()
eng.qf_fused = True
F, Tv = 7, 257
g = torch.Generator(device="cuda").manual_seed(3)
img = torch.zeros(F * Tv, 320, dtype=torch.bfloat16, device="cuda")
img.copy_(torch.randn(F * Tv, 320, generator=g, device="cuda").bfloat16())
for training in (False, True):
    for fused in (True, False):
        runs = [T._run(eng, img, F, fused, training) for _ in range(4)]
        bad = [k for k in runs[0] if any(not torch.equal(runs[0][k], r[k]) for r in runs[1:])]
        print("training", training, "fused", fused, "tensors that differ between 4 runs:", bad[:8], len(bad))
import bench
from mrblip import prompt as P
from mrblip.tokenizer import FixtureTokenizer
for rep in range(3):
    out = []
    for fused in (False, True):
        e = T._engine(layers=2, seed=9); e.qf_fused = fused; e.training = True
        tok = FixtureTokenizer(); repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
        s = bench.synthetic_samples(1, 6, 150.0, e.dev, 5); lay = P.build_layout(tok, s, repl, e.cfg.num_query, T=6)
        e.zero_grad(); out.append(e.forward_backward(s["video"], lay, backward=True).item())
    print("train-step losses chain / fused:", out, abs(out[0] - out[1]) / abs(out[0]))
