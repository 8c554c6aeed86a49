import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/mr-blip_amd"); sys.path.insert(0, "/root/repo/tests")
import bench
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
from mrblip import prompt as P
from mrblip.tokenizer import FixtureTokenizer
dev = torch.device("cuda:0")
eng = MrBlipEngine(EngineConfig.flan_t5_xl_qvh(), RandomSource(dev, seed=1234), dev, lora_init=bench.lora_init_nonzero, seed=42)
tok = FixtureTokenizer(); repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
eng.training = False
def rel(a, b): return ((a - b).norm() / b.norm()).item()
for B in (1, 4):
    samples = bench.synthetic_samples(B, 60, 150.0, dev, 1234)
    lay = P.build_layout(tok, samples, repl, eng.cfg.num_query, T=60)
    res = {}
    for name, setup in (("base", {}), ("decproj_off", {"dec_proj_enabled": False}), ("rows256", {"lora_rows_max_m": 256}), ("both", {"dec_proj_enabled": False, "lora_rows_max_m": 256})):
        saved = {k: getattr(eng, k) for k in setup}
        for k, v in setup.items(): setattr(eng, k, v)
        eng.zero_grad()
        l = eng.forward_backward(samples["video"], lay, backward=True).item()
        res[name] = (l, eng.grad.clone())
        for k, v in saved.items(): setattr(eng, k, v)
    for n in ("decproj_off", "rows256", "both"):
        print(f"B={B} {n:12s} vs base: loss rel {abs(res[n][0]-res['base'][0])/abs(res['base'][0]):.2e}  grad relerr {rel(res[n][1], res['base'][1]):.3f}")
    print(f"B={B} rows256 vs both (dec_proj on/off at rows256): grad relerr {rel(res['rows256'][1], res['both'][1]):.3f}")
