import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd")); sys.path.insert(0, ROOT)
import torch, bench
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
from mrblip import prompt as P
from mrblip.tokenizer import FixtureTokenizer
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["qvh"]
cfg = EngineConfig.flan_t5_xl_qvh(mean_pool=False)
eng = MrBlipEngine(cfg, RandomSource(dev, seed=1234), dev, lora_init=bench.lora_init_nonzero, seed=42)
tok = FixtureTokenizer()
repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
s = bench.synthetic_samples(2, wl["T"], wl["duration"], dev, 1234)
s["query_prompt"] = ["Query: a dog\n", bench.QUERY]
s["relevant_windows"] = ["[[8, 16]]", "[[8, 16], [100, 120]]"]
lay = P.build_layout(tok, s, repl, cfg.num_query, T=wl["T"])
print("S", lay.S, "mask zeros", int((lay.attention_mask == 0).sum()), "labels", lay.labels.shape)
for it in range(3):
    eng.zero_grad()
    loss = eng.forward_backward(s["video"], lay, backward=True, next_video=s["video"])
    eng.optimizer_step(lr=3e-4, weight_decay=0.05)
    print("loss", loss.item(), "grad finite", bool(torch.isfinite(eng.grad).all()), "grad norm", eng.grad.norm().item())
