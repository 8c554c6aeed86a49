"""Decoder-only loop (forward + backward of the 12-token T5 decoder against a fixed encoder output) for kernel profiling."""
import os, sys
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
eng.training = True
tok = FixtureTokenizer()
repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
samples = bench.synthetic_samples(1, wl["T"], wl["duration"], dev, 1234)
layout = P.build_layout(tok, samples, repl, cfg.num_query, T=wl["T"])
eng.zero_grad()
eng.forward_backward(samples["video"], layout, backward=True)
torch.cuda.synchronize()
S = layout.S
enc = eng.ws["e_out"] if "e_out" in eng.ws else None
L = eng._layout_dev(layout)
inp = eng.ws["inputs_embeds"]
enc = eng.t5_encoder_forward(inp, 1, S, L["mask"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(N):
    loss, _ = eng.t5_decoder_forward(layout.decoder_input_ids, layout.decoder_mask, enc, 1, S, L["mask"], layout.labels, want_grad=True)
    eng.t5_decoder_backward(enc, 1, S, layout.labels.shape[1], L["mask"], layout.decoder_mask)
e.record(); torch.cuda.synchronize()
print("decoder fwd+bwd: %.3f ms/iter" % (s.elapsed_time(e) / N))
