"""Evaluation-path timing at the QVH shape: encoder once, then beam-search decoding steps (5 beams, growing prefix) with the
cross-attention K/V cache vs the replicate-per-beam path, and one-position steps against the self-attention K/V cache."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
from mrblip import ops, prompt as P
from mrblip.tokenizer import FixtureTokenizer

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["qvh"]
cfg = EngineConfig.flan_t5_xl_qvh(mean_pool=False)
eng = MrBlipEngine(cfg, RandomSource(dev, seed=1234), dev, lora_init=bench.lora_init_nonzero, seed=42)
eng.training = False
tok = FixtureTokenizer()
repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
samples = bench.synthetic_samples(1, wl["T"], wl["duration"], dev, 1234)
layout = P.build_layout(tok, samples, repl, cfg.num_query, T=wl["T"])
B, S, d, K, STEPS = 1, layout.S, cfg.d_model, 5, 12


def run(cache_on, self_cache=False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fr, img, xv, qb = eng.frames_forward(samples["video"])
    L = eng._layout_dev(layout)
    inp = eng.buf("inputs_embeds", (B * S, d), torch.float32, zero=False)
    ops.row_copy(fr, L["frame_src"], inp, L["frame_dst"]); ops.row_copy(eng.emb, L["emb_src"], inp, L["emb_dst"])
    enc = eng.t5_encoder_forward(inp, B, S, L["mask"])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if cache_on:
        cross = eng.t5_cross_kv(enc, B, S); enc_k, mask_k = enc, L["mask"]
    else:
        cross = None
        enc_k = enc.view(B, S, -1).repeat_interleave(K, 0).reshape(B * K * S, -1).contiguous()
        mask_k = None if L["mask"] is None else L["mask"].repeat_interleave(K, 0).contiguous()
    seqs = torch.zeros(B * K, 1, dtype=torch.long)
    if self_cache:
        state = eng.t5_decode_begin(B * K, STEPS + 1)
        for step in range(STEPS):
            logits = eng.t5_decode_step(state, seqs[:, -1], None if step == 0 else torch.arange(B * K).flip(0), cross, B, mask_k)
            seqs = torch.cat([seqs.flip(0) if step else seqs, logits.argmax(-1).cpu()[:, None]], 1)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        return (t1 - t0) * 1e3, (t2 - t1) * 1e3
    for step in range(STEPS):
        Ld = seqs.shape[1]
        _, logits = eng.t5_decoder_forward(seqs, torch.ones(B * K, Ld, dtype=torch.int32), enc_k, B * K, S, mask_k, labels=None,
                                           cross_cache=cross, cross_batch=B if cross is not None else None)
        nxt = logits.view(B * K, Ld, -1)[:, -1].argmax(-1).cpu()
        seqs = torch.cat([seqs, nxt[:, None]], 1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3


for on, sc in ((True, True), (True, False), (False, False), (True, True), (True, False), (False, False)):
    e, dcd = run(on, sc)
    print(f"self K/V cache {'on ' if sc else 'off'} cross K/V cache {'on ' if on else 'off'}: encode {e:7.1f} ms, {STEPS} decoding steps x {K} beams {dcd:8.1f} ms, clip {e + dcd:8.1f} ms")
