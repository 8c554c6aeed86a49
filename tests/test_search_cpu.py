"""f1 — the beam search of generate (mrblip/search.py) pinned against HF's own implementation in this image (transformers 5.x
``generate`` on a small random T5, the third-party code the reference calls at blip2_mr.py:883-899): same encoder inputs, same model,
num_beams / length_penalty / max_new_tokens grids, EOS made likely so hypotheses finish at different lengths."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))


def _model(seed, eos_boost):
    from transformers import T5Config, T5ForConditionalGeneration

    torch.manual_seed(seed)
    cfg = T5Config(vocab_size=48, d_model=32, d_kv=8, d_ff=64, num_layers=2, num_decoder_layers=2, num_heads=4, feed_forward_proj="gated-gelu",
                   tie_word_embeddings=False, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1, dropout_rate=0.0)
    m = T5ForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(3.0)                       # peaky but not degenerate distributions
        m.lm_head.weight[1] *= eos_boost      # how often EOS wins
    return m


@pytest.mark.parametrize("seed,eos_boost", [(0, 1.0), (1, 2.5), (2, 4.0)])
def test_beam_search_equals_hf_generate(seed, eos_boost):
    from mrblip.search import beam_search

    m = _model(seed, eos_boost)
    B, S = 3, 7
    g = torch.Generator().manual_seed(100 + seed)
    emb = torch.randn(B, S, 32, generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, 5:] = 0
    with torch.no_grad():
        enc = m.encoder(inputs_embeds=emb, attention_mask=mask).last_hidden_state
    n_eos_end = 0
    for K in (1, 2, 5):
        for lp in (1.0, 0.5, 2.0):
            for max_new in (4, 9):
                with torch.no_grad():
                    ref = m.generate(inputs_embeds=emb, attention_mask=mask, do_sample=False, num_beams=K, max_new_tokens=max_new, min_length=1,
                                     length_penalty=lp, num_return_sequences=1, repetition_penalty=1.0, early_stopping=False)

                def step(seqs):
                    rep = seqs.shape[0] // B
                    with torch.no_grad():
                        out = m(encoder_outputs=(enc.repeat_interleave(rep, 0),), attention_mask=mask.repeat_interleave(rep, 0), decoder_input_ids=seqs)
                    return torch.log_softmax(out.logits[:, -1].float(), -1)

                got = beam_search(step, B, K, max_new, min_length=1, length_penalty=lp)
                for b in range(B):
                    r, mine = ref[b].tolist(), got[b].tolist()      # HF right-pads a finished row (pad id 0 — also a real token — or repeated EOS)
                    if 1 in r[1:]:
                        r = r[: r.index(1, 1) + 1]
                    assert mine == r[:len(mine)] and all(t == 0 for t in r[len(mine):]), (seed, K, lp, max_new, b, mine, r)
                    n_eos_end += int(mine[-1] == 1)
    assert eos_boost == 1.0 or n_eos_end > 0      # the boosted runs really exercise finished hypotheses
