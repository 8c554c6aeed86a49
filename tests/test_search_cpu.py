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


@pytest.mark.parametrize("seed,eos_boost", [(3, 1.0), (4, 2.5)])
def test_repetition_penalty_and_num_return_sequences_equal_hf_generate(seed, eos_boost):
    """generate(repetition_penalty != 1, num_captions > 1) as the reference passes them to HF (blip2_mr.py:883-899)"""
    from mrblip.search import beam_search

    m = _model(seed, eos_boost)
    B, S = 2, 6
    g = torch.Generator().manual_seed(200 + seed)
    emb = torch.randn(B, S, 32, generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    with torch.no_grad():
        enc = m.encoder(inputs_embeds=emb, attention_mask=mask).last_hidden_state

    def step(seqs):
        rep = seqs.shape[0] // B
        with torch.no_grad():
            out = m(encoder_outputs=(enc.repeat_interleave(rep, 0),), attention_mask=mask.repeat_interleave(rep, 0), decoder_input_ids=seqs)
        return torch.log_softmax(out.logits[:, -1].float(), -1)

    for K, n_ret, pen in ((3, 1, 1.3), (4, 3, 1.0), (5, 2, 2.0), (1, 1, 1.5)):
        with torch.no_grad():
            ref = m.generate(inputs_embeds=emb, attention_mask=mask, do_sample=False, num_beams=K, max_new_tokens=8, min_length=1, length_penalty=1.0,
                             num_return_sequences=n_ret, repetition_penalty=pen, early_stopping=False)
        if K == 1:   # one beam = HF's greedy decoding: the penalty acts on RAW logits (BLIP2_MR.generate routes it the same way)
            from mrblip.search import sample_search

            def step_raw(seqs):
                with torch.no_grad():
                    return m(encoder_outputs=(enc,), attention_mask=mask, decoder_input_ids=seqs).logits[:, -1].float()

            got = sample_search(step_raw, B, 1, 8, min_length=1, repetition_penalty=pen, greedy=True)
        else:
            got = beam_search(step, B, K, 8, min_length=1, length_penalty=1.0, repetition_penalty=pen, num_return=n_ret)
        assert len(got) == B * n_ret == ref.shape[0]
        for i in range(B * n_ret):
            r, mine = ref[i].tolist(), got[i].tolist()
            if 1 in r[1:]:
                r = r[: r.index(1, 1) + 1]
            assert mine == r[:len(mine)] and all(t == 0 for t in r[len(mine):]), (K, n_ret, pen, i, mine, r)


@pytest.mark.parametrize("seed", [5, 6])
def test_nucleus_sampling_equals_hf_generate(seed):
    """generate(use_nucleus_sampling=True, top_p, temperature, repetition_penalty, num_captions) -> HF multinomial sampling: the same
    warps in the same order and ONE multinomial draw per step over all rows, so with the same torch RNG state the sampled ids agree"""
    from mrblip.search import sample_search

    m = _model(seed, 2.0)
    B, S = 2, 6
    g = torch.Generator().manual_seed(300 + seed)
    emb = torch.randn(B, S, 32, generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    with torch.no_grad():
        enc = m.encoder(inputs_embeds=emb, attention_mask=mask).last_hidden_state
    # top_k: the reference never sets it, so HF 4.46's GenerationConfig default (50) applies -> sample_search's default; the small
    # fixture vocabulary needs small k's to make the warp bite (0 = off)
    for n_ret, top_p, temp, pen, top_k in ((1, 0.9, 1.0, 1.0, 0), (3, 0.7, 0.8, 1.2, 7), (2, 1.0, 1.5, 1.0, 3), (2, 0.9, 1.0, 1.0, 50)):
        torch.manual_seed(1000 + seed)
        with torch.no_grad():
            ref = m.generate(inputs_embeds=emb, attention_mask=mask, do_sample=True, num_beams=1, top_p=top_p, top_k=top_k, temperature=temp, max_new_tokens=8,
                             min_length=1, num_return_sequences=n_ret, repetition_penalty=pen)
        R = B * n_ret

        def step(seqs):
            with torch.no_grad():
                out = m(encoder_outputs=(enc.repeat_interleave(n_ret, 0),), attention_mask=mask.repeat_interleave(n_ret, 0), decoder_input_ids=seqs)
            return out.logits[:, -1].float()

        torch.manual_seed(1000 + seed)
        got = sample_search(step, B, n_ret, 8, min_length=1, top_p=top_p, temperature=temp, repetition_penalty=pen, top_k=top_k)
        assert len(got) == R == ref.shape[0]
        for i in range(R):
            r, mine = ref[i].tolist(), got[i].tolist()
            if 1 in r[1:]:
                r = r[: r.index(1, 1) + 1]
            assert mine == r[:len(mine)] and all(t == 0 for t in r[len(mine):]), (n_ret, top_p, temp, pen, top_k, i, mine, r)
