"""Beam search over a decoder step function — the host-side search of ``BLIP2_MR.generate`` (blip2_mr.py:883-899 calls HF
``t5_model.generate(num_beams=5, max_new_tokens=50, min_length=1, length_penalty=1.0, do_sample=False)``).

Semantics follow HF's beam search for an encoder-decoder (transformers 4.46 ``GenerationMixin._beam_search`` + ``BeamSearchScorer``,
early_stopping=False): scores are sums of log-probabilities; each step ranks the 2K best (beam, token) continuations of every batch
item; an EOS continuation ranked inside the first K becomes a finished hypothesis with score sum_logprobs / generated_len ** length_penalty
(generated_len counts the EOS, not the start token) — one ranked K..2K-1 is dropped; the first K non-EOS continuations are the next
beams; the hypothesis pool keeps the K best; a batch item is done when the pool is full and its worst score is at least
best_running_sum / generated_len ** length_penalty; at the length limit the running beams are added to the pool; the best hypothesis
wins.  HF itself is third-party (not in the reference tree): tests/test_search_cpu.py pins this module against the HF implementation
present in the image (transformers 5.x ``generate`` on a small random T5).
"""
from typing import Optional, Callable, List

import torch


def beam_search(step_fn: Callable[[torch.Tensor], torch.Tensor], batch: int, num_beams: int, max_new_tokens: int, min_length: int = 1,
                length_penalty: float = 1.0, eos_id: int = 1, pad_id: int = 0, start_id: int = 0, trace: Optional[list] = None) -> List[torch.Tensor]:
    """step_fn(seqs [batch * K, L] int64) -> log-probabilities [batch * K, V] (float32, CPU) of the next token.
    A step function with a truthy ``takes_parents`` attribute is called as step_fn(seqs, parents): parents [batch * K] int64 names, for
    every row of ``seqs``, the row of the PREVIOUS call it extends (None on the first call) — what an incremental decoder needs to
    re-order its self-attention K/V cache (HF's ``_reorder_cache(beam_idx)``).
    trace (diagnostics, optional list): receives one entry per step — for every batch item the 2K best candidate scores, best first
    (entry[b][K-1] - entry[b][K] is the margin by which the K-th beam survived the pruning) — and a last entry with every item's final
    pool scores, best first; tests use it to tell a genuine tie from a wrong result.
    Returns one 1-D tensor per batch item: start token, generated tokens, EOS if the hypothesis ended with one."""
    B, K = batch, max(1, int(num_beams))
    seqs = torch.full((B * K, 1), start_id, dtype=torch.long)
    beam_scores = torch.zeros(B, K)
    beam_scores[:, 1:] = -1e9
    pools = [[] for _ in range(B)]          # per item: list of (score, tensor), at most K, any order
    worst = [1e9] * B
    done = [False] * B

    def pool_add(b, hyp, sum_logprobs, generated_len):
        score = sum_logprobs / (generated_len ** length_penalty)
        if len(pools[b]) < K or score > worst[b]:
            pools[b].append((score, hyp))
            if len(pools[b]) > K:
                pools[b].remove(min(pools[b], key=lambda t: t[0]))
            worst[b] = min(s for s, _ in pools[b])

    takes_parents = bool(getattr(step_fn, "takes_parents", False))
    parents = None
    for _ in range(max_new_tokens):
        cur_len = seqs.shape[1]                                  # includes the start token
        lp = (step_fn(seqs, parents) if takes_parents else step_fn(seqs)).float()
        V = lp.shape[-1]
        if cur_len < min_length:
            lp[:, eos_id] = -float("inf")
        cand = (lp + beam_scores.view(B * K, 1)).view(B, K * V)
        top, idx = cand.topk(min(2 * K, K * V), dim=-1)
        if trace is not None:
            trace.append([None if done[b] else top[b].tolist() for b in range(B)])
        new_seqs = torch.full((B * K, cur_len + 1), pad_id, dtype=torch.long)
        new_scores = torch.zeros(B, K)
        parents = torch.arange(B * K)
        for b in range(B):
            if done[b]:                                          # padded beams with score 0, like HF
                new_seqs[b * K:(b + 1) * K, :cur_len] = seqs[b * K:(b + 1) * K]
                continue
            kept = 0
            for rank, (sc, ix) in enumerate(zip(top[b].tolist(), idx[b].tolist())):
                beam, tok = ix // V, ix % V
                if tok == eos_id:
                    if rank >= K:
                        continue
                    pool_add(b, torch.cat([seqs[b * K + beam], torch.tensor([eos_id])]), sc, cur_len)   # cur_len + 1 - 1 generated tokens
                    continue
                new_seqs[b * K + kept, :cur_len] = seqs[b * K + beam]
                new_seqs[b * K + kept, cur_len] = tok
                new_scores[b, kept] = sc
                parents[b * K + kept] = b * K + beam
                kept += 1
                if kept == K:
                    break
            assert kept == K, "fewer than K live continuations among the 2K best (cannot happen: at most K of them are EOS)"
            if len(pools[b]) >= K:                                # early_stopping=False: can any running beam still beat the pool's worst?
                best_running = top[b].max().item()
                done[b] = worst[b] >= best_running / (cur_len ** length_penalty)
        seqs, beam_scores = new_seqs, new_scores
        if all(done):
            break
    out = []
    for b in range(B):
        if not done[b]:                                           # finalize: running beams join the pool
            for k in range(K):
                pool_add(b, seqs[b * K + k], beam_scores[b, k].item(), seqs.shape[1] - 1)
        out.append(max(pools[b], key=lambda t: t[0])[1])
    if trace is not None:
        trace.append([sorted((sc for sc, _ in pools[b]), reverse=True) for b in range(B)])
    return out
