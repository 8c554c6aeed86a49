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


def repetition_penalty_(scores: torch.Tensor, seqs: torch.Tensor, penalty: float) -> torch.Tensor:
    """HF RepetitionPenaltyLogitsProcessor (in place): every token id already in a row's sequence (the decoder input ids, start token
    included) gets score * penalty if its score is negative, score / penalty otherwise."""
    if penalty != 1.0:
        sc = scores.gather(1, seqs)
        scores.scatter_(1, seqs, torch.where(sc < 0, sc * penalty, sc / penalty))
    return scores


def top_p_filter_(scores: torch.Tensor, top_p: float, min_tokens_to_keep: int = 1) -> torch.Tensor:
    """HF TopPLogitsWarper (in place): the smallest set of most-probable tokens whose mass reaches top_p survives, the rest -> -inf"""
    if top_p < 1.0:
        sorted_scores, sorted_idx = torch.sort(scores, descending=False)
        cum = sorted_scores.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[..., -min_tokens_to_keep:] = False
        scores.masked_fill_(remove.scatter(1, sorted_idx, remove), -float("inf"))
    return scores


def top_k_filter_(scores: torch.Tensor, top_k: int, min_tokens_to_keep: int = 1) -> torch.Tensor:
    """HF TopKLogitsWarper (in place): everything below the k-th largest logit of a row -> -inf (ties with the k-th survive)"""
    k = min(max(int(top_k), min_tokens_to_keep), scores.shape[-1])
    if k > 0 and k < scores.shape[-1]:
        kth = torch.topk(scores, k)[0][..., -1, None]
        scores.masked_fill_(scores < kth, -float("inf"))
    return scores


def sample_search(step_fn: Callable[[torch.Tensor], torch.Tensor], batch: int, num_return: int, max_new_tokens: int, min_length: int = 1,
                  top_p: float = 0.9, temperature: float = 1.0, repetition_penalty: float = 1.0, eos_id: int = 1, pad_id: int = 0,
                  start_id: int = 0, generator: Optional[torch.Generator] = None, greedy: bool = False,
                  top_k: int = 50) -> List[torch.Tensor]:
    """Multinomial ("nucleus") sampling, the reference's ``generate(use_nucleus_sampling=True, num_beams=1)`` -> HF ``_sample``
    (blip2_mr.py:883-899: do_sample=True, top_p, temperature, repetition_penalty, num_return_sequences=num_captions).  step_fn(seqs
    [batch * num_return, L]) -> RAW next-token logits [rows, V] (the repetition penalty depends on the logits' signs).  Order of the
    warps as in HF: repetition penalty, min-length EOS ban, temperature, top-k, top-p, softmax, one multinomial draw per row (top_k = 50
    is ``GenerationConfig``'s default in the reference's pinned transformers 4.46.1: the reference never overrides it, so HF puts
    TopKLogitsWarper(50) between the temperature and the top-p warps; 0 switches it off); a finished row
    keeps emitting the pad id.  greedy=True: HF's greedy decoding (do_sample=False, num_beams=1) — argmax of the penalised RAW logits,
    no temperature / top-p (HF ignores them without sampling); needed because with ONE beam HF applies the repetition penalty to raw
    logits, whose signs differ from the log-probabilities its beam search penalises.
    Returns batch * num_return sequences (item-major), start token first."""
    R = batch * max(1, int(num_return))
    seqs = torch.full((R, 1), start_id, dtype=torch.long)
    unfinished = torch.ones(R, dtype=torch.bool)
    takes_parents = bool(getattr(step_fn, "takes_parents", False))
    for _ in range(max_new_tokens):
        logits = (step_fn(seqs, None) if takes_parents else step_fn(seqs)).float().clone()
        repetition_penalty_(logits, seqs, float(repetition_penalty))
        if seqs.shape[1] < min_length:
            logits[:, eos_id] = -float("inf")
        if greedy:
            nxt = logits.argmax(dim=-1)
        else:
            if temperature != 1.0:
                logits = logits / temperature
            top_k_filter_(logits, int(top_k))
            top_p_filter_(logits, float(top_p))
            nxt = torch.multinomial(torch.softmax(logits, dim=-1), num_samples=1, generator=generator).squeeze(1)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad_id))
        seqs = torch.cat([seqs, nxt[:, None]], dim=1)
        unfinished = unfinished & (nxt != eos_id)
        if not bool(unfinished.any()):
            break
    out = []
    for r in range(R):
        row = seqs[r]
        hit = (row[1:] == eos_id).nonzero()
        out.append(row[: int(hit[0]) + 2] if hit.numel() else row)
    return out


def beam_search(step_fn: Callable[[torch.Tensor], torch.Tensor], batch: int, num_beams: int, max_new_tokens: int, min_length: int = 1,
                length_penalty: float = 1.0, eos_id: int = 1, pad_id: int = 0, start_id: int = 0, trace: Optional[list] = None,
                repetition_penalty: float = 1.0, num_return: int = 1) -> List[torch.Tensor]:
    """step_fn(seqs [batch * K, L] int64) -> log-probabilities [batch * K, V] (float32, CPU) of the next token.
    A step function with a truthy ``takes_parents`` attribute is called as step_fn(seqs, parents): parents [batch * K] int64 names, for
    every row of ``seqs``, the row of the PREVIOUS call it extends (None on the first call) — what an incremental decoder needs to
    re-order its self-attention K/V cache (HF's ``_reorder_cache(beam_idx)``).
    trace (diagnostics, optional list): receives one entry per step — for every batch item the 2K best candidate scores, best first
    (entry[b][K-1] - entry[b][K] is the margin by which the K-th beam survived the pruning) — and a last entry with every item's final
    pool scores, best first; tests use it to tell a genuine tie from a wrong result.
    repetition_penalty: HF's processor on the log-probabilities of the tokens already in each beam.  num_return (<= num_beams): the
    ``num_return_sequences`` best hypotheses of every item, best first (HF's BeamSearchScorer.finalize).
    Returns num_return 1-D tensors per batch item (item-major): start token, generated tokens, EOS if the hypothesis ended with one."""
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
        if repetition_penalty != 1.0:
            lp = repetition_penalty_(lp.clone(), seqs, float(repetition_penalty))
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
        ranked = sorted(pools[b], key=lambda t: t[0], reverse=True)   # (stable: ties keep pool order, as HF's sort of (score, hyp) does not matter here)
        assert 1 <= num_return <= K, "num_return_sequences has to be smaller or equal to num_beams"
        out.extend(h for _, h in ranked[:num_return])
    if trace is not None:
        trace.append([sorted((sc for sc, _ in pools[b]), reverse=True) for b in range(B)])
    return out
