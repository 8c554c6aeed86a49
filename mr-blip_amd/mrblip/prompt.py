"""Host-side integer logic of the Mr. BLIP encoder input: timestamp integers, "annoying number" remap, token ids and
the interleave INDEX MAP (which source row lands in which encoder row).  Everything here is exact integer/string
work on the host; the device side is two indexed row copies (ops.row_copy).

Mirrors (behaviour, not code) blip2_mr.py:572-824 (prompt_concatenation, interleave branch), :1497-1608
(find_annoying_numbers*, get_clean_timestamp_tokens_and_embs) and blip2_mr_models/utils.py:388-434.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch


def find_annoying_numbers(tokenizer, range_end: int = 200) -> Tuple[List[int], List[int]]:
    """Integers whose decimal string is split into >1 token; a leading id 3 (lone "▁") puts them in the 2nd list."""
    multi, multi_space = [], []
    for i in range(range_end):
        ids = tokenizer(str(i), add_special_tokens=False)["input_ids"]
        if len(ids) > 1:
            (multi_space if ids[0] == 3 else multi).append(i)
    return multi, multi_space


def annoying_replacement_dict(annoying: Sequence[int]) -> Dict[int, int]:
    bad = set(annoying)
    out = {}
    for i in annoying:
        for j in range(100):
            if i + j not in bad:
                out[i] = i + j
                break
            if i - j not in bad:
                out[i] = i - j
                break
    return out


def seconds_integers(timestamps: torch.Tensor, durations: torch.Tensor, repl: Dict[int, int]):
    """[B,T] fp32 timestamps, [B] durations -> per-sample int lists.  Python round() on the float64 of the fp32 value
    (round-half-even), then the remap — exactly the reference's arithmetic."""
    ts_out, d_out = [], []
    for row, d in zip(timestamps.tolist(), durations.tolist()):
        ints = []
        for x in row:
            r = round(x)
            ints.append(int(repl.get(r, r)))
        dr = round(d)
        ts_out.append(ints)
        d_out.append(int(repl.get(dr, dr)))
    return ts_out, d_out


def seconds_floats(timestamps: torch.Tensor, durations: torch.Tensor):
    """input_time_format "seconds_floats" (utils.py:464-485): each timestamp is round(t, 2) stored back into a float32 tensor and later
    printed with str(tensor_element.item()) (blip2_mr.py:1576-1578) — i.e. the float64 expansion of the float32 value ("22.49" becomes
    "22.489999771118164"; a reference quirk the tokens depend on).  Durations are NOT rounded: str(float32 duration)."""
    import numpy as np

    ts_out = [[str(float(np.float32(round(x, 2)))) for x in row] for row in timestamps.tolist()]
    d_out = [str(float(np.float32(d))) for d in durations.tolist()]
    return ts_out, d_out


def clean_number_tokens(tokenizer, values: Sequence) -> List[List[int]]:
    toks = tokenizer([str(v) for v in values], add_special_tokens=False)["input_ids"]
    return [t[1:] if (len(t) > 0 and t[0] == 3) else t for t in toks]


@dataclass
class EncoderLayout:
    """Index map of the encoder input [B, S, d]:  row (b, s) <- frame-token row, embedding row, or zeros."""
    S: int
    frame_src: torch.Tensor   # int32 [n_f]  rows of the [B*T*n, d] frame-token matrix
    frame_dst: torch.Tensor   # int32 [n_f]  rows of the [B*S, d] encoder input
    emb_src: torch.Tensor     # int32 [n_e]  token ids (rows of the embedding table); -1 = zero row (left padding)
    emb_dst: torch.Tensor     # int32 [n_e]
    attention_mask: torch.Tensor  # int32 [B, S]
    labels: torch.Tensor      # int64 [B, Ld], -100 = ignore
    decoder_input_ids: torch.Tensor  # int64 [B, Ld]
    decoder_mask: torch.Tensor  # int32 [B, Ld]


def shift_right(labels: torch.Tensor, start_id: int = 0, pad_id: int = 0) -> torch.Tensor:
    out = torch.zeros_like(labels)
    out[..., 1:] = labels[..., :-1]
    out[..., 0] = start_id
    return out.masked_fill(out == -100, pad_id)


def video_prompt_strings(samples: dict, repl: Dict[int, int], time_format: str) -> List[str]:
    """The textual video prompt of the NON-interleaved form (blip2_mr.py:783-822): seconds_integers -> ">t_1>t_2>...>t_T>duration"
    (utils.py:388-434, leading ">"), seconds_floats -> "t_1>...>t_T>round(duration)" with str(round(float64(t), 2)) per timestamp
    (utils.py:464-485: no leading ">", and NOT the float32-repr quirk of the interleaved form)."""
    out = []
    if time_format == "seconds_integers":
        ts, durs = seconds_integers(samples["timestamps"], samples["duration"], repl)
        for row, d in zip(ts, durs):
            out.append(">" + ">".join(str(v) for v in row) + ">" + str(d))
    elif time_format == "seconds_floats":
        for row, d in zip(samples["timestamps"].tolist(), samples["duration"].tolist()):
            out.append(">".join(str(round(x, 2)) for x in row) + ">" + str(round(d)))
    else:
        raise ValueError("Invalid input_time_format, please choose from ['seconds_integers', 'seconds_floats']")
    return out


def _plain_layout(tokenizer, samples: dict, repl: Dict[int, int], n_per_frame: int, T: int, max_txt_len: int, no_task_prompt: bool,
                  time_format: str) -> EncoderLayout:
    """interleave_data: False (blip2_mr.py:783-822): [ video_prompt tokens (right padded, mask 0 on the pads) | all T * n frame tokens |
    video_prompt_end | text (right padded) ] — every part tokenised with padding="longest" and its own attention mask."""
    vp = tokenizer(video_prompt_strings(samples, repl, time_format), padding="longest", add_special_tokens=False, truncation=True,
                   max_length=max_txt_len, return_tensors="pt")
    end_tok = tokenizer(list(samples["video_prompt_end"]), padding="longest", add_special_tokens=False, truncation=True,
                        max_length=max_txt_len, return_tensors="pt")
    text = list(samples["query_prompt"]) if no_task_prompt else [q + t for q, t in zip(samples["query_prompt"], samples["task_prompt"])]
    text_tok = tokenizer(text, padding="longest", truncation=True, max_length=max_txt_len, return_tensors="pt")
    B = vp.input_ids.shape[0]
    Lp, Lf, Le, Lt = vp.input_ids.shape[1], T * n_per_frame, end_tok.input_ids.shape[1], text_tok.input_ids.shape[1]
    S = Lp + Lf + Le + Lt
    f_src, f_dst, e_src, e_dst = [], [], [], []
    mask = torch.ones(B, S, dtype=torch.int32)
    for j in range(B):
        for part, off in ((vp, 0), (end_tok, Lp + Lf), (text_tok, Lp + Lf + Le)):
            ids = part.input_ids[j].tolist()
            e_src.extend(int(t) for t in ids)
            e_dst.extend(j * S + off + s for s in range(len(ids)))
            mask[j, off: off + len(ids)] = part.attention_mask[j].int()
        f_src.extend(j * Lf + r for r in range(Lf))
        f_dst.extend(j * S + Lp + r for r in range(Lf))
    ans = tokenizer(list(samples["relevant_windows"]), padding="longest", truncation=True, max_length=max_txt_len, return_tensors="pt")
    labels = ans.input_ids.masked_fill(ans.input_ids == tokenizer.pad_token_id, -100)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)  # noqa: E731
    return EncoderLayout(S=S, frame_src=i32(f_src), frame_dst=i32(f_dst), emb_src=i32(e_src), emb_dst=i32(e_dst), attention_mask=mask,
                         labels=labels, decoder_input_ids=shift_right(labels), decoder_mask=ans.attention_mask.int())


def build_layout(tokenizer, samples: dict, repl: Dict[int, int], n_per_frame: int, T: int, max_txt_len: int = 200,
                 no_task_prompt: bool = False, time_format: str = "seconds_integers", interleave: bool = True) -> EncoderLayout:
    """[ f_0(n) | ts_0 | f_1(n) | ts_1 | ... | ">" | duration | video_prompt_end | text(right padded) ], shorter video
    prompts LEFT padded with zero vectors whose attention mask stays 1 (reference quirk, blip2_mr.py:744-753, 769-774).
    interleave=False: the reference's other prompt form (its constructor default; no shipped Mr. BLIP config uses it), _plain_layout."""
    if not interleave:
        return _plain_layout(tokenizer, samples, repl, n_per_frame, T, max_txt_len, no_task_prompt, time_format)
    if time_format == "seconds_integers":
        ts, durs = seconds_integers(samples["timestamps"], samples["duration"], repl)
    elif time_format == "seconds_floats":
        ts, durs = seconds_floats(samples["timestamps"], samples["duration"])
    else:  # relative_* / framenumbers are broken in the reference itself (undefined helper / str + float: SURVEY.md §8c)
        raise ValueError("Invalid input_time_format, please choose from ['seconds_integers', 'seconds_floats']")
    B = len(ts)
    end_tok = tokenizer(list(samples["video_prompt_end"]), padding="longest", add_special_tokens=False, truncation=True,
                        max_length=max_txt_len, return_tensors="pt")
    if no_task_prompt:
        text = list(samples["query_prompt"])
    else:
        text = [q + t for q, t in zip(samples["query_prompt"], samples["task_prompt"])]
    text_tok = tokenizer(text, padding="longest", truncation=True, max_length=max_txt_len, return_tensors="pt")
    sep = tokenizer.convert_tokens_to_ids(">")
    dur_tokens = clean_number_tokens(tokenizer, durs)
    rows = []  # per sample: list of ("f", frame_row) | ("e", token_id)
    for j in range(B):
        tt = clean_number_tokens(tokenizer, ts[j])
        seq = []
        for i in range(T):
            base = (j * T + i) * n_per_frame
            seq.extend(("f", base + r) for r in range(n_per_frame))
            seq.extend(("e", t) for t in tt[i])
        seq.append(("e", sep))
        seq.extend(("e", t) for t in dur_tokens[j])
        rows.append(seq)
    Lv = max(len(r) for r in rows)
    Le, Lt = end_tok.input_ids.shape[1], text_tok.input_ids.shape[1]
    S = Lv + Le + Lt
    f_src, f_dst, e_src, e_dst = [], [], [], []
    mask = torch.ones(B, S, dtype=torch.int32)
    for j, seq in enumerate(rows):
        pad = Lv - len(seq)
        for s in range(pad):
            e_src.append(-1)
            e_dst.append(j * S + s)
        for s, (kind, v) in enumerate(seq):
            if kind == "f":
                f_src.append(v)
                f_dst.append(j * S + pad + s)
            else:
                e_src.append(v)
                e_dst.append(j * S + pad + s)
        for s in range(Le):
            e_src.append(int(end_tok.input_ids[j, s]))
            e_dst.append(j * S + Lv + s)
        for s in range(Lt):
            e_src.append(int(text_tok.input_ids[j, s]))
            e_dst.append(j * S + Lv + Le + s)
        mask[j, Lv:Lv + Le] = end_tok.attention_mask[j].int()
        mask[j, Lv + Le:] = text_tok.attention_mask[j].int()
    ans = tokenizer(list(samples["relevant_windows"]), padding="longest", truncation=True, max_length=max_txt_len, return_tensors="pt")
    labels = ans.input_ids.masked_fill(ans.input_ids == tokenizer.pad_token_id, -100)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)  # noqa: E731
    return EncoderLayout(S=S, frame_src=i32(f_src), frame_dst=i32(f_dst), emb_src=i32(e_src), emb_dst=i32(e_dst),
                         attention_mask=mask, labels=labels, decoder_input_ids=shift_right(labels),
                         decoder_mask=ans.attention_mask.int())


def build_qa_layout(tokenizer, qa_input: Sequence[str], qa_output: Sequence[str], n_frame_tokens: int, max_txt_len: int = 200) -> EncoderLayout:
    """Answerer input of the video-QA path (forward_QA, blip2_mr.py:365-405): [ all t * n frame tokens of the clip (mask 1) | question tokens
    (right padded, mask 0 on the pads) ]; labels = the answer's tokens with the pads at -100, decoder mask = the answer's attention mask."""
    q = tokenizer(list(qa_input), padding="longest", truncation=True, max_length=max_txt_len, return_tensors="pt")
    B, Lq, Lf = q.input_ids.shape[0], q.input_ids.shape[1], int(n_frame_tokens)
    S = Lf + Lq
    f_src, f_dst, e_src, e_dst = [], [], [], []
    mask = torch.ones(B, S, dtype=torch.int32)
    for j in range(B):
        f_src.extend(j * Lf + r for r in range(Lf))
        f_dst.extend(j * S + r for r in range(Lf))
        e_src.extend(int(t) for t in q.input_ids[j].tolist())
        e_dst.extend(j * S + Lf + s for s in range(Lq))
        mask[j, Lf:] = q.attention_mask[j].int()
    ans = tokenizer(list(qa_output), padding="longest", truncation=True, max_length=max_txt_len, return_tensors="pt")
    labels = ans.input_ids.masked_fill(ans.input_ids == tokenizer.pad_token_id, -100)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)  # noqa: E731
    return EncoderLayout(S=S, frame_src=i32(f_src), frame_dst=i32(f_dst), emb_src=i32(e_src), emb_dst=i32(e_dst), attention_mask=mask,
                         labels=labels, decoder_input_ids=shift_right(labels), decoder_mask=ans.attention_mask.int())


def relative_position_bucket(rel: int, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> int:
    """T5 bucket of one relative position (memory - query).  Integer restatement of modeling_t5.py:392-445: the log
    branch is evaluated in float32 like the reference's tensor code, then truncated."""
    import numpy as np

    ret = 0
    nb = num_buckets
    if bidirectional:
        nb //= 2
        if rel > 0:
            ret += nb
        n = abs(rel)
    else:
        n = -min(rel, 0)
    max_exact = nb // 2
    if n < max_exact:
        return ret + n
    v = np.float32(max_exact) + (np.log(np.float32(n) / np.float32(max_exact)) / np.float32(np.log(max_distance / max_exact))
                                * np.float32(nb - max_exact))
    return ret + min(int(v), nb - 1)


def bias_lut(table: torch.Tensor, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """[num_buckets, H] relative_attention_bias table -> [H, 257] LUT indexed by clamp(key - query, -128, 128) + 128
    (every |rel| >= max_distance shares one bucket, so the clamp is exact)."""
    assert max_distance == 128, "the attention kernel clamps relative positions at +-128"
    idx = torch.tensor([relative_position_bucket(r, bidirectional, num_buckets, max_distance) for r in range(-128, 129)])
    return table.float()[idx].t().contiguous()
