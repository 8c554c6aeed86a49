"""Deterministic fixture tokenizer with the subset of the HF ``T5TokenizerFast`` surface that
the Mr. BLIP hot path touches.

The real SentencePiece vocabulary of ``google/flan-t5-xl`` is a HF-hub artefact that is not in
the reference tree ("parity unpinned (i)"), so the model takes its tokenizer as
an injected object: the real ``T5TokenizerFast`` when its files are on disk, this class otherwise
(synthetic benchmark inputs, tests, golden generation).

Call sites in the reference that define the required surface:
  * ``blip2_mr.py:524-534``  answer tokenisation (padding="longest", return_tensors="pt")
  * ``blip2_mr.py:633-665``  prompt tokenisation with/without special tokens
  * ``blip2_mr.py:1497-1535`` ``find_annoying_numbers``: ``str(i)`` → ids, id 3 is the lone "▁"
  * ``blip2_mr.py:1576-1581`` list-of-str call without tensors, leading-3 stripping
  * ``blip2_mr.py:273-275``  ``convert_tokens_to_ids(">")``, ``pad_token_id``
  * ``blip2_mr.py:901-903``  ``batch_decode(..., skip_special_tokens=True)``

Vocabulary (32128 ids like flan-t5): 0 pad, 1 eos, 2 unk, 3 "▁"; punctuation 10..137;
integers < 200 mostly single tokens 200+n — except two deliberately awkward families that
exercise the reference's "annoying number" logic: n % 17 == 5 splits into two pieces and
n % 23 == 7 gets a leading lone "▁"; sentinels ``<extra_id_k>`` = 32099-k; words hash into
[1000, 31000).
"""
from __future__ import annotations

import re
from typing import Dict, List, Sequence, Union

import torch

_PIECE_RE = re.compile(r"<extra_id_\d+>|<vid>|[A-Za-z]+|\d+|[^\sA-Za-z\d]")

VOCAB_SIZE = 32128
PAD, EOS, UNK, SPACE = 0, 1, 2, 3
_OPTION_IDS = {"A": 71, "B": 272, "C": 205, "D": 309, "E": 262}    # the answer options of the video-QA path, as in the flan-t5 vocabulary


def _fnv1a(s: str) -> int:
    h = 0x811C9DC5
    for ch in s.encode("utf-8"):
        h ^= ch
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


class BatchEncoding(dict):
    """dict with attribute access and ``.to(device)`` like the HF class."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def to(self, device):
        for k, v in list(self.items()):
            if torch.is_tensor(v):
                self[k] = v.to(device)
        return self


class FixtureTokenizer:
    pad_token_id = PAD
    eos_token_id = EOS
    unk_token_id = UNK
    vocab_size = VOCAB_SIZE

    def __init__(self):
        self._id2word: Dict[int, str] = {}

    # ------------------------------------------------------------------ encode
    def _int_pieces(self, digits: str) -> List[int]:
        n = int(digits)
        if len(digits) <= 3 and n < 200 and str(n) == digits:
            if n % 17 == 5:  # "annoying": two real pieces
                return [200 + n // 10, 400 + n % 10]
            if n % 23 == 7:  # "space annoying": lone "▁" + one piece
                return [SPACE, 200 + n]
            return [200 + n]
        out = []
        for i in range(0, len(digits), 2):
            out.append(500 + int(digits[i : i + 2]) + (100 if len(digits[i : i + 2]) == 1 else 0))
        return out

    def _encode_one(self, text: str) -> List[int]:
        ids: List[int] = []
        for p in _PIECE_RE.findall(text):
            if p.startswith("<extra_id_"):
                ids.append(32099 - int(p[10:-1]))
            elif p == "<vid>":
                ids.append(9)
            elif p.isdigit():
                ids.extend(self._int_pieces(p))
            elif p in _OPTION_IDS:     # the one vocabulary fact the reference states: "A B C D E" -> [71, 272, 205, 309, 262] (blip2_mr.py:1299)
                ids.append(_OPTION_IDS[p])
            elif p[0].isalpha():
                wid = 1000 + _fnv1a(p) % 30000
                self._id2word.setdefault(wid, p)
                ids.append(wid)
            else:
                ids.append(10 + (ord(p) % 128))
        return ids

    def convert_tokens_to_ids(self, tok: Union[str, Sequence[str]]):
        if isinstance(tok, str):
            ids = self._encode_one(tok)
            return ids[0] if ids else UNK
        return [self.convert_tokens_to_ids(t) for t in tok]

    def __call__(
        self,
        text: Union[str, Sequence[str]],
        padding=False,
        add_special_tokens: bool = True,
        truncation: bool = False,
        max_length: int | None = None,
        return_tensors: str | None = None,
    ) -> BatchEncoding:
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        rows = []
        for t in texts:
            ids = self._encode_one(t)
            if truncation and max_length is not None:
                ids = ids[: max_length - (1 if add_special_tokens else 0)]
            if add_special_tokens:
                ids = ids + [EOS]
            rows.append(ids)
        masks = [[1] * len(r) for r in rows]
        if padding in (True, "longest") or return_tensors == "pt":
            m = max((len(r) for r in rows), default=0)
            masks = [mk + [0] * (m - len(r)) for r, mk in zip(rows, masks)]
            rows = [r + [PAD] * (m - len(r)) for r in rows]
        if return_tensors == "pt":
            return BatchEncoding(
                input_ids=torch.tensor(rows, dtype=torch.long).reshape(len(rows), -1),
                attention_mask=torch.tensor(masks, dtype=torch.long).reshape(len(rows), -1),
            )
        if single:
            return BatchEncoding(input_ids=rows[0], attention_mask=masks[0])
        return BatchEncoding(input_ids=rows, attention_mask=masks)

    # ------------------------------------------------------------------ decode
    def _piece(self, i: int) -> str:
        if i in (PAD, EOS):
            return ""
        if i == UNK:
            return "<unk>"
        if i == SPACE:
            return ""
        if i == 9:
            return "<vid>"
        if 10 <= i < 138:
            return chr(i - 10)
        if 200 <= i < 400:
            return str(i - 200)
        if 400 <= i < 410:
            return str(i - 400)
        if 500 <= i < 600:
            return "%02d" % (i - 500)
        if 600 <= i < 610:
            return str(i - 600)
        if 32000 <= i <= 32099:
            return "<extra_id_%d>" % (32099 - i)
        return self._id2word.get(i, "<w%d>" % i)

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        if torch.is_tensor(ids):
            ids = ids.tolist()
        if isinstance(ids, int):
            ids = [ids]
        out = ""
        prev_alnum = False
        prev_two_piece = False
        for i in ids:
            i = int(i)
            if skip_special_tokens and (i in (PAD, EOS) or 32000 <= i <= 32099):
                continue
            if not skip_special_tokens and i == PAD:
                p = "<pad>"
            elif not skip_special_tokens and i == EOS:
                p = "</s>"
            else:
                p = self._piece(i)
            if not p:
                continue
            alnum = p[0].isalnum()
            glue = 400 <= i < 410 and prev_two_piece  # second half of a split integer
            if out and ((alnum and prev_alnum and not glue) or out[-1] == ","):
                out += " "
            out += p
            prev_alnum = alnum
            prev_two_piece = 200 <= i < 400
        return out

    def batch_decode(self, seqs, skip_special_tokens: bool = False) -> List[str]:
        if torch.is_tensor(seqs):
            seqs = seqs.tolist()
        return [self.decode(s, skip_special_tokens=skip_special_tokens) for s in seqs]


def load_tokenizer(name_or_path: str = "google/flan-t5-xl", allow_fixture: bool = False):
    """The real ``T5TokenizerFast`` from local files (mirrors ``T5TokenizerFast.from_pretrained(t5_model)`` at ``blip2_mr.py:143``; no
    network here).  ``name_or_path == "fixture"`` or ``allow_fixture=True`` (synthetic / test configurations only) selects the
    deterministic ``FixtureTokenizer``; otherwise a missing vocabulary is an ERROR: a run that silently tokenised prompts and labels
    with a fake vocabulary would produce checkpoints and metrics that are incompatible with the reference."""
    import logging

    if name_or_path == "fixture":
        return FixtureTokenizer()
    try:
        from transformers import T5TokenizerFast  # type: ignore

        return T5TokenizerFast.from_pretrained(name_or_path, local_files_only=True)
    except Exception as e:  # noqa: BLE001
        if allow_fixture:
            logging.warning("tokenizer %r is not available locally (%s: %s): using the FixtureTokenizer — token ids, the "
                            "'annoying numbers' set and every checkpoint/metric of this run are NOT comparable with the reference",
                            name_or_path, type(e).__name__, e)
            return FixtureTokenizer()
        raise RuntimeError(f"tokenizer {name_or_path!r} is not available locally ({type(e).__name__}: {e}).  Put the flan-t5 tokenizer files in a "
                           "local directory and pass it as model.t5_model, or set model.synthetic_weights: True / t5_model: fixture for "
                           "synthetic runs") from e
