"""Deterministic, name-keyed synthetic weights shared by the golden generator and the tests, so the
fixtures only have to carry inputs/outputs and the (key, shape) manifest — not the weights."""
import re
import zlib

import numpy as np
import torch


def seeded_array(key: str, shape, std=None, wscale: float = 1.0, fast: bool = False) -> np.ndarray:
    """wscale scales the generic (Linear / conv) weights only: 1.0 -> std 0.08 (tiny fixtures), 0.25 -> std 0.02 (the real-width C1
    fixture: realistic pre-trained scale, so softmaxes are not saturated over 39 + 12 + 24 layers).  fast=True draws with numpy's
    Generator (float32 ziggurat, ~6x faster than RandomState: the C1 fixture regenerates 1.3 G weights at test time)."""
    # T5 ties encoder/decoder embed_tokens to `shared` (one Parameter, three state-dict names)
    key = re.sub(r"(encoder|decoder)\.embed_tokens\.weight$", "shared.weight", key)
    seed = zlib.crc32(key.encode()) & 0x7FFFFFFF
    if fast:
        x = np.random.Generator(np.random.PCG64(seed)).standard_normal(tuple(shape), dtype=np.float32)
    else:
        x = np.random.RandomState(seed).standard_normal(tuple(shape)).astype(np.float32)
    leaf = key.rsplit(".", 1)[-1]
    lower = key.lower()
    if std is not None:
        return x * std
    is_norm = ("norm" in lower or lower.startswith("ln_") or ".ln_" in lower) and leaf == "weight"
    if is_norm:
        return (1.0 + 0.1 * x).astype(np.float32)
    if "relative_attention_bias" in key:
        return (0.5 * x).astype(np.float32)
    if leaf == "bias" or leaf in ("q_bias", "v_bias"):
        return (0.05 * x).astype(np.float32)
    if key in ("query_tokens",) or leaf in ("cls_token", "pos_embed"):
        return (0.2 * x).astype(np.float32)
    if "lora_" in key:
        return (0.05 * x).astype(np.float32)
    if key.endswith("shared.weight") or key.endswith("embed_tokens.weight"):
        return (0.5 * x).astype(np.float32)
    return (0.08 * wscale * x).astype(np.float32)


def seeded_state_dict(manifest, wscale: float = 1.0, fast: bool = False):
    """manifest: iterable of (key, shape). Integer buffers (position_ids) are skipped by the caller."""
    return {k: torch.from_numpy(seeded_array(k, s, wscale=wscale, fast=fast)) for k, s in manifest}
