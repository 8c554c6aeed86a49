"""Checkpoint ingestion for the MI355X engine: the reference's weight files -> one flat state dict with the reference's key names,
which ``MrBlipEngine`` packs into its bf16 operand layout at construction (``StateDictSource``).

What the reference loads and how (all four sources are non-strict ``load_state_dict`` calls there):
  * ``eva_vit_g.pth`` — a plain state dict WITHOUT prefix, position embedding bicubically resized to the model's grid first
    (eva_vit.py:373-394 ``interpolate_pos_embed``, :415-441 ``create_eva_vit_g``)            -> keys ``visual_encoder.*``
  * ``blip2_pretrained_flant5xl.pth`` — ``checkpoint["model"]`` with ``Qformer.*``, ``query_tokens``, ``ln_vision.*``, ``t5_proj.*`` (and, in
    some releases, ``visual_encoder.*``) (blip2.py:86-104 ``load_from_pretrained``)
  * HF ``google/flan-t5-xl`` — ``shared.weight``, ``encoder.*``, ``decoder.*``, ``lm_head.weight`` (blip2_mr.py:144-151)  -> ``t5_model.*``
  * a fine-tuned Mr. BLIP checkpoint — ``checkpoint["model"]`` holding only the trainable tensors in peft naming
    (``t5_model.base_model.model.<linear>.lora_{A,B}.default.weight``, ``t5_proj.*``, ``ln_vision.*``; base_model.py:29-56, runner_base.py:572-600)

``reference_keys(cfg)`` is the contract: every (key, shape) the engine consumes, in the reference's naming.  tests/test_checkpoint_cpu.py
checks it against the (key, shape) manifest of the reference's own ``state_dict()`` (tests/golden/mr_c1.npz) — every key of the reference is
either consumed or listed in ``IGNORED_REFERENCE_KEYS`` with the reason.
"""
from __future__ import annotations

import glob
import json
import logging
import os
import re
from typing import Dict, List, Optional, Tuple

import torch

from .engine import EngineConfig

# keys present in the reference's state_dict that the hot path never reads
IGNORED_REFERENCE_KEYS = (
    (r"^t5_model\.(encoder|decoder)\.embed_tokens\.weight$", "tied to t5_model.shared.weight (one Parameter, three names)"),
    (r"^t5_model\.(base_model\.model\.)?(encoder|decoder)\.block\.[1-9]\d*\.layer\.0\.SelfAttention\.relative_attention_bias\.weight$",
     "only block 0 owns a relative_attention_bias (modeling_t5.py:1232-1235); HF never creates it for later blocks"),
    (r"^Qformer\.cls\.", "BertOnlyMLMHead: removed by BLIP2_MR.__init__ (blip2_mr.py, Qformer.cls = None)"),
    (r"^Qformer\.bert\.embeddings\.(word|position)_embeddings\.", "text branch of the Q-Former: set to None by BLIP2_MR.__init__"),
    (r"^Qformer\.bert\.embeddings\.position_ids$", "integer buffer of the text branch"),
    (r"^Qformer\.bert\.encoder\.layer\.\d+\.(intermediate|output)\.", "text-branch FFN of the Q-Former: set to None by BLIP2_MR.__init__"),
    (r"^visual_encoder\.(head|fc_norm|norm)\.", "EVA classification head / final norm: forward_features returns before them (eva_vit.py:324-340)"),
    (r"^visual_encoder\.rel_pos_bias\.", "not used by eva_vit_g (use_rel_pos_bias False)"),
    (r"^temp$|^itm_head\.|^vision_proj\.|^text_proj\.", "BLIP-2 stage-1 heads present in some pretrained files"),
)


def reference_keys(cfg: EngineConfig, peft: bool = False) -> List[Tuple[str, tuple]]:
    """Every float tensor the engine reads, with the reference's name and shape.  peft=True: T5 Linear weights under peft's wrapping
    (``t5_model.base_model.model.<name>.base_layer.weight``) plus the LoRA A/B tensors."""
    c = cfg
    D, P_, G = c.vit_dim, c.patch, c.img // c.patch
    out: List[Tuple[str, tuple]] = [("query_tokens", (1, c.num_query, c.qf_dim))]
    v = "visual_encoder."
    out += [(v + "cls_token", (1, 1, D)), (v + "pos_embed", (1, G * G + 1, D)), (v + "patch_embed.proj.weight", (D, 3, P_, P_)),
            (v + "patch_embed.proj.bias", (D,))]
    for i in range(c.vit_depth):
        b = v + f"blocks.{i}."
        out += [(b + "norm1.weight", (D,)), (b + "norm1.bias", (D,)), (b + "attn.q_bias", (D,)), (b + "attn.v_bias", (D,)),
                (b + "attn.qkv.weight", (3 * D, D)), (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,)),
                (b + "norm2.weight", (D,)), (b + "norm2.bias", (D,)), (b + "mlp.fc1.weight", (c.vit_mlp, D)), (b + "mlp.fc1.bias", (c.vit_mlp,)),
                (b + "mlp.fc2.weight", (D, c.vit_mlp)), (b + "mlp.fc2.bias", (D,))]
    out += [("ln_vision.weight", (D,)), ("ln_vision.bias", (D,))]
    Q, I = c.qf_dim, c.qf_inter
    q = "Qformer.bert."
    out += [(q + "embeddings.LayerNorm.weight", (Q,)), (q + "embeddings.LayerNorm.bias", (Q,))]
    for i in range(c.qf_layers):
        l = q + f"encoder.layer.{i}."
        atts = [("attention.", Q)] + ([("crossattention.", D)] if i % c.qf_cross_freq == 0 else [])
        for pref, kv in atts:
            a = l + pref
            out += [(a + "self.query.weight", (Q, Q)), (a + "self.query.bias", (Q,)), (a + "self.key.weight", (Q, kv)), (a + "self.key.bias", (Q,)),
                    (a + "self.value.weight", (Q, kv)), (a + "self.value.bias", (Q,)), (a + "output.dense.weight", (Q, Q)),
                    (a + "output.dense.bias", (Q,)), (a + "output.LayerNorm.weight", (Q,)), (a + "output.LayerNorm.bias", (Q,))]
        out += [(l + "intermediate_query.dense.weight", (I, Q)), (l + "intermediate_query.dense.bias", (I,)), (l + "output_query.dense.weight", (Q, I)),
                (l + "output_query.dense.bias", (Q,)), (l + "output_query.LayerNorm.weight", (Q,)), (l + "output_query.LayerNorm.bias", (Q,))]
    d, inner, ff, V, H = c.d_model, c.t5_heads * c.d_kv, c.d_ff, c.vocab, c.t5_heads
    t = "t5_model.base_model.model." if peft else "t5_model."

    def lin(name, o, i):
        if not peft:
            return [(t + name + ".weight", (o, i))]
        return [(t + name + ".base_layer.weight", (o, i)), (t + name + ".lora_A.default.weight", (c.lora_r, i)), (t + name + ".lora_B.default.weight", (o, c.lora_r))]

    out += [(t + "shared.weight", (V, d))]
    for stack, n in (("encoder", c.t5_layers), ("decoder", c.t5_dec_layers)):
        for i in range(n):
            b = f"{stack}.block.{i}."
            for x in "qkv":
                out += lin(b + "layer.0.SelfAttention." + x, inner, d)
            out += lin(b + "layer.0.SelfAttention.o", d, inner)
            if i == 0:
                out += [(t + b + "layer.0.SelfAttention.relative_attention_bias.weight", (32, H))]
            out += [(t + b + "layer.0.layer_norm.weight", (d,))]
            j = 1
            if stack == "decoder":
                for x in "qkv":
                    out += lin(b + "layer.1.EncDecAttention." + x, inner, d)
                out += lin(b + "layer.1.EncDecAttention.o", d, inner)
                out += [(t + b + "layer.1.layer_norm.weight", (d,))]
                j = 2
            out += lin(b + f"layer.{j}.DenseReluDense.wi_0", ff, d) + lin(b + f"layer.{j}.DenseReluDense.wi_1", ff, d)
            out += lin(b + f"layer.{j}.DenseReluDense.wo", d, ff)
            out += [(t + b + f"layer.{j}.layer_norm.weight", (d,))]
        out += [(t + f"{stack}.final_layer_norm.weight", (d,))]
    out += lin("lm_head", V, d)
    out += [("t5_proj.weight", (d, Q)), ("t5_proj.bias", (d,))]
    return out


def interpolate_pos_embed(pos_embed: torch.Tensor, num_patches: int, num_extra_tokens: int = 1) -> torch.Tensor:
    """eva_vit.py:373-394: the cls position is kept, the square grid of patch positions is resized with bicubic interpolation
    (align_corners=False) when the checkpoint's grid differs from the model's."""
    pe = pos_embed.float()
    emb = pe.shape[-1]
    orig = int((pe.shape[-2] - num_extra_tokens) ** 0.5)
    new = int(num_patches ** 0.5)
    if orig == new:
        return pe
    logging.info("Position interpolate from %dx%d to %dx%d", orig, orig, new, new)
    extra = pe[:, :num_extra_tokens]
    pos = pe[:, num_extra_tokens:].reshape(-1, orig, orig, emb).permute(0, 3, 1, 2)
    pos = torch.nn.functional.interpolate(pos, size=(new, new), mode="bicubic", align_corners=False)
    pos = pos.permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((extra, pos), dim=1)


def load_file(path: str) -> Dict[str, torch.Tensor]:
    """One weight file or a directory of them: torch pickles (``.pth/.pt/.bin``; ``{"model": sd}`` unwrapped like base_model.py:46-49),
    safetensors (single file, or an HF sharded checkpoint directory with its ``*.index.json``)."""
    if os.path.isdir(path):
        files = sorted(glob.glob(os.path.join(path, "*.safetensors"))) or sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
        if not files:
            raise RuntimeError(f"no weight files (*.safetensors / pytorch_model*.bin) in {path}")
        sd: Dict[str, torch.Tensor] = {}
        for f in files:
            sd.update(load_file(f))
        return sd
    if not os.path.isfile(path):
        raise RuntimeError("checkpoint url or path is invalid: %s" % path)  # (the reference's message; URLs are not fetchable here)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file as st_load

        return st_load(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)   # user-supplied path: tensors and plain containers only
    return ck["model"] if isinstance(ck, dict) and "model" in ck and isinstance(ck["model"], dict) else ck


def assemble_state_dict(cfg: EngineConfig, vit: Optional[str] = None, blip2: Optional[str] = None, t5: Optional[str] = None,
                        finetuned: Optional[str] = None):
    """Merge the reference's weight sources into one state dict with the reference's key names.  Returns (sd, report) with
    report = dict(missing=[...], unexpected=[...], ignored=[...]) — the non-strict semantics of the reference: nothing raises here,
    the caller decides (BLIP2_MR refuses to build from an incomplete backbone unless synthetic weights are asked for)."""
    sd: Dict[str, torch.Tensor] = {}
    G = cfg.img // cfg.patch
    if vit:
        raw = load_file(vit)
        if "pos_embed" in raw:
            raw["pos_embed"] = interpolate_pos_embed(raw["pos_embed"], G * G)
        sd.update({"visual_encoder." + k: v for k, v in raw.items()})
    if blip2:
        raw = load_file(blip2)
        if "visual_encoder.pos_embed" in raw:
            raw["visual_encoder.pos_embed"] = interpolate_pos_embed(raw["visual_encoder.pos_embed"], G * G)
        sd.update(raw)
    if t5:
        raw = load_file(t5)
        sd.update({(k if k.startswith("t5_model.") else "t5_model." + k): v for k, v in raw.items()})
    if finetuned:
        sd.update(load_file(finetuned))
    return sd, coverage_report(cfg, sd)


def coverage_report(cfg: EngineConfig, sd: Dict[str, torch.Tensor]):
    """which engine inputs are missing from ``sd``, which keys of ``sd`` nobody consumes (and which of those are known-ignorable)"""
    peft = any(k.startswith("t5_model.base_model.model.") for k in sd)
    want = reference_keys(cfg, peft=False)
    missing, bad_shape, used = [], [], set()
    for k, shape in want:
        hit = None
        cands = [k]
        if k.startswith("t5_model."):
            rest = k[len("t5_model."):]
            cands += ["t5_model.base_model.model." + rest, "t5_model.base_model.model." + rest.replace(".weight", ".base_layer.weight")]
        for c_ in cands:
            if c_ in sd:
                hit = c_
                break
        if hit is None:
            missing.append(k)
            continue
        used.add(hit)
        if tuple(sd[hit].shape) != tuple(shape):
            bad_shape.append((k, tuple(sd[hit].shape), tuple(shape)))
    if peft:
        for k in sd:
            if ".lora_A.default.weight" in k or ".lora_B.default.weight" in k:
                used.add(k)
    unexpected, ignored = [], []
    for k in sd:
        if k in used:
            continue
        if any(re.search(pat, k) for pat, _ in IGNORED_REFERENCE_KEYS):
            ignored.append(k)
        else:
            unexpected.append(k)
    return dict(missing=missing, unexpected=unexpected, ignored=ignored, bad_shape=bad_shape)


def describe(report: dict) -> str:
    return json.dumps({k: (v[:8] + ["... %d more" % (len(v) - 8)] if len(v) > 8 else v) for k, v in report.items()}, default=str)
