"""ORACLE — test infrastructure only.  CPU (PyTorch fp32 / numpy / pure Python) restatement of the
reference's Mr. BLIP train-step hot path, written from the reference's *behaviour*; every function cites
the reference file:line (relative to /root/reference) it follows.

Who may import this: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg —
as the CHECKER, never as the thing measured or shipped.  The product (``mr-blip_amd/``) never imports it.

Pinning: ``tests/test_oracle_golden.py`` checks this file against golden vectors captured by importing the
reference itself in the build container (``tests/golden/make_golden.py``): ViT, ln_vision+Q-Former, T5
(loss / logits / encoder output / input grads), the full ``forward_mr`` (interleaved encoder input, mask,
loss, grads of t5_proj and ln_vision), relative-position buckets, timestamp integers, ``post_process``,
``moment_str_to_list`` and the LR schedule.
Parity UNPINNED (third-party code absent from the reference tree): peft==0.13.0 LoRA
(restated here from its published algorithm: y = W x + (alpha/r) * B A dropout(x); A kaiming-uniform,
B zeros), the real flan-t5 SentencePiece vocabulary, and HF ``generate`` beam search.

``emu_bf16=True`` rounds every matmul operand (and the attention probabilities) to bfloat16 at the same
points where the HIP path stores/feeds bf16, keeping fp32 accumulation — this separates logic errors from
rounding when the HIP path is compared with this oracle.
"""
from __future__ import annotations

import ast
import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ====================================================================================== integer / string logic
def relative_position_bucket(rel: np.ndarray, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128):
    """modeling_t5.py:392-445.  rel = memory_position - query_position (int array).  Integer-exact restatement
    of the float-log formula (float32 log, truncation toward zero)."""
    rel = np.asarray(rel, dtype=np.int64)
    ret = np.zeros_like(rel)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        ret = ret + (rel > 0).astype(np.int64) * nb
        n = np.abs(rel)
    else:
        n = -np.minimum(rel, 0)
    max_exact = nb // 2
    is_small = n < max_exact
    with np.errstate(divide="ignore"):
        big = max_exact + (
            np.log(n.astype(np.float32) / np.float32(max_exact)) / np.float32(math.log(max_distance / max_exact))
            * np.float32(nb - max_exact)
        ).astype(np.float32)
    big = np.where(is_small, 0, big).astype(np.int64)
    big = np.minimum(big, nb - 1)
    return ret + np.where(is_small, n, big)


def shift_right(labels: Tensor, decoder_start_token_id: int = 0, pad_token_id: int = 0) -> Tensor:
    """modeling_t5.py:919-948."""
    out = torch.zeros_like(labels)
    out[..., 1:] = labels[..., :-1]
    out[..., 0] = decoder_start_token_id
    return out.masked_fill(out == -100, pad_token_id)


def find_annoying_numbers(tokenizer, range_end: int = 200) -> Tuple[List[int], List[int]]:
    """blip2_mr.py:1497-1535."""
    annoying, annoying_space = [], []
    for i in range(range_end):
        ids = tokenizer(str(i), padding="longest", add_special_tokens=False, truncation=True, max_length=300,
                        return_tensors="pt")["input_ids"].tolist()[0]
        if len(ids) > 1:
            (annoying_space if ids[0] == 3 else annoying).append(i)
    return annoying, annoying_space


def annoying_replacement_dict(annoying: Sequence[int]) -> Dict[int, int]:
    """blip2_mr.py:1537-1559: nearest non-annoying integer, i+j tried before i-j."""
    out = {}
    for i in annoying:
        for j in range(100):
            if (i + j) not in annoying:
                new = i + j
                break
            if (i - j) not in annoying:
                new = i - j
                break
        out[i] = new
    return out


def timestamps_as_seconds_integers(timestamps: Tensor, durations: Tensor, repl: Dict[int, int]):
    """utils.py:388-434.  Python round() = round-half-even on the float64 value of the fp32 timestamp."""
    new_ts, new_d, prompts = [], [], []
    for t, d in zip(timestamps, durations):
        ints = []
        for x in t:
            r = round(x.item())
            ints.append(int(repl.get(r, r)))
        dr = round(d.item())
        dr = repl.get(dr, dr)
        prompts.append(">" + ">".join(str(i) for i in ints) + ">" + str(dr))
        new_ts.append(torch.tensor(ints))
        new_d.append(dr)
    return new_ts, new_d, prompts


def timestamps_as_seconds_floats(timestamps: Tensor, durations: Tensor):
    """utils.py:464-485: round(t, 2) stored into a float32 tensor; durations untouched.  (The tokens are later taken from
    str(tensor_element.item()), blip2_mr.py:1576-1578 — see clean_timestamp_tokens.)"""
    new_ts = [torch.tensor([round(x.item(), 2) for x in t]) for t in timestamps]
    prompts = [">".join(str(round(x.item(), 2)) for x in t) + ">" + str(round(d.item())) for t, d in zip(timestamps, durations)]
    return new_ts, durations, prompts


def clean_timestamp_tokens(tokenizer, values) -> List[List[int]]:
    """blip2_mr.py:1576-1581: tokenise str(v) without specials, strip a leading id 3."""
    toks = tokenizer([str(v.item() if torch.is_tensor(v) else v) for v in values], add_special_tokens=False)["input_ids"]
    return [t[1:] if t[0] == 3 else t for t in toks]


def post_process(pred: str) -> str:
    """utils.py:18-83."""
    pred = pred.split("</s>")[0]
    if not re.match(r"\[\[.*\]\]", pred):
        return "[[-1, -1]]"
    pred = pred[1:-1]
    out = []
    for w in re.split(r"\s+(?=\[)", pred):
        w = re.sub(r",+$", "", w)
        w = re.sub(r"(\d) (\d)", r"\1, \2", w)
        w = re.sub(r",+", ",", w)
        nums = re.findall(r"\d+", w)
        if len(nums) == 2 and int(nums[0]) > int(nums[1]):
            w = "[" + nums[1] + ", " + nums[0] + "]"
        out.append(w)
    return "[" + ", ".join(out) + "]"


def moment_str_to_list(m: str):
    """utils.py:300-341."""
    if m == "[[-1, -1]]" or not re.match(r"\[\[.*\]\]", m):
        return [[-1, -1]]
    try:
        v = ast.literal_eval(m)
    except Exception:
        return [[-1, -1]]
    if not isinstance(v, list):
        return [[-1, -1]]
    for i in range(len(v)):
        if len(v[i]) != 2:
            v[i] = [-1, -1]
    return v


def lr_at(cur_epoch: int, cur_step: int, state: dict, *, max_epoch, min_lr, init_lr, warmup_steps, warmup_start_lr):
    """optims.py:79-119 (LinearWarmupCosineLRScheduler.step); ``state`` carries max_iters_per_epoch."""
    if cur_step > state.get("max_iters_per_epoch", 0):
        state["max_iters_per_epoch"] = cur_step
    g = cur_epoch * state.get("max_iters_per_epoch", 0) + cur_step
    if g < warmup_steps:
        return min(init_lr, warmup_start_lr + (init_lr - warmup_start_lr) * g / max(warmup_steps, 1))
    return (init_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * cur_epoch / max_epoch)) + min_lr


# ====================================================================================== floating-point path
class _GradRound(torch.autograd.Function):
    """identity whose backward rounds the incoming gradient (see Oracle.emu_grad)"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.fn(g), None


class Oracle:
    """Functional forward over a flat state dict with the reference's parameter names."""

    def __init__(self, sd: Dict[str, Tensor], cfg: dict, emu_bf16: bool = False, lora: Optional[dict] = None, dropout=None):
        self.sd = sd
        self.cfg = cfg
        self.emu = emu_bf16
        self.lora = lora  # dict(r=8, alpha=8) or None
        # training-mode parity: dropout(name, shape) -> multiplicative mask (keep / (1-p)) or None.  Site names follow the
        # reference's nn.Dropout modules (T5: modeling_t5.py:329,599,694,824,1151,1279; Q-Former: Qformer.py:107,262,288,374;
        # peft lora_dropout); the test maps them to the HIP engine's call-site ids and restates its counter hash.
        self.dropout = dropout

    def dm(self, x: Tensor, name: str) -> Tensor:
        if self.dropout is None:
            return x
        m = self.dropout(name, tuple(x.shape))
        return x if m is None else x * m

    def am(self, name: str, shape):
        return None if self.dropout is None else self.dropout(name, tuple(shape))

    # ---- helpers
    # bf16-operand emulation can be confined to some towers (error-budget experiments: which tower contributes how much of the
    # product path's ~1e-2 logits error): emu_towers = None (all) or a set out of {"vit", "qf", "proj", "enc", "dec", "head"}, with
    # "dec:<i>" naming one decoder layer; the forward methods below set the current tower.
    emu_towers = None
    _tower = None

    # Rounding experiments (tools/grad_noise_budget.py, round 5): ``round_fn`` replaces round-to-nearest-even (e.g. by a stochastic rounding
    # to one of the two bf16 neighbours: another, equally legitimate realisation of the same rounding points); ``emu_grad`` also rounds the
    # GRADIENT arriving at every linear layer's output in the backward pass — the HIP backward feeds its dX / dW GEMMs bf16 copies of dy
    # (mrblip/engine.py: cast_dropout / rmsnorm_bwd_cast), a rounding point the forward-only emulation does not have.
    round_fn = None
    emu_grad = False

    def rb(self, x: Tensor) -> Tensor:
        if not self.emu:
            return x
        if self.emu_towers is not None and self._tower not in self.emu_towers and (self._tower or "").split(":")[0] not in self.emu_towers:
            return x
        return x.bfloat16().float() if self.round_fn is None else self.round_fn(x)

    def lin(self, x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
        y = self.rb(x) @ self.rb(w).t()
        if self.emu and self.emu_grad and y.requires_grad:
            y = _GradRound.apply(y, self.round_fn or (lambda g: g.bfloat16().float()))
        return y if b is None else y + b

    def P(self, key: str) -> Tensor:
        return self.sd[key]

    # ---- ViT  (eva_vit.py:118-148, 173-180, 198-204, 324-340)
    def vit(self, image: Tensor, n_blocks: Optional[int] = None) -> Tensor:
        self._tower = "vit"
        c = self.cfg["vit"]
        D, H = c["embed_dim"], c["num_heads"]
        hd = D // H
        p = "visual_encoder."
        w = self.P(p + "patch_embed.proj.weight")
        B = image.shape[0]
        # Conv2d k=s=14 == patchify + GEMM (eva_vit.py:196-203)
        ps = w.shape[-1]
        g = image.shape[-1] // ps
        patches = image.reshape(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * ps * ps)
        x = self.lin(patches, w.reshape(D, -1), self.P(p + "patch_embed.proj.bias"))
        x = torch.cat([self.P(p + "cls_token").expand(B, -1, -1), x], 1) + self.P(p + "pos_embed")
        scale = hd ** -0.5
        depth = c["depth"] if n_blocks is None else n_blocks
        for i in range(depth):
            q = p + f"blocks.{i}."
            h = F.layer_norm(x, (D,), self.P(q + "norm1.weight"), self.P(q + "norm1.bias"), 1e-6)
            bias = torch.cat([self.P(q + "attn.q_bias"), torch.zeros(D), self.P(q + "attn.v_bias")])
            qkv = self.lin(h, self.P(q + "attn.qkv.weight"), bias)
            N = x.shape[1]
            qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
            o = self.attention(qkv[0], qkv[1], qkv[2], scale=scale)
            o = o.transpose(1, 2).reshape(B, N, D)
            x = x + self.lin(o, self.P(q + "attn.proj.weight"), self.P(q + "attn.proj.bias"))
            h = F.layer_norm(x, (D,), self.P(q + "norm2.weight"), self.P(q + "norm2.bias"), 1e-6)
            h = F.gelu(self.lin(h, self.P(q + "mlp.fc1.weight"), self.P(q + "mlp.fc1.bias")))
            x = x + self.lin(h, self.P(q + "mlp.fc2.weight"), self.P(q + "mlp.fc2.bias"))
        return x

    def attention(self, q, k, v, scale=1.0, bias=None, p_drop_mask=None):
        """softmax(q k^T * scale + bias) v with fp32 softmax (eva_vit.py:128-145, Qformer.py:195-262,
        modeling_t5.py:561-599).  emu: q, k, v and the probabilities are bf16-rounded."""
        s = (self.rb(q) @ self.rb(k).transpose(-1, -2)) * scale
        if bias is not None:
            s = s + bias
        pr = torch.softmax(s.float(), -1)
        if p_drop_mask is not None:
            pr = pr * p_drop_mask  # keep / (1 - p)
        return self.rb(pr) @ self.rb(v)

    def ln_vision(self, x: Tensor) -> Tensor:
        """blip2.py:113-119 (fp32 LayerNorm, eps 1e-5)."""
        return F.layer_norm(x.float(), (x.shape[-1],), self.P("ln_vision.weight"), self.P("ln_vision.bias"), 1e-5)

    # ---- Q-Former, query branch only (Qformer.py:51-108, 111-289, 378-484, 487-589)
    def qformer(self, image_embeds: Tensor) -> Tensor:
        self._tower = "qf"
        c = self.cfg["qf"]
        D, H = c["hidden_size"], c["num_attention_heads"]
        hd = D // H
        eps = 1e-12
        p = "Qformer.bert."
        Bq = image_embeds.shape[0]
        x = self.P("query_tokens").expand(Bq, -1, -1)
        x = F.layer_norm(x, (D,), self.P(p + "embeddings.LayerNorm.weight"), self.P(p + "embeddings.LayerNorm.bias"), eps)
        x = self.dm(x.reshape(-1, D), "qf.emb").reshape(x.shape)

        def heads(t):
            return t.reshape(t.shape[0], t.shape[1], H, hd).transpose(1, 2)

        def attn_block(pref, x, kv_src, tag):
            q = heads(self.lin(x, self.P(pref + "self.query.weight"), self.P(pref + "self.query.bias")))
            k = heads(self.lin(kv_src, self.P(pref + "self.key.weight"), self.P(pref + "self.key.bias")))
            v = heads(self.lin(kv_src, self.P(pref + "self.value.weight"), self.P(pref + "self.value.bias")))
            o = self.attention(q, k, v, scale=1.0 / math.sqrt(hd), p_drop_mask=self.am(tag + ".attn", (q.shape[0], H, q.shape[2], k.shape[2])))
            o = o.transpose(1, 2).reshape(x.shape[0], x.shape[1], D)
            o = self.lin(o, self.P(pref + "output.dense.weight"), self.P(pref + "output.dense.bias"))
            o = self.dm(o.reshape(-1, D), tag + ".out").reshape(o.shape)
            return F.layer_norm(o + x, (D,), self.P(pref + "output.LayerNorm.weight"), self.P(pref + "output.LayerNorm.bias"), eps)

        for i in range(c["num_hidden_layers"]):
            l = p + f"encoder.layer.{i}."
            x = attn_block(l + "attention.", x, x, f"qf.{i}.self")
            if i % c.get("cross_attention_freq", 2) == 0:
                x = attn_block(l + "crossattention.", x, image_embeds, f"qf.{i}.cross")
            h = F.gelu(self.lin(x, self.P(l + "intermediate_query.dense.weight"), self.P(l + "intermediate_query.dense.bias")))
            h = self.lin(h, self.P(l + "output_query.dense.weight"), self.P(l + "output_query.dense.bias"))
            h = self.dm(h.reshape(-1, D), f"qf.{i}.ffn.out").reshape(h.shape)
            x = F.layer_norm(h + x, (D,), self.P(l + "output_query.LayerNorm.weight"), self.P(l + "output_query.LayerNorm.bias"), eps)
        return x

    # ---- T5 with optional LoRA (modeling_t5.py:254-277, 314-329, 350-620, 697-826, 1021-1282, 1734-1893)
    # which T5 of the state dict the t5_* methods read: "t5_model." (the localizer / moment-retrieval model) or "answerer_model." (the second
    # T5 of the video-QA variants, blip2_mr.py:152-161)
    t5_prefix = "t5_model."

    def _t5w(self, name: str) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
        """weights of T5 Linear `name` (e.g. 'encoder.block.0.layer.0.SelfAttention.q'): plain or peft naming."""
        k = self.t5_prefix + name + ".weight"
        if k in self.sd:
            return self.sd[k], None, None
        b = self.t5_prefix + "base_model.model." + name
        return self.sd[b + ".base_layer.weight"], self.sd.get(b + ".lora_A.default.weight"), self.sd.get(b + ".lora_B.default.weight")

    def _t5p(self, name: str) -> Tensor:
        k = self.t5_prefix + name
        return self.sd[k] if k in self.sd else self.sd[self.t5_prefix + "base_model.model." + name]

    def t5lin(self, x: Tensor, name: str) -> Tensor:
        w, a, b = self._t5w(name)
        y = self.lin(x, w)
        if a is not None:
            scale = self.lora["alpha"] / self.lora["r"] if self.lora else 1.0
            xd = self.dm(self.rb(x).reshape(-1, x.shape[-1]), "lora:" + name).reshape(x.shape)  # lora_dropout on the (bf16) input
            y = y + self.lin(self.lin(xd, a), b) * scale  # peft 0.13.0 Linear.forward
        return y

    def rmsnorm(self, x: Tensor, w: Tensor, eps: float) -> Tensor:
        var = x.float().pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + eps))

    def t5_bias(self, table: Tensor, qlen: int, klen: int, bidirectional: bool) -> Tensor:
        c = self.cfg["t5"]
        rel = np.arange(klen)[None, :] - np.arange(qlen)[:, None]
        b = relative_position_bucket(rel, bidirectional, c.get("num_buckets", 32), c.get("max_distance", 128))
        return table[torch.from_numpy(b)].permute(2, 0, 1).unsqueeze(0)  # [1,H,q,k]

    def _t5_attn(self, pref: str, xq: Tensor, xkv: Tensor, bias: Tensor, tag: str = "") -> Tensor:
        c = self.cfg["t5"]
        H, dk = c["num_heads"], c["d_kv"]

        def heads(t):
            return t.reshape(t.shape[0], t.shape[1], H, dk).transpose(1, 2)

        q, k, v = heads(self.t5lin(xq, pref + ".q")), heads(self.t5lin(xkv, pref + ".k")), heads(self.t5lin(xkv, pref + ".v"))
        o = self.attention(q, k, v, scale=1.0, bias=bias, p_drop_mask=self.am(tag + ".attn", (q.shape[0], H, q.shape[2], k.shape[2])))  # no 1/sqrt(d)
        o = o.transpose(1, 2).reshape(xq.shape[0], xq.shape[1], H * dk)
        return self.t5lin(o, pref + ".o")

    def _t5_ff(self, pref: str, x: Tensor, tag: str = "") -> Tensor:
        h = F.gelu(self.t5lin(x, pref + ".wi_0")) * self.t5lin(x, pref + ".wi_1")
        h = self.dm(h.reshape(-1, h.shape[-1]), tag + ".ffn_inner").reshape(h.shape)
        return self.t5lin(h, pref + ".wo")

    def _d2(self, x: Tensor, name: str) -> Tensor:
        return self.dm(x.reshape(-1, x.shape[-1]), name).reshape(x.shape)

    def t5_encoder(self, inputs_embeds: Tensor, attention_mask: Tensor) -> Tensor:
        self._tower = "enc"
        c = self.cfg["t5"]
        eps = c.get("eps", 1e-6)
        x = self._d2(inputs_embeds, "t5.enc.emb")
        S = x.shape[1]
        neg = torch.finfo(torch.float32).min
        mask = (1.0 - attention_mask[:, None, None, :].float()) * neg
        bias = self.t5_bias(self._t5p("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"), S, S, True) + mask
        for i in range(c["num_layers"]):
            b = f"encoder.block.{i}."
            t = f"t5.enc.{i}"
            x = x + self._d2(self._t5_attn(b + "layer.0.SelfAttention", *(2 * [self.rmsnorm(x, self._t5p(b + "layer.0.layer_norm.weight"), eps)]), bias, t), t + ".attn_out")
            x = x + self._d2(self._t5_ff(b + "layer.1.DenseReluDense", self.rmsnorm(x, self._t5p(b + "layer.1.layer_norm.weight"), eps), t), t + ".ffn_out")
        return self._d2(self.rmsnorm(x, self._t5p("encoder.final_layer_norm.weight"), eps), "t5.enc.final")

    def t5_decoder(self, dec_ids: Tensor, dec_mask: Tensor, enc: Tensor, enc_mask: Tensor) -> Tensor:
        self._tower = "dec"
        c = self.cfg["t5"]
        eps = c.get("eps", 1e-6)
        x = self._d2(self._t5p("shared.weight")[dec_ids], "t5.dec.emb")
        L = x.shape[1]
        neg = torch.finfo(torch.float32).min
        causal = torch.tril(torch.ones(L, L))[None, None]
        ext = causal * dec_mask[:, None, None, :].float()
        self_bias = self.t5_bias(self._t5p("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"), L, L, False) + (1.0 - ext) * neg
        cross_bias = (1.0 - enc_mask[:, None, None, :].float()) * neg
        for i in range(c["num_decoder_layers"]):
            b = f"decoder.block.{i}."
            t = f"t5.dec.{i}"
            self._tower = f"dec:{i}"
            h = self.rmsnorm(x, self._t5p(b + "layer.0.layer_norm.weight"), eps)
            x = x + self._d2(self._t5_attn(b + "layer.0.SelfAttention", h, h, self_bias, t + ".self"), t + ".self_out")
            h = self.rmsnorm(x, self._t5p(b + "layer.1.layer_norm.weight"), eps)
            x = x + self._d2(self._t5_attn(b + "layer.1.EncDecAttention", h, enc, cross_bias, t + ".cross"), t + ".cross_out")
            x = x + self._d2(self._t5_ff(b + "layer.2.DenseReluDense", self.rmsnorm(x, self._t5p(b + "layer.2.layer_norm.weight"), eps), t), t + ".ffn_out")
        return self._d2(self.rmsnorm(x, self._t5p("decoder.final_layer_norm.weight"), eps), "t5.dec.final")

    def t5_loss(self, inputs_embeds: Tensor, attention_mask: Tensor, labels: Tensor, dec_mask: Tensor):
        """T5ForConditionalGeneration.forward (modeling_t5.py:1796-1877): untied lm_head, CE ignore_index=-100 mean."""
        enc = self.t5_encoder(inputs_embeds, attention_mask)
        dec = self.t5_decoder(shift_right(labels), dec_mask, enc, attention_mask)
        self._tower = "head"
        logits = self.t5lin(dec, "lm_head")
        loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), labels.reshape(-1), ignore_index=-100)
        return loss, logits, enc

    # ---- prompt construction (blip2_mr.py:572-783, interleave branch) -----------------------------------
    def prompt_concatenation(self, tok, timestamps, durations, frames_for_t5, video_prompt_end, query_prompt,
                             task_prompt, repl: Dict[int, int], n_per_frame: int, time_format: str = "seconds_integers", interleave: bool = True):
        emb = self._t5p("shared.weight")
        if not interleave:   # blip2_mr.py:783-822: video prompt as TEXT in front of all frame tokens, every part with its tokenizer mask
            if time_format == "seconds_floats":   # utils.py:464-485
                vps = [">".join(str(round(x.item(), 2)) for x in t) + ">" + str(round(d.item())) for t, d in zip(timestamps, durations)]
            else:                                  # utils.py:388-434
                _, _, vps = timestamps_as_seconds_integers(timestamps, durations, repl)
            vp = tok(vps, padding="longest", add_special_tokens=False, truncation=True, max_length=200, return_tensors="pt")
            end_tok = tok(video_prompt_end, padding="longest", add_special_tokens=False, truncation=True, max_length=200, return_tensors="pt")
            text_tok = tok([q + t for q, t in zip(query_prompt, task_prompt)], padding="longest", truncation=True, max_length=200, return_tensors="pt")
            embs = torch.cat([emb[vp.input_ids], frames_for_t5, emb[end_tok.input_ids], emb[text_tok.input_ids]], 1)
            atts = torch.cat([vp.attention_mask, torch.ones(frames_for_t5.shape[:2], dtype=torch.long), end_tok.attention_mask, text_tok.attention_mask], 1)
            return embs, atts
        if time_format == "seconds_floats":
            ts_int, dur_int, _ = timestamps_as_seconds_floats(timestamps, durations)
        else:
            ts_int, dur_int, _ = timestamps_as_seconds_integers(timestamps, durations, repl)
        end_tok = tok(video_prompt_end, padding="longest", add_special_tokens=False, truncation=True, max_length=200, return_tensors="pt")
        text_tok = tok([q + t for q, t in zip(query_prompt, task_prompt)], padding="longest", truncation=True, max_length=200, return_tensors="pt")
        dur_tokens = clean_timestamp_tokens(tok, dur_int)
        sep = tok.convert_tokens_to_ids(">")
        rows = []
        for j in range(frames_for_t5.shape[0]):
            tt = clean_timestamp_tokens(tok, ts_int[j])
            parts = []
            for i, tks in enumerate(tt):
                parts.append(frames_for_t5[j, i * n_per_frame:(i + 1) * n_per_frame])
                parts.append(emb[torch.tensor(tks)])
            parts.append(emb[torch.tensor([sep])])
            parts.append(emb[torch.tensor(dur_tokens[j])])
            rows.append(torch.cat(parts))
        mx = max(r.shape[0] for r in rows)
        rows = [torch.cat([torch.zeros(mx - r.shape[0], r.shape[1]), r]) for r in rows]  # left zero-pad, mask stays 1
        video = torch.stack(rows)
        embs = torch.cat([video, emb[end_tok.input_ids], emb[text_tok.input_ids]], 1)
        atts = torch.cat([torch.ones(video.shape[:2], dtype=torch.long), end_tok.attention_mask, text_tok.attention_mask], 1)
        return embs, atts

    # ---- whole train-step forward (blip2_mr.py:433-570) -------------------------------------------------
    def forward_mr(self, tok, samples: dict, repl: Dict[int, int], mean_pool: bool = False, time_format: str = "seconds_integers",
                   interleave: bool = True):
        video = samples["video"]
        b, t = video.shape[:2]
        with torch.no_grad():
            vit_out = self.vit(video.reshape(b * t, *video.shape[2:]))
        img = self.ln_vision(vit_out)
        q = self.qformer(img)
        self._tower = "proj"
        f = self.lin(q, self.P("t5_proj.weight"), self.P("t5_proj.bias"))
        if mean_pool:
            f = f.mean(dim=1, keepdim=True)
        n = f.shape[1]
        f = f.reshape(b, t * n, -1)
        embs, atts = self.prompt_concatenation(tok, samples["timestamps"], samples["duration"], f,
                                               samples["video_prompt_end"], samples["query_prompt"],
                                               samples["task_prompt"], repl, n, time_format=time_format, interleave=interleave)
        ans = tok(samples["relevant_windows"], padding="longest", truncation=True, max_length=200, return_tensors="pt")
        labels = ans.input_ids.masked_fill(ans.input_ids == tok.pad_token_id, -100)
        loss, logits, enc = self.t5_loss(embs, atts, labels, ans.attention_mask)
        return dict(loss=loss, logits=logits, inputs_embs=embs, inputs_atts=atts, labels=labels, enc=enc, frames=f)


    # ---- generate (blip2_mr.py:826-946): encoder once, then the decoder re-run on the growing prefix ---------
    def encode_for_generate(self, tok, samples: dict, repl: Dict[int, int], mean_pool: bool = False, time_format: str = "seconds_integers"):
        """the encoder side of BLIP2_MR.generate (blip2_mr.py:848-881: same frame encoding and prompt_concatenation as forward_mr, then
        t5_model.generate(inputs_embeds=..., attention_mask=...)): returns (encoder output [B,S,d], attention mask [B,S])"""
        video = samples["video"]
        b, t = video.shape[:2]
        f = self.lin(self.qformer(self.ln_vision(self.vit(video.reshape(b * t, *video.shape[2:])))), self.P("t5_proj.weight"), self.P("t5_proj.bias"))
        if mean_pool:
            f = f.mean(dim=1, keepdim=True)
        n = f.shape[1]
        embs, atts = self.prompt_concatenation(tok, samples["timestamps"], samples["duration"], f.reshape(b, t * n, -1), samples["video_prompt_end"],
                                               samples["query_prompt"], samples["task_prompt"], repl, n, time_format=time_format)
        return self.t5_encoder(embs, atts), atts

    def next_token_logprobs(self, seqs: Tensor, enc: Tensor, atts: Tensor, beams_per_clip: int = 1) -> Tensor:
        """log-probabilities of the next token for every row of ``seqs`` [B * K, L] (decoder start token first), rows b*K .. b*K+K-1
        attending to clip b's encoder output — what HF's generate evaluates per step (modeling_t5.py:1796-1877 without labels; the
        K/V cache HF uses is an optimisation of exactly this prefix re-run)."""
        K = beams_per_clip
        e, m = enc.repeat_interleave(K, 0), atts.repeat_interleave(K, 0)
        dec = self.t5_decoder(seqs, torch.ones_like(seqs), e, m)
        self._tower = "head"
        logits = self.t5lin(dec[:, -1:], "lm_head")[:, 0]
        return torch.log_softmax(logits.float(), -1)


    # ---- video-QA two-stage path (blip2_mr.py:309-431, 948-1314) -------------------------------------------------------------------
    @staticmethod
    def extract_frames(video: Tensor, timestamps: Tensor, durations: Tensor, moments, n_frames: int):
        """blip2_mr.py:1127-1164: per clip the frames whose timestamps are closest to [start, end] (start >= end: end = duration), padded
        with the last one / thinned with linspace(...).long() to n_frames.  Returns (frames [B, n, 3, H, W], index lists)."""
        out, idxs = [], []
        for i, (start, end) in enumerate(moments):
            if start >= end:
                end = durations[i].item()
            s_i = torch.argmin(torch.abs(timestamps[i] - start)).item()
            e_i = torch.argmin(torch.abs(timestamps[i] - end)).item()
            ids = list(range(s_i, e_i + 1))
            assert len(ids) > 0, "No frames found for the relevant moment."
            if len(ids) < n_frames:
                ids = ids + [ids[-1]] * (n_frames - len(ids))
            elif len(ids) > n_frames:
                ids = [ids[j] for j in torch.linspace(0, len(ids) - 1, n_frames).long().tolist()]
            idxs.append(ids)
            out.append(video[i, ids])
        return torch.stack(out), idxs

    @staticmethod
    def relevant_moments(preds: Sequence[str], durations: Tensor):
        """blip2_mr.py:1098-1117 (get_relevant_frames): the localizer's answer string -> ONE [start, end] per clip (no window -> the whole
        video; several -> the first; an end beyond the duration -> round(duration))."""
        out = []
        for i, sample in enumerate(preds):
            m = moment_str_to_list(sample)
            if m == [[-1, -1]]:
                m = [0, durations[i].item()]
            else:
                m = m[0]
            if m[1] > durations[i].item():
                m[1] = round(durations[i].item())
            out.append(m)
        return out

    def frame_tokens(self, frames: Tensor, mean_pool: bool = False) -> Tensor:
        """get_frame_embeddings_and_attentions (blip2_mr.py:948-988): [B, t, 3, H, W] -> [B, t * n, d] (attention mask: all ones)"""
        b, t = frames.shape[:2]
        with torch.no_grad():
            f = self.lin(self.qformer(self.ln_vision(self.vit(frames.reshape(b * t, *frames.shape[2:])))), self.P("t5_proj.weight"), self.P("t5_proj.bias"))
        if mean_pool:
            f = f.mean(dim=1, keepdim=True)
        return f.reshape(b, t * f.shape[1], -1)

    def qa_inputs(self, tok, frames_for_t5: Tensor, qa_input: Sequence[str], max_txt_len: int = 200):
        """[ frame tokens | question tokens (right padded, mask 0 on the pads) ] with the ANSWERER's embedding table (blip2_mr.py:365-383)"""
        q = tok(list(qa_input), padding="longest", truncation=True, max_length=max_txt_len, return_tensors="pt")
        emb = self._t5p("shared.weight")
        embs = torch.cat([frames_for_t5, emb[q.input_ids]], 1)
        atts = torch.cat([torch.ones(frames_for_t5.shape[:2], dtype=torch.long), q.attention_mask], 1)
        return embs, atts

    def forward_qa(self, tok, samples: dict, n_frames: int, moments=None, mean_pool: bool = False, max_txt_len: int = 200):
        """forward_QA (blip2_mr.py:309-431).  moments = None: uniform sampling over the whole video (use_localizer False); otherwise the
        [start, end] per clip the localizer stage produced (``relevant_moments`` of its generate() output).  The frame features come
        out of the shared ViT / ln_vision / Q-Former / t5_proj WITHOUT gradient; the loss is the answerer T5's."""
        if moments is None:
            moments = [[0, d.item()] for d in samples["duration"]]
        frames, idxs = self.extract_frames(samples["video"], samples["timestamps"], samples["duration"], moments, n_frames)
        f = self.frame_tokens(frames, mean_pool)
        prev, self.t5_prefix = self.t5_prefix, "answerer_model."
        try:
            embs, atts = self.qa_inputs(tok, f, samples["qa_input"], max_txt_len)
            ans = tok(list(samples["qa_output"]), padding="longest", truncation=True, max_length=max_txt_len, return_tensors="pt")
            labels = ans.input_ids.masked_fill(ans.input_ids == tok.pad_token_id, -100)
            loss, logits, enc = self.t5_loss(embs, atts, labels, ans.attention_mask)
        finally:
            self.t5_prefix = prev
        return dict(loss=loss, logits=logits, inputs_embs=embs, inputs_atts=atts, labels=labels, enc=enc, frame_idx=idxs, frames=f)

    ANSWER_IDS = (71, 272, 205, 309, 262)   # "A" .. "E" in the flan-t5 vocabulary (blip2_mr.py:1299)

    def qa_answer(self, tok, samples: dict, n_frames: int, moments=None, mean_pool: bool = False, min_length: int = 8, max_txt_len: int = 200):
        """videoQA_answer (blip2_mr.py:1237-1314): greedy answerer_model.generate(num_beams=1, min_length=8, output_scores=True) and the
        argmax of the SECOND step's scores over the five option tokens.  Restated: HF's greedy search evaluates step 0 from the decoder
        start token (EOS is suppressed while the sequence is shorter than min_length: MinLengthLogitsProcessor), appends its argmax and
        evaluates step 1; `scores` are the processed logits, whose option columns equal the raw ones.  Returns (predicted option index per
        clip, option logits [B, 5], first greedy token).  (The question is embedded with the LOCALIZER's table there, blip2_mr.py:1262.)"""
        if moments is None:
            moments = [[0, d.item()] for d in samples["duration"]]
        frames, _ = self.extract_frames(samples["video"], samples["timestamps"], samples["duration"], moments, n_frames)
        f = self.frame_tokens(frames, mean_pool)
        embs, atts = self.qa_inputs(tok, f, samples["qa_input"], max_txt_len)          # t5_model.encoder.embed_tokens
        prev, self.t5_prefix = self.t5_prefix, "answerer_model."
        try:
            with torch.no_grad():
                enc = self.t5_encoder(embs, atts)
                B = embs.shape[0]
                seqs = torch.zeros(B, 1, dtype=torch.long)
                lp0 = self.next_token_logprobs(seqs, enc, atts)
                if min_length > 1:
                    lp0[:, 1] = float("-inf")
                first = lp0.argmax(-1)
                dec = self.t5_decoder(torch.cat([seqs, first[:, None]], 1), torch.ones(B, 2, dtype=torch.long), enc, atts)
                logits1 = self.t5lin(dec[:, -1:], "lm_head")[:, 0].float()
        finally:
            self.t5_prefix = prev
        opt = logits1[:, list(self.ANSWER_IDS)]
        return opt.argmax(-1).tolist(), opt, first


# ====================================================================================== dropout mask restatement
def _u32(x):
    return x & 0xFFFFFFFF


def dropout_hash(idx: Tensor, seed: int, site: int) -> Tensor:
    """Restates csrc/common.h mrb_hash (uint32 arithmetic emulated in int64).  idx: int64 tensor of PAIR indices."""
    idx = idx.to(torch.int64)
    h = _u32(_u32((idx ^ (seed & 0xFFFFFFFF)) * 0x9E3779B1) + _u32(site * 0x85EBCA77))
    h = h ^ (h >> 15)
    h = _u32(h * 0xC2B2AE3D)
    h = h ^ (h >> 13)
    return h


def dropout_keep(shape, seed: int, site: int, p: float) -> Tensor:
    """keep mask (float 0/1) of an element-indexed dropout site: element idx = flat row-major index mod 2^32; one 32-bit hash
    per index pair, low half -> even index, high half -> odd index, keep iff the 16-bit draw >= round(p * 65536)."""
    n = 1
    for s in shape:
        n *= s
    idx = torch.arange(n, dtype=torch.int64) & 0xFFFFFFFF
    h = dropout_hash(idx >> 1, seed, site)
    draw = torch.where((idx & 1) == 1, h >> 16, h & 0xFFFF)
    thresh = int(p * 65536.0 + 0.5)
    return (draw >= thresh).reshape(shape).float()


def dropout_hash_lin(idx: Tensor, seed: int, site: int) -> Tensor:
    """Restates csrc/common.h mrb_lin_fin(idx * MRB_H1 + mrb_lin_base(seed, site)) — the linear-index form of the counter hash used
    by the attention-probability dropout (uint32 arithmetic emulated in int64)."""
    idx = idx.to(torch.int64)
    base = _u32(_u32((seed & 0xFFFFFFFF) * 0x9E3779B1) + _u32((site & 0xFFFFFFFF) * 0x85EBCA77))
    t = _u32(_u32(idx * 0x9E3779B1) + base)
    t = t ^ (t >> 15)
    t = _u32(t * 0xC2B2AE3D)
    t = t ^ (t >> 13)
    return t


def dropout_hash_lin24(idx: Tensor, seed: int, site: int) -> Tensor:
    """Restates csrc/common.h mrb_lin_fin24(idx * MRB_H1 + mrb_lin_base(seed, site)): the linear-index counter hash whose finaliser runs
    on the 24-bit multiplier (uint32 arithmetic emulated in int64)."""
    idx = idx.to(torch.int64)
    base = _u32(_u32((seed & 0xFFFFFFFF) * 0x9E3779B1) + _u32((site & 0xFFFFFFFF) * 0x85EBCA77))
    t = _u32(_u32(idx * 0x9E3779B1) + base)
    t = t ^ (t >> 15)
    t = _u32((t & 0xFFFFFF) * 0x9E3779)
    t = t ^ (t >> 13)
    return t


def dropout_keep_attn(B: int, H: int, Sq: int, Sk: int, seed: int, site: int, p: float) -> Tensor:
    """keep mask [B,H,Sq,Sk] of the attention-probability dropout (csrc/attention.hip, "draws v3"): one hash h per
    (row = (b*H+h)*Sq+q, key QUAD): index = row * ceil(Sk/4) + key/4; g = rotr(h, 8); key 4i + j reads the 16-bit draw
    j = 0: h & 0xffff, 1: h >> 16, 2: g & 0xffff, 3: g >> 16; keep iff draw >= max(1, round(p * 65536))."""
    skq = (Sk + 3) // 4
    row = torch.arange(B * H * Sq, dtype=torch.int64)[:, None]
    key = torch.arange(Sk, dtype=torch.int64)[None, :]
    h = dropout_hash_lin24((row * skq + (key >> 2)) & 0xFFFFFFFF, seed, site)
    g = _u32((h >> 8) | (h << 24))
    j = key & 3
    w = torch.where(j >= 2, g, h)
    draw = torch.where((j & 1) == 1, w >> 16, w & 0xFFFF)
    return (draw >= max(1, int(p * 65536.0 + 0.5))).reshape(B, H, Sq, Sk).float()
