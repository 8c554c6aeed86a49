"""``blip2_mr`` — the reference's registry entry (blip2_mr.py:49-50) backed by the MI355X-native engine.

Same surface as the reference class: ``PRETRAINED_MODEL_CONFIG_DICT``, ``from_config(cfg)`` (reads the keys of
blip2_mr.py:1422-1441), ``forward(samples) -> {"loss"}``, ``generate(samples, ...) -> {"duration","prediction",
"raw_prediction","answer","qid"}``, ``load_checkpoint``, ``state_dict`` with the reference's parameter names for the
trainable tensors (LoRA in peft naming, t5_proj, ln_vision).  The frozen backbones live inside ``mrblip.engine`` as
packed bf16 operands; ``loss.backward()`` hands autograd the gradient the HIP backward already produced.
"""
import logging
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from lavis.common.registry import registry
from lavis.models.base_model import BaseModel
from lavis.models.blip2_mr_models.utils import post_process
from mrblip import prompt as P
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource, StateDictSource
from mrblip.tokenizer import load_tokenizer


class _TrainStep(torch.autograd.Function):
    """forward = whole HIP train step (loss AND gradients); backward = hand the flat gradients to autograd.

    Two modes.  Generic (any caller, any upstream scale g): the engine's gradient buffer holds THIS micro-step only and backward()
    returns g * grad; autograd accumulates it into the two flat Parameters' .grad, which are views of ONE buffer (model.flat_grad) —
    one all-reduce per optimizer step.  Fused accumulation (``model.begin_accumulation``; the LAVIS train loop of this package): the
    engine accumulates straight into its own flat buffer across the window's micro-steps (the Parameters' .grad alias it) and the
    upstream scale must be the announced one (1: the reference calls loss.backward() unscaled, moment_retrieval.py:219-223); it is
    applied once, inside AdamW, together with 1/world — no extra passes over the 78 MB buffer, and the exchange of the window's last
    micro-step overlaps with its t5_proj / Q-Former backward."""

    @staticmethod
    def forward(ctx, decay, no_decay, model, video, layout, need_grad, next_video=None, frame_tokens=None):
        eng = model.train_engine  # (grad mode is always off inside Function.forward, hence the explicit flag)
        if model._fused_scale is None:
            eng.zero_grad()
        if frame_tokens is not None:   # video-QA: the ANSWERER's step on frame tokens the shared towers produced without gradient
            loss = eng.forward_backward(None, layout, backward=need_grad, frames=frame_tokens, train_frames=False)
        else:
            loss = eng.forward_backward(video, layout, backward=need_grad, next_video=next_video)
        ctx.model = model
        return loss.clone().reshape(())

    @staticmethod
    def backward(ctx, g):
        model = ctx.model
        eng = model.train_engine
        if model._fused_scale is not None:
            model._check_fused_scale(g)   # (device-side compare, read one step late: no host sync behind ~2900 queued launches)
            return None, None, None, None, None, None, None, None
        model._ensure_named_grads()
        nd = eng.n_decay
        return eng.grad[:nd] * g, eng.grad[nd:] * g, None, None, None, None, None, None


@registry.register_model("blip2_mr")
class BLIP2_MR(BaseModel):
    PRETRAINED_MODEL_CONFIG_DICT = {
        "pretrain_flant5xl": "configs/models/blip2/blip2_pretrain_flant5xl.yaml",
        "tiny_synthetic": "configs/models/blip2/blip2_tiny_synthetic.yaml",
    }

    def __init__(self, img_size=224, drop_path_rate=0, use_grad_checkpoint=False, vit_precision="fp16", freeze_vit=True,
                 num_query_token=32, t5_model="google/flan-t5-xl", num_beams=5, prompt="", max_txt_len=200, apply_lemmatizer=False,
                 input_time_format="seconds_integers", interleave_data=False, frame_token_aggregation=None, task="lora",
                 num_frames_for_answer=4, resample_frames=False, engine_config: Optional[EngineConfig] = None, weights=None,
                 tokenizer=None, device=None, seed=None, synthetic_weights=False):
        super().__init__()
        if not freeze_vit:
            raise NotImplementedError("the MI355X engine keeps the ViT frozen (every Mr. BLIP config sets freeze_vit: True)")
        if "QA" in task and resample_frames:
            # get_relevant_frames_resampled (blip2_mr.py:1166-1235) re-decodes the clip from samples["video_path"] for the localized window
            raise NotImplementedError("video-QA with resample_frames=True re-reads the video file through the eval video processor: not built "
                                      "(resample_frames=False selects the answerer's frames from the frames already in the batch)")
        if input_time_format not in ("seconds_integers", "seconds_floats"):
            raise NotImplementedError(f"input_time_format={input_time_format!r}: 'seconds_integers' (every shipped config) and 'seconds_floats' are "
                                      "implemented; the reference's relative_*/framenumbers formats are broken upstream (SURVEY.md §8c)")
        # interleave_data: False (the reference constructor's default, blip2_mr.py:82; every shipped config sets True) = the plain prompt
        # [video_prompt text | all frame tokens | video_prompt_end | text] of blip2_mr.py:783-822: mrblip.prompt._plain_layout
        if "lora" not in task or "qformer_freeze" not in task:
            raise NotImplementedError("task must contain 'lora' and 'qformer_freeze' (the shipped Mr. BLIP fine-tuning recipe)")
        assert frame_token_aggregation in (None, False, "mean"), "Invalid aggregation method, please choose from ['mean']"
        self.task, self.num_beams, self.max_txt_len = task, num_beams, max_txt_len
        self.input_time_format, self.interleave_data = input_time_format, interleave_data
        self.frame_token_aggregation = frame_token_aggregation
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        cfg = engine_config or EngineConfig(img=img_size, num_query=num_query_token)
        cfg.mean_pool = frame_token_aggregation == "mean"
        self.t5_tokenizer = tokenizer or load_tokenizer(t5_model, allow_fixture=bool(synthetic_weights))
        self.annoying_numbers, _ = P.find_annoying_numbers(self.t5_tokenizer, 200)
        self.annoying_numbers_replacement_dict = P.annoying_replacement_dict(self.annoying_numbers)
        if weights is None:
            # the reference downloads eva_vit_g / BLIP-2 / flan-t5 here (blip2_mr.py:127-151); without them the only honest options are
            # an explicit synthetic run or an error — never a silently random backbone
            if not synthetic_weights:
                raise RuntimeError("blip2_mr: no backbone weights.  Give model.vit_weights (eva_vit_g.pth), model.pretrained (BLIP-2 stage-2 "
                                   "checkpoint) and model.t5_weights (HF flan-t5 directory), or pass weights=<state dict>; set "
                                   "model.synthetic_weights: True for a random-weight (benchmark / plumbing) run")
            logging.warning("blip2_mr: synthetic_weights — ViT, Q-Former and T5 are RANDOM (seeded); only throughput/plumbing is meaningful")
            src = RandomSource(self._device, seed=1234)
        else:
            src = StateDictSource(weights) if isinstance(weights, dict) else weights
        self.engine = MrBlipEngine(cfg, src, self._device, seed=seed)
        # ---- video-QA variants (blip2_mr.py:103-118, 152-161, 206-236): the moment-retrieval model above becomes the frozen LOCALIZER and a
        # second T5 with its own LoRA — the ANSWERER, keys answerer_model.* — is what trains; ViT / ln_vision / Q-Former / t5_proj are shared
        self.is_qa = "QA" in task
        self.use_localizer = "with_localizer" in task
        self.use_oracle_localizer = "oracle_localizer" in task
        self.resample_frames, self.num_frames_for_answer = resample_frames, int(num_frames_for_answer)
        self.ANS_MAPPING_C_TO_I = {"A": 0, "B": 1, "C": 2, "D": 3, "E": 4}
        self.ANS_MAPPING_I_TO_C = {0: "A", 1: "B", 2: "C", 3: "D", 4: "E"}
        self.answerer = None
        if self.is_qa:
            self.answerer = MrBlipEngine(cfg, src, self._device, seed=seed, t5_prefix="answerer_model.", share_towers=self.engine)
            self.answerer_tokenizer = self.t5_tokenizer
            # the reference freezes the localizer's T5 by casting it to bf16 (blip2_mr.py:206-209): its embedding rows are bf16 values
            self.engine.emb.copy_(self.engine.emb.bfloat16().float())
        self.train_engine = self.answerer if self.is_qa else self.engine       # the engine whose flat buffer the optimizer updates
        nd = self.train_engine.n_decay
        # the trainable tensors as two flat Parameters that ALIAS the engine's buffer (AdamW decay / no-decay groups)
        self.trainable_decay = nn.Parameter(self.train_engine.flat[:nd])
        self.trainable_no_decay = nn.Parameter(self.train_engine.flat[nd:])
        # ONE flat gradient buffer behind both Parameters (generic mode: autograd accumulates g * engine.grad into it)
        self.flat_grad = torch.zeros_like(self.train_engine.flat)
        self._fused_scale = None
        self._named = None                                   # reference-named per-tensor Parameters, built on first use
        self._flat_version = self.train_engine.flat._version       # torch-side writes to the trainable buffer bump it (see forward)
        self._bind_grads(self.flat_grad)
        self.post_process = post_process

    # ------------------------------------------------------------------ parameters under the reference's names
    per_tensor_parameters = True   # named_parameters() / parameters(): reference names (False: the two flat Parameters)

    def _trainable_views(self, buf):
        """reference parameter name -> view of ``buf``, a flat fp32 buffer laid out like the engine's trainable buffer.  LoRA in peft's
        naming (blip2_mr.py:182-200 wraps t5_model in a PeftModel): lora_A.default.weight [r, in], lora_B.default.weight [out, r] —
        the engine stores B transposed, so that one is a transposed (non-contiguous) view."""
        eng = self.train_engine
        base = eng.flat.data_ptr()

        def like(t, transpose=False):
            off = (t.data_ptr() - base) // 4
            v = buf[off: off + t.numel()].view(t.shape)
            return v.t() if transpose else v

        out = OrderedDict()
        if not self.is_qa:   # (video-QA: the frame path runs without gradient, only the answerer's LoRA tensors train)
            out["ln_vision.weight"], out["ln_vision.bias"] = like(eng.lnv_w), like(eng.lnv_b)
        for a in eng.adapters:
            name = eng.t5_prefix + "base_model.model." + a.name
            out[name + ".lora_A.default.weight"] = like(a.A)
            out[name + ".lora_B.default.weight"] = like(a.Bt, True)
        if not self.is_qa:
            out["t5_proj.weight"], out["t5_proj.bias"] = like(eng.proj_w), like(eng.proj_b)
        return out

    def reference_named_parameters(self):
        """The trainable tensors as individual ``nn.Parameter``s under the reference's ``named_parameters()`` names, ALIASING the flat
        buffer the engine trains (and their ``.grad`` aliasing the flat gradient): anything that selects parameters by name — the
        reference's weight-decay grouping (runner_base.py:111-122: ``p.ndim < 2 or "bias" in n or "ln" in n or "bn" in n``), gradient
        clipping, a stock ``torch.optim`` optimizer — works on the engine's own memory.  forward() notices torch-side updates of the
        buffer (tensor version counter) and re-derives the bf16 operand copies."""
        if self._named is None:
            self._named = OrderedDict((n, nn.Parameter(v)) for n, v in self._trainable_views(self.train_engine.flat).items())
            self._rebind_named_grads()
        return self._named

    def _rebind_named_grads(self):
        if self._named is not None:
            for p, g in zip(self._named.values(), self._trainable_views(self.grad_buffer()).values()):
                p.grad = g

    def _ensure_named_grads(self):
        """``optimizer.zero_grad(set_to_none=True)`` on the per-tensor Parameters drops their aliases of the flat gradient WITHOUT
        zeroing it: do what the caller meant (zero the buffer) and re-alias"""
        if self._named is not None and next(iter(self._named.values())).grad is None:
            if self._fused_scale is None:
                self.flat_grad.zero_()
            self._rebind_named_grads()

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        if not self.per_tensor_parameters:
            yield from super().named_parameters(prefix=prefix, recurse=recurse, remove_duplicate=remove_duplicate)
            return
        for n, p in self.reference_named_parameters().items():
            yield (prefix + "." if prefix else "") + n, p

    # ------------------------------------------------------------------ gradients: one flat buffer
    def _bind_grads(self, buf):
        nd = self.train_engine.n_decay
        self.trainable_decay.grad = buf[:nd]
        self.trainable_no_decay.grad = buf[nd:]
        self._rebind_named_grads()

    def zero_grad(self, set_to_none: bool = False):
        """keeps the flat views (set_to_none would un-alias them)"""
        self.grad_buffer().zero_()

    def grad_buffer(self) -> torch.Tensor:
        """the flat fp32 gradient of every trainable tensor: the only thing data-parallel ranks exchange"""
        return self.train_engine.grad if self._fused_scale is not None else self.flat_grad

    def grad_scale(self) -> float:
        """factor AdamW applies to grad_buffer() (fused mode: the loss scale the engine did not apply)"""
        return self._fused_scale if self._fused_scale is not None else 1.0

    # The upstream scale of loss.backward() must be the announced one in fused mode.  ``g`` lives on the device and the stream is ~2900
    # launches deep at that point: float(g) would drain it (and the host would then re-fill the queue while the GPU idles).  The
    # comparison therefore runs on the device, its verdict travels to pinned host memory behind the step, and it is READ one step late
    # (or at end_accumulation / by check_fused_scale_now) — the same trick as the train loop's one-step-late loss.item().
    _scale_flag_dev = None
    _scale_flag_host = None
    _scale_flag_event = None

    def _check_fused_scale(self, g):
        if not g.is_cuda:   # a host scalar costs nothing to read
            if abs(float(g) - self._fused_scale) > 1e-6 * abs(self._fused_scale):
                raise RuntimeError(self._fused_scale_message(float(g)))
            return
        self._raise_if_scale_violation(block=False)   # verdict of an EARLIER micro-step, if it has arrived
        if self._scale_flag_dev is None:
            self._scale_flag_dev = torch.zeros(2, dtype=torch.float32, device=g.device)
            self._scale_flag_host = torch.zeros(2, dtype=torch.float32).pin_memory()
        bad = ((g.detach().reshape(()).float() - self._fused_scale).abs() > 1e-6 * abs(self._fused_scale)).float()
        self._scale_flag_dev[0] += bad                 # sticky: counts the violations of the run
        self._scale_flag_dev[1] = torch.where(bad > 0, g.detach().reshape(()).float(), self._scale_flag_dev[1])
        self._scale_flag_host.copy_(self._scale_flag_dev, non_blocking=True)
        self._scale_flag_event = torch.cuda.Event()
        self._scale_flag_event.record()

    def _fused_scale_message(self, got):
        return (f"fused gradient accumulation expects the loss scale {self._fused_scale}, got {got}; "
                "call model.end_accumulation() to use arbitrary loss scaling")

    def _raise_if_scale_violation(self, block: bool):
        ev = self._scale_flag_event
        if ev is None:
            return
        if block:
            ev.synchronize()
        elif not ev.query():
            return
        if float(self._scale_flag_host[0]) > 0:
            got = float(self._scale_flag_host[1])
            self._scale_flag_dev.zero_()
            self._scale_flag_event = None
            raise RuntimeError(self._fused_scale_message(got))

    def check_fused_scale_now(self):
        """blocking form of the deferred loss-scale check (tests; end of an epoch)"""
        self._raise_if_scale_violation(block=True)

    def begin_accumulation(self, loss_scale: float = 1.0):
        """fused mode (see _TrainStep): micro-step gradients accumulate in the engine's buffer; every loss.backward() of the window
        must come with the upstream scale ``loss_scale``"""
        self._fused_scale = float(loss_scale)
        self._bind_grads(self.train_engine.grad)

    def end_accumulation(self):
        self._raise_if_scale_violation(block=True)
        self._fused_scale = None
        self._bind_grads(self.flat_grad)

    # ------------------------------------------------------------------ reference surface
    @classmethod
    def from_config(cls, cfg):
        ecfg = None
        if cfg.get("engine"):
            ecfg = EngineConfig(**dict(cfg.engine))
        # frozen backbones: the reference's three downloads as local paths (checkpoint.py); `pretrained` keeps its reference meaning
        # (the BLIP-2 stage-2 file with Q-Former / query_tokens / ln_vision / t5_proj, blip2.py:86-104)
        weights = None
        paths = dict(vit=cfg.get("vit_weights") or None, blip2=cfg.get("pretrained") or None, t5=cfg.get("t5_weights") or None)
        if any(paths.values()):
            from mrblip import checkpoint as CK
            probe = ecfg or EngineConfig(img=cfg.get("image_size", 224), num_query=cfg.get("num_query_token", 32))
            weights, report = CK.assemble_state_dict(probe, **paths)
            logging.info("Missing keys {}".format(report["missing"]))  # (the reference's non-strict log line, blip2.py:100)
            if report["unexpected"]:
                logging.info("unexpected keys (ignored) %s", CK.describe(dict(unexpected=report["unexpected"])))
            if report["bad_shape"]:
                raise RuntimeError("checkpoint tensors with the wrong shape: %s" % CK.describe(dict(bad_shape=report["bad_shape"])))
            backbone_missing = [k for k in report["missing"] if not (k.startswith("t5_proj.") or k.startswith("ln_vision."))]
            if backbone_missing:
                raise RuntimeError("incomplete backbone weights (vit_weights / pretrained / t5_weights): %s" % CK.describe(dict(missing=backbone_missing)))
            if "t5_proj.weight" in report["missing"]:   # a BLIP-2 file without t5_proj: the reference keeps its fresh nn.Linear init (blip2_mr.py:270-272)
                # nn.Linear's default init (kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)) for weight and bias) from a FIXED-seed
                # generator: the global RNG is seeded with run.seed + rank (train.py) and nothing broadcasts the trainable tensors from
                # rank 0 as the reference's DDP wrapper does — data-parallel replicas must start from identical t5_proj weights
                gen = torch.Generator().manual_seed(4322)
                bound = 1.0 / (probe.qf_dim ** 0.5)
                weights["t5_proj.weight"] = (torch.rand(probe.d_model, probe.qf_dim, generator=gen) * 2 - 1) * bound
                weights["t5_proj.bias"] = (torch.rand(probe.d_model, generator=gen) * 2 - 1) * bound
            if "ln_vision.weight" in report["missing"]:
                weights["ln_vision.weight"], weights["ln_vision.bias"] = torch.ones(probe.vit_dim), torch.zeros(probe.vit_dim)
        model = cls(
            img_size=cfg.get("image_size", 224), num_query_token=cfg.get("num_query_token", 32), t5_model=cfg.get("t5_model", "google/flan-t5-xl"),
            num_beams=cfg.get("num_beams", 5), drop_path_rate=cfg.get("drop_path_rate", 0), use_grad_checkpoint=cfg.get("use_grad_checkpoint", False),
            vit_precision=cfg.get("vit_precision", "fp16"), freeze_vit=cfg.get("freeze_vit", True), prompt=cfg.get("prompt", ""),
            max_txt_len=cfg.get("max_txt_len", 200), apply_lemmatizer=cfg.get("apply_lemmatizer", False),
            input_time_format=cfg.get("input_time_format", "seconds_integers"), interleave_data=cfg.get("interleave_data", False),
            frame_token_aggregation=cfg.get("frame_token_aggregation", None), task=cfg.get("task", "lora"),
            num_frames_for_answer=cfg.get("num_frames_for_answer", 4), resample_frames=cfg.get("resample_frames", False), engine_config=ecfg,
            weights=weights, synthetic_weights=cfg.get("synthetic_weights", False),
            seed=cfg.get("seed", None),  # None: the engine's dropout stream follows torch's seed = run.seed + rank (train.py setup_seeds)
        )
        model.load_checkpoint_from_config(cfg)
        return model

    def train(self, mode=True):
        super().train(mode)
        self.engine.training = mode
        if self.answerer is not None:
            self.answerer.training = mode
        return self

    def _layout(self, samples):
        T = samples["video"].shape[1]
        n = 1 if self.engine.cfg.mean_pool else self.engine.cfg.num_query
        return P.build_layout(self.t5_tokenizer, samples, self.annoying_numbers_replacement_dict, n, T, self.max_txt_len,
                              no_task_prompt="no_task_prompt" in self.task, time_format=self.input_time_format, interleave=bool(self.interleave_data))

    def _frames_to_device(self, v):
        """fp32 frames already normalised by the processor (the reference's contract), or raw uint8 frames [B,T,3,H,W]: those stay uint8 —
        a quarter of the H2D bytes — and ToTensor + Normalize(CLIP mean/std) is fused into the patch-embed load."""
        return v.to(self._device) if v.dtype == torch.uint8 else v.to(self._device, torch.float32)

    _staged_next = None  # (host tensor of the next batch's frames, its device copy): see forward()
    _reserved = False    # engine workspaces sized for the longest step (first forward)
    generate_cross_cache = True  # project the decoder's cross-attention K/V once per clip (False: per step and beam, for the A/B test)
    generate_self_cache = True   # self-attention K/V cache: one new position per decoding step (False: re-run the prefix, for the A/B test)

    def forward(self, samples):
        """samples: the reference's dict (blip2_mr.py:433-445).  Optional extra key ``next_video``: the NEXT batch's frames (the train
        loop's one-batch look-ahead, tasks/moment_retrieval.py); its frozen-ViT forward is overlapped with this step's T5 decoder."""
        if self.is_qa:
            return self.forward_QA(samples)
        src = samples["video"]
        if self._staged_next is not None and self._staged_next[0] is src:
            video = self._staged_next[1]  # the device copy whose ViT features were prefetched during the previous step
        else:
            video = self._frames_to_device(src)
        self._staged_next = None
        layout = self._layout(samples)
        if not self._reserved:
            # the encoder length follows the query's token count, the decoder length the answer's (blip2_mr.py:572-824): size every
            # workspace ONCE for the longest step this configuration can produce (text truncated to max_txt_len tokens; timestamps of
            # later clips may tokenise a little longer than this first one: +12 %), so that the epoch's stream of different lengths
            # neither allocates nor synchronises (engine.buf / engine.reserve)
            # (the decoder's few-row buffers are left to grow on demand — a longer answer than any seen so far costs one re-allocation)
            self.engine.reserve(layout.S, int(layout.S * 1.12) + int(self.max_txt_len))
            self._reserved = True
        need_grad = torch.is_grad_enabled() and (self.trainable_decay.requires_grad or self.trainable_no_decay.requires_grad)
        self._sync_trainable_operands()
        if need_grad:
            self._ensure_named_grads()
        nxt = samples.get("next_video") if need_grad else None
        nxt_dev = None
        if nxt is not None:
            nxt_dev = self._frames_to_device(nxt)
            self._staged_next = (nxt, nxt_dev)
        loss = _TrainStep.apply(self.trainable_decay, self.trainable_no_decay, self, video, layout, need_grad, nxt_dev)
        return {"loss": loss}

    def _sync_trainable_operands(self):
        eng = self.train_engine
        if eng.flat._version != self._flat_version:
            # a torch-side optimizer / load wrote the trainable buffer through one of its aliases (the engine's own AdamW kernel does
            # not go through torch and re-packs by itself): re-derive the bf16 GEMM operand copies of LoRA A / B and t5_proj
            eng.refresh_trainable()
            self._flat_version = eng.flat._version

    # ------------------------------------------------------------------ video-QA two-stage path (blip2_mr.py:309-431, 948-1314)
    def extract_frames(self, samples, relevant_moments, num_frames_for_answer):
        """blip2_mr.py:1127-1164: per clip the batch's frames whose timestamps are closest to [start, end] (start >= end: end = duration),
        padded with the last frame / thinned with linspace(...).long() to num_frames_for_answer -> [B, n, 3, H, W]"""
        video, ts = samples["video"], samples["timestamps"]
        out = []
        for i, (start, end) in enumerate(relevant_moments):
            if start >= end:
                end = samples["duration"][i].item()
            s_i = torch.argmin(torch.abs(ts[i] - start)).item()
            e_i = torch.argmin(torch.abs(ts[i] - end)).item()
            frames = video[i, s_i: e_i + 1]
            assert frames.shape[0] > 0, "No frames found for the relevant moment."
            if frames.shape[0] < num_frames_for_answer:
                frames = torch.cat([frames, frames[-1:].expand(num_frames_for_answer - frames.shape[0], *frames.shape[1:])])
            elif frames.shape[0] > num_frames_for_answer:
                frames = frames[torch.linspace(0, frames.shape[0] - 1, num_frames_for_answer).long().to(frames.device)]
            out.append(frames)
        return torch.stack(out)

    def get_relevant_frames(self, samples, relevant_moments_out, num_frames_for_answer):
        """blip2_mr.py:1098-1125: the localizer's answer strings -> one [start, end] per clip (none -> the whole video, several -> the
        first, an end beyond the duration -> round(duration)) and the frames extract_frames selects for it"""
        from lavis.models.blip2_mr_models.utils import moment_str_to_list
        relevant_moments = []
        for i, sample in enumerate(relevant_moments_out):
            m = moment_str_to_list(sample)
            if m == [[-1, -1]]:
                m = [0, samples["duration"][i].item()]
            else:
                m = m[0]
            if m[1] > samples["duration"][i].item():
                m[1] = round(samples["duration"][i].item())
            relevant_moments.append(m)
        assert len(relevant_moments) == samples["video"].shape[0]
        return relevant_moments, self.extract_frames(samples, relevant_moments, num_frames_for_answer)

    def get_relevant_frames_resampled(self, samples, relevant_moments, num_frames_for_answer):
        raise NotImplementedError("resample_frames=True re-reads samples['video_path'] through the eval video processor (blip2_mr.py:1166-1235): not built")

    @torch.no_grad()
    def get_frame_embeddings_and_attentions(self, image):
        """blip2_mr.py:948-988: [B, t, 3, H, W] -> frame tokens [B * t * n, d_model] (fp32, on the device; the attention mask is all ones) through
        the SHARED ViT / ln_vision / Q-Former / t5_proj of the localizer's engine.  The rows stay in that engine's workspace: consume them
        before its next call."""
        if isinstance(image, (list, tuple)):
            image = torch.stack(list(image))
        return self.engine.frames_forward(self._frames_to_device(image))[0]

    def _qa_layout(self, samples, n_frames):
        n = 1 if self.engine.cfg.mean_pool else self.engine.cfg.num_query
        answers = samples.get("qa_output") or [""] * len(samples["qa_input"])
        return P.build_qa_layout(self.t5_tokenizer, list(samples["qa_input"]), list(answers), n_frames * n, self.max_txt_len)

    def _localize(self, samples, generate_kwargs=None):
        """stage 1 of the QA path: [start, end] per clip and the frames the answerer sees"""
        n = self.num_frames_for_answer
        B = samples["video"].shape[0]
        if self.use_localizer:
            s2 = dict(samples)
            s2["relevant_windows"] = ["[[0, 0]]"] * B          # (the reference's dummy answer, blip2_mr.py:314)
            s2.setdefault("query_id", samples.get("question_id"))
            out_mr = self.generate(s2, **(generate_kwargs or {}))
            return self.get_relevant_frames(samples, out_mr["prediction"], n)
        if self.use_oracle_localizer and generate_kwargs is not None:   # (videoQA_generate only: the ground-truth windows, first one per clip)
            rw = samples["relevant_windows"]
            moments = [list(m[0]) for m in (rw.tolist() if torch.is_tensor(rw) else rw)]
            return moments, self.extract_frames(samples, moments, n)
        moments = [[0, d.item()] for d in samples["duration"]]      # uniform sampling over the whole video
        return moments, self.extract_frames(samples, moments, n)

    def forward_QA(self, samples):
        """blip2_mr.py:309-431.  Stage 1 (no gradient): the localizer's generate() — or the whole video — picks num_frames_for_answer
        frames per clip, which the shared towers turn into frame tokens.  Stage 2: the answerer T5 (its own LoRA: the only trainable
        tensors) on [frame tokens | question] -> CE loss of the answer; the HIP step returns loss and the LoRA gradients."""
        need_grad = torch.is_grad_enabled() and self.trainable_decay.requires_grad
        with torch.no_grad():
            moments, frames = self._localize(samples)
            fr = self.get_frame_embeddings_and_attentions(frames)
        self.last_relevant_moments = moments
        layout = self._qa_layout(samples, frames.shape[1])
        self._sync_trainable_operands()
        if need_grad:
            self._ensure_named_grads()
        loss = _TrainStep.apply(self.trainable_decay, self.trainable_no_decay, self, None, layout, need_grad, None, fr)
        return {"loss": loss}

    ANSWER_IDS = (71, 272, 205, 309, 262)     # "A" .. "E" in the flan-t5 vocabulary (blip2_mr.py:1299)

    @torch.no_grad()
    def videoQA_answer(self, samples, use_nucleus_sampling=False, num_beams=5, max_length=50, min_length=8, top_p=0.9, repetition_penalty=1.0,
                       length_penalty=1.0, num_captions=1, temperature=1, output_attentions=False):
        """blip2_mr.py:1237-1314: the answerer decodes greedily (num_beams=1 there, whatever the argument says) and the answer is the argmax
        of the SECOND step's scores over the five option tokens.  On the HIP decoder: encoder once, cross K/V once, two cached decoding
        steps; EOS is suppressed at step 0 while min_length > 1 (HF's MinLengthLogitsProcessor), sampling warps are not applied to the
        option argmax.  The question is embedded with the LOCALIZER's table, as the reference does (blip2_mr.py:1262)."""
        from mrblip import ops
        ans, loc = self.answerer, self.engine
        was = (ans.training, loc.training)
        ans.training = loc.training = False
        try:
            fr = self.get_frame_embeddings_and_attentions(samples["relevant_frames"])
            layout = self._qa_layout(samples, samples["relevant_frames"].shape[1])
            B, S, d = layout.attention_mask.shape[0], layout.S, ans.cfg.d_model
            L = ans._layout_dev(layout)
            inp = ans.buf("inputs_embeds", (B * S, d), torch.float32, zero=False)
            ops.row_copy(fr, L["frame_src"], inp, L["frame_dst"])
            ops.row_copy(loc.emb, L["emb_src"], inp, L["emb_dst"])
            enc = ans.t5_encoder_forward(inp, B, S, L["mask"], want_grad=False)
            cross = ans.t5_cross_kv(enc, B, S)
            state = ans.t5_decode_begin(B, 3)
            start = torch.zeros(B, dtype=torch.long)
            lg0 = ans.t5_decode_step(state, start, None, cross, B, L["mask"]).float()
            if int(min_length) > 1:
                lg0[:, 1] = float("-inf")
            first = lg0.argmax(-1).cpu()
            lg1 = ans.t5_decode_step(state, first, torch.arange(B), cross, B, L["mask"]).float()
            opt = lg1[:, list(self.ANSWER_IDS)]
            self.last_option_logits, self.last_first_token = opt.cpu(), first
            return {"output_text": opt.argmax(-1).cpu().tolist(), "answer": samples.get("qa_output"), "qid": samples.get("question_id"),
                    "relevant_moments_gt": samples.get("relevant_windows")}
        finally:
            ans.training, loc.training = was

    @torch.no_grad()
    def videoQA_generate(self, samples, num_frames_for_answer=4, use_nucleus_sampling=False, num_beams=5, max_length=50, min_length=8, top_p=0.9,
                         repetition_penalty=1.0, length_penalty=1.0, num_captions=1, temperature=1, output_attentions=False):
        """blip2_mr.py:990-1096: stage 1 picks the frames (localizer / oracle windows / whole video), stage 2 answers on them"""
        samples = dict(samples)
        samples.setdefault("relevant_windows", [[0, 0]])
        samples["query_id"] = samples.get("question_id")
        kw = dict(use_nucleus_sampling=use_nucleus_sampling, num_beams=num_beams, max_length=max_length, min_length=min_length, top_p=top_p,
                  repetition_penalty=repetition_penalty, length_penalty=length_penalty, num_captions=num_captions, temperature=temperature)
        moments, frames = self._localize(samples, generate_kwargs=kw)
        samples["relevant_frames"] = frames
        out = self.videoQA_answer(samples)
        out["relevant_moments"] = [moments]
        return out

    @torch.no_grad()
    def generate(self, samples, use_nucleus_sampling=False, num_beams=5, max_length=50, min_length=1, top_p=0.9, repetition_penalty=1.0,
                 length_penalty=1.0, num_captions=1, temperature=1, output_attentions=False):
        """Beam search over the HIP decoder (blip2_mr.py:826-946).  The encoder runs once; the cross-attention K/V of all 24 decoder
        layers are projected once per clip and shared by every step and every beam (engine.t5_cross_kv); each step runs ONE new
        position per beam against a self-attention K/V cache (engine.t5_decode_step).  ``generate_self_cache = False`` re-runs the
        whole decoder prefix each step instead (kept for the A/B test)."""
        # use_nucleus_sampling (HF do_sample=True with top_p / temperature), repetition_penalty and num_captions (num_return_sequences) follow
        # HF's generate as the reference calls it (blip2_mr.py:883-899): mrblip/search.py.  Sampling combined with num_beams > 1 is HF's
        # beam-sample mode, which no Mr. BLIP config uses: sampling here means num_beams = 1 (the HIP decoder is the same either way).
        # (temperature only warps logits when sampling in HF generate: with do_sample=False it is ignored, as here)
        n_ret = max(1, int(num_captions))
        if use_nucleus_sampling and int(num_beams) > 1:
            raise NotImplementedError("generate: use_nucleus_sampling with num_beams > 1 is HF's beam-sample mode (not implemented; pass num_beams=1)")
        if not use_nucleus_sampling and n_ret > max(1, int(num_beams)):
            raise ValueError("num_captions has to be smaller or equal to num_beams")
        eng = self.engine
        was_training = eng.training
        eng.training = False
        try:
            video = self._frames_to_device(samples["video"])
            s2 = dict(samples)
            s2.setdefault("relevant_windows", ["[[0, 0]]"] * video.shape[0])
            s2["relevant_windows"] = [str(w) for w in s2["relevant_windows"]]
            layout = self._layout(s2)
            B, S, d = video.shape[0], layout.S, eng.cfg.d_model
            fr, img, xv, qb = eng.frames_forward(video)
            L = eng._layout_dev(layout)
            inp = eng.buf("inputs_embeds", (B * S, d), torch.float32, zero=False)
            from mrblip import ops
            ops.row_copy(fr, L["frame_src"], inp, L["frame_dst"])
            ops.row_copy(eng.emb, L["emb_src"], inp, L["emb_dst"])
            enc = eng.t5_encoder_forward(inp, B, S, L["mask"], want_grad=False)
            K = n_ret if use_nucleus_sampling else max(1, int(num_beams))   # decoder rows per clip: beams, or sampled sequences
            # one beam with a repetition penalty is HF's GREEDY decoding, which penalises raw logits (beam search penalises log-probabilities)
            greedy_pen = (not use_nucleus_sampling) and K == 1 and float(repetition_penalty) != 1.0
            raw_logits = bool(use_nucleus_sampling) or greedy_pen            # the sampler warps RAW logits (repetition penalty is sign dependent)
            cross = eng.t5_cross_kv(enc, B, S) if self.generate_cross_cache else None
            if cross is None:  # reference-shaped fallback (kept for the A/B test): replicate the encoder rows per beam
                enc_k = enc.view(B, S, -1).repeat_interleave(K, 0).reshape(B * K * S, -1).contiguous()
                mask_k = None if L["mask"] is None else L["mask"].repeat_interleave(K, 0).contiguous()
            else:
                enc_k, mask_k = enc, L["mask"]
            # HF beam search semantics (mrblip/search.py, pinned against transformers' generate in tests/test_search_cpu.py); the step
            # function re-runs the short decoder prefix on the HIP decoder (weight-streaming bound: <= 64 rows cost what one row costs)
            from mrblip.search import beam_search

            if cross is not None and self.generate_self_cache and int(max_length) < 128:
                # incremental decoding (HF use_cache=True): one new position per step against the self-attention K/V cache, which the
                # search re-orders through `parents` (HF's _reorder_cache)
                state = eng.t5_decode_begin(B * K, int(max_length) + 1)

                def step_fn(seqs, parents):
                    logits = eng.t5_decode_step(state, seqs[:, -1], parents, cross, B, mask_k)
                    return logits.float().cpu() if raw_logits else torch.log_softmax(logits.float(), -1).cpu()

                step_fn.takes_parents = True
            else:
                def step_fn(seqs):
                    Ld = seqs.shape[1]
                    _, logits = eng.t5_decoder_forward(seqs, torch.ones(B * K, Ld, dtype=torch.int32), enc_k, B * K, S, mask_k, labels=None,
                                                       cross_cache=cross, cross_batch=B if cross is not None else None)
                    last = logits.view(B * K, Ld, -1)[:, -1].float()
                    return last.cpu() if raw_logits else torch.log_softmax(last, -1).cpu()

            if use_nucleus_sampling or greedy_pen:
                from mrblip.search import sample_search
                best = sample_search(step_fn, B, n_ret, int(max_length), min_length=int(min_length), top_p=float(top_p), temperature=float(temperature),
                                     repetition_penalty=float(repetition_penalty), eos_id=1, pad_id=0, start_id=0,
                                     generator=getattr(self, "sampling_generator", None), greedy=greedy_pen)
            else:
                best = beam_search(step_fn, B, K, int(max_length), min_length=int(min_length), length_penalty=float(length_penalty),
                                   eos_id=1, pad_id=0, start_id=0, repetition_penalty=float(repetition_penalty), num_return=n_ret)
            self.last_sequences = [seq.clone() for seq in best]   # token ids of the winning hypotheses (start token first): parity tests
            out_text = [self.t5_tokenizer.decode(seq[1:], skip_special_tokens=True) for seq in best]
            raw = list(out_text)
            pred = [self.post_process(t) for t in out_text]
            # (num_captions > 1: B * num_captions predictions, item-major, beside B durations / answers / qids — as the reference returns them)
            return {"duration": [float(x) for x in samples["duration"]], "prediction": pred, "raw_prediction": raw,
                    "answer": samples.get("relevant_windows", [""] * B), "qid": samples.get("query_id", [str(i) for i in range(B)])}
        finally:
            eng.training = was_training

    # ------------------------------------------------------------------ checkpoints: trainable tensors only, reference key names
    def state_dict(self, *args, **kwargs):
        """the tensors that require grad in the reference (runner_base.py:572-598 keeps exactly those): t5_proj, ln_vision and the LoRA
        tensors of the model that trains — the answerer's in the video-QA variants, whose localizer LoRA is frozen (blip2_mr.py:206-209)"""
        eng = self.engine
        sd = {"t5_proj.weight": eng.proj_w.detach().clone(), "t5_proj.bias": eng.proj_b.detach().clone(),
              "ln_vision.weight": eng.lnv_w.detach().clone(), "ln_vision.bias": eng.lnv_b.detach().clone()}
        te = self.train_engine
        for a in te.adapters:
            base = te.t5_prefix + "base_model.model." + a.name
            sd[base + ".lora_A.default.weight"] = a.A.detach().clone()
            sd[base + ".lora_B.default.weight"] = a.Bt.detach().t().contiguous()
        return sd

    def load_state_dict(self, state_dict, strict=False):
        eng = self.engine
        own = self.state_dict()
        engines = [eng] + ([self.answerer] if self.answerer is not None else [])
        # (video-QA: a moment-retrieval checkpoint brings the LOCALIZER's LoRA under t5_model.*, a QA checkpoint the answerer's)
        known = set(own) | {e.t5_prefix + "base_model.model." + a.name + sfx for e in engines for a in e.adapters
                            for sfx in (".lora_A.default.weight", ".lora_B.default.weight")}
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in known]
        with torch.no_grad():
            for k, dst in (("t5_proj.weight", eng.proj_w), ("t5_proj.bias", eng.proj_b), ("ln_vision.weight", eng.lnv_w), ("ln_vision.bias", eng.lnv_b)):
                if k in state_dict:
                    dst.copy_(state_dict[k])
            for e in engines:
                for a in e.adapters:
                    base = e.t5_prefix + "base_model.model." + a.name
                    if base + ".lora_A.default.weight" in state_dict:
                        a.A.copy_(state_dict[base + ".lora_A.default.weight"])
                    if base + ".lora_B.default.weight" in state_dict:
                        a.Bt.copy_(state_dict[base + ".lora_B.default.weight"].t())
        for e in engines:
            e.refresh_trainable()
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing {missing[:5]} unexpected {unexpected[:5]}")
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def load_from_pretrained(self, url_or_filename):
        """blip2.py:86-104: non-strict load of checkpoint["model"].  The frozen tensors of that file were packed into the engine when the
        model was built (from_config reads the same path as ``pretrained``); what is (re)loaded here are the trainable ones it holds
        (t5_proj, ln_vision — and LoRA tensors if it is a fine-tuned file)."""
        msg = self.load_checkpoint(url_or_filename)
        return msg
