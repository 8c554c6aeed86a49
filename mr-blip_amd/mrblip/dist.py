"""Data-parallel gradient exchange of the MI355X engine: clips are independent, so the ONLY cross-rank step of a train
step is the average of the flat fp32 gradient buffer (~19.5 M floats = 78 MB) once per OPTIMIZER step — over RCCL/xGMI
(torch.distributed backend "nccl" is RCCL on ROCm).  The reference reaches the same result with DDP bucket hooks on every micro-step
(runner_base.py:89-96).

Overlap: the buffer is laid out [ LoRA A/B^T ... | t5_proj.weight || t5_proj.bias | ln_vision ].  The LoRA segment (92 % of the bytes) is
final as soon as the T5 encoder backward has been enqueued; the engine calls ``grad_ready_hook("lora")`` there and the segment's
all-reduce is issued asynchronously beside the t5_proj / Q-Former backward that is still to run (~6 ms of compute vs < 1 ms of xGMI
time).  The small tail segment follows at ``"all"``; ``finish()`` makes the compute stream wait for both before AdamW.  The 1/world
factor is folded into AdamW's grad_scale — no separate division pass.

Streams: the collectives are issued from a dedicated communication stream of NORMAL priority that waits on an event of the compute
stream — never from the compute stream itself, which bench.py creates with high priority.  (torch's NCCL/RCCL process group runs the
collective kernels on its own internal stream, ordered behind whatever stream is current at the call: issuing from the comm stream
keeps RCCL's channel kernels at normal priority, beside — not ahead of — the look-ahead ViT GEMMs that hold 192 CUs, and keeps the
compute stream's queue free of the collective's dependency.)

Frame sharding (first half): ``FrameShard`` splits ONE clip's frames across ranks through ViT + ln_vision + Q-Former +
t5_proj (frames are independent there: blip2_mr.py:444-445 flattens [B, T] into a batch), all-gathers the [T*n, d_model] frame tokens,
runs the replicated T5 on every rank, and reduce-scatters (sum) the frame-token gradient back to the owners.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def _initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


class GradExchange:
    """buffer: None = the engine's own flat gradient (fused accumulation), or a callable returning the flat buffer to exchange at
    finish() time (generic mode of the LAVIS model: ``model.grad_buffer``) — in that mode nothing can be sent early (autograd only
    adds the micro-step into that buffer after the engine's backward has returned), so finish() does one blocking-ordered all-reduce."""

    def __init__(self, engine, group=None, overlap: bool = True, buffer: Optional[Callable[[], torch.Tensor]] = None):
        self.eng, self.group, self.overlap, self.buffer = engine, group, overlap, buffer
        self.world = dist.get_world_size(group) if _initialized() else 1
        self._works = []
        self._armed = False
        self._lora_sent = False
        self._comm_stream = None

    def _buf(self) -> torch.Tensor:
        return self.eng.grad if self.buffer is None else self.buffer()

    def _early_ok(self) -> bool:
        """segments may be sent from inside the backward only when the exchanged buffer IS the one the engine accumulates into"""
        return self.buffer is None or self._buf().data_ptr() == self.eng.grad.data_ptr()

    def _issue(self, seg: torch.Tensor):
        if seg.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=seg.device)   # default (normal) priority
            ev = torch.cuda.Event()
            ev.record()                                # everything that wrote `seg` is ahead of this point on the compute stream
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                w = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((w, self._comm_stream))
        else:
            self._works.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))

    # ---- per optimizer step -------------------------------------------------------------------------------------------------
    def arm(self):
        """call before the LAST micro-step of an accumulation window: its backward triggers the overlapped exchange"""
        if self.world == 1:
            return
        self._armed, self._lora_sent, self._works = True, False, []
        self.eng.grad_ready_hook = self._on_ready if self._early_ok() else None

    def _on_ready(self, what: str):
        if not self._armed:
            return
        g, nl = self.eng.grad, self.eng.n_lora
        if what == "lora" and self.overlap:
            self._issue(g[:nl])
            self._lora_sent = True
        elif what == "all":
            self._issue(g[nl:] if self._lora_sent else g)
            self._armed = False
            self.eng.grad_ready_hook = None

    def finish(self) -> float:
        """the compute stream waits for the exchange; returns the factor AdamW must apply to the summed gradient (1 / world)"""
        if self.world == 1:
            return 1.0
        if self._armed:  # nothing was sent from inside the backward (generic mode, or the caller accumulated outside forward_backward)
            self._issue(self._buf())
            self._armed = False
            self.eng.grad_ready_hook = None
        for w, st in self._works:
            w.wait()                                   # the CURRENT (compute) stream waits for the collective
            if st is not None:
                torch.cuda.current_stream().wait_stream(st)
        self._works = []
        return 1.0 / self.world


def broadcast_trainable(engine, src: int = 0, group=None):
    """what the reference's DDP wrapper does at construction (torch DDP broadcasts module state from rank 0): every replica starts from
    rank 0's trainable tensors, whatever its local RNG produced"""
    if not _initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(engine.flat, src=src, group=group)
    engine.refresh_trainable()


def rccl_selftest(device, group=None, n: int = 1 << 20) -> dict:
    """Start-up check of the collective path the step depends on: all-reduce(SUM) of a flat fp32 buffer whose value is rank + 1 must
    come back as world * (world + 1) / 2 on every rank; also times it (the same call pattern as GradExchange: comm stream + event).
    Returns {"backend", "ranks", "ok", "allreduce_ms", "bytes"}; raises if the result is wrong."""
    if not _initialized():
        return {"backend": None, "ranks": 1, "ok": True}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = device if backend == "nccl" else torch.device("cpu")
    x = torch.full((n,), float(rank + 1), dtype=torch.float32, device=dev)
    dist.all_reduce(x, group=group)                    # warm-up: communicator / channel set-up
    x.fill_(float(rank + 1))
    if dev.type == "cuda":
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dist.all_reduce(x, group=group)
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b)
    else:
        import time
        t0 = time.perf_counter()
        dist.all_reduce(x, group=group)
        ms = (time.perf_counter() - t0) * 1e3
    want = world * (world + 1) / 2
    ok = bool((x == want).all())
    if not ok:
        raise RuntimeError(f"collective self-test failed on rank {rank}: all-reduce of rank+1 over {world} ranks gave {x[:4].tolist()}, expected {want}")
    return {"backend": backend, "ranks": world, "ok": ok, "allreduce_ms": round(ms, 3), "bytes": n * 4}


class FrameShard:
    """One long clip, frames split across the ranks of ``group`` (blip2_mr.py:444-445: [B, T] is just a batch through
    ViT + Q-Former).  Rank r owns frames [t0_r, t1_r) (contiguous, sizes differ by at most one).

    forward:  local frame tokens [T_r * n, d] --all-gather--> [T * n, d] on every rank (fp32, the t5_proj output)
    backward: every rank holds the SAME full gradient [T * n, d] of the replicated T5 (same clip, same prompt, same dropout seed) ->
              no reduction is needed for it: each rank simply keeps its own rows.  The LoRA / t5_proj / ln_vision gradients are a
              different matter: the T5 (LoRA) gradients are computed identically on every rank, the t5_proj / ln_vision ones only from
              the local frames — ``combine_grads`` sums the local-frame segments across ranks and leaves the replicated ones alone."""

    def __init__(self, T: int, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if _initialized() else 1
        self.rank = dist.get_rank(group) if _initialized() else 0
        base, rem = divmod(T, self.world)
        self.counts: List[int] = [base + (1 if r < rem else 0) for r in range(self.world)]
        self.starts: List[int] = [sum(self.counts[:r]) for r in range(self.world)]
        self.T = T
        self._attached = None

    def attach(self, engine, check: bool = True):
        """Make ``engine`` a member of this shard group (called by forward_backward on the first sharded step; idempotent).  The slice-
        instead-of-reduce-scatter of the frame-token gradient and the un-reduced LoRA gradients are valid ONLY if every rank runs a
        bit-identical T5, i.e. the same dropout seed at the same position of its per-step sequence: rank 0's device seed is broadcast to
        the group (train.py seeds every rank with run.seed + rank — those streams must not be used here), and later steps bump it
        identically on every rank (ops.seed_bump).  The Q-Former, whose rows are this rank's LOCAL frames, gets rank-salted call-site
        ids so that different frames do not draw the same masks on different ranks."""
        if self._attached is engine:
            return
        self._attached = engine
        if self.world == 1:
            return
        backend = dist.get_backend(self.group)
        seed = engine.seed if backend == "nccl" else engine.seed.cpu()
        dist.broadcast(seed, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        engine.seed.copy_(seed)
        engine.qf_site_salt = self.rank << 20      # call-site ids are small integers (engine.new_site): no collision with any other site
        if check:
            self.assert_same_seed(engine)

    def assert_same_seed(self, engine):
        """cheap guard (one 2-element all-reduce): every rank of the group holds the same device seed"""
        if self.world == 1:
            return
        # RCCL ("nccl") reduces DEVICE tensors only, gloo (the CPU / shared-GPU test path) host tensors: build the pair where the
        # group's backend can reach it (a CPU tensor on the production backend raised at the first sharded step)
        v = engine.seed.to(torch.int64).reshape(-1)
        if dist.get_backend(self.group) != "nccl":
            v = v.cpu()
        mm = torch.stack([v[0], -v[0]])
        dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=self.group)
        mm = mm.cpu()
        if int(mm[0]) != -int(mm[1]):
            raise RuntimeError(f"frame-sharded mode: dropout seeds differ across the shard group (max {int(mm[0])}, min {-int(mm[1])}): "
                               "the replicated T5 would draw different masks and the sliced frame-token gradient would be wrong")

    @property
    def t0(self) -> int:
        return self.starts[self.rank]

    @property
    def t1(self) -> int:
        return self.starts[self.rank] + self.counts[self.rank]

    def gather_rows(self, local: torch.Tensor, rows_per_frame: int, out: torch.Tensor) -> torch.Tensor:
        """local [T_r * rows_per_frame, d] -> out [T * rows_per_frame, d] (every rank's rows, in frame order)"""
        if self.world == 1:
            out.copy_(local)
            return out
        spans = [(self.starts[r] * rows_per_frame, (self.starts[r] + self.counts[r]) * rows_per_frame) for r in range(self.world)]
        nccl = dist.get_backend(self.group) == "nccl"
        if nccl and len(set(self.counts)) == 1:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)   # ONE RCCL all-gather over xGMI
            return out
        # ragged split (T not divisible by the group size) and/or gloo (the CPU / shared-GPU test path: no all_gather on device
        # tensors): every rank contributes max(counts) frames' worth of rows, zero padded; gloo stages through the host
        rows_max = max(self.counts) * rows_per_frame
        stage_dev = local.device if nccl else torch.device("cpu")
        mine = torch.zeros(rows_max, out.shape[1], dtype=out.dtype, device=stage_dev)
        mine[: local.shape[0]].copy_(local)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        for (a, b), part in zip(spans, parts):
            out[a:b].copy_(part[: b - a])
        return out

    def local_rows(self, full: torch.Tensor, rows_per_frame: int) -> torch.Tensor:
        return full[self.t0 * rows_per_frame: self.t1 * rows_per_frame]

    def combine_grads(self, engine):
        """after a frame-sharded backward: t5_proj.{weight,bias} and ln_vision.{weight,bias} gradients are partial sums over the LOCAL
        frames -> SUM over ranks (two small all-reduces on views of the flat buffer: [n_lora, n_decay) and the no-decay tail are adjacent,
        so it is ONE contiguous range); LoRA gradients [0, n_lora) were computed identically by every rank from the full sequence."""
        if self.world == 1:
            return
        dist.all_reduce(engine.grad[engine.n_lora:], op=dist.ReduceOp.SUM, group=self.group)
