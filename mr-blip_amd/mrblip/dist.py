"""Data-parallel gradient exchange of the MI355X engine: clips are independent, so the ONLY cross-rank step of a train
step is the average of the flat fp32 gradient buffer (~19.5 M floats = 78 MB) once per OPTIMIZER step — over RCCL/xGMI
(torch.distributed backend "nccl" is RCCL on ROCm).  The reference reaches the same result with DDP bucket hooks on every micro-step
(runner_base.py:89-96).

Overlap: the buffer is laid out [ LoRA A/B^T ... | t5_proj.weight || t5_proj.bias | ln_vision ].  The LoRA segment (92 % of the bytes) is
final as soon as the T5 encoder backward has been enqueued; the engine calls ``grad_ready_hook("lora")`` there and the segment's
all-reduce is issued asynchronously (RCCL runs it on its own stream after an event on the compute stream) beside the t5_proj / Q-Former
backward that is still to run (~6 ms of compute vs < 1 ms of xGMI time).  The small tail segment follows at ``"all"``; ``finish()`` makes the
compute stream wait for both before AdamW.  The 1/world factor is folded into AdamW's grad_scale — no separate division pass.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class GradExchange:
    def __init__(self, engine, group=None, overlap: bool = True):
        self.eng, self.group, self.overlap = engine, group, overlap
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self._works = []
        self._armed = False
        self._lora_sent = False

    # ---- per optimizer step -------------------------------------------------------------------------------------------------
    def arm(self):
        """call before the LAST micro-step of an accumulation window: its backward triggers the overlapped exchange"""
        if self.world == 1:
            return
        self._armed, self._lora_sent, self._works = True, False, []
        self.eng.grad_ready_hook = self._on_ready

    def _on_ready(self, what: str):
        if not self._armed:
            return
        g, nl = self.eng.grad, self.eng.n_lora
        if what == "lora" and self.overlap:
            self._works.append(dist.all_reduce(g[:nl], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._lora_sent = True
        elif what == "all":
            seg = g[nl:] if self._lora_sent else g
            self._works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._armed = False
            self.eng.grad_ready_hook = None

    def finish(self) -> float:
        """the compute stream waits for the exchange; returns the factor AdamW must apply to the summed gradient (1 / world)"""
        if self.world == 1:
            return 1.0
        if self._armed:  # the hook never fired (e.g. the caller accumulated outside forward_backward): exchange everything now
            self._works.append(dist.all_reduce(self.eng.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._armed = False
            self.eng.grad_ready_hook = None
        for w in self._works:
            w.wait()
        self._works = []
        return 1.0 / self.world
