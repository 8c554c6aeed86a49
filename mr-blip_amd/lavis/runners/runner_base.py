"""``runner_base``: epochs, fused AdamW on the flat trainable buffer, cosine+warmup LR, one RCCL all-reduce of the flat gradient per
optimizer step, checkpoints holding only the trainable tensors (behaviour of lavis/runners/runner_base.py:47-658)."""
import json
import logging
import math
import os
import time

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, DistributedSampler

from lavis.common.dist_utils import get_rank, get_world_size, is_dist_avail_and_initialized, is_main_process, main_process
from lavis.common.registry import registry
from mrblip import ops


class FlatAdamW:
    """torch.optim.AdamW semantics (betas (0.9, 0.999), weight decay on the >= 2-D non-bias/ln tensors only; runner_base.py:102-132)
    executed by the fused HIP kernel over the model's flat parameter / gradient buffers (decay segment | no-decay segment).
    ``grad_scale`` (1/world after a SUM all-reduce, x the model's own loss scale) is applied inside the kernel."""

    def __init__(self, model, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.wd, self.betas, self.eps = model, lr, weight_decay, betas, eps
        flat = getattr(model, "train_engine", model.engine).flat
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.t = 0
        self.grad_scale = 1.0
        self.hyper = torch.zeros(4, device=flat.device)

    def set_lr(self, lr):
        self.lr = lr

    def zero_grad(self):
        self.model.zero_grad()

    @torch.no_grad()
    def step(self):
        eng = getattr(self.model, "train_engine", self.model.engine)
        # (the verdict itself is only consumed at a step's ENTRY or at a blocking check point — never between a step and its own AdamW, which
        # must stay guarded by the error word: mrblip/engine.py check_thin_role)
        self.t = max(0, self.t - eng.consume_thin_skipped())    # steps the guarded kernel dropped on the device do not count (bias correction)
        self.t += 1
        eng.note_optimizer_step()
        b1, b2 = self.betas
        scale = self.grad_scale * self.model.grad_scale()
        self.hyper.copy_(torch.tensor([self.lr, 1.0 / (1 - b1 ** self.t), 1.0 / math.sqrt(1 - b2 ** self.t), scale]).pin_memory(), non_blocking=True)
        g, nd = self.model.grad_buffer(), eng.n_decay
        # guard: the error word of the in-GEMM thin role — the update is dropped on the device if a tile's bounded wait ran out during the
        # step (mrblip/engine.py check_thin_role; the host raises one step later / at the next blocking check point)
        guard = eng.thin_guard() if eng.gemm_thin_enabled else None
        ops.adamw(eng.flat[:nd], g[:nd], self.m[:nd], self.v[:nd], self.hyper, b1, b2, self.eps, self.wd, guard=guard)
        ops.adamw(eng.flat[nd:], g[nd:], self.m[nd:], self.v[nd:], self.hyper, b1, b2, self.eps, 0.0, guard=guard)
        eng.refresh_trainable()

    def state_dict(self):
        return {"t": self.t, "m": self.m, "v": self.v, "lr": self.lr}

    def load_state_dict(self, sd):
        self.t, self.lr = sd["t"], sd["lr"]
        m, v = sd["m"], sd["v"]
        if isinstance(m, (list, tuple)):  # round-1 checkpoints kept the two segments apart
            m, v = torch.cat([x.reshape(-1) for x in m]), torch.cat([x.reshape(-1) for x in v])
        self.m.copy_(m)
        self.v.copy_(v)


@registry.register_runner("runner_base")
class RunnerBase:
    def __init__(self, cfg, task, model, datasets, job_id):
        self.config, self.task, self.model, self.datasets, self.job_id = cfg, task, model, datasets, job_id
        self.start_epoch = 0
        run = cfg.run_cfg
        self.output_dir = os.path.join(registry.get_path("repo_root") or ".", run.get("output_dir", "output"), str(job_id))
        os.makedirs(self.output_dir, exist_ok=True)
        if registry.get_path("result_dir") is None:
            registry.register_path("result_dir", os.path.join(self.output_dir, "result"))
            registry.register_path("output_dir", self.output_dir)
        self.optimizer = FlatAdamW(model, lr=float(run.init_lr), weight_decay=float(run.get("weight_decay", 0.05)))
        sched_cls = registry.get_lr_scheduler_class(run.lr_sched)
        self.lr_scheduler = sched_cls(optimizer=self.optimizer, max_epoch=run.max_epoch, min_lr=float(run.min_lr), init_lr=float(run.init_lr),
                                      decay_rate=run.get("lr_decay_rate", None), warmup_start_lr=float(run.get("warmup_lr", -1)),
                                      warmup_steps=run.get("warmup_steps", 0))
        # data parallel: fused accumulation (the gradients of a window's micro-steps add up in the engine's flat buffer) + ONE overlapped exchange
        self.exchange = None
        self._loaders = {}
        multi = is_dist_avail_and_initialized() and get_world_size() > 1
        if hasattr(model, "begin_accumulation"):
            model.begin_accumulation(1.0)
            if multi:
                from mrblip.dist import GradExchange, broadcast_trainable
                # what the reference's DDP wrapper does at construction (runner_base.py:89-96): replicas start from rank 0's trainable tensors
                broadcast_trainable(getattr(model, "train_engine", model.engine))
                # the buffer AdamW reads: the engine's own flat gradient in fused mode (segments are then sent from inside the backward),
                # model.flat_grad after end_accumulation() (one all-reduce at finish())
                self.exchange = GradExchange(getattr(model, "train_engine", model.engine), buffer=model.grad_buffer)
        elif multi:
            raise RuntimeError("runner_base: data-parallel training needs a model with a flat gradient buffer (begin_accumulation / grad_buffer, "
                               "i.e. blip2_mr on the MI355X engine); no generic per-parameter all-reduce is built")
        if run.get("resume_ckpt_path"):
            self._load_checkpoint(run.resume_ckpt_path)

    # ---- data
    def _loader(self, split, is_train):
        run = self.config.run_cfg
        if (split, is_train) in self._loaders:   # one loader (worker pool + pinned-memory thread) per split for the whole run
            return self._loaders[(split, is_train)]
        ds = None
        for name, splits in self.datasets.items():
            if split in splits:
                ds = splits[split]
        if ds is None:
            return None
        sampler = DistributedSampler(ds, shuffle=is_train, num_replicas=get_world_size(), rank=get_rank()) if is_dist_avail_and_initialized() else None
        bs = run.batch_size_train if is_train else run.batch_size_eval
        nw = run.get("num_workers", 0)
        loader = DataLoader(ds, batch_size=bs, num_workers=nw, shuffle=(sampler is None and is_train), sampler=sampler,
                            collate_fn=getattr(ds, "collater", None), drop_last=is_train, pin_memory=torch.cuda.is_available(),
                            persistent_workers=nw > 0)
        if torch.cuda.is_available() and run.get("prefetch_to_device", True):
            from lavis.datasets.dataloader_utils import PrefetchLoader
            loader = PrefetchLoader(loader, device=getattr(self.model, "device", None))   # dataloader_utils.py:46-125 (side-stream H2D)
        self._loaders[(split, is_train)] = loader
        return loader

    def _reduce_grads(self):
        """ONE flat-buffer exchange per optimizer step (the reference's DDP reduces on every micro-step, runner_base.py:89-96): finishes the
        all-reduce that the last micro-step's backward started beside its t5_proj / Q-Former backward (mrblip/dist.py) — or does it now —
        and hands 1/world to AdamW as its gradient scale."""
        if self.exchange is None:
            return
        self.optimizer.grad_scale = self.exchange.finish()

    def _arm_exchange(self):
        """called by the task before the last micro-step of an accumulation window"""
        if self.exchange is not None:
            self.exchange.arm()

    # ---- loops
    def train(self):
        run = self.config.run_cfg
        best, best_epoch = -1.0, 0   # (the reference starts at 0: a run whose validation score stays 0 saves NO checkpoint and then
        #                               crashes in its testing phase on the missing checkpoint_best.pth; here the first epoch always counts)
        epoch = self.start_epoch
        t0 = time.time()
        train_splits = run.get("train_splits", ["train"])
        for epoch in range(self.start_epoch, run.max_epoch):
            if not run.get("evaluate", False):
                loader = self._loader(train_splits[0], True)
                assert loader is not None, (f"no dataset provides the train split {train_splits[0]!r}: check datasets.<name>.build_info "
                                            "(annotation files) of the run config")
                if hasattr(loader.sampler, "set_epoch"):
                    loader.sampler.set_epoch(epoch)
                stats = self.task.train_epoch(epoch=epoch, model=self.model, data_loader=loader, optimizer=self.optimizer,
                                              lr_scheduler=self.lr_scheduler, log_freq=run.get("log_freq", 50),
                                              accum_grad_iters=run.get("accum_grad_iters", 1), reduce_grads=self._reduce_grads,
                                              arm_exchange=self._arm_exchange)
                self.log_stats(stats, "train")
                self._check_loss_scale()
            for split in run.get("valid_splits", []):
                m = self.eval_epoch(split, epoch)
                if m is not None and is_main_process() and split == "val":
                    if m["agg_metrics"] > best:
                        best, best_epoch = m["agg_metrics"], epoch
                        self._save_checkpoint(epoch, is_best=True)
                    self.log_stats({**m, "best_epoch": best_epoch}, split)
            if not run.get("valid_splits"):
                self._save_checkpoint(epoch, is_best=False)
            if run.get("evaluate", False):
                break
            if is_dist_avail_and_initialized():
                dist.barrier()
        # testing phase (runner_base.py:413-415): the test splits, on the best validation checkpoint when there was a validation
        test_epoch = "best" if run.get("valid_splits") else epoch
        self.evaluate(cur_epoch=test_epoch, skip_reload=bool(run.get("evaluate", False)))
        logging.info("Training time {:.0f}s".format(time.time() - t0))

    def evaluate(self, cur_epoch="best", skip_reload=False):
        return {s: self.eval_epoch(s, cur_epoch, skip_reload=skip_reload) for s in self.config.run_cfg.get("test_splits", [])}

    def _reload_best_model(self, model):
        """runner_base.py:602-620: load checkpoint_best.pth; a strict load that fails (only part of the model is saved) falls back to
        strict=False.  A run whose validation never beat 0 has no such file: the current weights are kept, with a warning."""
        path = os.path.join(self.output_dir, "checkpoint_best.pth")
        if not os.path.isfile(path):
            logging.warning("no %s (validation never produced a best checkpoint): evaluating the current weights", path)
            return model
        logging.info("Loading checkpoint from {}.".format(path))
        ck = torch.load(path, map_location="cpu", weights_only=True)   # tensors + plain containers only: never unpickle code from a path
        try:
            model.load_state_dict(ck["model"], strict=True)
        except RuntimeError:
            logging.warning("Key mismatch when loading checkpoint. This is expected if only part of the model is saved. "
                            "Trying to load the model with strict=False.")
            model.load_state_dict(ck["model"], strict=False)
        return model

    @torch.no_grad()
    def eval_epoch(self, split_name, cur_epoch, skip_reload=False):
        loader = self._loader(split_name, False)
        if loader is None:
            return None
        if not skip_reload and cur_epoch == "best":
            self._reload_best_model(self.model)
        self.model.eval()
        results = self.task.evaluation(self.model, loader)
        return self.task.after_evaluation(val_result=results, split_name=split_name, epoch=cur_epoch)

    # ---- checkpoints: trainable tensors only (runner_base.py:572-600)
    def _check_loss_scale(self):
        """The fused-accumulation loss-scale check of the model is deferred (its verdict is read one step late so that the host never
        drains the stream, blip2_mr.py: _check_fused_scale): a violation on the LAST micro-steps of an epoch would otherwise surface only
        in the next epoch — or never, at the end of the run.  Blocking form at every epoch end and before every checkpoint: weights that a
        wrongly scaled backward may have touched are never written out silently.  (On every rank: the error must not leave the others
        waiting in a barrier.)"""
        m = self.model.module if hasattr(self.model, "module") else self.model
        chk = getattr(m, "check_fused_scale_now", None)
        if chk is not None:
            chk()
        for eng in (getattr(m, "engine", None), getattr(m, "answerer", None)):
            if eng is not None and hasattr(eng, "check_thin_role"):
                eng.check_thin_role(block=True)   # (the same deferral, the same two blocking points: end of epoch, before a checkpoint)

    def _save_checkpoint(self, cur_epoch, is_best=False):
        self._check_loss_scale()
        self._write_checkpoint(cur_epoch, is_best)

    @main_process
    def _write_checkpoint(self, cur_epoch, is_best=False):
        obj = {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(), "config": self.config.to_dict(), "epoch": cur_epoch}
        path = os.path.join(self.output_dir, "checkpoint_{}.pth".format("best" if is_best else cur_epoch))
        logging.info("Saving checkpoint at epoch {} to {}.".format(cur_epoch, path))
        torch.save(obj, path)

    def _load_checkpoint(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        self.model.load_state_dict(ck["model"], strict=True)
        self.optimizer.load_state_dict(ck["optimizer"])
        self.start_epoch = ck["epoch"] + 1
        logging.info("Resume checkpoint from {}".format(path))

    @main_process
    def log_stats(self, stats, split_name):
        with open(os.path.join(self.output_dir, "log.txt"), "a") as f:
            f.write(json.dumps({f"{split_name}_{k}": v for k, v in stats.items()}, default=str) + "\n")
