"""The drop-in surface (registry / Config / LR schedulers / post-processing / data-parallel gradient exchange) without a GPU."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_has_the_reference_entry_points():
    import lavis  # noqa: F401
    from lavis.common.registry import registry
    from lavis.models.base_model import BaseModel

    cls = registry.get_model_class("blip2_mr")
    assert cls is not None and issubclass(cls, BaseModel)
    assert "pretrain_flant5xl" in cls.PRETRAINED_MODEL_CONFIG_DICT
    assert os.path.isfile(cls.default_config_path("pretrain_flant5xl"))
    assert registry.get_task_class("moment_retrieval") is not None
    assert registry.get_runner_class("runner_base") is not None
    assert registry.get_lr_scheduler_class("linear_warmup_cosine_lr") is not None
    with pytest.raises(KeyError):
        registry.register_task("moment_retrieval")(object)
    for m in ("from_config", "forward", "generate", "load_checkpoint", "load_from_pretrained"):
        assert hasattr(cls, m)


def test_config_merge_and_options():
    import lavis  # noqa: F401
    from lavis.common.config import Config

    args = types.SimpleNamespace(cfg_path=os.path.join(ROOT, "mr-blip_amd/lavis/projects/mr_BLIP/train/qvh.yaml"),
                                 options=["run.init_lr=1e-4", "model.frame_token_aggregation=mean", "datasets.qvh.vis_processor.train.n_frms=20"])
    cfg = Config(args)
    assert cfg.run_cfg.init_lr == 1e-4 and cfg.run_cfg.accum_grad_iters == 8 and cfg.run_cfg.task == "moment_retrieval"
    assert cfg.model_cfg.arch == "blip2_mr" and cfg.model_cfg.frame_token_aggregation == "mean" and cfg.model_cfg.interleave_data is True
    assert cfg.model_cfg.num_query_token == 32 and cfg.model_cfg.t5_model == "google/flan-t5-xl"      # from the arch default yaml
    assert cfg.datasets_cfg.qvh.vis_processor.train.n_frms == 20 and cfg.datasets_cfg.qvh.vis_processor.eval.n_frms == 60
    assert cfg.datasets_cfg.qvh.build_info.annotations.train.storage.endswith("train.json")          # builder default yaml
    # "k v k v" form of --options
    cfg2 = Config(types.SimpleNamespace(cfg_path=args.cfg_path, options=["run.seed", "7"]))
    assert cfg2.run_cfg.seed == 7


def test_lr_scheduler_matches_reference_golden():
    import lavis  # noqa: F401
    from lavis.common.registry import registry

    g = load_golden("lr_sched")

    class Opt:
        lr = None

        def set_lr(self, lr):
            self.lr = lr

    opt = Opt()
    s = registry.get_lr_scheduler_class("linear_warmup_cosine_lr")(opt, max_epoch=5, min_lr=0.0, init_lr=3e-4, warmup_steps=30, warmup_start_lr=1e-8)
    lrs = []
    for ep in range(5):
        for it in range(20):
            s.step(cur_epoch=ep, cur_step=ep * 20 + it)
            lrs.append(opt.lr)
    assert np.allclose(lrs, g["lrs"], rtol=1e-12, atol=0)


def test_post_process_matches_reference_golden():
    from lavis.models.blip2_mr_models.utils import moment_str_to_list, post_process

    g = load_golden("post_process")["cases"]
    for c, p, m in zip(g["cases"], g["post"], g["moments"]):
        assert post_process(c) == p, c
        assert moment_str_to_list(p) == m, c


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
    from lavis.common.dist_utils import get_rank, get_world_size, init_distributed_mode
    from lavis.common.config import Node
    from lavis.runners.runner_base import RunnerBase
    from lavis.datasets import SyntheticMomentRetrievalDataset
    from torch.utils.data import DistributedSampler

    run = Node({"dist_url": "env://"})
    init_distributed_mode(run)
    assert run.distributed and get_world_size() == world and get_rank() == rank and dist.get_backend() == "gloo"
    # the data-parallel exchange (mrblip/dist.py) as the runner drives it: armed before the window's last micro-step, the LoRA segment's
    # all-reduce issued from inside the backward ("lora" hook), the tail at its end ("all"), finish() -> 1/world for AdamW's grad_scale
    from mrblip.dist import GradExchange
    eng = types.SimpleNamespace(grad=torch.cat([torch.full((10,), float(rank + 1)), torch.arange(3.0) * (rank + 1)]), n_lora=10, grad_ready_hook=None)
    runner = RunnerBase.__new__(RunnerBase)
    runner.exchange = GradExchange(eng)
    runner.optimizer = types.SimpleNamespace(grad_scale=1.0)
    runner._arm_exchange()
    assert eng.grad_ready_hook is not None
    eng.grad_ready_hook("lora")
    eng.grad_ready_hook("all")
    runner._reduce_grads()
    assert eng.grad_ready_hook is None and runner.optimizer.grad_scale == 1.0 / world
    avg = eng.grad * runner.optimizer.grad_scale
    model = types.SimpleNamespace(trainable_decay=types.SimpleNamespace(grad=avg[:10]), trainable_no_decay=types.SimpleNamespace(grad=avg[10:]))
    # a caller whose backward never reaches the hooks (accumulated elsewhere): finish() exchanges the whole buffer
    eng2 = types.SimpleNamespace(grad=torch.full((13,), float(rank + 1)), n_lora=10, grad_ready_hook=None)
    ex2 = GradExchange(eng2, overlap=False)
    ex2.arm()
    assert ex2.finish() == 0.5 and eng2.grad.tolist() == [3.0] * 13 and eng2.grad_ready_hook is None
    # generic mode (the exchanged buffer is NOT the engine's own: model.grad_buffer() = flat_grad): nothing may be sent early, finish()
    # reduces the buffer the optimizer will read
    other = torch.full((13,), float(10 * (rank + 1)))
    eng3 = types.SimpleNamespace(grad=torch.zeros(13), n_lora=10, grad_ready_hook=None)
    ex3 = GradExchange(eng3, buffer=lambda: other)
    ex3.arm()
    assert eng3.grad_ready_hook is None
    assert ex3.finish() == 0.5 and other.tolist() == [30.0] * 13 and eng3.grad.abs().sum() == 0
    # start-up self-test of the collective path + DDP-style broadcast of the trainable tensors from rank 0
    from mrblip.dist import FrameShard, broadcast_trainable, rccl_selftest
    st = rccl_selftest(torch.device("cpu"), n=1024)
    assert st["ok"] and st["ranks"] == world and st["backend"] == "gloo"
    eng4 = types.SimpleNamespace(flat=torch.full((5,), float(rank + 7)), refresh_trainable=lambda: None)
    broadcast_trainable(eng4)
    assert eng4.flat.tolist() == [7.0] * 5
    # frame sharding: 5 frames over 2 ranks = 3 + 2, 4 rows per frame; the gather restores frame order on every rank
    fs = FrameShard(5)
    assert fs.counts == [3, 2] and fs.starts == [0, 3] and (fs.t0, fs.t1) == ((0, 3) if rank == 0 else (3, 5))
    full = torch.arange(5 * 4 * 2, dtype=torch.float32).view(20, 2)
    got = fs.gather_rows(fs.local_rows(full, 4).clone(), 4, torch.zeros(20, 2))
    assert torch.equal(got, full)
    eng5 = types.SimpleNamespace(grad=torch.cat([torch.full((10,), 5.0), torch.full((3,), float(rank + 1))]), n_lora=10)
    fs.combine_grads(eng5)
    assert eng5.grad.tolist() == [5.0] * 10 + [3.0] * 3     # replicated LoRA segment untouched, local-frame segment summed
    # clips are sharded over ranks with no overlap (DistributedSampler, seed + rank)
    ds = SyntheticMomentRetrievalDataset(n_items=8, n_frms=2, image_size=14)
    idx = list(DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False))
    q.put((rank, model.trainable_decay.grad.tolist(), model.trainable_no_decay.grad.tolist(), idx))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_exchange_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from util import free_port
    port = free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gd, gn, idx in out:
        assert gd == [1.5] * 10 and gn == [0.0, 1.5, 3.0]      # mean of rank grads (1, 2)
    assert sorted(out[0][3] + out[1][3]) == list(range(8)) and not set(out[0][3]) & set(out[1][3])
