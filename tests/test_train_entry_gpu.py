"""End-to-end through the reference's entry points on a GPU: train.py --cfg-path ... (Config -> task -> blip2_mr -> runner_base),
generate-based validation, trainable-only checkpoints, and loss.backward() through the autograd bridge."""
import glob
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_py_synthetic_tiny(tmp_path):
    import train
    from lavis.common.registry import registry
    from mrblip.engine import MrBlipEngine
    hits0, miss0 = MrBlipEngine.vit_prefetch_hits, MrBlipEngine.vit_prefetch_misses

    cfg = os.path.join(ROOT, "mr-blip_amd/lavis/projects/mr_BLIP/train/synthetic_tiny.yaml")
    train.main(["--cfg-path", cfg, "--options", f"run.output_dir={tmp_path}/out", "run.max_epoch=2"])
    out = registry.get_path("output_dir")
    log = open(os.path.join(out, "log.txt")).read().strip().splitlines()
    assert any("train_loss" in l for l in log) and any("val_agg_metrics" in l for l in log)
    import json
    losses = [float(json.loads(l)["train_loss"]) for l in log if "train_loss" in l]
    assert len(losses) == 2 and losses[1] < losses[0], losses          # the optimizer really moves the trainable tensors
    # the train loop's one-batch look-ahead reached the engine (next clip's frozen-ViT forward overlapped with the decoder)
    from mrblip.engine import MrBlipEngine
    assert MrBlipEngine.vit_prefetch_hits > hits0 and MrBlipEngine.vit_prefetch_misses == miss0
    ck = glob.glob(os.path.join(out, "checkpoint_*.pth"))
    assert ck
    # the testing phase reloaded checkpoint_best.pth and scored the test split with it (runner_base.py:413-415, 602-620)
    assert os.path.isfile(os.path.join(out, "checkpoint_best.pth")) and os.path.isfile(os.path.join(out, "result", "val_epochbest.json"))
    sd = torch.load(ck[0], map_location="cpu")["model"]
    # trainable tensors only, reference key names (peft naming for LoRA)
    assert "t5_proj.weight" in sd and "ln_vision.bias" in sd
    assert "t5_model.base_model.model.encoder.block.0.layer.0.SelfAttention.q.lora_A.default.weight" in sd
    assert sd["t5_model.base_model.model.lm_head.lora_B.default.weight"].shape == (32128, 8)
    assert not any(k.startswith("visual_encoder") or k.startswith("Qformer") for k in sd)
    # evaluate.py's path (evaluate.py:65-119): same config + the fine-tuned checkpoint -> RunnerBase.evaluate(skip_reload=True) over the
    # test splits: metrics of the reference's evaluator, one result file per split
    best = os.path.join(out, "checkpoint_best.pth")
    logs = train.main(["--cfg-path", cfg, "--options", f"run.output_dir={tmp_path}/eval", "run.evaluate=True", "model.load_finetuned=True",
                       f"model.finetuned={best}"], evaluate=True)
    assert set(logs) == {"val"} and {"agg_metrics", "r1", "mAP", "mIoU", "invalid_predictions", "total"} <= set(logs["val"])
    assert logs["val"]["total"] == 2


def test_forward_backward_bridge_and_generate():
    import lavis  # noqa: F401
    from lavis.common.config import load_yaml
    from lavis.common.registry import registry
    from lavis.datasets import SyntheticMomentRetrievalDataset, collate

    cls = registry.get_model_class("blip2_mr")
    mcfg = load_yaml(cls.default_config_path("tiny_synthetic")).model
    mcfg.update(dict(task="qformer_freeze_lora", input_time_format="seconds_integers", interleave_data=True))
    model = cls.from_config(mcfg)
    ds = SyntheticMomentRetrievalDataset(n_items=2, n_frms=4, image_size=56, duration=60.0)
    samples = collate([ds[0], ds[1]])
    model.train()
    out = model(samples)
    assert out["loss"].dim() == 0 and out["loss"].requires_grad
    out["loss"].backward()
    g = model.trainable_decay.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0
    assert model.trainable_no_decay.grad.abs().sum() > 0
    # checkpoint round trip keeps the loss
    model.eval()
    with torch.no_grad():
        l0 = model(samples)["loss"].item()
    sd = model.state_dict()
    m2 = cls.from_config(mcfg)
    m2.load_state_dict(sd, strict=True)
    m2.eval()
    with torch.no_grad():
        assert abs(m2(samples)["loss"].item() - l0) < 1e-4
    res = model.generate(samples, num_beams=2, max_length=6)
    assert set(res) == {"duration", "prediction", "raw_prediction", "answer", "qid"} and len(res["prediction"]) == 2
    # the cross-attention K/V cache (default) and the replicate-per-beam path decode the same text
    model.generate_cross_cache = False
    res2 = model.generate(samples, num_beams=2, max_length=6)
    model.generate_cross_cache = True
    assert res2["raw_prediction"] == res["raw_prediction"]
    # ... and so does the prefix re-run without the self-attention K/V cache
    model.generate_self_cache = False
    res3 = model.generate(samples, num_beams=3, max_length=9)
    model.generate_self_cache = True
    assert res3["raw_prediction"] == model.generate(samples, num_beams=3, max_length=9)["raw_prediction"]


def test_incremental_decode_matches_prefix_rerun():
    """engine.t5_decode_step (self-attention K/V cache, shifted relative-position LUT, cache re-ordering by `parents`) against the
    full-prefix decoder forward on the same token sequences: next-token logits of every step, beams permuted between steps."""
    import lavis  # noqa: F401
    from lavis.common.registry import registry
    from lavis.common.config import load_yaml
    from lavis.datasets import SyntheticMomentRetrievalDataset, collate
    from mrblip import ops
    from util import check

    cls = registry.get_model_class("blip2_mr")
    mcfg = load_yaml(cls.default_config_path("tiny_synthetic")).model
    mcfg.update(dict(task="qformer_freeze_lora", input_time_format="seconds_integers", interleave_data=True))
    model = cls.from_config(mcfg).eval()
    eng = model.engine
    eng.training = False
    ds = SyntheticMomentRetrievalDataset(n_items=2, n_frms=4, image_size=56, duration=60.0)
    samples = collate([ds[0], ds[1]])
    video = model._frames_to_device(samples["video"])
    layout = model._layout(dict(samples, relevant_windows=[str(w) for w in samples["relevant_windows"]]))
    B, S, d, K = 2, layout.S, eng.cfg.d_model, 3
    fr = eng.frames_forward(video)[0]
    L = eng._layout_dev(layout)
    inp = eng.buf("inputs_embeds", (B * S, d), torch.float32, zero=False)
    ops.row_copy(fr, L["frame_src"], inp, L["frame_dst"])
    ops.row_copy(eng.emb, L["emb_src"], inp, L["emb_dst"])
    enc = eng.t5_encoder_forward(inp, B, S, L["mask"])
    cross = eng.t5_cross_kv(enc, B, S)
    R, steps = B * K, 7
    # Two passes.  (i) both sides on the SAME kernels (the fused one-launch decoder projection serves <= 16 rows: the 6 rows of a decode
    # step but not the 6 x (t + 1) rows of the prefix re-run — switched off here): bit-identical logits.  (ii) the product setting: the
    # step takes the fused kernel, the re-run the two-launch path — equal up to the fp32 summation order of bf16 products.
    for fused, name, tol in ((False, "generate.incremental_vs_prefix_logits", 1e-6), (True, "generate.incremental (fused projections) vs prefix re-run", 5e-3)):
        eng.dec_proj_enabled = fused
        g = torch.Generator().manual_seed(5)
        state = eng.t5_decode_begin(R, steps + 1)
        seqs = torch.zeros(R, 1, dtype=torch.long)
        parents, worst = None, 0.0
        for t in range(steps):
            inc = eng.t5_decode_step(state, seqs[:, -1], parents, cross, B, L["mask"]).float().clone()
            _, full = eng.t5_decoder_forward(seqs, torch.ones(R, t + 1, dtype=torch.int32), enc, R, S, L["mask"], labels=None, cross_cache=cross, cross_batch=B)
            ref = full.view(R, t + 1, -1)[:, -1].float()
            worst = max(worst, ((inc - ref).norm() / ref.norm()).item())
            if not fused:
                assert (inc.argmax(-1) == ref.argmax(-1)).all()
            # next step: every sequence extends a random beam of ITS clip with a random token (what the search does through `parents`)
            parents = torch.cat([b * K + torch.randint(0, K, (K,), generator=g) for b in range(B)])
            seqs = torch.cat([seqs[parents], torch.randint(2, eng.cfg.vocab, (R, 1), generator=g)], 1)
        check(name, worst, tol)   # (i) measured: bit-identical (same kernels, same operands)
    eng.dec_proj_enabled = type(eng).dec_proj_enabled


def test_bench_two_ranks_share_one_gpu():
    """bench.py's N > 1 path (process group, per-rank clips, gradient all-reduce of the flat buffer, barrier-bracketed timing, max over
    ranks, rank-0 JSON line) with two ranks on ONE GPU over gloo — the RCCL run itself needs the multi-GPU node."""
    import json
    import subprocess
    import sys
    from util import free_port

    env = dict(os.environ, MRB_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1


def test_bench_frame_shard_two_ranks_share_one_gpu():
    """bench.py --shard-frames: two ranks on ONE GPU over gloo split the frames of one Charades-shaped clip (the RCCL
    all-gather itself needs the multi-GPU node); one clip per step whatever N, "strong" scaling, loss finite at the random-init level."""
    import json
    import subprocess
    import sys
    from util import free_port

    env = dict(os.environ, MRB_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workload", "charades", "--shard-frames"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 1 and d["scaling"] == "strong" and "frame-shard2" in d["config"]["parallelism"]
    assert 5.0 < d["loss"] < 14.0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert d["collective_selftest"]["ok"] and d["collective_selftest"]["ranks"] == 2


def test_reference_named_parameters_with_a_stock_optimizer():
    """named_parameters() carries the reference's names (peft naming for LoRA) on Parameters that alias the engine's flat buffer: the
    reference's name-based weight-decay grouping (runner_base.py:111-122) + a stock torch.optim.AdamW, with the default
    zero_grad(set_to_none=True), train the same tensors to the same values as the engine's fused AdamW."""
    import lavis  # noqa: F401
    from lavis.common.config import load_yaml
    from lavis.common.registry import registry
    from lavis.datasets import SyntheticMomentRetrievalDataset, collate
    from util import check

    cls = registry.get_model_class("blip2_mr")
    mcfg = load_yaml(cls.default_config_path("tiny_synthetic")).model
    mcfg.update(dict(task="qformer_freeze_lora", input_time_format="seconds_integers", interleave_data=True, seed=11))
    a, b = cls.from_config(mcfg).eval(), cls.from_config(mcfg).eval()      # eval: no dropout, gradients still flow
    assert torch.equal(a.engine.flat, b.engine.flat)
    named = dict(a.named_parameters())
    sd = a.state_dict()
    assert set(named) == set(sd) and all(tuple(named[k].shape) == tuple(sd[k].shape) for k in sd)
    assert all(isinstance(p, torch.nn.Parameter) and p.is_leaf and p.requires_grad for p in named.values())
    assert sum(p.numel() for p in a.parameters()) == a.engine.flat.numel()
    decay, no_decay = [], []
    for n, p in a.named_parameters():
        (no_decay if (p.ndim < 2 or "bias" in n or "ln" in n or "bn" in n) else decay).append(p)
    assert sum(p.numel() for p in decay) == a.engine.n_decay           # the flat layout's decay | no-decay split is the reference's rule
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    ds = SyntheticMomentRetrievalDataset(n_items=2, n_frms=4, image_size=56, duration=60.0)
    samples = collate([ds[0], ds[1]])
    la, lb = [], []
    for _ in range(3):
        loss = a(samples)["loss"]
        loss.backward()
        opt.step()
        opt.zero_grad()                                                 # set_to_none=True: the model re-aliases and zeroes the flat gradient
        la.append(loss.item())
    video = b._frames_to_device(samples["video"])
    layout = b._layout(dict(samples, relevant_windows=[str(w) for w in samples["relevant_windows"]]))
    for _ in range(3):
        b.engine.zero_grad()
        lb.append(b.engine.forward_backward(video, layout).item())
        b.engine.optimizer_step(lr=1e-3, weight_decay=0.05)
    assert la[2] != la[0]
    check("named-params: losses of 3 steps, torch AdamW over named Parameters vs the engine's fused AdamW (rel)",
          max(abs(x - y) / abs(y) for x, y in zip(la, lb)), 4e-5)     # (two AdamW roundings + the bf16 re-pack of LoRA A/B: measured 1.5e-5)
    check("named-params: trainable buffer after 3 steps, torch AdamW vs fused AdamW",
          ((a.engine.flat - b.engine.flat).norm() / b.engine.flat.norm()).item(), 1e-4)


@pytest.mark.parametrize("name,builder", [("charades", "charades_sta"), ("anet", "anet")])
def test_train_py_charades_and_anet_configs(tmp_path, name, builder):
    """BASELINE.json configs 4 / 5 through the REAL entry point (VERDICT r2 missing 3): train.py --cfg-path projects/mr_BLIP/train/{charades,anet}.yaml
    with the annotation-JSON dataset on frame dumps (no codec / corpus in the image), tiny engine dimensions through --options; Charades additionally
    with the 32 -> 1 mean-pool of configs[3].  One epoch of train + generate-based validation + checkpoint."""
    import json

    import numpy as np
    import train
    from lavis.common.registry import registry

    rs = np.random.RandomState(7)
    for v in range(2):
        np.savez(tmp_path / f"vid{v}.npz", frames=rs.randint(0, 256, (48, 64, 64, 3), dtype=np.uint8), fps=np.float64(4.0))
    ann = [{"video": f"vid{k % 2}", "qid": k, "query": ["a person opens the door", "someone sits down on a chair and reads"][k % 2], "duration": 12.0,
            "relevant_windows": [[2 + k % 3, 6 + k % 3]]} for k in range(4)]
    for split in ("train", "val", "test"):
        json.dump(ann if split == "train" else ann[:2], open(tmp_path / f"{split}.json", "w"))
    cfg = os.path.join(ROOT, "mr-blip_amd/lavis/projects/mr_BLIP/train", name + ".yaml")
    opts = [f"datasets.{builder}.build_info.annotations.{s}.storage={tmp_path}/{s}.json" for s in ("train", "val", "test")]
    opts += [f"datasets.{builder}.build_info.videos.storage={tmp_path}", "model.model_type=tiny_synthetic",
             f"datasets.{builder}.vis_processor.train.n_frms=4", f"datasets.{builder}.vis_processor.eval.n_frms=4",
             f"datasets.{builder}.vis_processor.train.image_size=56", f"datasets.{builder}.vis_processor.eval.image_size=56",
             f"run.output_dir={tmp_path}/out", "run.max_epoch=1", "run.batch_size_train=2", "run.batch_size_eval=2", "run.num_workers=0",
             "run.accum_grad_iters=1", "run.warmup_steps=1", "run.distributed=False", "run.max_len=8", "run.num_beams=2"]
    if name == "charades":
        opts.append("model.frame_token_aggregation=mean")
    train.main(["--cfg-path", cfg, "--options"] + opts)
    out = registry.get_path("output_dir")
    log = open(os.path.join(out, "log.txt")).read().strip().splitlines()
    assert any("train_loss" in l for l in log) and any("val_agg_metrics" in l for l in log)
    loss = [float(json.loads(l)["train_loss"]) for l in log if "train_loss" in l][0]
    assert 5.0 < loss < 14.0   # ~ln(32128) for random weights
    assert os.path.isfile(os.path.join(out, "checkpoint_best.pth"))


def test_fused_loss_scale_check_is_deferred_not_blocking():
    """in fused-accumulation mode backward() must not read the upstream gradient on the host (a D2H sync behind the whole queued
    step).  The comparison runs on the device and surfaces one step late / at end_accumulation."""
    import lavis  # noqa: F401
    from lavis.common.config import load_yaml
    from lavis.common.registry import registry
    from lavis.datasets import SyntheticMomentRetrievalDataset, collate

    cls = registry.get_model_class("blip2_mr")
    mcfg = load_yaml(cls.default_config_path("tiny_synthetic")).model
    mcfg.update(dict(task="qformer_freeze_lora", input_time_format="seconds_integers", interleave_data=True))
    model = cls.from_config(mcfg)
    ds = SyntheticMomentRetrievalDataset(n_items=2, n_frms=4, image_size=56, duration=60.0)
    samples = collate([ds[0], ds[1]])
    model.train()
    model.begin_accumulation(1.0)
    model(samples)["loss"].backward()            # the announced scale: fine
    model.check_fused_scale_now()
    (model(samples)["loss"] * 0.5).backward()    # a different scale: NOT raised inside backward() ...
    with pytest.raises(RuntimeError, match="expects the loss scale"):
        model.end_accumulation()                 # ... but surfaced at the next blocking point
