#!/usr/bin/env python
"""Headline benchmark: video-clips/sec of the Mr. BLIP QVHighlights train step (ViT-g/14 + Q-Former(32) + Flan-T5-XL with
LoRA, 60 frames, bf16 operands / fp32 accumulate) on N MI355X — forward, backward, gradient all-reduce and AdamW.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one optimizer step over one batch of synthetic clips resident in HBM.  Data-parallel:
clips are sharded across ranks (weak scaling, --batch-per-gpu clips each), the only exchange is ONE RCCL all-reduce of the
flat fp32 gradient buffer of the ~19.5 M trainable parameters per step.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
WORKLOADS = {  # BASELINE.json configs[1] (QVH), [3] (Charades 32->1 mean pool), [4] (ActivityNet 120 frames)
    "qvh": dict(T=60, mean_pool=False, duration=150.0, step_tflop_per_clip=45.47),
    "charades": dict(T=20, mean_pool=True, duration=30.0, step_tflop_per_clip=11.45),
    "anet": dict(T=120, mean_pool=False, duration=120.0, step_tflop_per_clip=93.73),
}
QUERY = "Query: a person opens the red door and walks into the kitchen\n"
TASK = "Given the video and the query, find the relevant windows.\nRelevant windows: "


def synthetic_samples(B, T, duration, device, seed):
    """uint8 frames -> /255 -> CLIP mean/std normalise (blip_processors.py:63-66) -> fp32 [B,T,3,224,224]."""
    g = torch.Generator(device=device).manual_seed(seed)
    u8 = torch.randint(0, 256, (B, T, 3, 224, 224), device=device, generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=device).view(1, 1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=device).view(1, 1, 3, 1, 1)
    video = (u8.float() / 255.0 - mean) / std
    ts = torch.tensor([[round((i + 0.5) * duration / T, 2) for i in range(T)]] * B, dtype=torch.float32)
    return dict(video=video.contiguous(), timestamps=ts, duration=torch.tensor([duration] * B), query_prompt=[QUERY] * B,
                task_prompt=[TASK] * B, video_prompt_end=["<extra_id_0>"] * B, relevant_windows=["[[8, 16]]"] * B)


def lora_init_nonzero(a, gen):
    """benchmark init: A and B ~ N(0, 0.02) so every LoRA gradient path is exercised."""
    a.A.copy_(torch.randn(a.A.shape, generator=gen) * 0.02)
    a.Bt.copy_(torch.randn(a.Bt.shape, generator=gen) * 0.02)


def cpu_baseline(budget_s=20.0):
    """CPU port (the oracle, oracle/mrblip_oracle.py) timed on this host on a bounded sample: full-width, full-depth
    ViT-g/14 forward of a few frames (the ViT is 69 % of the step's FLOPs and, like the rest, GEMM bound); clips/s is that
    sustained TFLOP/s divided by the 45.47 TFLOP of one QVH clip step."""
    from oracle.mrblip_oracle import Oracle

    torch.manual_seed(0)
    D, H, mlp, depth = 1408, 16, 6144, 39
    blk = {"norm1.weight": torch.ones(D), "norm1.bias": torch.zeros(D), "norm2.weight": torch.ones(D), "norm2.bias": torch.zeros(D),
           "attn.q_bias": torch.zeros(D), "attn.v_bias": torch.zeros(D), "attn.qkv.weight": torch.randn(3 * D, D) * 0.02,
           "attn.proj.weight": torch.randn(D, D) * 0.02, "attn.proj.bias": torch.zeros(D), "mlp.fc1.weight": torch.randn(mlp, D) * 0.02,
           "mlp.fc1.bias": torch.zeros(mlp), "mlp.fc2.weight": torch.randn(D, mlp) * 0.02, "mlp.fc2.bias": torch.zeros(D)}
    sd = {"visual_encoder.cls_token": torch.zeros(1, 1, D), "visual_encoder.pos_embed": torch.randn(1, 257, D) * 0.02,
          "visual_encoder.patch_embed.proj.weight": torch.randn(D, 3, 14, 14) * 0.02, "visual_encoder.patch_embed.proj.bias": torch.zeros(D)}
    for i in range(depth):  # the same tensors aliased for every block: identical arithmetic, 1/39 of the memory
        for k, v in blk.items():
            sd[f"visual_encoder.blocks.{i}.{k}"] = v
    orc = Oracle(sd, dict(vit=dict(embed_dim=D, depth=depth, num_heads=H)))
    frames = 2
    img = torch.randn(frames, 3, 224, 224)
    with torch.no_grad():
        orc.vit(img, n_blocks=2)  # warm-up
        t0 = time.time()
        n = 0
        while True:
            orc.vit(img)
            n += frames
            if time.time() - t0 > budget_s * 0.6 or n >= 8:
                break
        dt = time.time() - t0
    tflops = n * 0.52072 / dt
    return dict(value=round(tflops / 45.47, 5), unit="clips/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle ViT-g/14 fp32 forward, {n} frames of 224x224 in {dt:.1f}s = {tflops:.3f} TFLOP/s sustained; "
                       f"clips/s = TFLOP/s / 45.47 TFLOP per QVH clip step (GEMM-bound path, ViT = 69% of the FLOPs)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="qvh", choices=list(WORKLOADS))
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vit-chunk", type=int, default=0, help="frames per ViT pass (0 = engine default)")
    ap.add_argument("--no-dropout", action="store_true", help="debug only: the headline number keeps the reference's dropouts on")
    ap.add_argument("--no-lookahead", action="store_true", help="do not overlap the next clip's frozen-ViT forward with this step's decoder")
    ap.add_argument("--lookahead-blocks", type=int, default=0, help="ViT blocks run ahead beside the decoder (0 = engine default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("MRB_BENCH_SHARE_GPU"):  # test hook: several ranks on ONE GPU (gloo) to exercise the N > 1 code path on a 1-GPU box
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("MRB_BENCH_SHARE_GPU"):
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" == RCCL on ROCm

    from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
    from mrblip import ops, prompt as P
    from mrblip.tokenizer import FixtureTokenizer

    wl = WORKLOADS[args.workload]
    cfg = EngineConfig.flan_t5_xl_qvh(mean_pool=wl["mean_pool"])
    eng = MrBlipEngine(cfg, RandomSource(dev, seed=1234), dev, lora_init=lora_init_nonzero, seed=42 + rank)
    eng.training = not args.no_dropout
    if args.vit_chunk > 0:
        eng.vit_chunk = args.vit_chunk
    if args.lookahead_blocks > 0:
        eng.vit_lookahead_blocks = args.lookahead_blocks
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    B = args.batch_per_gpu
    samples = synthetic_samples(B, wl["T"], wl["duration"], dev, 1234 + rank)
    layout = P.build_layout(tok, samples, repl, 1 if wl["mean_pool"] else cfg.num_query, T=wl["T"])
    video = samples["video"]

    # HIP events around the dominant kernel's launches (ViT fc1: gemm_tile_kernel 15420x6144x1408 at T=60, B=1)
    probe_events = []

    def step(lr=3e-4, record=False):
        eng.zero_grad()
        eng.probe = probe_events if record else None
        loss = eng.forward_backward(video, layout, backward=True, next_video=None if args.no_lookahead else video)
        if world > 1:
            dist.all_reduce(eng.grad, op=dist.ReduceOp.SUM)
        eng.optimizer_step(lr=lr, weight_decay=0.05, grad_scale=1.0 / world)
        return loss

    # the step runs on a HIGH-priority stream, the look-ahead ViT on a default (low) priority one: the decoder's small kernels are
    # dispatched ahead of the thousands of GEMM workgroups they share the CUs with
    main_stream = torch.cuda.Stream(device=dev, priority=-1) if os.environ.get("MRB_BENCH_PRIO", "1") == "1" else torch.cuda.current_stream()
    main_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main_stream):
        for _ in range(args.warmup):
            loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(main_stream):
        for _ in range(args.steps):
            loss = step(record=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    loss_v = float(loss.item())
    # outside the timed region: the same kernel with the GPU to itself (one plain ViT pass on the main stream) — with the look-ahead
    # the timed launches share the CUs with the previous clip's decoder / encoder-backward kernels
    excl = []
    if rank == 0 and not args.no_lookahead:
        eng.probe = excl
        eng.vit_forward(video.reshape(-1, 3, 224, 224), slot=2)
        eng.probe = None
        torch.cuda.synchronize()

    if rank == 0:
        global_batch = B * world
        clips_s = global_batch * args.steps / elapsed
        durs = [s.elapsed_time(e) * 1e-3 for s, e in probe_events]
        F_ = min(B * wl["T"], eng.vit_chunk)
        m, n, k = F_ * 257, cfg.vit_mlp, cfg.vit_dim
        traffic = None  # HBM-side bytes per launch of the same kernel from the committed PMC pass (tools/pmc_fc1.sh), QVH B=1 shape only
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_fc1.json")
        if os.path.exists(pmc) and args.workload == "qvh" and B == 1 and F_ == 60:
            traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
        if durs:
            avg = sum(durs) / len(durs)
            ach = 2.0 * m * n * k / avg / 1e12
            roof = dict(bound="mfma", achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4),
                        traffic=traffic, kernel="gemm_w4_kernel<bf16 out, bias+GELU> (ViT fc1 %dx%dx%d)" % (m, n, k), launches=len(durs),
                        avg_us=round(avg * 1e6, 1))
            if not args.no_lookahead:
                # the timed launches are the look-ahead's: persistent blocks on (CUs - reserve) CUs, the rest is left to the other stream
                roof["cus"] = torch.cuda.get_device_properties(dev).multi_processor_count - eng.vit_lookahead_reserve
                roof["note"] = ("timed launches run on %d of %d CUs beside the clip being trained (look-ahead CU reserve); frac is against the "
                                "whole chip's peak" % (roof["cus"], torch.cuda.get_device_properties(dev).multi_processor_count))
            if excl:
                ea = sum(a.elapsed_time(b) * 1e-3 for a, b in excl) / len(excl)
                roof["exclusive"] = dict(avg_us=round(ea * 1e6, 1), achieved=round(2.0 * m * n * k / ea / 1e12, 1),
                                         frac=round(2.0 * m * n * k / ea / 1e12 / PEAK_BF16_TFLOPS, 4),
                                         note="same launches with the GPU to themselves, all CUs (untimed extra ViT pass); the timed ones run "
                                              "beside the previous clip's decoder/backward kernels (frozen-ViT look-ahead on a second stream)")
        else:
            roof = None
        out = {
            "metric": "video-clips/sec (train step) QVH 60-frame BLIP-2+T5-XL @1/2/4/8 GPU" if args.workload == "qvh" else f"video-clips/sec (train step) {args.workload}",
            "value": round(clips_s, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: ViT-g/14 + Q-Former(32) + Flan-T5-XL LoRA r=8 train step, {wl['T']} frames, S_enc={layout.S}, "
                                   f"L_dec={layout.labels.shape[1]}, random-init weights, dropout {'on' if eng.training else 'off'}",
                       "global_batch": global_batch, "batch_per_gpu": B, "frames": wl["T"], "parallelism": f"dp{world}",
                       "vit_lookahead": not args.no_lookahead},
            "step_tflop_per_clip": wl["step_tflop_per_clip"],
            "step_mfu": round(clips_s * wl["step_tflop_per_clip"] / (world * PEAK_BF16_TFLOPS), 4),
            "loss": round(loss_v, 4),
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
