"""Worker of tests/test_frame_shard_gpu.py: one rank of the frame-sharded long-video mode (mrblip/dist.py: FrameShard)
on the REAL tiny engine.  Ranks share GPU 0 over gloo; each takes ITS frames of clip 0 of the mr_tiny fixture (3 frames over 2 ranks =
2 + 1: the ragged case), runs ViT / ln_vision / Q-Former / t5_proj on them, all-gathers the frame tokens, runs the replicated T5 and
continues the backward with its rows of the frame-token gradient.  Rank 0 saves loss and the combined flat gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mr-blip_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out_path, mean_pool, training="0"):
    from mrblip import prompt as P
    from mrblip.dist import FrameShard
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from util import load_golden, golden_state_dict
    from test_model_gpu import _peft_sd, _samples

    rank = int(os.environ["RANK"])
    mean = bool(int(mean_pool))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    g = load_golden("mr_tiny_mean" if mean else "mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = {k: v[:1] for k, v in _samples(g).items()}
    T = s["video"].shape[1]
    train = bool(int(training))
    # training mode: the ranks are DELIBERATELY built with different dropout seeds (what train.py's run.seed + rank gives every engine);
    # FrameShard.attach must put the group on rank 0's stream, or the replicated T5 draws different masks per rank
    eng = MrBlipEngine(EngineConfig.tiny(mean_pool=mean), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"),
                       seed=42 + (rank if train else 0))
    eng.training = train
    lay = P.build_layout(tok, s, repl, 1 if mean else 8, T=T)
    fs = FrameShard(T)
    local = s["video"][:, fs.t0: fs.t1].cuda().contiguous()
    eng.zero_grad()
    loss = eng.forward_backward(local, lay, backward=True, shard=fs)
    fs.combine_grads(eng)
    torch.cuda.synchronize()
    if train:   # a second step: the per-step seed bump must keep the ranks together
        eng.zero_grad()
        loss = eng.forward_backward(local, lay, backward=True, shard=fs)
        fs.combine_grads(eng)
        torch.cuda.synchronize()
    grads = [torch.zeros_like(eng.grad, device="cpu") for _ in range(fs.world)]
    dist.all_gather(grads, eng.grad.cpu())
    losses = [torch.zeros(1) for _ in range(fs.world)]
    dist.all_gather(losses, loss.detach().float().cpu().reshape(1))
    seeds = [torch.zeros(1, dtype=torch.int32) for _ in range(fs.world)]
    dist.all_gather(seeds, eng.seed.cpu())
    if rank == 0:
        torch.save({"grad": grads[0], "grad_other": grads[1], "loss": loss.item(), "n_lora": eng.n_lora, "counts": fs.counts,
                    "losses": [float(x) for x in losses], "seeds": [int(x) for x in seeds], "qf_salt": eng.qf_site_salt}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(*sys.argv[1:4])
