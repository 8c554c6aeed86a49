"""Worker of tests/test_dp_shard_4rank_gpu.py: data parallelism ACROSS clips combined with frame sharding INSIDE a clip, four ranks on one GPU
over gloo.  World = 4 = 2 clips x 2 frame shards: ranks {0, 1} hold clip 0, ranks {2, 3} hold clip 1; inside a pair the clip's 3 frames are
split 2 + 1 (the ragged case) through ViT + ln_vision + Q-Former + t5_proj, the frame tokens are all-gathered in the pair's sub-group and the
replicated T5 runs on both ranks (mrblip/dist.py: FrameShard); the pair's t5_proj / ln_vision partial gradients are summed in the sub-group;
then ONE flat all-reduce over all four ranks with the 1/world factor (GradExchange) — every clip's gradient is present on both ranks of its
pair, so sum / 4 is exactly the mean over the two clips.  Rank 0 saves the exchanged gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mr-blip_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def clips_for(s, n):
    """n clips with ONE layout: the fixture's two equal-layout clips, then copies of them with the frames in reverse order (different
    videos, same prompt / timestamps / answer — see dp_worker.equal_layout_clips for why the layouts must agree)"""
    if n == 2:
        return s
    assert n == 4
    out = {}
    for k, v in s.items():
        if k == "video":
            out[k] = torch.cat([v, v.flip(1)])
        elif torch.is_tensor(v):
            out[k] = torch.cat([v, v])
        else:
            out[k] = list(v) + list(v)
    return out


def main(out_path):
    from mrblip import prompt as P
    from mrblip.dist import FrameShard, GradExchange
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from util import load_golden, golden_state_dict
    from test_model_gpu import _peft_sd, _samples
    from dp_worker import equal_layout_clips

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world in (4, 8)          # 2 or 4 clips x 2 frame shards (round 5: the 8-rank shape of one node)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    pairs = [dist.new_group([2 * c, 2 * c + 1]) for c in range(world // 2)]        # every rank creates every group (collective)
    clip = rank // 2
    g = load_golden("mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = clips_for(equal_layout_clips(_samples(g)), world // 2)
    mine = {k: v[clip:clip + 1] for k, v in s.items()}
    T = mine["video"].shape[1]
    eng = MrBlipEngine(EngineConfig.tiny(), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"), seed=42 + rank)
    eng.training = False
    lay = P.build_layout(tok, mine, repl, 8, T=T)
    fs = FrameShard(T, group=pairs[clip])
    local = mine["video"][:, fs.t0: fs.t1].cuda().contiguous()
    ex = GradExchange(eng, overlap=False)
    eng.zero_grad()
    loss = eng.forward_backward(local, lay, backward=True, shard=fs)   # (first step: workspaces, shard attach)
    torch.cuda.synchronize()
    dist.barrier()
    import time
    eng.zero_grad()
    t_h = time.perf_counter()
    loss = eng.forward_backward(local, lay, backward=True, shard=fs)   # eval mode: the same loss and gradient again
    host_ms = (time.perf_counter() - t_h) * 1e3                        # host time to ENQUEUE a step with `world` processes sharing the cores
    fs.combine_grads(eng)                     # t5_proj / ln_vision: partial sums over the pair's local frames
    ex.arm()
    scale = ex.finish()                       # one flat all-reduce over the 4 ranks
    torch.cuda.synchronize()
    out = (eng.grad * scale).cpu()
    losses = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(losses, loss.detach().float().cpu().reshape(1))
    hosts = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(hosts, torch.tensor([host_ms]))
    if rank == 0:
        torch.save({"grad": out, "losses": [float(x) for x in losses], "counts": fs.counts, "n_lora": eng.n_lora,
                    "host_enqueue_ms": [float(x) for x in hosts]}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
