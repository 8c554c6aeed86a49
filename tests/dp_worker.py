"""Worker of tests/test_dp_gpu.py: one data-parallel rank of the REAL (tiny) engine.  Several ranks share GPU 0 over gloo (the RCCL run
itself needs the multi-GPU node); each takes clip[rank] of the mr_tiny fixture, runs the HIP train step with the overlapped gradient
exchange armed (mrblip/dist.py) and rank 0 saves the exchanged, 1/world-scaled flat gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mr-blip_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def equal_layout_clips(s):
    """two DIFFERENT clips with the SAME prompt / timestamps / answer, so that batching them needs no padding: the reference left-pads the
    shorter video prompt of a batch with zero vectors that stay attended (blip2_mr.py:744-753), which makes a clip's gradient depend on
    its batch mates — only equal-length clips make "N ranks x 1 clip" comparable with "1 rank x N clips"."""
    out = {}
    for k, v in s.items():
        if k == "video":
            out[k] = v
        elif torch.is_tensor(v):
            out[k] = torch.cat([v[:1], v[:1]])
        else:
            out[k] = [v[0], v[0]]
    out["relevant_windows"] = ["[[8, 16]]"] * 2
    return out


def clips_n(s, n):
    """n different clips with ONE layout (see equal_layout_clips): the fixture's two, their frame-reversed copies, and the sign-flipped
    versions of those four"""
    vids = [s["video"], s["video"].flip(1), -s["video"], -s["video"].flip(1)]
    v = torch.cat(vids)[:n]
    assert v.shape[0] == n, "at most 8 clips"
    out = {"video": v}
    for k, x in s.items():
        if k == "video":
            continue
        out[k] = torch.cat([x[:1]] * n) if torch.is_tensor(x) else [x[0]] * n
    return out


def main_graph(out_path):
    """world ranks, ONE clip each, the CAPTURED step (hipGraph, engine.graph_mode = "1") with the overlapped exchange armed on every step: visit
    1 of the shape bucket runs eager, visit 2 captures, visit 3 REPLAYS — the replayed step's grad_ready_hook("lora") must fire behind the
    second graph launch and its all-reduce must see the replayed gradients.  Rank 0 saves the exchanged gradient and the
    losses of the replayed step."""
    from mrblip import prompt as P
    from mrblip.dist import GradExchange
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from util import load_golden, golden_state_dict
    from test_model_gpu import _peft_sd, _samples

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    g = load_golden("mr_tiny_mean")                 # the Charades-STA form: 32 -> 1 mean-pooled frame tokens
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = clips_n(equal_layout_clips(_samples(g)), world)
    mine = {k: v[rank:rank + 1] for k, v in s.items()}
    eng = MrBlipEngine(EngineConfig.tiny(mean_pool=True), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"), seed=42 + rank)
    eng.training = False
    eng.graph_mode = "1"
    lay = P.build_layout(tok, mine, repl, 1, T=3)
    ex = GradExchange(eng, overlap=True)
    video = mine["video"].cuda()
    fired = []
    for step in range(3):
        eng.zero_grad()
        ex.arm()
        hook = eng.grad_ready_hook
        eng.grad_ready_hook = lambda what, hook=hook, step=step: (fired.append((step, what)), hook(what))[1]
        loss = eng.forward_backward(video, lay, backward=True)
        scale = ex.finish()
        torch.cuda.synchronize()
    assert MrBlipEngine.graph_replays >= 1, "the third visit of the bucket must replay the captured graphs"
    assert fired == [(0, "lora"), (0, "all"), (1, "lora"), (1, "all"), (2, "lora"), (2, "all")], fired
    assert scale == 1.0 / world
    losses = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(losses, loss.detach().cpu().reshape(1))
    if rank == 0:
        torch.save({"grad": (eng.grad * scale).cpu(), "losses": torch.cat(losses), "n_lora": eng.n_lora, "replays": MrBlipEngine.graph_replays}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def main(out_path, overlap):
    from mrblip import prompt as P
    from mrblip.dist import GradExchange
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from util import load_golden, golden_state_dict
    from test_model_gpu import _peft_sd, _samples

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    g = load_golden("mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = _samples(g)
    s = equal_layout_clips(s)
    mine = {k: v[rank:rank + 1] for k, v in s.items()}
    eng = MrBlipEngine(EngineConfig.tiny(), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"), seed=42 + rank)
    eng.training = False
    lay = P.build_layout(tok, mine, repl, 8, T=3)
    ex = GradExchange(eng, overlap=bool(int(overlap)))
    eng.zero_grad()
    ex.arm()
    loss = eng.forward_backward(mine["video"].cuda(), lay, backward=True)
    if os.environ.get("MRB_DP_DEBUG"):
        print("DPDBG rank", rank, "overlap", overlap, "loss right after the step", loss.item(), "hook", eng.grad_ready_hook, flush=True)
    scale = ex.finish()
    torch.cuda.synchronize()
    if os.environ.get("MRB_DP_DEBUG"):
        l_fin = loss.item()
        l_again = eng.forward_backward(mine["video"].cuda(), lay, backward=False).item()
        print("DPDBG rank", rank, "loss after finish", l_fin, "fresh forward", l_again, flush=True)
    assert eng.grad_ready_hook is None and scale == 1.0 / world
    losses = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(losses, loss.detach().cpu().reshape(1))
    if rank == 0:
        torch.save({"grad": (eng.grad * scale).cpu(), "losses": torch.cat(losses), "n_lora": eng.n_lora}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if sys.argv[2] == "graph":
        main_graph(sys.argv[1])
    else:
        main(sys.argv[1], sys.argv[2])
