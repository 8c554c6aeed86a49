"""N-rank data parallel == 1 rank on the concatenated batch, with the REAL engine.  Two ranks share the one GPU of
the test box over gloo (tests/dp_worker.py); their exchanged gradient (two async all-reduces: the LoRA segment issued from inside the
backward, the tail at its end; 1/world folded into the scale) must equal (a) the gradient of the 2-clip batch on one rank and (b) the
accumulation of the two clips as micro-steps."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from util import check, free_port, load_golden, golden_state_dict, relerr  # noqa: E402


@pytest.mark.parametrize("overlap", [0, 1])
def test_two_rank_gradient_equals_single_rank_batch(tmp_path, overlap):
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_model_gpu import _peft_sd, _samples

    out = str(tmp_path / "dp.pt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "tests", "dp_worker.py"), out, str(overlap)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    dp = torch.load(out)
    g = load_golden("mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    from dp_worker import equal_layout_clips
    s = equal_layout_clips(_samples(g))
    eng = MrBlipEngine(EngineConfig.tiny(), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"))
    eng.training = False
    lay2 = P.build_layout(tok, s, repl, 8, T=3)
    eng.zero_grad()
    l2 = eng.forward_backward(s["video"].cuda(), lay2, backward=True).item()
    g_batch = eng.grad.clone().cpu()
    eng.zero_grad()
    ls = []
    for i in range(2):
        one = {k: v[i:i + 1] for k, v in s.items()}
        lay1 = P.build_layout(tok, one, repl, 8, T=3)
        ls.append(eng.forward_backward(one["video"].cuda(), lay1, backward=True).item())
    g_acc = (eng.grad / 2).cpu()
    tag = "dp2 (overlap=%d): " % overlap
    print(tag, "rank losses", dp["losses"].tolist(), "single-rank per-clip losses", ls, "batch loss", l2)
    # (the CE kernel sums its rows with fp32 atomics: a loss of ~10.45 moves by 1-2 ulp = 1-2e-6 between two runs of the same step)
    check(tag + "rank losses vs single-rank per-clip losses", float((dp["losses"] - torch.tensor(ls)).abs().max()), 2e-7)    # (measured 0)
    check(tag + "exchanged grad vs accumulated micro-steps / 2", relerr(dp["grad"], g_acc), 1e-7)
    check(tag + "exchanged grad vs 2-clip batch on one rank", relerr(dp["grad"], g_batch), 1.5e-2)
    check(tag + "batch loss vs mean of rank losses", abs(l2 - float(dp["losses"].mean())) / abs(l2), 6e-5)
    nl = dp["n_lora"]
    assert dp["grad"][:nl].abs().sum() > 0 and dp["grad"][nl:].abs().sum() > 0


def test_eight_ranks_captured_step_with_the_overlapped_exchange(tmp_path):
    """the process layout of one 8-GPU node (here: eight ranks on the test box's one GPU over gloo) running the CAPTURED
    Charades-form step (mean-pooled frame tokens, hipGraph replay) with GradExchange armed — the graph replay followed by
    grad_ready_hook("lora") and the two asynchronous all-reduces.  The exchanged gradient of the REPLAYED step must be the mean of the eight
    clips' own gradients, its losses the clips' own losses."""
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_model_gpu import _peft_sd, _samples
    from dp_worker import equal_layout_clips, clips_n

    world = 8
    out = str(tmp_path / "dp8.pt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "tests", "dp_worker.py"), out, "graph"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    dp = torch.load(out)
    assert dp["replays"] >= 1
    g = load_golden("mr_tiny_mean")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = clips_n(equal_layout_clips(_samples(g)), world)
    eng = MrBlipEngine(EngineConfig.tiny(mean_pool=True), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"))
    eng.training = False
    eng.graph_mode = "0"
    ref = torch.zeros_like(eng.grad, device="cpu")
    ls = []
    for i in range(world):
        one = {k: v[i:i + 1] for k, v in s.items()}
        lay1 = P.build_layout(tok, one, repl, 1, T=3)
        eng.zero_grad()
        ls.append(eng.forward_backward(one["video"].cuda(), lay1, backward=True).item())
        ref += eng.grad.cpu() / world
    tag = "dp8, captured step + overlapped exchange: "
    check(tag + "rank losses (replayed graph) vs the clips' own eager losses", float((dp["losses"] - torch.tensor(ls)).abs().max()), 2e-6)
    check(tag + "exchanged gradient vs mean of the clips' own gradients", relerr(dp["grad"], ref), 1e-6)
    nl = dp["n_lora"]
    assert dp["grad"][:nl].abs().sum() > 0 and dp["grad"][nl:].abs().sum() > 0
