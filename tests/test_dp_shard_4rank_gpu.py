"""data parallelism and the ragged frame shard TOGETHER on more than two ranks — four ranks share the test box's GPU
over gloo (tests/dp_shard_worker.py): 2 clips x 2 frame shards (3 frames = 2 + 1).  The exchanged gradient must equal the mean of the two
clips' unsharded single-process gradients; the losses of a pair must be equal and equal the clip's own loss."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from util import check, free_port, load_golden, golden_state_dict, relerr  # noqa: E402


@pytest.mark.parametrize("world", [4, 8])
def test_clips_times_two_frame_shards_on_four_and_eight_ranks(tmp_path, world):
    """world 8 = 4 clips x 2 shards: the process layout of one 8-GPU node, here on one GPU over gloo; also records
    every rank's host time to enqueue a step while `world` processes share the box's cores (tests/util.check log)"""
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_model_gpu import _peft_sd, _samples
    from dp_worker import equal_layout_clips
    from dp_shard_worker import clips_for

    out = str(tmp_path / "dp_shard.pt")
    nclip = world // 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "tests", "dp_shard_worker.py"), out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    sh = torch.load(out)
    assert sh["counts"] == [2, 1]
    g = load_golden("mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = clips_for(equal_layout_clips(_samples(g)), nclip)
    eng = MrBlipEngine(EngineConfig.tiny(), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"), seed=42)
    eng.training = False
    ref = torch.zeros_like(eng.grad, device="cpu")
    losses = []
    for c in range(nclip):
        mine = {k: v[c:c + 1] for k, v in s.items()}
        lay = P.build_layout(tok, mine, repl, 8, T=3)
        eng.zero_grad()
        losses.append(eng.forward_backward(mine["video"].cuda(), lay, backward=True).item())
        ref += eng.grad.cpu() / nclip
    nl = sh["n_lora"]
    tag = "DP x frame shard (%d clips x 2 shards, %d ranks): " % (nclip, world)
    assert all(sh["losses"][2 * c] == sh["losses"][2 * c + 1] for c in range(nclip))         # a pair runs one replicated T5
    check(tag + "pair losses vs the clips' own losses", max(abs(sh["losses"][2 * c] - losses[c]) / abs(losses[c]) for c in range(nclip)), 2e-7)   # (measured 0)
    check(tag + "slowest rank's host enqueue of one tiny step [ms] (recorded, not a parity bound)", max(sh["host_enqueue_ms"]), 5e3)
    check(tag + "LoRA gradients vs mean of the unsharded clips", relerr(sh["grad"][:nl], ref[:nl]), 3e-7)     # (measured 2.9e-8 / 5.3e-8: the mean over clips in another order)
    check(tag + "t5_proj / ln_vision gradients vs mean of the unsharded clips", relerr(sh["grad"][nl:], ref[nl:]), 3.5e-7)   # (measured 2.9e-8 / 6.2e-8)
    assert ref[nl:].abs().sum() > 0 and ref[:nl].abs().sum() > 0
