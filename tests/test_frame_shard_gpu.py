"""SURVEY.md §8(f4), first half: ONE clip's frames split across ranks through ViT + ln_vision + Q-Former + t5_proj (blip2_mr.py:444-445:
[B, T] is just a batch there), one all-gather of the [T * n, d_model] frame tokens, replicated T5.  Two ranks share the test box's GPU
over gloo (tests/shard_worker.py); the sharded step must equal the unsharded step of the same clip: loss, LoRA gradients (computed
identically on every rank) and the t5_proj / ln_vision gradients (summed over the ranks' local frames)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from util import check, free_port, load_golden, golden_state_dict, relerr  # noqa: E402


@pytest.mark.parametrize("mean", [0, 1])
def test_frame_sharded_step_equals_unsharded(tmp_path, mean):
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_model_gpu import _peft_sd, _samples

    out = str(tmp_path / "shard.pt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "tests", "shard_worker.py"), out, str(mean)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    sh = torch.load(out)
    assert sh["counts"] == [2, 1]   # 3 frames over 2 ranks: the ragged split
    g = load_golden("mr_tiny_mean" if mean else "mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = {k: v[:1] for k, v in _samples(g).items()}
    eng = MrBlipEngine(EngineConfig.tiny(mean_pool=bool(mean)), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"), seed=42)
    eng.training = False
    lay = P.build_layout(tok, s, repl, 1 if mean else 8, T=3)
    eng.zero_grad()
    loss = eng.forward_backward(s["video"].cuda(), lay, backward=True).item()
    ref = eng.grad.cpu()
    nl = sh["n_lora"]
    tag = "frame-shard x2 (mean_pool=%d): " % mean
    check(tag + "loss vs unsharded", abs(sh["loss"] - loss) / abs(loss), 2e-7)    # (measured 0: the replicated T5 sees the same bits)
    check(tag + "LoRA grads vs unsharded", relerr(sh["grad"][:nl], ref[:nl]), 2e-7)    # (measured 0)
    check(tag + "LoRA grads rank 1 vs rank 0 (replicated T5)", relerr(sh["grad_other"][:nl], sh["grad"][:nl]), 1e-7)
    check(tag + "t5_proj / ln_vision grads (summed over ranks) vs unsharded", relerr(sh["grad"][nl:], ref[nl:]), 2e-7)    # (measured 1.6e-8 / 3.4e-8: one more fp32 add per element)
    check(tag + "combined tail identical on both ranks", relerr(sh["grad_other"][nl:], sh["grad"][nl:]), 1e-7)
    assert ref[nl:].abs().sum() > 0


def test_frame_sharded_training_step_keeps_the_replicated_t5_identical(tmp_path):
    """the slice that replaces the reduce-scatter of the frame-token gradient, and the un-reduced LoRA gradients, are
    valid only if every rank runs a bit-identical T5 — same dropout seed, same bump position.  The two ranks are built with DIFFERENT
    seeds (run.seed + rank, as train.py seeds them) and run two TRAINING steps: FrameShard.attach must have put them on rank 0's stream
    (equal device seeds after two bumps), the replicated T5's loss and every LoRA gradient must agree bit for bit across the ranks, and the
    Q-Former call sites must be rank-salted (local frames start at row 0 on every rank: unsalted, different frames would share masks)."""
    out = str(tmp_path / "shard_train.pt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "tests", "shard_worker.py"), out, "0", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    sh = torch.load(out)
    nl = sh["n_lora"]
    assert sh["seeds"][0] == sh["seeds"][1], sh["seeds"]
    assert sh["losses"][0] == sh["losses"][1], sh["losses"]
    assert torch.equal(sh["grad"][:nl], sh["grad_other"][:nl])           # replicated T5: identical LoRA gradients, no reduction needed
    assert torch.equal(sh["grad"][nl:], sh["grad_other"][nl:])           # t5_proj / ln_vision: summed over the ranks' local frames
    assert sh["grad"][:nl].abs().sum() > 0 and sh["grad"][nl:].abs().sum() > 0 and torch.isfinite(sh["grad"]).all()
    assert sh["qf_salt"] == 0   # rank 0's salt; rank r uses r << 20 (mrblip/dist.py: FrameShard.attach)


def test_seed_guard_reduces_a_device_tensor_on_the_rccl_backend(monkeypatch):
    """FrameShard.assert_same_seed all-reduced a CPU tensor whatever the backend; RCCL ("nccl") has no CPU backend, so the
    first frame-sharded step on real multi-GPU raised.  With the backend reported as "nccl" the reduced pair must live on the seed's
    device (and on the host for gloo); a seed mismatch must still raise."""
    import types
    import torch.distributed as dist
    from mrblip.dist import FrameShard

    seen = []

    def fake_all_reduce(t, op=None, group=None):
        seen.append(t.device.type)

    fs = FrameShard.__new__(FrameShard)
    fs.group, fs.world, fs.rank = None, 2, 0
    eng = types.SimpleNamespace(seed=torch.tensor([1234567], dtype=torch.int64, device="cuda:0"))
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    fs.assert_same_seed(eng)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "gloo")
    fs.assert_same_seed(eng)
    assert seen == ["cuda", "cpu"], seen

    def mismatch(t, op=None, group=None):
        t[1] += 5       # as if another rank held a smaller seed: max(-seed) grows

    monkeypatch.setattr(dist, "all_reduce", mismatch)
    with pytest.raises(RuntimeError, match="dropout seeds differ"):
        fs.assert_same_seed(eng)
