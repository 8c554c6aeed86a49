"""the encoder length follows the query's token count (blip2_mr.py:572-824), so S changes on nearly every
step of a real run.  Workspaces are capacity-based views (engine.buf): a stream of different S must (a) give exactly the losses and
gradients of engines that only ever saw one shape, (b) stop allocating once the longest shape has been seen."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import check, load_golden, golden_state_dict, relerr  # noqa: E402


def _prompts():
    return ["Query: a dog\n", "Query: a person opens the red door and walks into the kitchen while the cat watches from the sofa\n",
            "Query: someone is cooking pasta in a large pot\n"]


@pytest.mark.parametrize("training", [False, True])
def test_stream_of_different_sequence_lengths(training):
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_model_gpu import _peft_sd, _samples

    g = load_golden("mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    base = {k: v[:1] for k, v in _samples(g).items()}
    sd = _peft_sd(golden_state_dict(g))
    dev = torch.device("cuda:0")
    layouts = []
    for q, win in zip(_prompts(), ["[[8, 16]]", "[[1, 3], [10, 22]]", "[[4, 9]]"]):
        s = dict(base)
        s["query_prompt"], s["relevant_windows"] = [q], [win]
        layouts.append(P.build_layout(tok, s, repl, 8, T=3))
    lens = [l.S for l in layouts]
    assert len(set(lens)) == 3, lens
    video = base["video"].cuda()

    def fresh():
        e = MrBlipEngine(EngineConfig.tiny(), StateDictSource(sd), dev, seed=42)
        e.training = training
        return e

    # reference: one engine per shape, each at the same position of the dropout seed stream as the mixed run below
    order = [0, 1, 2, 1, 0, 2]
    mixed = fresh()
    mixed.reserve(min(lens), max(lens), Ld=min(l.labels.shape[1] for l in layouts), Ld_max=max(l.labels.shape[1] for l in layouts))
    got = []
    allocs = []
    for step, k in enumerate(order):
        mixed.zero_grad()
        loss = mixed.forward_backward(video, layouts[k], backward=True)
        got.append((loss.item(), mixed.grad.clone()))
        allocs.append(mixed.ws_allocations)
    for step, k in enumerate(order):
        ref = fresh()
        if training:   # advance the per-step dropout seed exactly as the mixed run did
            from mrblip import ops
            for _ in range(step):
                ops.seed_bump(ref.seed)
        ref.zero_grad()
        l = ref.forward_backward(video, layouts[k], backward=True).item()
        tag = f"variable-S ({'train' if training else 'eval'}) step {step} S={lens[k]}: "
        check(tag + "loss vs fixed-shape engine", abs(got[step][0] - l) / abs(l), 2e-7)      # (measured 0: capacity-based workspaces change no bit)
        check(tag + "flat grad vs fixed-shape engine", relerr(got[step][1], ref.grad), 2e-7)  # (measured 0)
    # no (re)allocation after the first step: reserve() sized every workspace for the longest prompt
    assert allocs[0] > 0 and allocs[-1] == allocs[0], allocs


def test_growth_without_reserve_converges():
    """without reserve(): buffers grow when a longer prompt arrives, and stop growing once the longest has been seen"""
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_model_gpu import _peft_sd, _samples

    g = load_golden("mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    base = {k: v[:1] for k, v in _samples(g).items()}
    eng = MrBlipEngine(EngineConfig.tiny(), StateDictSource(_peft_sd(golden_state_dict(g))), torch.device("cuda:0"), seed=42)
    eng.training = False
    video = base["video"].cuda()
    lays = []
    for q in _prompts():
        s = dict(base)
        s["query_prompt"] = [q]
        lays.append(P.build_layout(tok, s, repl, 8, T=3))
    counts, losses = [], []
    for k in [0, 2, 1, 0, 2, 1]:
        eng.zero_grad()
        losses.append(eng.forward_backward(video, lays[k], backward=True).item())
        counts.append(eng.ws_allocations)
    assert counts[1] > counts[0] and counts[2] > counts[1] and counts[5] == counts[2], counts
    assert abs(losses[3] - losses[0]) < 1e-6 * abs(losses[0]) and abs(losses[4] - losses[1]) < 1e-6 * abs(losses[1])
