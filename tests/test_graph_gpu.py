"""Round 5 (VERDICT r4 missing 5): the captured T5 part of the train step.  Both halves of the T5 (interleave + encoder forward | decoder forward
+ loss, decoder backward, encoder backward) are captured as hipGraphs per shape bucket and replayed; the step's integer inputs (index maps,
masks, decoder ids, labels) live in per-bucket static device buffers that every step refills, so a graph serves every later step of its
bucket whatever the text says.  A replayed step runs the SAME kernels with the same arguments as the eager step: losses and the whole
flat gradient must be bit-identical, with dropout on, with and without the look-ahead, and across different contents of one bucket."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(graph_mode):
    import bench
    from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
    dev = torch.device("cuda:0")
    cfg = EngineConfig(vit_dim=320, vit_depth=2, vit_heads=5, vit_mlp=512, qf_dim=256, qf_heads=4, qf_inter=512, qf_layers=4, num_query=32,
                       d_model=256, d_kv=64, t5_heads=4, d_ff=512, t5_layers=3, t5_dec_layers=3)
    eng = MrBlipEngine(cfg, RandomSource(dev, seed=77), dev, lora_init=bench.lora_init_nonzero, seed=11)
    eng.graph_mode = graph_mode
    eng.training = True
    return eng, cfg, dev


def _layouts(cfg, dev, T):
    import bench
    from mrblip import prompt as P
    from mrblip.tokenizer import FixtureTokenizer
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    out = []
    for seed, words, win in ((5, "a person opens the red door", "[[8, 16]]"), (6, "a child builds the wooden tower", "[[30, 44]]"),
                             (7, "a person opens the red door and walks into the kitchen", "[[2, 10], [40, 52]]")):
        s = bench.synthetic_samples(1, T, 150.0, dev, seed)
        s["query_prompt"] = ["Query: " + words + "\n"]
        s["relevant_windows"] = [win]
        out.append((s["video"], P.build_layout(tok, s, repl, cfg.num_query, T=T)))
    return out


@pytest.mark.parametrize("T,lookahead", [(2, False), (2, True), (40, True)])
def test_captured_step_is_bit_identical_to_the_eager_step(T, lookahead):
    """T = 2: a 70-row encoder (few-row kernels, the Charades-STA regime); T = 40: S > 1024 rows — the in-GEMM thin role, whose flag words a
    replay must find cleared, the tall-input thin kernels and the per-layer batched weight-gradient launches are inside the graphs."""
    from mrblip.engine import MrBlipEngine
    runs = {}
    for mode in ("0", "1"):
        eng, cfg, dev = _engine(mode)
        lays = _layouts(cfg, dev, T)
        # per bucket: first visit eager (workspaces), second captures, later ones replay — unless a longer bucket made a workspace grow in
        # between (its store moved: the graph is captured again)
        order = [0, 0, 0, 1, 1, 1, 2, 2, 2, 0, 1, 2, 0, 0]
        losses, grads = [], []
        r0 = MrBlipEngine.graph_replays
        for step, li in enumerate(order):
            video, lay = lays[li]
            nxt = lays[order[(step + 1) % len(order)]][0] if lookahead else None
            eng.zero_grad()
            losses.append(eng.forward_backward(video, lay, backward=True, next_video=nxt).item())
            grads.append(eng.grad.detach().clone())
            eng.optimizer_step(1e-3)
        torch.cuda.synchronize()
        eng.check_thin_role(block=True)
        runs[mode] = (losses, grads, MrBlipEngine.graph_replays - r0, eng)
    (l0, g0, _, _), (l1, g1, replays, eng1) = runs["0"], runs["1"]
    assert replays >= 4, replays                          # visits 3.. of a bucket are replays
    assert l0 == l1, (l0, l1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert torch.equal(runs["0"][3].flat, eng1.flat)      # ... and so are the trained parameters after all optimizer steps
    keys = list(eng1._static_sets)
    assert 1 <= len(keys) <= 3


def test_auto_mode_captures_short_encoders_only():
    eng, cfg, dev = _engine("auto")
    assert eng._graph_wanted(1, 72, True, False) and not eng._graph_wanted(1, 2012, True, False)
    assert not eng._graph_wanted(1, 72, False, False) and not eng._graph_wanted(1, 72, True, True)
