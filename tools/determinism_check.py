"""Run-to-run determinism of the train step (training mode, dropout on): the SAME seed must give the SAME bits — loss and the flat gradient — on
every repetition.  A difference means a race between the engine's streams (or a read of stale workspace contents), which no parity tolerance
would show — or a reduction whose order follows arrival (round 4 removed the three the step had: the CE loss sum, LayerNorm's weight
gradients, the bias column sum).  tiny config: many repetitions; QVH config: a few.   usage: determinism_check.py [tiny_reps=200] [qvh_reps=12]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mr-blip_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402

dev = torch.device("cuda:0")


def run(eng, video, lay, reps, seed0, next_video=None):
    ref = None
    bad = 0
    for i in range(reps):
        eng.seed.fill_(seed0)
        eng.zero_grad()
        loss = eng.forward_backward(video, lay, backward=True, next_video=next_video)
        torch.cuda.synchronize()
        cur = (loss.detach().clone(), eng.grad.detach().clone())
        if ref is None:
            ref = cur
        elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
            bad += 1
            d = (cur[1] - ref[1]).abs()
            print(f"  rep {i}: loss {cur[0].item():.9f} vs {ref[0].item():.9f}; grad max |diff| {d.max().item():.3e} at {int(d.argmax())}, "
                  f"{int((d > 0).sum())} of {d.numel()} entries differ", flush=True)
    return bad


def tiny(reps):
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from util import load_golden, golden_state_dict
    from test_model_gpu import _peft_sd, _samples
    g = load_golden("mr_tiny")
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = {k: v[:1] for k, v in _samples(g).items()}
    eng = MrBlipEngine(EngineConfig.tiny(), StateDictSource(_peft_sd(golden_state_dict(g))), dev, seed=42)
    eng.training = True
    lay = P.build_layout(tok, s, repl, 8, T=3)
    bad = run(eng, s["video"].cuda(), lay, reps, 42)
    print(f"tiny config, training mode: {bad} of {reps - 1} repetitions differ from the first")
    return bad


def qvh(reps):
    import bench
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
    from mrblip.tokenizer import FixtureTokenizer
    wl = bench.WORKLOADS["qvh"]
    cfg = EngineConfig.flan_t5_xl_qvh(mean_pool=False)
    eng = MrBlipEngine(cfg, RandomSource(dev, seed=1234), dev, lora_init=bench.lora_init_nonzero, seed=42)
    eng.training = True
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = bench.synthetic_samples(1, wl["T"], wl["duration"], dev, 1234)
    layout = P.build_layout(tok, samples, repl, cfg.num_query, T=wl["T"])
    bad = run(eng, samples["video"], layout, reps, 42, next_video=samples["video"])
    print(f"QVH config (60 frames, XL), training mode, look-ahead on: {bad} of {reps - 1} repetitions differ from the first")
    return bad


if __name__ == "__main__":
    a = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    n = tiny(a) + (qvh(b) if b > 0 else 0)
    sys.exit(1 if n else 0)
