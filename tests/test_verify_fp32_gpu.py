"""The north star's "fp logits within 1e-3 relative" against the reference's fp32 CPU run, shown on the engine's OWN forward path fed
with fp32-accurate (split-bf16) operands — tests/verify_fp32.py.  If the bf16 gap of the product path (measured ~1e-2, tests/
test_model_gpu.py / test_fullsize_gpu.py) came from a logic error, it would still be there with 16-bit-mantissa operands; it is not."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import check, load_golden, golden_state_dict, relerr  # noqa: E402


def _run(eng, video, lay):
    from verify_fp32 import Fp32Verify

    with Fp32Verify(eng):
        loss = eng.forward_backward(video, lay, backward=False).item()
        logits = eng.ws["d_logits"].clone()
        emb = eng.ws["inputs_embeds"].clone()
        xv = eng.ws["vit_x"].clone()
        qf = eng._qf_last_f32.clone()
    return loss, logits, emb, xv, qf


@pytest.mark.parametrize("tag,mean", [("mr_tiny", False), ("mr_tiny_mean", True)])
def test_fp32_operand_mode_tiny(tag, mean):
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_model_gpu import _samples

    g = load_golden(tag)
    sd = golden_state_dict(g)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = _samples(g)
    eng = MrBlipEngine(EngineConfig.tiny(mean_pool=mean), StateDictSource(sd), torch.device("cuda:0"))   # LoRA: peft init, B = 0
    eng.training = False
    eng._verify_src = StateDictSource(sd)
    lay = P.build_layout(tok, samples, repl, 1 if mean else 8, T=3)
    l_bf16 = eng.forward_backward(samples["video"].cuda(), lay, backward=False).item()
    lg_bf16 = eng.ws["d_logits"].clone()
    loss, logits, emb, xv, qf = _run(eng, samples["video"].cuda(), lay)
    lg = logits.cpu().view(g["logits_sub"].shape[0], g["logits_sub"].shape[1], -1)
    check(tag + ".verify-fp32: logits vs reference-fp32", relerr(lg[..., ::64], g["logits_sub"]), 5e-5)
    check(tag + ".verify-fp32: logits_lse vs reference-fp32", relerr(torch.logsumexp(lg, -1), g["logits_lse"]), 5e-7)
    check(tag + ".verify-fp32: loss vs reference-fp32 (rel)", abs(loss - float(g["loss"])) / abs(float(g["loss"])), 2e-6)
    check(tag + ".verify-fp32: inputs_embeds vs reference-fp32", relerr(emb.cpu().view(g["inputs_embs"].shape), g["inputs_embs"]), 2.5e-5)
    # and the product (bf16-operand) path on the same engine, for the record: same code, only the operand precision differs
    check(tag + ".product-bf16: logits vs reference-fp32", relerr(lg_bf16.cpu().view(lg.shape)[..., ::64], g["logits_sub"]), 1.4e-2)
    # leaving the mode restores the product path bit for bit
    l_again = eng.forward_backward(samples["video"].cuda(), lay, backward=False).item()
    assert abs(l_again - l_bf16) <= 1e-6 * abs(l_bf16)   # (atomic loss reduction: last-bit order effects)


def test_fp32_operand_mode_c1_real_depth():
    """39 ViT blocks + 12 Q-Former layers + 12 + 12 T5 layers at real width (BASELINE configs[0]) against the reference's own fp32 run"""
    from weights import seeded_state_dict
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_fullsize_gpu import _c1_samples

    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = _c1_samples(g)
    cfg = EngineConfig(d_model=768, d_kv=64, t5_heads=12, d_ff=2048, t5_layers=12, t5_dec_layers=12)
    eng = MrBlipEngine(cfg, StateDictSource(sd), torch.device("cuda:0"))
    eng.training = False
    eng._verify_src = StateDictSource(sd)
    lay = P.build_layout(tok, samples, repl, 32, T=4)
    loss, logits, emb, xv, qf = _run(eng, samples["video"].cuda(), lay)
    lg = logits.cpu().view(1, -1, 32128)
    check("c1.verify-fp32: vit.out (39 blocks) vs reference-fp32", relerr(xv.view(4, 257, 1408)[:, ::8, ::4].cpu(), g["vit_sub"]), 3e-5)
    check("c1.verify-fp32: qformer.out vs reference-fp32", relerr(qf.view(4, 32, 768)[:, :, ::2].cpu(), g["qf_out"]), 2.5e-5)
    check("c1.verify-fp32: inputs_embeds vs reference-fp32", relerr(emb.view(1, lay.S, 768)[..., ::4].cpu(), g["inputs_embs_sub"]), 2.5e-5)
    check("c1.verify-fp32: logits vs reference-fp32", relerr(lg[..., ::64], g["logits_sub"]), 5e-5)
    check("c1.verify-fp32: loss vs reference-fp32 (rel)", abs(loss - float(g["loss"])) / abs(float(g["loss"])), 2e-6)


def test_fp32_operand_mode_c2_benched_size():
    """VERDICT r3 missing 6 / next 3(b): the same verification AT THE BENCHED SIZE (BASELINE.json configs[1]: 60 frames, Flan-T5-XL dims, 24 + 24
    layers, S_enc = 2012) against the reference's own fp32 run of that size (tests/golden/mr_c2.npz): the 9e-3 logits gap of the product
    path at C2 is operand rounding, not logic."""
    from test_fullsize_gpu import _c2_setup

    eng, src, lay, video, g, T = _c2_setup()
    eng._verify_src = src
    loss, logits, emb, xv, qf = _run(eng, video.cuda(), lay)
    lg = logits.cpu().view(1, -1, 32128)
    S, d = lay.S, 2048
    check("c2.verify-fp32: vit.out (60 frames) vs reference-fp32", relerr(xv.view(T, 257, 1408)[::6, ::16, ::16].cpu(), g["vit_sub"]), 3e-5)
    check("c2.verify-fp32: qformer.out vs reference-fp32", relerr(qf.view(T, 32, 768)[::6, ::4, ::8].cpu(), g["qf_sub"]), 3e-5)
    check("c2.verify-fp32: inputs_embeds vs reference-fp32", relerr(emb.view(1, S, d)[:, ::4, ::16].cpu(), g["inputs_embs_sub"]), 3e-5)
    check("c2.verify-fp32: logits vs reference-fp32", relerr(lg[..., ::64], g["logits_sub"]), 1e-4)
    check("c2.verify-fp32: loss vs reference-fp32 (rel)", abs(loss - float(g["loss"])) / abs(float(g["loss"])), 5e-6)
    del eng
    torch.cuda.empty_cache()


def test_finite_differences_of_the_fp32_operand_forward_confirm_the_product_gradient_c1():
    """VERDICT r4 missing 1, without the oracle: the gradient the PRODUCT path (bf16 operands) hands to AdamW, projected on a direction v,
    against the central finite difference (L(theta + eps v) - L(theta - eps v)) / (2 eps) of the engine's own forward run with fp32-accurate
    (split-bf16) operands — a loss known to 2e-6 of the reference's fp32 run (test above).  Directions live in the non-LoRA trainables
    (t5_proj weight / bias, ln_vision gamma / beta: their gradients come out of the WHOLE backward chain — decoder, encoder, Q-Former, every
    dX kernel; the verification mode pins LoRA B = 0, so LoRA tensors cannot be moved).  v = the product gradient's own direction, as a
    whole and restricted to each of the four tensors: the projection is then that gradient's norm, and a missing term, a wrong scale or a
    biased rounding shows as a ratio != 1 (the element-wise 2-3e-2 of an UNBIASED rounding error moves a norm by its square, ~5e-4).  A random
    direction says nothing new: its projection is |g| / sqrt(n) and carries the element-wise error itself (measured 4.6e-2 on one draw).
    Real depth (C1: 39 + 12 + 12 + 12 layers), eval mode (no dropout)."""
    from weights import seeded_state_dict
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from test_fullsize_gpu import _c1_samples

    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = _c1_samples(g)
    cfg = EngineConfig(d_model=768, d_kv=64, t5_heads=12, d_ff=2048, t5_layers=12, t5_dec_layers=12)
    eng = MrBlipEngine(cfg, StateDictSource(sd), torch.device("cuda:0"))
    eng.training = False
    eng._verify_src = StateDictSource(sd)
    lay = P.build_layout(tok, samples, repl, 32, T=4)
    video = samples["video"].cuda()
    eng.zero_grad()
    eng.forward_backward(video, lay, backward=True)
    torch.cuda.synchronize()
    n0 = eng.n_lora
    grad = eng.grad[n0:].double().clone()
    theta = eng.flat[n0:].clone()
    sizes = [("t5_proj.weight", eng.proj_w.numel()), ("t5_proj.bias", eng.proj_b.numel()), ("ln_vision.weight", eng.lnv_w.numel()), ("ln_vision.bias", eng.lnv_b.numel())]
    assert sum(n for _, n in sizes) == grad.numel()

    def loss_at(delta):
        eng.flat[n0:].copy_((theta.double() + delta).float())
        eng.refresh_trainable()
        return _run(eng, video, lay)[0]

    off = 0
    spans = [("all four tensors", 0, grad.numel())]
    for name, n in sizes:
        spans.append((name, off, off + n))
        off += n
    l0 = loss_at(torch.zeros_like(grad))
    for name, a, b in spans:
        v = torch.zeros_like(grad)
        v[a:b] = grad[a:b]
        nrm = float(v.norm())
        v /= nrm
        # a step that moves the loss by ~0.03 along the gradient (central difference: the second-order term cancels; the loss is known to
        # ~2e-6 relative and its error mostly cancels in the difference), and half of it to show that the quotient has converged
        eps = 0.03 / nrm
        fd = [(loss_at(e * v) - loss_at(-e * v)) / (2 * e) for e in (eps, eps / 2)]
        rel = abs(nrm - fd[1]) / abs(fd[1])
        print(f"finite differences, own direction in {name}: |g_product| = {nrm:.6e}, fd(eps) = {fd[0]:.6e}, fd(eps/2) = {fd[1]:.6e}, rel {rel:.2e} (loss {l0:.5f})")
        # measured (profiles/r05_finite_difference_c1.txt): all 3.7e-4, t5_proj.weight 4.4e-4, .bias 4.0e-4, ln_vision.weight 2.6e-3, .bias 1.6e-3
        # (its gradient also crosses the Q-Former backward); eps vs eps / 2: 0.5-3.5e-4
        check(f"c1.finite-difference (fp32-operand forward) vs product gradient norm, {name} (rel)", rel, 6e-3 if name.startswith("ln_vision") else 1.5e-3)
        check(f"c1.finite-difference convergence eps vs eps/2, {name} (rel)", abs(fd[0] - fd[1]) / abs(fd[1]), 1.5e-3)
    eng.flat[n0:].copy_(theta)
    eng.refresh_trainable()
    del eng
    torch.cuda.empty_cache()


def test_finite_differences_confirm_the_lora_gradients_c1():
    """The same finite-difference check for what the optimizer mostly updates: the LoRA tensors (92 % of the trainable floats), with every
    adapter of the real-depth C1 model in a NON-ZERO state (N(0, 0.02)).  The verification mode takes the LoRA branch in fp32 from the
    master tensors (tests/verify_fp32.py), so A and B can be moved: its loss is first pinned to the oracle's fp32 run of the same state,
    then the product path's dA / dB, projected on their own direction — all LoRA floats, all A, all B, the encoder's, the decoder's
    adapters — must equal the central difference of that loss.  (peft itself is absent: the LoRA semantics are the oracle's restatement
    of its published algorithm, blip2_mr.py:182-200; this test shows that the HIP backward differentiates the HIP forward, and that
    forward equals the restatement.)"""
    from weights import seeded_state_dict
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from oracle import mrblip_oracle as O
    from test_fullsize_gpu import _c1_samples, C1_CFG
    from test_model_gpu import _peft_sd

    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    sdl = _peft_sd(sd, lora_std=0.02)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = _c1_samples(g)
    cfg = EngineConfig(d_model=768, d_kv=64, t5_heads=12, d_ff=2048, t5_layers=12, t5_dec_layers=12)
    eng = MrBlipEngine(cfg, StateDictSource(sdl), torch.device("cuda:0"))
    eng.training = False
    eng._verify_src = StateDictSource(sdl)
    lay = P.build_layout(tok, samples, repl, 32, T=4)
    video = samples["video"].cuda()
    with torch.no_grad():
        ref_loss = float(O.Oracle(sdl, C1_CFG, emu_bf16=False, lora=dict(r=8, alpha=8)).forward_mr(tok, samples, repl)["loss"])
    l0 = _run(eng, video, lay)[0]
    check("c1.lora!=0.verify-fp32: loss vs oracle-fp32 (rel)", abs(l0 - ref_loss) / abs(ref_loss), 1e-5)
    eng.zero_grad()
    l_prod = eng.forward_backward(video, lay, backward=True).item()
    torch.cuda.synchronize()
    n0 = eng.n_lora
    grad = eng.grad[:n0].double().clone()
    theta = eng.flat[:n0].clone()
    is_a = torch.zeros(n0, dtype=torch.bool, device=grad.device)
    is_enc = torch.zeros(n0, dtype=torch.bool, device=grad.device)
    for a in eng.adapters:
        is_a[a.a_off: a.a_off + a.A.numel()] = True
        if a.name.startswith("encoder."):
            is_enc[a.a_off: a.a_off + a.A.numel()] = True
            is_enc[a.bt_off: a.bt_off + a.Bt.numel()] = True
    assert a.A.data_ptr() == eng.flat.data_ptr() + 4 * a.a_off
    spans = [("all LoRA tensors", torch.ones_like(is_a)), ("every lora_A", is_a), ("every lora_B", ~is_a), ("encoder adapters", is_enc), ("decoder + lm_head adapters", ~is_enc)]

    def loss_at(delta):
        eng.flat[:n0].copy_((theta.double() + delta).float())
        eng.refresh_trainable()
        return _run(eng, video, lay)[0]

    for name, mask in spans:
        v = torch.where(mask, grad, torch.zeros_like(grad))
        nrm = float(v.norm())
        v /= nrm
        eps = 0.03 / nrm
        fd = [(loss_at(e * v) - loss_at(-e * v)) / (2 * e) for e in (eps, eps / 2)]
        rel = abs(nrm - fd[1]) / abs(fd[1])
        print(f"finite differences, own direction in {name}: |g_product| = {nrm:.6e}, fd(eps) = {fd[0]:.6e}, fd(eps/2) = {fd[1]:.6e}, rel {rel:.2e} (loss {l0:.5f}, product {l_prod:.5f})")
        # measured (profiles/r05_finite_difference_c1.txt): all 5.0e-4, lora_A 5.3e-4, lora_B 5.0e-4, encoder 3.7e-4, decoder + lm_head 1.3e-3;
        # eps vs eps / 2: 0.6-2.2e-4
        check(f"c1.lora!=0.finite-difference (fp32-operand forward) vs product gradient norm, {name} (rel)", rel, 3e-3 if name.startswith("decoder") else 1.5e-3)
        check(f"c1.lora!=0.finite-difference convergence eps vs eps/2, {name} (rel)", abs(fd[0] - fd[1]) / abs(fd[1]), 1e-3)
    eng.flat[:n0].copy_(theta)
    eng.refresh_trainable()
    del eng
    torch.cuda.empty_cache()


def test_finite_differences_confirm_the_gradients_at_the_benched_size_c2():
    """The finite-difference check AT THE BENCHED SIZE (C2: 60 frames, Flan-T5-XL width, 24 + 24 layers, S_enc = 2012, every one of the 433
    adapters non-zero: the state of tests/golden/mr_c2_lora.npz) — the 4-wave kernel's K-split input gradients, the stacked cross K / V
    backward, the thin launches above 512 rows, which C1 does not reach.
    ALWAYS ON since round 6 in its short form: ONE direction (all LoRA tensors, 92 % of the trainable floats), ONE step size — two fp32-operand
    forwards, about a minute — so that the driver's GPUTEST carries a benched-size gradient check that does not go through the oracle.  The
    long form (MRB_FD_C2=1, ~5 minutes: the loss pin of the non-zero-LoRA state, eps and eps / 2, the t5_proj + ln_vision direction;
    `MRB_FD_C2=1 python -m pytest tests/test_verify_fp32_gpu.py -k benched_size_c2 -s`; measured: profiles/r05_finite_difference_c2.txt)."""
    import os
    full = os.environ.get("MRB_FD_C2", "0") == "1"
    from util import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "mr_c2_lora.npz")):
        pytest.skip("tests/golden/mr_c2_lora.npz not generated")
    from weights import seeded_array
    from test_fullsize_gpu import _c2_setup
    gl = load_golden("mr_c2_lora")
    std = float(gl["strings"]["lora_std"])

    def lora_init(a, gen):
        base = "t5_model.base_model.model." + a.name
        a.A.copy_(torch.from_numpy(seeded_array(base + ".lora_A.default.weight", (8, a.in_dim), std=std)))
        a.Bt.copy_(torch.from_numpy(seeded_array(base + ".lora_B.default.weight", (a.out, 8), std=std)).t())

    eng, src, lay, video, g, T = _c2_setup(lora_init=lora_init)
    eng._verify_src = src
    video = video.cuda()
    l0 = float("nan")
    if full:
        l0 = _run(eng, video, lay)[0]
        ref_loss = float(gl["loss"])
        print(f"finite differences c2: fp32-operand forward loss {l0:.6f}, oracle-fp32 loss of the same state {ref_loss:.6f}, rel {abs(l0 - ref_loss) / abs(ref_loss):.2e}")
        check("c2.lora!=0.verify-fp32: loss vs oracle-fp32 (rel)", abs(l0 - ref_loss) / abs(ref_loss), 1e-5)
    eng.zero_grad()
    l_prod = eng.forward_backward(video, lay, backward=True).item()
    torch.cuda.synchronize()
    assert eng._enc_bwd_w4_ok(lay.S) and "eb_dxn_p_wi" in eng.ws           # the K-split parts were what ran
    n0 = eng.n_lora
    grad = eng.grad.double().clone()
    theta = eng.flat.clone()
    idx = torch.arange(grad.numel(), device=grad.device)

    def loss_at(delta):
        eng.flat.copy_((theta.double() + delta).float())
        eng.refresh_trainable()
        return _run(eng, video, lay)[0]

    for name, mask, steps in ((("all LoRA tensors", idx < n0, 2), ("t5_proj + ln_vision", idx >= n0, 1)) if full else (("all LoRA tensors", idx < n0, 1),)):
        v = torch.where(mask, grad, torch.zeros_like(grad))
        nrm = float(v.norm())
        v /= nrm
        eps = 0.03 / nrm
        fd = [(loss_at(e * v) - loss_at(-e * v)) / (2 * e) for e in (eps, eps / 2)[:steps]]
        rel = abs(nrm - fd[-1]) / abs(fd[-1])
        print(f"finite differences c2, own direction in {name}: |g_product| = {nrm:.6e}, fd = {[f'{x:.6e}' for x in fd]}, rel {rel:.2e} (loss {l0:.5f}, product {l_prod:.5f})")
        check(f"c2.lora!=0.finite-difference (fp32-operand forward) vs product gradient norm, {name} (rel)", rel, 5e-3)     # measured 2.0e-3 / 1.2e-3
    eng.flat.copy_(theta)
    eng.refresh_trainable()
    del eng
    torch.cuda.empty_cache()
