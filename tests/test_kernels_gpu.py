"""Per-kernel parity tests of libmrblip_hip.so, called through the C ABI (ctypes, mrblip.ops) on a real MI355X,
against plain PyTorch fp32 references of the same op evaluated on the same bf16-rounded operands, plus the
oracle's restatement of the dropout hash.  Tolerances are stated per test.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mrblip import ops as _ops

    return _ops


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(x):
    return x.bfloat16()


def keep_mask(shape, seed, site, p):
    from oracle.mrblip_oracle import dropout_keep

    return dropout_keep(shape, seed, site, p).to(dev())


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (64, 520, 192), (513, 1408, 576), (33, 264, 2304), (1, 8, 64), (257, 8, 128)])
def test_gemm_plain(ops, cfg, M, N, K):
    torch.manual_seed(0)
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.1)
    ref = a.float() @ w.float().t()
    for dt in (torch.float32, torch.bfloat16):
        out = torch.full((M, N), float("nan"), dtype=dt, device=dev())
        ops.gemm(a, w, out, tile_cfg=cfg)
        tol = 2e-6 if dt == torch.float32 else 3e-3  # fp32 accumulate; bf16 output rounding
        assert rel(out.float(), ref) < tol, (cfg, dt)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12])
def test_gemm_epilogues(ops, cfg):
    torch.manual_seed(1)
    M, N, K = 200, 328, 256
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.1)
    ae = bf(torch.randn(M, 64, device=dev()))
    we = bf(torch.randn(N, 64, device=dev()) * 0.1)
    bias = torch.randn(N, device=dev())
    res = torch.randn(M, N, device=dev())
    base = a.float() @ w.float().t() + ae.float() @ we.float().t() + bias
    # bias + K-extension + gelu + pre-activation copy
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(a, w, out, aext=ae, wext=we, bias=bias, act=1, out2=pre, tile_cfg=cfg)
    assert rel(pre.float(), base) < 3e-3
    assert rel(out.float(), torch.nn.functional.gelu(base)) < 3e-3
    # residual + dropout, fp32 out (in place on the residual stream)
    seed = torch.tensor([1234567], dtype=torch.int32, device=dev())
    x = res.clone()
    ops.gemm(a, w, x, aext=ae, wext=we, bias=bias, residual=x, drop=ops.Dropout(seed, 17, 0.1), tile_cfg=cfg)
    mask = keep_mask((M, N), 1234567, 17, 0.1)
    assert 0.85 < mask.mean().item() < 0.95
    assert rel(x, res + base * mask / 0.9) < 2e-6


def test_gemm_four_wave_tile_epilogues(ops):
    """cfg 13 (256x256, four waves of 128x128, hand-pipelined K loop) takes the plain epilogues of the frozen-ViT GEMMs only."""
    torch.manual_seed(5)
    M, N, K = 700, 520, 320  # 3 x 3 tiles with ragged edges, 5 K-tiles (prologue + steady state + both tail forms)
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.1)
    bias = torch.randn(N, device=dev())
    res = torch.randn(M, N, device=dev())
    base = a.float() @ w.float().t() + bias
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    for cfg in (13, 14):
        ops.gemm(a, w, out, bias=bias, act=1, tile_cfg=cfg)
        assert rel(out.float(), torch.nn.functional.gelu(base)) < 3e-3
        x = res.clone()
        ops.gemm(a, w, x, bias=bias, residual=x, tile_cfg=cfg)
        assert rel(x, res + base) < 2e-6
        for k in (64, 128):  # one and two K-tiles
            ops.gemm(a[:, :k].contiguous(), w[:, :k].contiguous(), x, tile_cfg=cfg)
            assert rel(x, a[:, :k].float() @ w[:, :k].float().t()) < 2e-6
        with pytest.raises(ops.MrblipError):
            ops.gemm(a, w, out, gated=True, tile_cfg=cfg)


@pytest.mark.parametrize("M,N,K", [(700, 520, 320), (1000, 768, 64), (515, 304, 128), (2100, 1408, 1408), (300, 264, 192)])
def test_gemm_four_wave_tile_with_three_w_stages_is_bit_identical(ops, M, N, K):
    """cfg 17 (round 4): the 4-wave 256x256 kernel with a THIRD stage buffer for the W operand (its pieces go out a K-tile earlier, the
    hand-over waits with vmcnt(8)) — the same MFMA order as cfg 13, so the same bits, for every epilogue of the frozen ViT, with ragged
    edges and 1 / 2 / 3 / 5 / 22 K-tiles (every prologue / tail form of the pipelined loop), and for a persistent block walking several tiles."""
    torch.manual_seed(21)
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.1)
    bias = torch.randn(N, device=dev())
    res = torch.randn(M, N, device=dev())
    for kw in (dict(bias=bias), dict(bias=bias, act=1), dict(bias=bias, f32=True), dict(bias=bias, f32=True, res=True)):
        outs = []
        for cfg in (13, 17):
            out = res.clone() if kw.get("res") else torch.full((M, N), 3.0, dtype=torch.float32 if kw.get("f32") else torch.bfloat16, device=dev())
            with ops.gemm_cu_reserve(248 if M > 2000 else 0):       # 8 blocks: every block walks several tiles
                ops.gemm(a, w, out, bias=kw.get("bias"), act=kw.get("act", 0), residual=out if kw.get("res") else None, tile_cfg=cfg)
            outs.append(out)
        assert torch.equal(outs[0], outs[1]), kw
    assert rel(outs[1], res + a.float() @ w.float().t() + bias) < 2e-6


def test_vit_gemms_full_size_against_fp32_reference(ops):
    """The frozen-ViT GEMMs at the QVH shapes (60 frames x 257 tokens = 15420 rows), as the library dispatches them itself (the
    persistent four-wave kernel), against fp32 torch: fc1 (bias + exact-erf GELU, bf16 out), fc2 (bias + fp32 residual, in place),
    qkv (bias, bf16 out); and the same bits when the launch leaves 64 CUs to another stream (the look-ahead's setting)."""
    torch.manual_seed(11)
    M, D, F = 15420, 1408, 6144
    h = bf(torch.randn(M, D, device=dev()))
    w1 = bf(torch.randn(F, D, device=dev()) * 0.03)
    b1 = torch.randn(F, device=dev()) * 0.1
    f = torch.empty(M, F, dtype=torch.bfloat16, device=dev())
    ops.gemm(h, w1, f, bias=b1, act=1)
    ref = torch.nn.functional.gelu(h.float() @ w1.float().t() + b1)
    assert rel(f.float(), ref) < 3e-3
    with ops.gemm_cu_reserve(64):
        f2 = torch.empty_like(f)
        ops.gemm(h, w1, f2, bias=b1, act=1)
    assert torch.equal(f2, f)
    del ref, f2
    w2 = bf(torch.randn(D, F, device=dev()) * 0.02)
    b2 = torch.randn(D, device=dev()) * 0.1
    x = torch.randn(M, D, device=dev())
    want = x + f.float() @ w2.float().t() + b2
    ops.gemm(f, w2, x, bias=b2, residual=x)
    assert rel(x, want) < 2e-6
    del want
    wq = bf(torch.randn(3 * D, D, device=dev()) * 0.03)
    bq = torch.randn(3 * D, device=dev()) * 0.1
    qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev())
    ops.gemm(h, wq, qkv, bias=bq)
    assert rel(qkv.float(), h.float() @ wq.float().t() + bq) < 3e-3


@pytest.mark.parametrize("cfg", [13, 14])
def test_gemm_cu_reserve_keeps_results(ops, cfg):
    """mrblip_gemm_set_cu_reserve only changes how many persistent blocks walk the tiles: same bits with 0, 64 and 248 CUs reserved
    (8 blocks walk all 48 tiles) and the previous value comes back."""
    torch.manual_seed(6)
    M, N, K = 1500, 1800, 256
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.1)
    bias = torch.randn(N, device=dev())
    ref = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(a, w, ref, bias=bias, act=1, tile_cfg=cfg)
    assert rel(ref.float(), torch.nn.functional.gelu(a.float() @ w.float().t() + bias)) < 3e-3
    # the ragged fp32-residual form too (ViT proj / fc2 epilogue): M = 1500 is 5.9 row tiles, N = 1800 7.03 / 9.4 column tiles
    res = torch.randn(M, N, device=dev())
    ref32 = torch.empty(M, N, device=dev())
    ops.gemm(a, w, ref32, bias=bias, residual=res, tile_cfg=cfg)
    assert rel(ref32, a.float() @ w.float().t() + bias + res) < 3e-3
    for r in (8, 64, 72, 136, 248, 250):
        with ops.gemm_cu_reserve(r):
            for _ in range(2):
                out = torch.full_like(ref, float("nan"))
                ops.gemm(a, w, out, bias=bias, act=1, tile_cfg=cfg)
                assert torch.equal(out, ref), (cfg, r)
            out32 = torch.full_like(ref32, float("nan"))
            ops.gemm(a, w, out32, bias=bias, residual=res, tile_cfg=cfg)
            assert torch.equal(out32, ref32), (cfg, r)
    with ops.gemm_cu_reserve(32) as g:
        assert g.prev == 0
    out = torch.full_like(ref, float("nan"))
    ops.gemm(a, w, out, bias=bias, act=1, tile_cfg=cfg)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("cfg", [1, 2, 4, 7, 8, 9])
def test_gemm_gated(ops, cfg):
    torch.manual_seed(2)
    M, Nh, K = 300, 200, 128
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(2 * Nh, K, device=dev()) * 0.1)
    y = torch.empty(M, Nh, dtype=torch.bfloat16, device=dev())
    h = torch.empty(M, 2 * Nh, dtype=torch.bfloat16, device=dev())
    seed = torch.tensor([99], dtype=torch.int32, device=dev())
    ops.gemm(a, w, y, out2=h, gated=True, drop=ops.Dropout(seed, 3, 0.1), tile_cfg=cfg)
    hr = a.float() @ w.float().t()
    assert rel(h.float(), hr) < 3e-3
    mask = keep_mask((M, Nh), 99, 3, 0.1)
    ref = torch.nn.functional.gelu(hr[:, :Nh]) * hr[:, Nh:] * mask / 0.9
    assert rel(y.float(), ref) < 4e-3
    # backward kernel against autograd on the saved bf16 pre-activations
    dy = bf(torch.randn(M, Nh, device=dev()))
    dh = torch.empty(M, 2 * Nh, dtype=torch.bfloat16, device=dev())
    ops.gated_gelu_bwd(dy, h, dh, drop=ops.Dropout(seed, 3, 0.1))
    hh = h.float().requires_grad_(True)
    (torch.nn.functional.gelu(hh[:, :Nh]) * hh[:, Nh:] * mask / 0.9 * dy.float()).sum().backward()
    assert rel(dh.float(), hh.grad) < 4e-3


@pytest.mark.parametrize("D", [1408, 768, 2048, 96])
def test_norms(ops, D):
    torch.manual_seed(3)
    M = 77
    x = torch.randn(M, D, device=dev()) * 2 + 0.5
    g = torch.randn(D, device=dev()) * 0.1 + 1
    b = torch.randn(D, device=dev()) * 0.1
    dy = torch.randn(M, D, device=dev())
    add = torch.randn(M, D, device=dev())
    # LayerNorm
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    ref.backward(dy)
    of = torch.empty_like(x)
    ob = torch.empty(M, D, dtype=torch.bfloat16, device=dev())
    ops.layernorm_fwd(x, g, b, 1e-6, out_bf16=ob, out_f32=of)
    assert rel(of, ref) < 2e-6 and rel(ob.float(), ref) < 3e-3
    dx = torch.empty_like(x)
    dg, db = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    ops.layernorm_bwd(dy, x, g, 1e-6, dx, dx_add=add, dgamma=dg, dbeta=db)
    assert rel(dx, xr.grad + add) < 1e-5
    assert rel(dg, gr.grad) < 1e-5 and rel(db, br.grad) < 1e-5
    # T5 RMSNorm
    xr = x.clone().requires_grad_(True)
    ref = g * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    ref.backward(dy)
    ops.rmsnorm_fwd(x, g, 1e-6, out_bf16=ob, out_f32=of)
    assert rel(of, ref) < 2e-6 and rel(ob.float(), ref) < 3e-3
    ops.rmsnorm_bwd(dy, x, g, 1e-6, dx, dx_add=add)
    assert rel(dx, xr.grad + add) < 1e-5


def test_weight_gradient_reductions_are_ordered(ops):
    """Round 4: the two reductions of the step that used fp32 atomics — LayerNorm's dgamma / dbeta (ln_vision, 15420 rows) and the
    column sum behind the t5_proj bias gradient — add their blocks' partial sums in BLOCK order (last-arriver ticket): correct against
    torch, accumulating (+=) semantics kept, and the same bits on every launch while another stream keeps the chip busy."""
    torch.manual_seed(17)
    M, D = 15420, 1408
    x, dy = torch.randn(M, D, device=dev()), torch.randn(M, D, device=dev())
    g = torch.randn(D, device=dev()) * 0.1 + 1
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), torch.zeros(D, device=dev(), requires_grad=True)
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy)
    a = bf(torch.randn(4096, 4096, device=dev()))
    c = torch.empty(4096, 4096, device=dev())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(10):
            ops.gemm(a, a, c)
    outs = []
    for rep in range(12):
        dx = torch.empty_like(x)
        dg, db = torch.full((D,), 0.5, device=dev()), torch.full((D,), -0.25, device=dev())
        ops.layernorm_bwd(dy, x, g, 1e-6, dx, dgamma=dg, dbeta=db)
        outs.append((dg.clone(), db.clone()))
    side.synchronize()
    assert rel(outs[0][0] - 0.5, gr.grad) < 2e-5 and rel(outs[0][1] + 0.25, br.grad) < 2e-5
    for dg, db in outs[1:]:
        assert torch.equal(dg, outs[0][0]) and torch.equal(db, outs[0][1])
    for Mc, Nc in ((1920, 2048), (77, 300), (9000, 768)):
        y = torch.randn(Mc, Nc, device=dev())
        sums = []
        for rep in range(8):
            out = torch.full((Nc,), 2.0, device=dev())
            ops.colsum(y, out)
            sums.append(out.clone())
        assert rel(sums[0] - 2.0, y.sum(0)) < 1e-5
        assert all(torch.equal(s_, sums[0]) for s_ in sums[1:])


def _attn_ref(q, k, v, scale, bias=None, mask=None, drop_mask=None, p=0.0):
    """q,k,v: [B,H,S,D] fp32 (already bf16-rounded values).  fp32 softmax, probabilities bf16-rounded before PV."""
    s = q @ k.transpose(-1, -2) * scale
    if bias is not None:
        s = s + bias
    if mask is not None:
        s = s.masked_fill(~mask, -1e30)
    pr = torch.softmax(s, -1)
    if drop_mask is not None:
        pr = pr * drop_mask / (1 - p)
    return pr @ v, pr


def _lut_bias(lut, Sq, Sk):
    rel_ = (torch.arange(Sk, device=dev())[None, :] - torch.arange(Sq, device=dev())[:, None]).clamp(-128, 128) + 128
    return lut[:, rel_]  # [H,Sq,Sk]


@pytest.mark.parametrize("B,H,Sq,Sk,D", [(2, 3, 257, 257, 88), (1, 2, 32, 257, 64), (2, 4, 300, 300, 64), (1, 2, 12, 333, 16), (2, 2, 17, 17, 24),
                                         (1, 1, 1, 1, 64), (1, 1, 33, 65, 64), (1, 2, 65, 1, 64), (3, 1, 64, 64, 88)])
def test_attention_fwd(ops, B, H, Sq, Sk, D):
    torch.manual_seed(4)
    scale = D ** -0.5
    q = bf(torch.randn(B, Sq, H, D, device=dev()))
    k = bf(torch.randn(B, Sk, H, D, device=dev()))
    v = bf(torch.randn(B, Sk, H, D, device=dev()))
    vt = ops.head_transpose(v)
    assert torch.equal(vt[:, :, :D, :Sk], v.permute(0, 2, 3, 1)) and vt[:, :, D:].abs().sum() == 0 and vt[..., Sk:].abs().sum() == 0
    o = torch.empty(B, Sq, H, D, dtype=torch.bfloat16, device=dev())
    lse = torch.zeros(B, H, ops.rup32(Sq), device=dev())
    ops.attention_fwd(q, k, vt, o, lse, scale=scale)
    ref, pr = _attn_ref(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3), scale)
    assert rel(o.float().permute(0, 2, 1, 3), ref) < 6e-3  # P and O are bf16-rounded
    s = q.float().permute(0, 2, 1, 3) @ k.float().permute(0, 2, 1, 3).transpose(-1, -2) * scale
    assert (lse[..., :Sq] - torch.logsumexp(s, -1)).abs().max() < 1e-3


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("B,H,Sq,Sk,D", [(2, 4, 150, 150, 64), (1, 2, 40, 40, 16), (2, 3, 12, 200, 64), (3, 2, 32, 257, 64), (2, 4, 8, 700, 64),
                                         (1, 4, 8, 2012, 64), (2, 2, 14, 1030, 64)])   # (the last two: long key ranges — 8 waves split them with MRB_ATTN_SPLIT8=1)
def test_attention_bias_mask_dropout_fwd_bwd(ops, causal, B, H, Sq, Sk, D):
    if causal and Sq != Sk:
        pytest.skip("causal only for self-attention")
    torch.manual_seed(5)
    scale = 1.0 if D == 64 else 0.25
    p = 0.1
    q = bf(torch.randn(B, Sq, H, D, device=dev()) * 0.5)
    k = bf(torch.randn(B, Sk, H, D, device=dev()) * 0.5)
    v = bf(torch.randn(B, Sk, H, D, device=dev()))
    do = bf(torch.randn(B, Sq, H, D, device=dev()))
    lut = torch.randn(H, 257, device=dev())
    kmask = torch.zeros(B, ops.rup32(Sk), dtype=torch.int32, device=dev())  # rows padded to a multiple of 32 entries
    kmask[:, :Sk] = 1
    kmask[0, Sk - 7:] = 0
    seed = torch.tensor([4242], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 11, p)
    vt = ops.head_transpose(v)
    o = torch.empty_like(q)
    lse = torch.zeros(B, H, ops.rup32(Sq), device=dev())
    ops.attention_fwd(q, k, vt, o, lse, scale=scale, bias_lut=lut, kmask=kmask, causal=causal, drop=drop)
    # reference with autograd
    qr, kr, vr = (t.float().permute(0, 2, 1, 3).clone().requires_grad_(True) for t in (q, k, v))
    bias = _lut_bias(lut, Sq, Sk)[None]
    mask = kmask[:, :Sk].bool()[:, None, None, :].expand(B, H, Sq, Sk)
    if causal:
        mask = mask & torch.tril(torch.ones(Sq, Sk, dtype=torch.bool, device=dev()))[None, None]
    from oracle.mrblip_oracle import dropout_keep_attn
    dmask = dropout_keep_attn(B, H, Sq, Sk, 4242, 11, p).to(dev())
    ref, _ = _attn_ref(qr, kr, vr, scale, bias, mask, dmask, p)
    assert rel(o.float().permute(0, 2, 1, 3), ref) < 6e-3
    ref.backward(do.float().permute(0, 2, 1, 3))
    kt, qt, dot = ops.head_transpose(k), ops.head_transpose(q), ops.head_transpose(do)
    delta = torch.zeros_like(lse)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.attention_bwd(q, k, v, o, do, kt, qt, dot, lse, delta, dq, dk, dv, scale=scale, bias_lut=lut, kmask=kmask, causal=causal, drop=drop)
    assert rel(dq.float().permute(0, 2, 1, 3), qr.grad) < 1.5e-2
    assert rel(dk.float().permute(0, 2, 1, 3), kr.grad) < 1.5e-2
    assert rel(dv.float().permute(0, 2, 1, 3), vr.grad) < 1.5e-2


@pytest.mark.parametrize("n_split", [0, 3, 8, 15])
@pytest.mark.parametrize("B,H,Sq,Sk,masked,p", [(1, 32, 8, 2012, False, 0.1), (1, 32, 14, 2012, True, 0.1), (2, 32, 12, 1030, True, 0.0), (1, 8, 32, 4003, False, 0.0)])
def test_attention_cross_block_key_split(ops, n_split, B, H, Sq, Sk, masked, p):
    """Round 4: the decoder's cross attention (few queries, ~2000 keys, one block per head in the key-split form) with its key range cut over
    several BLOCKS per head (mrblip_attention_set_split_workspace): partial (m, l, O) / partial dQ through a workspace, the last arriver
    combines them in chunk order.  Forward (O, LSE) and backward (dQ; dK / dV do not change form) against the one-block form and against fp32
    torch autograd; repeated launches are bit-identical (the combination order does not depend on which block arrives last) while a large
    GEMM keeps every CU busy on another stream; the tickets are left at zero."""
    torch.manual_seed(15)
    D = 64
    q = bf(torch.randn(B, Sq, H, D, device=dev()) * 0.5)
    k = bf(torch.randn(B, Sk, H, D, device=dev()) * 0.5)
    v = bf(torch.randn(B, Sk, H, D, device=dev()))
    do = bf(torch.randn(B, Sq, H, D, device=dev()))
    kmask = None
    if masked:
        kmask = torch.zeros(B, ops.rup32(Sk), dtype=torch.int32, device=dev())
        kmask[:, :Sk] = 1
        kmask[0, Sk - 40:] = 0
        kmask[0, 100:170] = 0          # two whole key tiles of one chunk masked out
    seed = torch.tensor([777], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 9, p) if p > 0 else None
    vt, kt, qt, dot = ops.head_transpose(v), ops.head_transpose(k), ops.head_transpose(q), ops.head_transpose(do)

    def run():
        o = torch.zeros_like(q)
        lse = torch.zeros(B, H, ops.rup32(Sq), device=dev())
        ops.attention_fwd(q, k, vt, o, lse, scale=1.0, kmask=kmask, drop=drop)
        delta = torch.zeros_like(lse)
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        ops.attention_bwd(q, k, v, o, do, kt, qt, dot, lse, delta, dq, dk, dv, scale=1.0, kmask=kmask, drop=drop)
        return o, lse, dq, dk, dv, delta

    ops.attention_split_workspace(None)
    base = run()
    ws = torch.zeros((16384 + B * H * 64 * 9216) // 4, dtype=torch.int32, device=dev())
    ops.attention_split_workspace(ws, n_split)
    try:
        got = run()
        # a busy chip on another stream: blocks of one head arrive in any order
        a = bf(torch.randn(4096, 4096, device=dev()))
        c = torch.empty(4096, 4096, device=dev())
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(12):
                ops.gemm(a, a, c)
        reps = [run() for _ in range(25)]
        side.synchronize()
    finally:
        ops.attention_split_workspace(None)
    torch.cuda.synchronize()
    assert int(ws[:4096].abs().sum()) == 0                                   # the tickets are back at zero
    for r in reps:
        for x, y in zip(r, got):
            assert torch.equal(x, y)
    names = ("o", "lse", "dq", "dk", "dv", "delta")
    for n, x, y in zip(names, got, base):
        e = rel(x.float()[..., :Sq] if n in ("lse", "delta") else x.float(), y.float()[..., :Sq] if n in ("lse", "delta") else y.float())
        assert e < (3e-3 if n in ("o", "dq", "dk", "dv") else 5e-3 if n == "delta" else 1e-5), (n, e)   # bf16 outputs: rounding flips under another fp32 summation order (dK / dV see it through LSE / Delta = rowsum(dO * bf16 O))
    qr, kr, vr = (t.float().permute(0, 2, 1, 3).clone().requires_grad_(True) for t in (q, k, v))
    mask = kmask[:, :Sk].bool()[:, None, None, :].expand(B, H, Sq, Sk) if masked else None
    dmask = None
    if p > 0:
        from oracle.mrblip_oracle import dropout_keep_attn
        dmask = dropout_keep_attn(B, H, Sq, Sk, 777, 9, p).to(dev())
    ref, _ = _attn_ref(qr, kr, vr, 1.0, None, mask, dmask, p)
    assert rel(got[0].float().permute(0, 2, 1, 3), ref) < 6e-3
    ref.backward(do.float().permute(0, 2, 1, 3))
    assert rel(got[2].float().permute(0, 2, 1, 3), qr.grad) < 1.5e-2


@pytest.mark.parametrize("use_bits", [False, True])
@pytest.mark.parametrize("use_lut", [False, True])
@pytest.mark.parametrize("B,H,Sq,Sk,D", [(1, 3, 300, 300, 64), (2, 2, 257, 190, 64), (1, 2, 70, 333, 64)])
def test_attention_lds_path_dropout_bits(ops, use_bits, use_lut, B, H, Sq, Sk, D):
    """T5-encoder form (no key mask -> interior tiles take the check-free path; several 64-key stages; ragged tails), with the
    dropout keep mask either re-hashed in the backward or carried forward->backward in the drop_bits scratch."""
    torch.manual_seed(7)
    p = 0.1
    q = bf(torch.randn(B, Sq, H, D, device=dev()) * 0.5)
    k = bf(torch.randn(B, Sk, H, D, device=dev()) * 0.5)
    v = bf(torch.randn(B, Sk, H, D, device=dev()))
    do = bf(torch.randn(B, Sq, H, D, device=dev()))
    lut = torch.randn(H, 257, device=dev()) if use_lut else None
    seed = torch.tensor([977], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 5, p)
    bits = torch.full(ops.drop_bits_shape(B, H, Sq, Sk), 0x55555555, dtype=torch.int32, device=dev()) if use_bits else None
    vt = ops.head_transpose(v)
    o = torch.empty_like(q)
    lse = torch.zeros(B, H, ops.rup32(Sq), device=dev())
    ops.attention_fwd(q, k, vt, o, lse, scale=1.0, bias_lut=lut, drop=drop, drop_bits=bits)
    qr, kr, vr = (t.float().permute(0, 2, 1, 3).clone().requires_grad_(True) for t in (q, k, v))
    bias = _lut_bias(lut, Sq, Sk)[None] if use_lut else None
    from oracle.mrblip_oracle import dropout_keep_attn
    dmask = dropout_keep_attn(B, H, Sq, Sk, 977, 5, p).to(dev())
    ref, _ = _attn_ref(qr, kr, vr, 1.0, bias, None, dmask, p)
    assert rel(o.float().permute(0, 2, 1, 3), ref) < 6e-3
    if use_bits:  # the stored words are exactly the oracle's keep mask, key kk of a tile at bit attn_bitpos_key(kk) (csrc/attention.hip:
        # the forward collects the bits of PACKED probability pairs, so a word holds even keys in its low and odd keys in its high half)
        w = bits.view(B, H, ops.rup32(Sk) // 32, ops.rup32(Sq))
        kk = torch.arange(32, device=dev())
        pos = 16 * (kk & 1) + 8 * ((kk >> 3) & 1) + 4 * (kk >> 4) + ((kk & 7) >> 1)
        got = ((w[:, :, :, :Sq, None] >> pos) & 1).permute(0, 1, 3, 2, 4).reshape(B, H, Sq, -1)[..., :Sk]
        assert torch.equal(got.bool(), dmask.bool())
    ref.backward(do.float().permute(0, 2, 1, 3))
    kt, qt, dot = ops.head_transpose(k), ops.head_transpose(q), ops.head_transpose(do)
    delta = torch.zeros_like(lse)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.attention_bwd(q, k, v, o, do, kt, qt, dot, lse, delta, dq, dk, dv, scale=1.0, bias_lut=lut, drop=drop, drop_bits=bits)
    assert rel(dq.float().permute(0, 2, 1, 3), qr.grad) < 1.5e-2
    assert rel(dk.float().permute(0, 2, 1, 3), kr.grad) < 1.5e-2
    assert rel(dv.float().permute(0, 2, 1, 3), vr.grad) < 1.5e-2


@pytest.mark.parametrize("H,S,masked", [(8, 2012, False), (4, 2012, True), (2, 4003, False), (2, 4003, True)])
def test_attention_full_size_t5_encoder(ops, H, S, masked):
    """The T5-encoder attention at the bench's real sequence lengths (QVH S = 2012, ActivityNet S = 4003: VERDICT r1 weak #2): LUT
    bias, dropout with the keep bits carried forward -> backward, with and without a key mask (masked tail = right-padded text),
    against fp32 torch autograd evaluated head by head."""
    from util import check
    torch.manual_seed(9)
    B, D, p = 1, 64, 0.1
    q = bf(torch.randn(B, S, H, D, device=dev()) * 0.5)
    k = bf(torch.randn(B, S, H, D, device=dev()) * 0.5)
    v = bf(torch.randn(B, S, H, D, device=dev()))
    do = bf(torch.randn(B, S, H, D, device=dev()))
    lut = torch.randn(H, 257, device=dev())
    kmask = None
    if masked:
        kmask = torch.zeros(B, ops.rup32(S), dtype=torch.int32, device=dev())
        kmask[:, :S - 11] = 1
    seed = torch.tensor([31337], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 77, p)
    bits = torch.zeros(ops.drop_bits_shape(B, H, S, S), dtype=torch.int32, device=dev())
    vt = ops.head_transpose(v)
    o = torch.empty_like(q)
    lse = torch.zeros(B, H, ops.rup32(S), device=dev())
    ops.attention_fwd(q, k, vt, o, lse, scale=1.0, bias_lut=lut, kmask=kmask, drop=drop, drop_bits=bits)
    kt, qt, dot = ops.head_transpose(k), ops.head_transpose(q), ops.head_transpose(do)
    delta = torch.zeros_like(lse)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.attention_bwd(q, k, v, o, do, kt, qt, dot, lse, delta, dq, dk, dv, scale=1.0, bias_lut=lut, kmask=kmask, drop=drop, drop_bits=bits)
    from oracle.mrblip_oracle import dropout_keep_attn
    num = dict(o=0.0, dq=0.0, dk=0.0, dv=0.0)
    den = dict(o=0.0, dq=0.0, dk=0.0, dv=0.0)
    keep_all = dropout_keep_attn(B, H, S, S, 31337, 77, p)   # the oracle's restatement of the kernels' draws (one hash per key quad)
    for h in range(H):  # one head at a time: [S, S] fp32 scores
        qr, kr, vr = (t[:, :, h].float().clone().requires_grad_(True) for t in (q, k, v))  # [B,S,D]
        dmask = keep_all[:, h].to(dev())
        bias = _lut_bias(lut[h:h + 1], S, S)
        mask = None if kmask is None else kmask[:, :S].bool()[:, None, :].expand(B, S, S)
        ref, _ = _attn_ref(qr, kr, vr, 1.0, bias, mask, dmask, p)
        ref.backward(do[:, :, h].float())
        for nm, got, want in (("o", o[:, :, h], ref.detach()), ("dq", dq[:, :, h], qr.grad), ("dk", dk[:, :, h], kr.grad), ("dv", dv[:, :, h], vr.grad)):
            num[nm] += (got.float() - want).double().pow(2).sum().item()
            den[nm] += want.double().pow(2).sum().item()
    tag = "attn.t5enc S=%d H=%d %s: " % (S, H, "masked" if masked else "no-mask")
    check(tag + "o vs fp32 torch", (num["o"] / den["o"]) ** 0.5, 4e-3)
    check(tag + "dq vs fp32 autograd", (num["dq"] / den["dq"]) ** 0.5, 5.5e-3)
    check(tag + "dk vs fp32 autograd", (num["dk"] / den["dk"]) ** 0.5, 5.5e-3)
    check(tag + "dv vs fp32 autograd", (num["dv"] / den["dv"]) ** 0.5, 5.5e-3)


def test_attention_strided_qkv_buffer(ops):
    """ViT/T5 layout: q, k, v are column slices of one [B*S, 3*H*D] GEMM output."""
    torch.manual_seed(6)
    B, S, H, D = 2, 70, 4, 64
    qkv = bf(torch.randn(B, S, 3, H, D, device=dev()))
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    o = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=dev())
    ops.attention_fwd(q, k, ops.head_transpose(v), o, None, scale=0.125)
    ref, _ = _attn_ref(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3), 0.125)
    assert rel(o.float().permute(0, 2, 1, 3), ref) < 6e-3


def test_patchify_assemble_rowcopy_meanpool(ops):
    torch.manual_seed(7)
    F_, IMG, P, D = 3, 56, 14, 96
    G = IMG // P
    video = torch.randn(F_, 3, IMG, IMG, device=dev())
    out = torch.full((F_ * G * G, 640), 7.0, dtype=torch.bfloat16, device=dev())
    ops.patchify(video, out, P)
    ref = video.reshape(F_, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(F_ * G * G, 3 * P * P)
    assert torch.equal(out[:, :588], ref.bfloat16()) and out[:, 588:].abs().max() == 0
    # uint8 frames with the processor's ToTensor + Normalize fused into the load: bit-identical to normalising first
    u8 = torch.randint(0, 256, (F_, 3, IMG, IMG), device=dev(), dtype=torch.uint8)
    mean = torch.tensor(ops.CLIP_MEAN, device=dev()).view(1, 3, 1, 1)
    std = torch.tensor(ops.CLIP_STD, device=dev()).view(1, 3, 1, 1)
    norm = (u8.float() / 255.0 - mean) / std
    a = torch.full((F_ * G * G, 640), 7.0, dtype=torch.bfloat16, device=dev())
    b = torch.full_like(a, 5.0)
    ops.patchify(norm.contiguous(), a, P)
    ops.patchify(u8, b, P)
    assert torch.equal(a, b)
    patch = torch.randn(F_ * G * G, D, device=dev())
    cls, pos = torch.randn(D, device=dev()), torch.randn(G * G + 1, D, device=dev())
    x = torch.empty(F_, G * G + 1, D, device=dev())
    ops.vit_assemble(patch, cls, pos, x)
    assert torch.equal(x, torch.cat([cls.expand(F_, 1, D), patch.reshape(F_, G * G, D)], 1) + pos)
    # indexed row copy: bit exact, zero rows, accumulate
    src = torch.randn(10, D, device=dev())
    dst = torch.full((6, D), 3.0, device=dev())
    si = torch.tensor([4, -1, 9, 0], dtype=torch.int32, device=dev())
    di = torch.tensor([0, 2, 5, 3], dtype=torch.int32, device=dev())
    ops.row_copy(src, si, dst, di)
    assert torch.equal(dst[0], src[4]) and dst[2].abs().max() == 0 and torch.equal(dst[5], src[9]) and torch.equal(dst[3], src[0]) and (dst[1] == 3).all()
    ops.row_copy(src, si, dst, di, accumulate=True)
    assert torch.equal(dst[0], src[4] * 2)
    # mean pool 32 -> 1
    t = torch.randn(5, 32, 2048, device=dev())
    m = torch.empty(5, 2048, device=dev())
    ops.mean_pool(t, m)
    assert rel(m, t.mean(1)) < 1e-6
    dx = torch.empty_like(t)
    ops.mean_pool_bwd(m, dx)
    assert torch.allclose(dx, (m / 32)[:, None].expand_as(t))


def test_cast_gelu_ce_adamw(ops):
    torch.manual_seed(8)
    M, N = 33, 256
    x = torch.randn(M, N, device=dev())
    seed = torch.tensor([5], dtype=torch.int32, device=dev())
    ob = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.cast_dropout(x, out_bf16=ob, drop=ops.Dropout(seed, 2, 0.1))
    assert torch.equal(ob, (x * keep_mask((M, N), 5, 2, 0.1) / 0.9).bfloat16())
    ops.seed_bump(seed)
    assert seed.item() == (5 * 1664525 + 1013904223) % 2 ** 32 - (2 ** 32 if (5 * 1664525 + 1013904223) % 2 ** 32 >= 2 ** 31 else 0)
    h = bf(torch.randn(M, N, device=dev()))
    dy = bf(torch.randn(M, N, device=dev()))
    dh = torch.empty_like(h)
    ops.gelu_bwd(dy, h, dh)
    hh = h.float().requires_grad_(True)
    torch.nn.functional.gelu(hh).backward(dy.float())
    assert rel(dh.float(), hh.grad) < 4e-3
    # cross entropy with ignore_index
    R, V = 9, 32128
    logits = torch.randn(R, V, device=dev()) * 2
    labels = torch.randint(0, V, (R,), device=dev())
    labels[3] = -100
    lg = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lg, labels, ignore_index=-100)
    ref.backward()
    loss = torch.zeros(1, device=dev())
    dl = torch.empty(R, V, dtype=torch.bfloat16, device=dev())
    ops.cross_entropy(logits, labels.int(), 1.0 / 8, loss, dl)
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    assert rel(dl.float(), lg.grad) < 4e-3
    # AdamW vs torch.optim.AdamW, 3 steps
    n = 1000
    p0 = torch.randn(n, device=dev())
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    p, m, v = p0.clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    for step in range(1, 4):
        g = torch.randn(n, device=dev())
        pt.grad = g.clone()
        opt.step()
        hyper = torch.tensor([3e-4, 1 / (1 - 0.9 ** step), 1 / math.sqrt(1 - 0.999 ** step), 1.0], device=dev())
        ops.adamw(p, g, m, v, hyper, weight_decay=0.05)
    assert rel(p, pt.detach()) < 1e-6


def test_lora_pieces(ops):
    """LoRA side kernels: lora_dropout (bf16->bf16), the transposing copy with the same mask, dx += mask*(G A), operand packing."""
    torch.manual_seed(9)
    M, K = 101, 256
    x = bf(torch.randn(M, K, device=dev()))
    seed = torch.tensor([77], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 21, 0.05)
    mask = keep_mask((M, K), 77, 21, 0.05)
    xd = torch.zeros_like(x)
    ops.dropout_bf16(x, xd, drop)
    ref = (x.float() * mask / 0.95).bfloat16()
    assert torch.equal(xd, ref)
    # transposed copy of dropout(x): [K, Mpad] through the head-transpose kernel with the same mask
    Mp = 128
    xt = torch.full((K, Mp), 9.0, dtype=torch.bfloat16, device=dev())
    v = torch.as_strided(x, (1, M, K // 64, 64), (0, x.stride(0), 64, 1))
    ops.head_transpose(v, out=xt.view(1, K // 64, 64, Mp), spad=Mp, drop=drop)
    assert torch.equal(xt[:, :M], ref.t()) and xt[:, M:].abs().sum() == 0
    # dx += mask * (G Acat)
    R = 16
    G = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
    G[:, :R] = bf(torch.randn(M, R, device=dev()))
    A = bf(torch.randn(R, K, device=dev()) * 0.1)
    for dt in (torch.float32, torch.bfloat16):
        dx0 = torch.randn(M, K, device=dev()).to(dt)
        dx = dx0.clone()
        ops.lora_dx_add(dx, G, A, drop=drop)
        want = dx0.float() + (G[:, :R].float() @ A.float()) * mask / 0.95
        assert rel(dx.float(), want) < (1e-6 if dt == torch.float32 else 4e-3)
    # packing: A [8,K], Bt [8,out] fp32 -> acat (scaled), wext, bblk (block-diagonal, scaled)
    out_, Ntot, row0, col0 = 72, 200, 64, 8
    flat = torch.randn(8 * K + 8 * out_, device=dev())
    acat = torch.zeros(16 * K, dtype=torch.bfloat16, device=dev())
    wext = torch.zeros(Ntot * 64, dtype=torch.bfloat16, device=dev())
    bblk = torch.zeros(16 * Ntot, dtype=torch.bfloat16, device=dev())
    acatt = torch.zeros(K, 64, dtype=torch.bfloat16, device=dev())
    desc = torch.tensor([[0, 8 * K, K, out_, 8 * K, row0 * 64 + col0, 8 * Ntot + row0, Ntot, 8, 0]], dtype=torch.int64, device=dev())
    ops.lora_pack(flat, acat, wext, bblk, acatt, desc, 1, scale=2.0)
    Af, Bt = flat[: 8 * K].view(8, K), flat[8 * K:].view(8, out_)
    assert torch.equal(acatt[:, 8:16], (Af * 2).bfloat16().t()) and acatt[:, :8].abs().sum() == 0 and acatt[:, 16:].abs().sum() == 0
    assert torch.equal(acat.view(16, K)[8:], (Af * 2).bfloat16()) and acat.view(16, K)[:8].abs().sum() == 0
    assert torch.equal(wext.view(Ntot, 64)[row0: row0 + out_, col0: col0 + 8], Bt.t().bfloat16())
    assert torch.equal(bblk.view(16, Ntot)[8:, row0: row0 + out_], (Bt * 2).bfloat16())
    assert wext.float().abs().sum() == Bt.t().bfloat16().float().abs().sum()
    # the thin rank-8 products run on the GEMM kernel: u = x A^T (N=8), dB^T += u^T dy via transposed operands (M'=8 rows)
    u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
    ops.gemm(x, acat.view(16, K), u, tile_cfg=3)
    assert rel(u[:, :16].float(), x.float() @ acat.view(16, K).float().t()) < 3e-3 and u[:, 16:].abs().sum() == 0
    uT = torch.zeros(64, Mp, dtype=torch.bfloat16, device=dev())
    ops.head_transpose(torch.as_strided(u, (1, M, 1, 64), (0, 64, 64, 1)), out=uT.view(1, 1, 64, Mp), spad=Mp)
    dW = torch.ones(8, K, device=dev())
    ops.gemm(uT[8:16], xt, dW, residual=dW, tile_cfg=3, K=Mp)
    assert rel(dW, 1 + u[:, 8:16].float().t() @ ref.float()) < 1e-5


@pytest.mark.parametrize("M", [12, 333])
def test_lora_fused_launches(ops, M):
    """The one-launch forms of the LoRA side products against their definitions (same masks as the separate kernels):
    u = dropout(x) Acat^T, dx = dy Wt^T + residual + mask * (g AcatT^T) on the tile and the skinny kernel, and both weight gradients."""
    torch.manual_seed(12)
    K, N, p = 256, 320, 0.05       # in_features, out_features
    x = bf(torch.randn(M, K, device=dev()))
    seed = torch.tensor([55], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 9, p)
    mask = keep_mask((M, K), 55, 9, p)
    acat = bf(torch.randn(24, K, device=dev()) * 0.1)
    # down
    u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
    ops.lora_down(x, acat, u, K, drop=drop)
    want = (x.float() * mask) @ acat.float().t() / (1 - p)
    assert rel(u[:, :24].float(), want) < 4e-3 and u[:, 24:].abs().sum() == 0
    ops.lora_down(x, acat, u, K)
    assert rel(u[:, :24].float(), x.float() @ acat.float().t()) < 4e-3
    # dx
    dy = bf(torch.randn(M, N, device=dev()))
    wt = bf(torch.randn(K, N, device=dev()) * 0.1)             # [in, out]
    g = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
    g[:, :24] = bf(torch.randn(M, 24, device=dev()))
    acatt = torch.zeros(K, 64, dtype=torch.bfloat16, device=dev())
    acatt[:, :24] = acat.t()
    res = torch.randn(M, K, device=dev())
    want = dy.float() @ wt.float().t() + res + (g[:, :24].float() @ acat.float()) * mask / (1 - p)
    for cfg in ([3] if M <= 64 else [2, 4, 5, 1, 7, 8, 9]):
        dx = torch.full((M, K), float("nan"), device=dev())
        ops.lora_dx(dy, wt, g, acatt, dx, N, residual=res, drop=drop, tile_cfg=cfg)
        assert rel(dx, want) < 1e-5, cfg
        dxb = torch.zeros(M, K, dtype=torch.bfloat16, device=dev())
        ops.lora_dx(dy, wt, g, acatt, dxb, N, drop=None, tile_cfg=cfg)
        assert rel(dxb.float(), dy.float() @ wt.float().t() + g[:, :24].float() @ acat.float()) < 4e-3, cfg
    # both weight gradients, one launch
    outs_, col0 = [128, 64, 128], [0, 128, 192]
    dB = [torch.ones(8, o, device=dev()) for o in outs_]
    dA = [torch.ones(8, K, device=dev()) for _ in range(3)]
    ops.lora_grads(dy, u, x, g, dB, col0, outs_, dA, K, drop=drop)
    xd = (x.float() * mask / (1 - p)).bfloat16().float()
    for j, o in enumerate(outs_):
        assert rel(dB[j], 1 + u[:, 8 * j: 8 * j + 8].float().t() @ dy[:, col0[j]: col0[j] + o].float()) < 1e-5, j
        assert rel(dA[j], 1 + g[:, 8 * j: 8 * j + 8].float().t() @ xd) < 1e-5, j


def test_lora_tn_weight_gradients(ops):
    """dB^T = u^T dy (block-diagonal over the adapters of a fused group) and dA = g^T dropout(x), contraction over rows, no transposes."""
    torch.manual_seed(10)
    M, K, outs_ = 333, 192, [64, 64, 72]
    Ntot = sum(outs_)
    dy = bf(torch.randn(M, Ntot, device=dev()))
    u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
    u[:, :24] = bf(torch.randn(M, 24, device=dev()))
    dB = [torch.ones(8, o, device=dev()) for o in outs_]
    col0 = [0, 64, 128]
    ops.lora_tn(dy, u, dB, col0, outs_, outs_)
    for j, o in enumerate(outs_):
        ref = 1 + u[:, 8 * j: 8 * j + 8].float().t() @ dy[:, col0[j]: col0[j] + o].float()
        assert rel(dB[j], ref) < 1e-5, j
    x = bf(torch.randn(M, K, device=dev()))
    seed = torch.tensor([31], dtype=torch.int32, device=dev())
    dA = [torch.zeros(8, K, device=dev()) for _ in range(3)]
    ops.lora_tn(x, u, dA, [0] * 3, [K] * 3, [K] * 3, drop=ops.Dropout(seed, 5, 0.05))
    xd = (x.float() * keep_mask((M, K), 31, 5, 0.05) / 0.95).bfloat16().float()
    for j in range(3):
        assert rel(dA[j], u[:, 8 * j: 8 * j + 8].float().t() @ xd) < 1e-5, j


@pytest.mark.parametrize("M,K,R", [(12, 2048, 24), (2012, 2048, 24), (2012, 5120, 8), (333, 64, 16), (8, 32128, 8), (1100, 10240, 16), (70, 768, 32)])
def test_lora_rows_kernel(ops, M, K, R):
    """u = dropout(x) A^T (csrc/lora.hip) against fp32 torch with the oracle's restatement of the mask, and against the MFMA skinny
    kernel it replaces (same operands, same mask: equal up to the summation order)."""
    from oracle.mrblip_oracle import dropout_keep
    from util import check
    torch.manual_seed(21)
    p = 0.05
    ldx = K + 64
    xbuf = bf(torch.randn(M, ldx, device=dev()))
    x = xbuf[:, :K]
    a = bf(torch.randn(R, K, device=dev()) * 0.05)
    seed = torch.tensor([555], dtype=torch.int32, device=dev())
    for drop in (None, ops.Dropout(seed, 9, p)):
        u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
        ops.lora_rows(xbuf, a, u, K, drop=drop)
        xf = x.float()
        if drop is not None:
            xf = xf * dropout_keep((M, K), 555, 9, p).to(dev()) / (1 - p)
        ref = xf @ a.float().t()
        tag = "lora_rows M=%d K=%d R=%d %s: " % (M, K, R, "drop" if drop else "plain")
        check(tag + "vs fp32 torch", rel(u[:, :R].float(), ref), 4e-3)
        assert u[:, R:].abs().sum() == 0
        if K % 64 == 0:
            u2 = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
            ops.lora_down(xbuf, a, u2, K, drop=drop)
            check(tag + "vs the MFMA skinny kernel", rel(u[:, :R].float(), u2[:, :R].float()), 1.3e-4)   # (bf16 last-bit flips: measured 0 ... 2.7e-5)
    if R >= 16 and K % (R // 8) == 0:  # block-diagonal A (the backward's s*B^T of a fused group): segment skipping == dense evaluation
        ng, w = R // 8, K // (R // 8)
        ad = torch.zeros_like(a)
        for j in range(ng):
            ad[8 * j: 8 * j + 8, j * w: (j + 1) * w] = a[8 * j: 8 * j + 8, j * w: (j + 1) * w]
        ud, us = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev()), torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
        ops.lora_rows(xbuf, ad, ud, K)
        ops.lora_rows(xbuf, ad, us, K, seg=[v for j in range(ng) for v in (j * w, (j + 1) * w)])
        assert torch.equal(ud, us)
        assert rel(us[:, :R].float(), x.float() @ ad.float().t()) < 4e-3


@pytest.mark.parametrize("M,D,R", [(12, 2048, 24), (2012, 2048, 16), (2012, 2048, 24), (2012, 2048, 8), (77, 768, 24), (5, 64, 8)])
def test_rmsnorm_lora_fused(ops, M, D, R):
    """fused T5 RMSNorm + LoRA down == rmsnorm_fwd followed by lora_rows (bit-identical xn; u identical: same arithmetic order per row)"""
    torch.manual_seed(22)
    x = torch.randn(M, D, device=dev()) * 1.7
    w = torch.randn(D, device=dev()) * 0.1 + 1
    a = bf(torch.randn(R, D, device=dev()) * 0.05)
    seed = torch.tensor([777], dtype=torch.int32, device=dev())
    Dp = (D + 63) // 64 * 64
    for drop in (None, ops.Dropout(seed, 3, 0.05)):
        xn1 = torch.zeros(M, Dp, dtype=torch.bfloat16, device=dev())
        u1 = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
        ops.rmsnorm_fwd(x, w, 1e-6, out_bf16=xn1)
        ops.lora_rows(xn1, a, u1, D, drop=drop)
        xn2 = torch.zeros_like(xn1)
        u2 = torch.zeros_like(u1)
        ops.rmsnorm_lora_fwd(x, w, 1e-6, xn2, a, u2, drop=drop)
        assert torch.equal(xn1, xn2)
        assert rel(u2.float(), u1.float()) < 4e-3 and u2[:, R:].abs().sum() == 0


@pytest.mark.parametrize("M,N,K,ks", [(8, 2048, 2048, 4), (12, 2048, 5120, 4), (32, 768, 2048, 8), (5, 2048, 10240, 4), (8, 96, 1024, 2)])
def test_gemm_k_split_skinny_with_preinitialised_output(ops, M, N, K, ks):
    """decoder-row GEMMs: lora_rows pre-initialises the fp32 output with the residual (side job of the LoRA "down" launch), the K-split
    skinny GEMM adds its partial products atomically (bias once, LoRA K-extension once, output dropout on every partial) — equal to the
    one-block-per-tile form up to the fp32 summation order."""
    from util import check
    torch.manual_seed(31)
    x = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.03)
    acat = bf(torch.randn(16, K, device=dev()) * 0.05)
    wext = torch.zeros(N, 64, dtype=torch.bfloat16, device=dev())
    wext[:, :16] = bf(torch.randn(N, 16, device=dev()) * 0.05)
    res = torch.randn(M, N, device=dev())
    bias = torch.randn(N, device=dev())
    seed = torch.tensor([99], dtype=torch.int32, device=dev())
    ldrop, odrop = ops.Dropout(seed, 4, 0.05), ops.Dropout(seed, 8, 0.1)
    # reference: the unsplit path
    u0 = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
    ops.lora_rows(x, acat, u0, K, drop=ldrop)
    ref = torch.empty(M, N, device=dev())
    ops.gemm(x, w, ref, aext=u0, wext=wext, bias=bias, residual=res, drop=odrop)
    # split path
    u1 = torch.zeros_like(u0)
    out = torch.full((M, N), 7.0, device=dev())
    ops.lora_rows(x, acat, u1, K, drop=ldrop, init_dst=out, init_src=res)
    assert torch.equal(u0, u1)
    ops.gemm(x, w, out, aext=u1, wext=wext, bias=bias, drop=odrop, k_splits=ks)
    check("gemm k-split M=%d N=%d K=%d ks=%d vs unsplit" % (M, N, K, ks), rel(out, ref), 1e-6)
    # zero initialisation + the LoRA backward form (masked K-extension first)
    g = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
    g[:, :16] = bf(torch.randn(M, 16, device=dev()))
    acatt = torch.zeros(N, 64, dtype=torch.bfloat16, device=dev())
    acatt[:, :16] = bf(torch.randn(N, 16, device=dev()) * 0.05)
    ref2 = torch.empty(M, N, device=dev())
    ops.lora_dx(x, w, g, acatt, ref2, K, residual=None, drop=ldrop)
    out2 = torch.full((M, N), -3.0, device=dev())
    ops.lora_rows(x, acat, u1, K, init_dst=out2, init_src=None)
    ops.lora_dx(x, w, g, acatt, out2, K, residual=None, drop=ldrop, k_splits=ks)
    check("lora_dx k-split M=%d N=%d K=%d ks=%d vs unsplit" % (M, N, K, ks), rel(out2, ref2), 1e-6)


@pytest.mark.parametrize("M,N,K", [(700, 520, 320), (513, 1408, 1408), (300, 264, 128), (1200, 1024, 2560), (257, 8, 64), (2000, 776, 1152)])
def test_gemm_deferred_epilogue_tile(ops, M, N, K):
    """cfg 15 = cfg 13's tile with the deferred epilogue (results parked as bf16 in VGPRs, activated / stored in the shadow of the NEXT
    tile's MFMAs, FIFO of 16 blocks).  bias only: cfg 13's result (up to the summation order of the bias).  bias + GELU: the pre-activation is rounded to bf16 before
    the GELU, so compare with fp32 torch on that rounding (exact restatement) and loosely with the un-rounded GELU."""
    torch.manual_seed(12)
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.05)
    bias = torch.randn(N, device=dev())
    base = a.float() @ w.float().t() + bias
    o13 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev())
    o15 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev())
    ops.gemm(a, w, o13, bias=bias, tile_cfg=13)
    ops.gemm(a, w, o15, bias=bias, tile_cfg=15)
    # (cfg 15 starts its accumulators AT the bias, cfg 13 adds it last: same value up to the fp32 summation order, i.e. a rare last-bit
    # flip of the bf16 result)
    assert rel(o15.float(), o13.float()) < 2e-4 and rel(o15.float(), base) < 3e-3
    assert (o15.float() - o13.float()).abs().max() <= 0.0079 * base.abs().max()
    ops.gemm(a, w, o15, tile_cfg=15)                       # no bias
    assert rel(o15.float(), a.float() @ w.float().t()) < 3e-3
    ops.gemm(a, w, o15, bias=bias, act=1, tile_cfg=15)
    want = torch.nn.functional.gelu(base.bfloat16().float())
    assert rel(o15.float(), want) < 3e-3 and rel(o15.float(), torch.nn.functional.gelu(base)) < 5e-3
    with ops.gemm_cu_reserve(64):
        o2 = torch.empty_like(o15)
        ops.gemm(a, w, o2, bias=bias, act=1, tile_cfg=15)
    assert torch.equal(o2, o15)
    with pytest.raises(ops.MrblipError):
        ops.gemm(a, w, torch.empty(M, N, device=dev()), bias=bias, tile_cfg=15)     # fp32 out is cfg 13's job


@pytest.mark.parametrize("M,N,K", [(700, 520, 320), (513, 1408, 1408), (300, 264, 128), (1200, 1024, 2560), (257, 8, 64), (2000, 776, 1152)])
def test_gemm_16x16x32_tile(ops, M, N, K):
    """cfg 16 = cfg 13's 256x256x64 tile on v_mfma_f32_16x16x32_bf16 (64 accumulator tiles of 16x16 per wave): the same products in the
    same k order, so every epilogue form equals cfg 13 bit for bit — ragged M / N, bias, GELU, fp32 residual, CU reserve."""
    torch.manual_seed(13)
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.05)
    bias = torch.randn(N, device=dev())
    res = torch.randn(M, N, device=dev())
    for kw, dt in ((dict(bias=bias), torch.bfloat16), (dict(), torch.bfloat16), (dict(bias=bias, act=1), torch.bfloat16),
                   (dict(bias=bias, residual=res), torch.float32), (dict(residual=res), torch.float32), (dict(bias=bias), torch.float32)):
        o13 = torch.full((M, N), float("nan"), dtype=dt, device=dev())
        o16 = torch.full((M, N), float("nan"), dtype=dt, device=dev())
        ops.gemm(a, w, o13, tile_cfg=13, **kw)
        ops.gemm(a, w, o16, tile_cfg=16, **kw)
        assert torch.equal(o13, o16), kw.keys()
    with ops.gemm_cu_reserve(64):
        o2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
        ops.gemm(a, w, o2, bias=bias, act=1, tile_cfg=16)
    o13b = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(a, w, o13b, bias=bias, act=1, tile_cfg=13)
    assert torch.equal(o2, o13b)


@pytest.mark.parametrize("B,H,S,D", [(3, 16, 257, 88), (2, 4, 300, 88), (1, 2, 64, 72), (2, 3, 129, 96), (2, 2, 258, 88), (1, 3, 260, 80), (2, 2, 131, 88)])
def test_attention_fwd_row_major_v(ops, B, H, S, D):
    """mrblip_attention_fwd_rowv (V read row-major from the fused qkv buffer through LDS transpose reads) against the transposed-copy
    path — the same bf16 products in the same order: bit-identical — and against fp32 torch; q / k / v are column slices of ONE
    [B*S, 3*H*D] buffer, as in the ViT."""
    torch.manual_seed(21)
    qkv = bf(torch.randn(B * S, 3 * H * D, device=dev()))
    view = lambda c0: torch.as_strided(qkv, (B, S, H, D), (S * 3 * H * D, 3 * H * D, D, 1), c0)  # noqa: E731
    q, k, v = view(0), view(H * D), view(2 * H * D)
    scale = D ** -0.5
    o_ref = torch.full((B, S, H, D), float("nan"), dtype=torch.bfloat16, device=dev())
    o_new = torch.full((B, S, H, D), float("nan"), dtype=torch.bfloat16, device=dev())
    ops.attention_fwd(q, k, ops.head_transpose(v), o_ref, None, scale=scale)
    lse = torch.zeros(B, H, ops.rup32(S), device=dev())
    ops.attention_fwd_rowv(q, k, v, o_new, lse, scale=scale)
    assert torch.equal(o_new, o_ref)
    want = torch.nn.functional.scaled_dot_product_attention(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3),
                                                            v.float().permute(0, 2, 1, 3), scale=scale).permute(0, 2, 1, 3)
    assert rel(o_new.float(), want) < 4e-3
    s = (q.float().permute(0, 2, 1, 3) @ k.float().permute(0, 2, 3, 1)) * scale
    assert rel(lse[:, :, :S], torch.logsumexp(s, -1)) < 1e-5
    with pytest.raises(ops.MrblipError):      # head_dim <= 64 / few queries keep the transposed-copy entry
        ops.attention_fwd_rowv(q[..., :64], k[..., :64], v[..., :64], o_new[..., :64], None, scale=scale)


@pytest.mark.parametrize("D,p", [(2048, 0.1), (768, 0.1), (64, 0.0)])
def test_rmsnorm_bwd_with_fused_operand_cast(ops, D, p):
    """mrblip_rmsnorm_bwd_cast: dx as mrblip_rmsnorm_bwd writes it, plus bf16(dropout-backward(dx)) — bit for bit what a following
    mrblip_cast_dropout launch produces (same element index, same hash, same rounding)."""
    torch.manual_seed(31)
    M = 301
    x = torch.randn(M, D, device=dev()) * 1.5
    w = torch.randn(D, device=dev()) * 0.1 + 1
    dy = torch.randn(M, D, device=dev())
    add = torch.randn(M, D, device=dev())
    seed = torch.tensor([777], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 23, p) if p > 0 else None
    dx_ref = torch.empty_like(x)
    ops.rmsnorm_bwd(dy, x, w, 1e-6, dx_ref, dx_add=add)
    want = torch.zeros(M, D + 64, dtype=torch.bfloat16, device=dev())[:, :D]
    ops.cast_dropout(dx_ref, out_bf16=want, drop=drop)
    dx = torch.empty_like(x)
    got = torch.zeros(M, D + 64, dtype=torch.bfloat16, device=dev())[:, :D]     # (row stride != D, like the padded operand buffers)
    ops.rmsnorm_bwd(dy, x, w, 1e-6, dx, dx_add=add, out_bf16=got, out_drop=drop)
    assert torch.equal(dx, dx_ref) and torch.equal(got, want)
    if p > 0:
        assert 0.85 < (got != 0).float().mean().item() < 0.95


@pytest.mark.parametrize("D,p", [(768, 0.1), (768, 0.0), (64, 0.1)])
def test_layernorm_bwd_with_fused_operand_cast(ops, D, p):
    """mrblip_layernorm_bwd_cast (round 6, the Q-Former's post-LayerNorm sub-layers): dx as mrblip_layernorm_bwd writes it, plus
    bf16(dropout-backward(dx)) — bit for bit what a following mrblip_cast_dropout launch produces."""
    torch.manual_seed(32)
    M = 1920 if D == 768 else 77
    x = torch.randn(M, D, device=dev()) * 1.5 + 0.3
    w = torch.randn(D, device=dev()) * 0.1 + 1
    dy = torch.randn(M, D, device=dev())
    seed = torch.tensor([778], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 24, p) if p > 0 else None
    dx_ref = torch.empty_like(x)
    ops.layernorm_bwd(dy, x, w, 1e-12, dx_ref)
    want = torch.zeros(M, D + 64, dtype=torch.bfloat16, device=dev())[:, :D]
    ops.cast_dropout(dx_ref, out_bf16=want, drop=drop)
    dx = torch.empty_like(x)
    got = torch.zeros(M, D + 64, dtype=torch.bfloat16, device=dev())[:, :D]
    ops.layernorm_bwd(dy, x, w, 1e-12, dx, out_bf16=got, out_drop=drop)
    assert torch.equal(dx, dx_ref) and torch.equal(got, want)
    if p > 0:
        assert 0.85 < (got != 0).float().mean().item() < 0.95


@pytest.mark.parametrize("M,N,K,cfg", [(1920, 3072, 768, 0), (1920, 3072, 768, 5), (640, 3072, 768, 0), (77, 128, 64, 0), (1920, 3072, 768, 2)])
def test_gemm_gelu_backward_epilogue_equals_gemm_plus_gelu_bwd(ops, M, N, K, cfg):
    """act = 2 (round 6): dh = (dy W^T) * gelu'(hpre) from the GEMM's epilogue, out2 READ as the saved pre-activation — bit for bit the bf16 GEMM
    followed by mrblip_gelu_bwd (the accumulator is rounded through bf16 first); the Q-Former FFN's backward (Qformer.py:349-360)."""
    torch.manual_seed(33)
    a = torch.randn(M, K, device=dev()).bfloat16()
    w = (torch.randn(N, K, device=dev()) * 0.05).bfloat16()
    hpre = (torch.randn(M, N, device=dev()) * 1.5).bfloat16()
    hpre0 = hpre.clone()
    dh = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(a, w, dh, tile_cfg=cfg)
    want = torch.empty_like(dh)
    ops.gelu_bwd(dh, hpre, want)
    got = torch.empty_like(dh)
    ops.gemm(a, w, got, out2=hpre, act=2, tile_cfg=cfg)
    assert torch.equal(got, want) and torch.equal(hpre, hpre0)      # (the pre-activation is read, never written)
    with pytest.raises(ops.MrblipError):
        ops.gemm(a, w, got, act=2)                                  # no saved pre-activation


# ---- round 5: K-split form of the 4-wave kernel and the consumers that add its parts -------------------------------------------------
@pytest.mark.parametrize("f32out", [True, False])
@pytest.mark.parametrize("M,N,K,ks,cfg,ext", [(2012, 2048, 2560, 4, 13, True), (1312, 256, 768, 6, 14, True), (300, 264, 128, 2, 13, False),
                                               (2012, 1024, 1024, 1, 14, True), (513, 520, 192, 3, 13, True), (257, 8, 64, 1, 13, True),
                                               (517, 264, 384, 3, 22, True), (2012, 2048, 1024, 2, 22, False)])
def test_gemm_k_split_parts_of_the_four_wave_kernel(ops, M, N, K, ks, cfg, ext, f32out):
    """mrblip_gemm_ksplit: part s = A[:, K range s] W[:, K range s]^T, the K extension's product as the last part — against fp32 matmuls of
    the bf16 operands (fp32 parts: accumulation order only, 2e-6 measured; bf16 parts: one rounding); rows beyond M and columns beyond N of
    the last tiles must not be written (canary)."""
    torch.manual_seed(5)
    a = bf(torch.randn(M, K + 64, device=dev()))[:, :K]           # (row stride != K)
    w = bf(torch.randn(N, K, device=dev()) * 0.1)
    g = bf(torch.randn(M, 64, device=dev())) if ext else None
    at = bf(torch.randn(N, 64, device=dev()) * 0.1) if ext else None
    nparts = ks + (1 if ext else 0)
    dt = torch.float32 if f32out else torch.bfloat16
    buf = torch.full((nparts, M + 3, N + 8), 7.0, dtype=dt, device=dev())
    parts = buf[:, :M, :N]
    ops.gemm_ksplit(a, w, parts, K, ks, ext=(g, at) if ext else None, tile_cfg=cfg)
    torch.cuda.synchronize()
    kp = K // ks
    for s_ in range(ks):
        ref = a[:, s_ * kp:(s_ + 1) * kp].float() @ w[:, s_ * kp:(s_ + 1) * kp].float().t()
        assert rel(parts[s_].float(), ref) < (5e-6 if f32out else 4e-3), (s_, rel(parts[s_].float(), ref))
    if ext:
        ref = g.float() @ at.float().t()
        assert rel(parts[ks].float(), ref) < (5e-6 if f32out else 4e-3)
    assert torch.all(buf[:, M:, :] == 7.0) and torch.all(buf[:, :, N:] == 7.0)
    # the same launch again: the same bits (fixed K ranges, no atomics)
    again = torch.full_like(buf, 3.0)
    ops.gemm_ksplit(a, w, again[:, :M, :N], K, ks, ext=(g, at) if ext else None, tile_cfg=cfg)
    assert torch.equal(again[:, :M, :N], parts)
    with pytest.raises(ops.MrblipError):
        ops.gemm_ksplit(a, w, parts, K, ks, ext=(g, at) if ext else None, tile_cfg=2)
    if f32out:     # the plain reduce: residual + parts in part order
        res = torch.randn(M, N, device=dev())
        want = res.clone()
        for s_ in range(nparts):
            want = want + parts[s_]
        out = res.clone()
        ops.sum_parts(parts, out, residual=out)
        assert torch.equal(out, want)


@pytest.mark.parametrize("D,p,pe", [(2048, 0.1, 0.05), (768, 0.0, 0.05), (256, 0.1, 0.0)])
def test_rmsnorm_bwd_adds_k_split_parts_and_masks_the_lora_part(ops, D, p, pe):
    """mrblip_rmsnorm_bwd_parts = mrblip_rmsnorm_bwd(_cast) on dy = part 0 + ... + mask (.) last part, the parts added in part order in
    fp32: bit for bit the launch on a dy summed the same way by torch (the mask from the oracle's restatement of the hash)."""
    torch.manual_seed(33)
    M, nparts = 301, 5
    x = torch.randn(M, D, device=dev()) * 1.5
    w = torch.randn(D, device=dev()) * 0.1 + 1
    parts = torch.randn(nparts, M + 2, D, device=dev())[:, :M]
    add = torch.randn(M, D, device=dev())
    seed = torch.tensor([4242], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 23, p) if p > 0 else None
    edrop = ops.Dropout(seed, 57, pe) if pe > 0 else None
    dy = parts[0].clone()
    for s_ in range(1, nparts - 1):
        dy = dy + parts[s_]
    last = parts[nparts - 1]
    if pe > 0:
        inv_keep = (torch.tensor(1.0) / (torch.tensor(1.0) - torch.tensor(pe, dtype=torch.float32))).item()     # (fp32 arithmetic, as the C side)
        last = torch.where(keep_mask((M, D), 4242, 57, pe) > 0, last * inv_keep, torch.zeros_like(last))
    dy = dy + last
    dx_ref, dx = torch.empty_like(x), torch.empty_like(x)
    want = torch.zeros(M, D, dtype=torch.bfloat16, device=dev())
    got = torch.zeros(M, D, dtype=torch.bfloat16, device=dev())
    ops.rmsnorm_bwd(dy, x, w, 1e-6, dx_ref, dx_add=add, out_bf16=want, out_drop=drop)
    ops.rmsnorm_bwd(parts, x, w, 1e-6, dx, dx_add=add, out_bf16=got, out_drop=drop, ext_drop=edrop, ext_part=True)
    assert torch.equal(dx, dx_ref) and torch.equal(got, want)
    ops.rmsnorm_bwd(parts, x, w, 1e-6, dx, dx_add=add, ext_drop=edrop, ext_part=True)          # without the bf16 operand
    assert torch.equal(dx, dx_ref)
    # no masked part: a plain sum
    ops.rmsnorm_bwd(parts[:1], x, w, 1e-6, dx, dx_add=add)
    ops.rmsnorm_bwd(parts[0], x, w, 1e-6, dx_ref, dx_add=add)
    assert torch.equal(dx, dx_ref)


@pytest.mark.parametrize("D,p,nparts", [(2048, 0.1, 5), (768, 0.0, 1), (64, 0.1, 2)])
def test_rmsnorm_bwd_also_writes_the_lora_g_product_of_its_operand(ops, D, p, nparts):
    """mrblip_rmsnorm_bwd_parts_g (round 6): dx and the bf16 operand bit for bit as mrblip_rmsnorm_bwd_parts writes them, plus
    g_out[:, :8] = operand @ g_b^T — against the fp32 product of the same bf16 operands (one bf16 rounding of the result + summation order) and
    against the lora_rows launch it replaces."""
    torch.manual_seed(35)
    M = 2012 if D == 2048 else 301
    x = torch.randn(M, D, device=dev()) * 1.5
    w = torch.randn(D, device=dev()) * 0.1 + 1
    parts = torch.randn(nparts, M + 2, D, device=dev())[:, :M]
    add = torch.randn(M, D, device=dev())
    seed = torch.tensor([991], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 23, p) if p > 0 else None
    edrop = ops.Dropout(seed, 57, 0.05)
    gb = bf(torch.randn(8, D + 64, device=dev()) * 0.05)[:, :D]
    dx_ref, dx = torch.empty_like(x), torch.empty_like(x)
    want = torch.zeros(M, D, dtype=torch.bfloat16, device=dev())
    got = torch.zeros(M, D, dtype=torch.bfloat16, device=dev())
    gout = torch.full((M + 1, 64), 3.0, dtype=torch.bfloat16, device=dev())
    dyp = parts if nparts > 1 else parts[0]
    kw = dict(ext_drop=edrop, ext_part=True) if nparts > 1 else {}
    ops.rmsnorm_bwd(dyp, x, w, 1e-6, dx_ref, dx_add=add, out_bf16=want, out_drop=drop, **kw)
    ops.rmsnorm_bwd(dyp, x, w, 1e-6, dx, dx_add=add, out_bf16=got, out_drop=drop, g_prod=(gb, gout[:M]), **kw)
    assert torch.equal(dx, dx_ref) and torch.equal(got, want)
    ref = want.float() @ gb.float().t()
    from util import check
    check(f"rmsnorm_bwd g product D={D} vs fp32 matmul of the bf16 operands", rel(gout[:M, :8].float(), ref), 4e-3)
    assert torch.all(gout[:M, 8:] == 3.0) and torch.all(gout[M:] == 3.0)       # only the 8 product columns of the M rows are written
    if D % 32 == 0:
        thin = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
        ops.lora_rows(want, gb, thin, D)
        check(f"rmsnorm_bwd g product D={D} vs the lora_rows launch", rel(gout[:M, :8].float(), thin[:, :8].float()), 2.1e-3)   # (bf16 last-bit flips of a fp32 sum in another order: measured 5.5e-8 ... 4e-4)


@pytest.mark.parametrize("p,pe", [(0.1, 0.05), (0.0, 0.05), (0.1, 0.0)])
def test_gated_gelu_bwd_adds_the_masked_lora_part(ops, p, pe):
    """mrblip_gated_gelu_bwd_parts: dy + mask (.) dy_ext in fp32 before the gate's derivative — against the one-part launch on the sum
    (which rounds the sum to bf16 first: one bf16 rounding of dy apart)"""
    torch.manual_seed(35)
    M, Nh = 203, 512
    parts = bf(torch.randn(2, M, Nh, device=dev()))
    parts[1] *= 0.05
    h = bf(torch.randn(M, 2 * Nh, device=dev()))
    seed = torch.tensor([99], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 3, p) if p > 0 else None
    edrop = ops.Dropout(seed, 9, pe) if pe > 0 else None
    e = parts[1].float()
    if pe > 0:
        e = torch.where(keep_mask((M, Nh), 99, 9, pe) > 0, e * (1.0 / (1.0 - pe)), torch.zeros_like(e))
    dy = parts[0].float() + e
    # fp32 reference of the gate's backward on the exact sum
    hf = h.float()
    h0, h1 = hf[:, :Nh], hf[:, Nh:]
    gy = dy
    if p > 0:
        gy = torch.where(keep_mask((M, Nh), 99, 3, p) > 0, gy * (1.0 / (1.0 - p)), torch.zeros_like(gy))
    cdf = 0.5 * (1 + torch.erf(h0 / math.sqrt(2)))
    pdf = torch.exp(-0.5 * h0 * h0) / math.sqrt(2 * math.pi)
    ref = torch.cat([gy * h1 * (cdf + h0 * pdf), gy * h0 * cdf], 1)
    dh = torch.empty(M, 2 * Nh, dtype=torch.bfloat16, device=dev())
    ops.gated_gelu_bwd(parts[0], h, dh, drop=drop, dy_ext=parts[1], ext_drop=edrop)
    assert rel(dh.float(), ref) < 4e-3, rel(dh.float(), ref)
    one = torch.empty_like(dh)
    ops.gated_gelu_bwd(bf(dy), h, one, drop=drop)
    assert rel(dh.float(), one.float()) < 4e-3
    assert rel(dh.float(), ref) <= rel(one.float(), ref) * 1.05          # no extra rounding of the sum: at least as close


# ---- fused decoder projection (csrc/decproj.hip) against the two-launch paths it replaces ------------------------------------------------
def _dp_operands(R, N, K, Rk, gated=False, seed=41):
    torch.manual_seed(seed)
    rows = 2 * N if gated else N
    w = bf(torch.randn(rows, K + 64, device=dev()) * 0.03)[:, :K]
    acat = bf(torch.randn(Rk, K, device=dev()) * 0.05)
    wext = torch.zeros(rows, 64, dtype=torch.bfloat16, device=dev())
    wext[:, :Rk] = bf(torch.randn(rows, Rk, device=dev()) * 0.05)
    return w, acat, wext


@pytest.mark.parametrize("R,N,K,Rk", [(8, 2048, 2048, 8), (14, 2048, 5120, 8), (16, 2048, 2048, 16), (3, 96, 64, 24),
                                      (32, 2048, 2048, 8), (40, 2048, 5120, 8), (72, 2048, 2048, 8), (80, 2048, 5120, 16)])   # (> 16 rows: 3 / 5 row tiles)
def test_dec_proj_plain_input_residual_out(ops, R, N, K, Rk):
    """o / co / wo of a decoder layer: out = residual + dropout(x W^T + u B^T), u = dropout_lora(x) A^T — one launch vs lora_rows + gemm"""
    from util import check
    w, acat, wext = _dp_operands(R, N, K, Rk)
    x = bf(torch.randn(R, K, device=dev()))
    res = torch.randn(R, N, device=dev())
    seed = torch.tensor([99], dtype=torch.int32, device=dev())
    for ldrop, odrop in ((None, None), (ops.Dropout(seed, 4, 0.05), ops.Dropout(seed, 8, 0.1))):
        u0 = torch.zeros(R, 64, dtype=torch.bfloat16, device=dev())
        ref = torch.empty(R, N, device=dev())
        ops.lora_rows(x, acat, u0, K, drop=ldrop)
        ops.gemm(x, w, ref, aext=u0, wext=wext, residual=res, drop=odrop)
        u1 = torch.zeros_like(u0)
        out = torch.full((R, N), 7.0, device=dev())
        ops.dec_proj(x, w, acat, wext, u1, out, K, residual=res, in_drop=ldrop, out_drop=odrop)
        tag = "dec_proj R=%d N=%d K=%d Rk=%d %s: " % (R, N, K, Rk, "drop" if ldrop else "plain")
        check(tag + "u vs lora_rows", rel(u1.float(), u0.float()), 1.2e-6)   # (measured 0 ... 2.4e-7: the same products, one more fp32 summation order)
        assert u1[:, Rk:].abs().sum() == 0
        check(tag + "out vs lora_rows + gemm", rel(out, ref), 2e-6)    # (fp32 output: measured 2e-8 ... 4.1e-7)
        if odrop is not None:   # the same output mask: dropped elements are exactly the residual
            assert torch.equal((out == res), (ref == res))


@pytest.mark.parametrize("R,N,Rk", [(8, 6144, 24), (14, 2048, 8), (16, 2048, 8)])
def test_dec_proj_fused_rmsnorm_bf16_out(ops, R, N, Rk):
    """q/k/v and the cross-attention q: RMSNorm -> (saved) bf16 rows -> projection, one launch vs rmsnorm_lora_fwd + gemm"""
    from util import check
    K = 2048
    w, acat, wext = _dp_operands(R, N, K, Rk, seed=43)
    x32 = torch.randn(R, K, device=dev()) * 1.7
    gamma = torch.randn(K, device=dev()) * 0.1 + 1
    seed = torch.tensor([5], dtype=torch.int32, device=dev())
    for ldrop in (None, ops.Dropout(seed, 3, 0.05)):
        xn0 = torch.zeros(R, K, dtype=torch.bfloat16, device=dev())
        u0 = torch.zeros(R, 64, dtype=torch.bfloat16, device=dev())
        ref = torch.empty(R, N, dtype=torch.bfloat16, device=dev())
        ops.rmsnorm_lora_fwd(x32, gamma, 1e-6, xn0, acat, u0, drop=ldrop)
        ops.gemm(xn0, w, ref, aext=u0, wext=wext)
        xn1, u1, out = torch.zeros_like(xn0), torch.zeros_like(u0), torch.zeros_like(ref)
        ops.dec_proj(xn1, w, acat, wext, u1, out, K, x32=x32, gamma=gamma, eps=1e-6, in_drop=ldrop)
        tag = "dec_proj norm R=%d N=%d Rk=%d %s: " % (R, N, Rk, "drop" if ldrop else "plain")
        assert torch.equal(xn1, xn0), tag + "xn vs rmsnorm_lora_fwd: the same bits"
        check(tag + "xn vs rmsnorm_lora_fwd", rel(xn1.float(), xn0.float()), 2e-7)
        check(tag + "u", rel(u1.float(), u0.float()), 2e-7)               # (measured 0: the same bits)
        check(tag + "out", rel(out.float(), ref.float()), 1.6e-4)       # (bf16 output; measured 3.4e-6 ... 3.3e-5: a few last-bit flips)
        if N % 2048 == 0:   # head-transposed copies of the 2048-wide column ranges (32 heads x 64): what head_transpose writes from the output
            nj = N // 2048
            # (stale contents of another layout in the tiles: the kernel must write the pad columns too — capacity-based workspaces)
            touts = [torch.full((1, 32, 64, 32), 7.5, dtype=torch.bfloat16, device=dev()) for _ in range(nj)]
            out_t = torch.zeros_like(ref)
            ops.dec_proj(xn1, w, acat, wext, u1, out_t, K, x32=x32, gamma=gamma, eps=1e-6, in_drop=ldrop, tout=touts, t_rows=R)
            assert torch.equal(out_t, out)
            for j in range(nj):
                want = ops.head_transpose(out_t[:, j * 2048:(j + 1) * 2048].unflatten(1, (32, 64)).unsqueeze(0))
                assert torch.equal(touts[j], want), j


def test_dec_proj_head_transposed_copies_for_a_batch_of_clips(ops):
    """4 clips x 8 label rows = 32 rows (3 row tiles): the q | k | v copies must land at [clip, head, d, position] exactly as head_transpose
    of the [4, 8, 32, 64] views of the output would write them"""
    B, Ld, N, K, Rk = 4, 8, 6144, 2048, 24
    R = B * Ld
    w, acat, wext = _dp_operands(R, N, K, Rk, seed=61)
    x = bf(torch.randn(R, K, device=dev()))
    u = torch.zeros(R, 64, dtype=torch.bfloat16, device=dev())
    out = torch.zeros(R, N, dtype=torch.bfloat16, device=dev())
    touts = [torch.full((B, 32, 64, 32), float("nan"), dtype=torch.bfloat16, device=dev()) for _ in range(3)]   # stale tiles of another layout
    ops.dec_proj(x, w, acat, wext, u, out, K, tout=touts, t_rows=Ld)
    u0, ref = torch.zeros_like(u), torch.zeros_like(out)
    ops.lora_rows(x, acat, u0, K)
    ops.gemm(x, w, ref, aext=u0, wext=wext)
    assert rel(out.float(), ref.float()) < 4e-3
    for j in range(3):
        want = ops.head_transpose(out[:, j * 2048:(j + 1) * 2048].unflatten(1, (32, 64)).unflatten(0, (B, Ld)))
        assert torch.equal(touts[j], want), j


def test_dec_proj_head_transposed_copies_when_the_label_length_changes(ops):
    """the head-transposed copies live in capacity-based workspaces (engine.buf hands out VIEWS of one backing store), so a step
    with another label length finds the previous layout's values in the tile.  Ld 40 -> 12 -> 40 crosses a multiple of 32 in both
    directions (t_spad 64 -> 32 -> 64) on the d_kv = 64 path: every call must equal head_transpose of its own output, pads included."""
    N, K, Rk = 2048, 2048, 8
    store = torch.zeros(32 * 64 * 64, dtype=torch.bfloat16, device=dev())      # one backing store, sized for the longest layout
    for Ld in (40, 12, 40):
        w, acat, wext = _dp_operands(Ld, N, K, Rk, seed=70 + Ld)
        x = bf(torch.randn(Ld, K, device=dev()))
        u = torch.zeros(Ld, 64, dtype=torch.bfloat16, device=dev())
        out = torch.zeros(Ld, N, dtype=torch.bfloat16, device=dev())
        spad = ops.rup32(Ld)
        tile = store[: 32 * 64 * spad].view(1, 32, 64, spad)                     # what engine.buf returns for this shape
        ops.dec_proj(x, w, acat, wext, u, out, K, tout=(tile,), t_rows=Ld)
        want = ops.head_transpose(out.unflatten(1, (32, 64)).unsqueeze(0))
        assert torch.equal(tile, want), Ld


@pytest.mark.parametrize("R", [8, 14])
def test_dec_proj_fused_rmsnorm_gated(ops, R):
    """wi_0 / wi_1: y = dropout(gelu(h0) * h1), out2 = [h0 | h1] — one launch vs rmsnorm_lora_fwd + the gated tile GEMM"""
    from util import check
    K, Nh, Rk = 2048, 5120, 16
    w, acat, wext = _dp_operands(R, Nh, K, Rk, gated=True, seed=47)
    x32 = torch.randn(R, K, device=dev())
    gamma = torch.randn(K, device=dev()) * 0.1 + 1
    seed = torch.tensor([17], dtype=torch.int32, device=dev())
    for ldrop, odrop in ((None, None), (ops.Dropout(seed, 3, 0.05), ops.Dropout(seed, 6, 0.1))):
        xn0 = torch.zeros(R, K, dtype=torch.bfloat16, device=dev())
        u0 = torch.zeros(R, 64, dtype=torch.bfloat16, device=dev())
        y0 = torch.zeros(R, Nh, dtype=torch.bfloat16, device=dev())
        h0 = torch.zeros(R, 2 * Nh, dtype=torch.bfloat16, device=dev())
        ops.rmsnorm_lora_fwd(x32, gamma, 1e-6, xn0, acat, u0, drop=ldrop)
        ops.gemm(xn0, w, y0, aext=u0, wext=wext, out2=h0, gated=True, drop=odrop, tile_cfg=2)
        xn1, u1, y1, h1 = torch.zeros_like(xn0), torch.zeros_like(u0), torch.zeros_like(y0), torch.zeros_like(h0)
        ops.dec_proj(xn1, w, acat, wext, u1, y1, K, x32=x32, gamma=gamma, eps=1e-6, out2=h1, gated=True, in_drop=ldrop, out_drop=odrop)
        tag = "dec_proj gated R=%d %s: " % (R, "drop" if ldrop else "plain")
        check(tag + "pre-activations [h0 | h1]", rel(h1.float(), h0.float()), 2.7e-4)   # (bf16; measured 1.7e-5 ... 5.5e-5)
        check(tag + "y", rel(y1.float(), y0.float()), 7e-5)                                   # (measured 1.7e-6 ... 1.45e-5)
        if odrop is not None:
            assert torch.equal(y1 == 0, y0 == 0) or ((y1 == 0) != (y0 == 0)).sum() <= 2   # same mask (up to a value that rounds to 0)


@pytest.mark.parametrize("R,N,K,Rk,f32out", [(8, 2048, 6144, 24, True), (14, 2048, 10240, 16, True), (16, 5120, 2048, 8, False), (8, 2048, 2048, 8, False),
                                             (72, 2048, 6144, 24, True), (33, 2048, 10240, 16, True), (72, 5120, 2048, 8, False)])
def test_dec_proj_backward_form(ops, R, N, K, Rk, f32out):
    """the input gradient of an adapted decoder projection: g = dy (sB), dx = dy W + mask_lora (.) (g (sA)) [+ residual] vs lora_rows + lora_dx"""
    from util import check
    torch.manual_seed(53)
    dy = bf(torch.randn(R, K, device=dev()))
    wt = bf(torch.randn(N, K, device=dev()) * 0.03)
    bblk = bf(torch.randn(Rk, K, device=dev()) * 0.05)
    acatt = torch.zeros(N, 64, dtype=torch.bfloat16, device=dev())
    acatt[:, :Rk] = bf(torch.randn(N, Rk, device=dev()) * 0.05)
    res = torch.randn(R, N, device=dev()) if f32out else None
    seed = torch.tensor([23], dtype=torch.int32, device=dev())
    for ldrop in (None, ops.Dropout(seed, 4, 0.05)):
        g0 = torch.zeros(R, 64, dtype=torch.bfloat16, device=dev())
        dx0 = torch.empty(R, N, dtype=torch.float32 if f32out else torch.bfloat16, device=dev())
        ops.lora_rows(dy, bblk, g0, K)
        ops.lora_dx(dy, wt, g0, acatt, dx0, K, residual=res, drop=ldrop)
        g1, dx1 = torch.zeros_like(g0), torch.zeros_like(dx0)
        ops.dec_proj(dy, wt, bblk, acatt, g1, dx1, K, residual=res, ext_drop=ldrop)
        tag = "dec_proj bwd R=%d N=%d K=%d Rk=%d %s: " % (R, N, K, Rk, "mask" if ldrop else "plain")
        check(tag + "g vs lora_rows", rel(g1.float(), g0.float()), 2e-7)     # (measured 0: the same bits)
        check(tag + "dx vs lora_rows + lora_dx", rel(dx1.float(), dx0.float()), 2.6e-4 if not f32out else 2.6e-6)   # (measured: bf16 out <= 5.3e-5, fp32 out <= 5.1e-7)


@pytest.mark.parametrize("B,S,N,K,ext,bias,cfg", [(1, 2012, 6144, 2048, True, False, 0), (1, 333, 2048, 512, False, False, 0), (60, 32, 2304, 768, False, True, 0),
                                                  (1, 2012, 2048, 2048, True, False, 4), (1, 40, 2048, 256, False, False, 2), (3, 64, 4096, 128, True, False, 0)])
def test_gemm_writes_head_transposed_copies(ops, B, S, N, K, ext, bias, cfg):
    """Round 4: the tile GEMM's epilogue also writes the head-transposed copies of its bf16 output (q | k | v ranges of a fused projection)
    that the attention kernels read — bit-identical to mrblip_head_transpose of the output, pad columns included (the tiles start out
    holding another layout's values), for one clip of any length and for batches whose clips are multiples of 32 rows (the Q-Former's 32
    query tokens per frame); the row-major output itself is unchanged."""
    torch.manual_seed(33)
    M = B * S
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(N, K, device=dev()) * 0.05)
    aext = bf(torch.randn(M, 64, device=dev())) if ext else None
    wext = bf(torch.randn(N, 64, device=dev()) * 0.05) if ext else None
    bvec = torch.randn(N, device=dev()) if bias else None
    ref = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm(a, w, ref, aext=aext, wext=wext, bias=bvec, tile_cfg=cfg)
    H = 2048 // 64 if N % 2048 == 0 else 768 // 64
    inner = H * 64
    nj = min(3, N // inner)
    spad = ops.rup32(S)
    guard = torch.full((nj, B + 2, H, 64, spad), 7.5, dtype=torch.bfloat16, device=dev())     # a clip of guard space on either side of every tile
    touts = [guard[j, 1:B + 1] for j in range(nj)]
    out = torch.empty_like(ref)
    ops.gemm(a, w, out, aext=aext, wext=wext, bias=bvec, tile_cfg=cfg, tout=touts, t_rows=S)
    assert torch.equal(out, ref)
    assert bool((guard[:, 0] == 7.5).all()) and bool((guard[:, B + 1] == 7.5).all())               # nothing written outside the tiles
    for j in range(nj):
        want = ops.head_transpose(out[:, j * inner:(j + 1) * inner].unflatten(1, (H, 64)).unflatten(0, (B, S)))
        assert torch.equal(touts[j], want), j
    # the LoRA-backward form (masked K extension first): dO^T of the o-projection's input gradient
    if ext and N == 2048:
        seed = torch.tensor([5], dtype=torch.int32, device=dev())
        d0 = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
        ops.lora_dx(a, w, aext, wext, d0, K, drop=ops.Dropout(seed, 2, 0.05))
        t = torch.full((B, H, 64, spad), -3.0, dtype=torch.bfloat16, device=dev())
        d1 = torch.empty_like(d0)
        ops.lora_dx(a, w, aext, wext, d1, K, drop=ops.Dropout(seed, 2, 0.05), tout=(t,), t_rows=S)
        assert torch.equal(d1, d0)
        assert torch.equal(t, ops.head_transpose(d1.unflatten(1, (H, 64)).unflatten(0, (B, S))))
    with pytest.raises(ops.MrblipError):       # fp32 output: no transposed copies
        ops.gemm(a, w, torch.empty(M, N, device=dev()), tout=touts, t_rows=S)
    ops.gemm(a, w, out, aext=aext, wext=wext, bias=bvec, tile_cfg=cfg)     # the extras are one-shot: the failed call consumed them
    assert torch.equal(out, ref)
    with pytest.raises(ops.MrblipError):       # ... also when the call is rejected before a kernel form is chosen (K not a multiple of 64)
        ops.gemm(a, w, out, K=K - 8, tout=touts, t_rows=S)
    before = guard.clone()
    ops.gemm(a, w, out, aext=aext, wext=wext, bias=bvec, tile_cfg=cfg)
    assert torch.equal(out, ref) and torch.equal(guard, before)


def test_gemm_grouped_k_extension(ops):
    """Round 4: one GEMM for several LoRA groups that share the input — output-column group g takes ITS 64-column slot of Aext as the K
    extension (the cross-attention K / V projections of all decoder layers on one encoder output): bit-identical to one GEMM per group."""
    torch.manual_seed(35)
    M, K, G, Ng = 2012, 2048, 5, 4096
    a = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(G * Ng, K, device=dev()) * 0.03)
    u = bf(torch.randn(M, G * 64, device=dev()))
    wext = bf(torch.randn(G * Ng, 64, device=dev()) * 0.05)
    out = torch.empty(M, G * Ng, dtype=torch.bfloat16, device=dev())
    H, spad = 32, ops.rup32(M)
    ops.gemm(a, w, out, aext=u, wext=wext, ext_group_n=Ng)
    for g in range(G):
        ref = torch.empty(M, Ng, dtype=torch.bfloat16, device=dev())
        ops.gemm(a, w[g * Ng:(g + 1) * Ng], ref, aext=u[:, g * 64:(g + 1) * 64], wext=wext[g * Ng:(g + 1) * Ng])
        assert torch.equal(out[:, g * Ng:(g + 1) * Ng], ref), g
    # with the transposed copies of the first three 2048-wide ranges (K, V of group 0, K of group 1)
    touts = [torch.zeros(1, H, 64, spad, dtype=torch.bfloat16, device=dev()) for _ in range(3)]
    out2 = torch.empty_like(out)
    ops.gemm(a, w, out2, aext=u, wext=wext, ext_group_n=Ng, tout=touts, t_rows=M)
    assert torch.equal(out2, out)
    for j in range(3):
        assert torch.equal(touts[j], ops.head_transpose(out[:, j * 2048:(j + 1) * 2048].unflatten(1, (H, 64)).unsqueeze(0)))
    # ... and of ALL ranges into one buffer [ranges, B, H, 64, Spad]
    tall = torch.full((2 * G, 1, H, 64, spad), 9.0, dtype=torch.bfloat16, device=dev())
    ops.gemm(a, w, out2, aext=u, wext=wext, ext_group_n=Ng, tout=tall, t_rows=M)
    assert torch.equal(out2, out)
    for j in range(2 * G):
        assert torch.equal(tall[j], ops.head_transpose(out[:, j * 2048:(j + 1) * 2048].unflatten(1, (H, 64)).unsqueeze(0))), j


@pytest.mark.parametrize("cfg,gated,f32", [(4, False, True), (2, True, False), (8, False, False), (5, False, False), (1, False, True)])
def test_gemm_prefetch_workgroups_change_no_result(ops, cfg, gated, f32):
    """Round 4: a tile GEMM may start extra workgroups that stream a later launch's weights through the memory-side cache
    (mrblip_gemm_set_prefetch).  They take the first block ids of the grid; the tiles behind them must produce the same bits, for
    one-tile-per-block and persistent forms, any number of prefetch blocks (rounded up to 8), odd byte counts, and the hint must be
    one-shot.  The stand-alone prefetch launch only reads."""
    torch.manual_seed(36)
    M, K, N = 1000, 512, 1536
    rows = 2 * N if gated else N
    a = bf(torch.randn(M, K, device=dev())); w = bf(torch.randn(rows, K, device=dev()) * 0.05)
    u = bf(torch.randn(M, 64, device=dev())); wext = bf(torch.randn(rows, 64, device=dev()) * 0.05)
    res = torch.randn(M, N, device=dev()) if f32 else None
    far = bf(torch.randn(3 * 2**20 + 24, device=dev()))      # the "later launch's weights"
    far0 = far.clone()

    def run(pf=None):
        out = torch.full((M, N), 7.0, dtype=torch.float32 if f32 else torch.bfloat16, device=dev())
        h = torch.zeros(M, 2 * N, dtype=torch.bfloat16, device=dev()) if gated else None
        if pf is not None:
            ops.gemm_prefetch(far, n_blocks=pf[0], nbytes=pf[1], t2=far[4096:4096 + 100 * 1024] if pf[0] == 64 else None)   # (a second range once)
        ops.gemm(a, w, out, aext=u, wext=wext, residual=res, out2=h, gated=gated, tile_cfg=cfg)
        return out, h
    ref, href = run()
    for pf in ((8, None), (3, None), (64, 2**20 + 16), (1024, None), (16, 32), (16, 0)):
        out, h = run(pf)
        assert torch.equal(out, ref), pf
        assert href is None or torch.equal(h, href), pf
    out, _ = run()                      # the hint above was consumed by its launch
    assert torch.equal(out, ref)
    ops.prefetch(far, 16)
    torch.cuda.synchronize()
    assert torch.equal(far, far0)
    with pytest.raises(ops.MrblipError):
        ops.gemm_prefetch(far, n_blocks=2000)


@pytest.mark.parametrize("M,N,K,nad,cfg,gated,f32", [(2012, 2048, 2048, 1, 0, False, True), (2012, 5120, 2048, 2, 0, True, False),
                                                   (2012, 6144, 2048, 3, 8, False, False), (2012, 2048, 5120, 1, 0, False, True),
                                                   (1000, 1536, 512, 3, 1, False, False), (600, 512, 256, 1, 5, False, False),
                                                   (4100, 6144, 512, 3, 8, False, False), (8048, 2048, 512, 1, 0, False, True)])
def test_gemm_thin_role_equals_the_lora_rows_launch(ops, M, N, K, nad, cfg, gated, f32):
    """(the last two shapes: more tiles than workgroup slots — a persistent 16-wave grid and a 64x128 grid of several rounds: roles in FRONT)
    Round 4: the GEMM that consumes the LoRA "down" product u = dropout(x) (sA)^T as its K extension computes it in its own first
    workgroups (mrblip_gemm_set_thin; body shared with lora_thin_kernel: csrc/lora_thin.h) and hands it to the tiles through
    write-through stores and per-row-block flags.  u and the output must have the bits of lora_rows + gemm — for 4-, 8- and 16-wave
    tiles, one and two r tiles (8 / 16 / 24 adapters' rows), persistent and one-tile-per-block grids, with the head-transposed copies and
    with prefetch workgroups in the same launch; repeated, because a broken hand-over would show as a race."""
    torch.manual_seed(37)
    seed = torch.tensor([41], dtype=torch.int32, device=dev())
    rows = 2 * N if gated else N
    R = 8 * nad
    x = bf(torch.randn(M, K, device=dev())); w = bf(torch.randn(rows, K, device=dev()) * 0.03)
    acat = bf(torch.randn(R, K, device=dev()) * 0.05); wext = bf(torch.randn(rows, 64, device=dev()) * 0.05)
    wext[:, R:] = 0
    res = torch.randn(M, N, device=dev()) if f32 else None
    far = bf(torch.randn(2**20, device=dev()))
    in_drop = ops.Dropout(seed, 11, 0.05)
    out_drop = ops.Dropout(seed, 12, 0.1) if (f32 or gated) else None
    H = N // 64
    t_ok = not f32 and not gated and cfg != 5
    spad = ops.rup32(M)

    def run(fused, pf=False):
        u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
        out = torch.full((M, N), 3.0, dtype=torch.float32 if f32 else torch.bfloat16, device=dev())
        h = torch.zeros(M, 2 * N, dtype=torch.bfloat16, device=dev()) if gated else None
        tt = torch.zeros(1, H, 64, spad, dtype=torch.bfloat16, device=dev()) if t_ok else None
        if not fused:
            ops.lora_rows(x, acat, u, K, drop=in_drop)
        if pf:
            ops.gemm_prefetch(far, n_blocks=24)
        ops.gemm(x, w, out, aext=u, wext=wext, residual=res, out2=h, gated=gated, drop=out_drop, tile_cfg=cfg, tout=(tt,) if t_ok else None, t_rows=M,
                 thin=(acat, K, in_drop) if fused else None)
        return u, out, h, tt
    ref = run(False)
    for rep in range(6):
        got = run(True, pf=bool(rep & 1))
        for name, a, b in zip(("u", "out", "h", "tout"), got, ref):
            assert (a is None and b is None) or torch.equal(a, b), (name, rep)
    # no mask (eval): same
    in_drop = None
    ref = run(False)
    got = run(True)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert ops.gemm_thin_timeouts() == 0
    with pytest.raises(ops.MrblipError):     # <= 64 rows take the skinny kernel, which has no such role
        ops.gemm(x[:32], w, torch.empty(32, N, dtype=torch.bfloat16, device=dev()), aext=torch.zeros(32, 64, dtype=torch.bfloat16, device=dev()), wext=wext,
                 thin=(acat, K, None)) if not gated else (_ for _ in ()).throw(ops.MrblipError("n/a"))


def test_gemm_thin_role_ticket_mode_in_a_fresh_process():
    """Round 5: MRB_GEMM_THIN_TICKET=1 hands the roles of a thin-role launch out by TICKET (the order in which the workgroups start to
    run) instead of by block id — the provably live form for a GPU that several processes share (csrc/gemm.hip).  The switch is read once
    per process: the equality test above runs again in a fresh interpreter with it set (same bits as the lora_rows launch, counters back
    at zero after every launch or the repetitions would hang)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MRB_GEMM_THIN_TICKET="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "test_gemm_thin_role_equals_the_lora_rows_launch"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "8 passed" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("grid", [1, 5, 64, 0])
def test_dec_proj_streaming_kernel_is_bit_identical_to_the_tile_kernel(ops, grid):
    """Round 4: for <= 16 rows mrblip_dec_proj runs as a streaming kernel (a block owns a range of 16-column tiles; rows and LoRA "down"
    product once per block).  Same K split over the waves and same summation order as the one-tile-per-block kernel of round 3: every
    output must be bit-identical — forward forms (RMSNorm-fed bf16 with head-transposed copies, gated, plain-input residual, long K) and
    the backward form, with and without dropout, for 1 / 5 / 64 blocks and one block per CU (ragged tile ranges included)."""
    torch.manual_seed(71)
    seed = torch.tensor([31], dtype=torch.int32, device=dev())

    def both(fn):
        outs = []
        for ver in (0, 1):
            ops.dec_proj_config(grid if ver else -1, ver)
            outs.append(fn())
        ops.dec_proj_config(0, 1)
        for a, b in zip(*outs):
            assert torch.equal(a, b)

    for R, N, K, Rk in ((8, 6144, 2048, 24), (13, 2048, 2048, 8), (16, 48, 64, 8)):     # RMSNorm-fed, bf16 out (+ head-transposed copies)
        w, acat, wext = _dp_operands(R, N, K, Rk, seed=43)
        x32, gamma = torch.randn(R, K, device=dev()) * 1.7, torch.randn(K, device=dev()) * 0.1 + 1
        for ldrop in (None, ops.Dropout(seed, 3, 0.05)):
            def run():
                xn, u, out = (torch.full((R, K), 3.0, dtype=torch.bfloat16, device=dev()), torch.zeros(R, 64, dtype=torch.bfloat16, device=dev()),
                              torch.full((R, N), 5.0, dtype=torch.bfloat16, device=dev()))
                touts = [torch.full((1, 32, 64, 32), 7.5, dtype=torch.bfloat16, device=dev()) for _ in range(N // 2048)] if N % 2048 == 0 else None
                ops.dec_proj(xn, w, acat, wext, u, out, K, x32=x32, gamma=gamma, eps=1e-6, in_drop=ldrop, tout=touts, t_rows=R)
                return [xn, u, out] + (touts or [])
            both(run)
    for R in (8, 14):                                                                      # gated (wi_0 / wi_1)
        K, Nh, Rk = 2048, 5120, 16
        w, acat, wext = _dp_operands(R, Nh, K, Rk, gated=True, seed=47)
        x32, gamma = torch.randn(R, K, device=dev()), torch.randn(K, device=dev()) * 0.1 + 1
        for ldrop, odrop in ((None, None), (ops.Dropout(seed, 3, 0.05), ops.Dropout(seed, 6, 0.1))):
            def run():
                xn, u = torch.zeros(R, K, dtype=torch.bfloat16, device=dev()), torch.zeros(R, 64, dtype=torch.bfloat16, device=dev())
                y, h = torch.zeros(R, Nh, dtype=torch.bfloat16, device=dev()), torch.zeros(R, 2 * Nh, dtype=torch.bfloat16, device=dev())
                ops.dec_proj(xn, w, acat, wext, u, y, K, x32=x32, gamma=gamma, eps=1e-6, out2=h, gated=True, in_drop=ldrop, out_drop=odrop)
                return [xn, u, y, h]
            both(run)
    for R, N, K, Rk in ((8, 2048, 2048, 8), (12, 2048, 5120, 8), (16, 80, 96, 8)):        # plain bf16 input rows, fp32 residual out (o / co / wo)
        w, acat, wext = _dp_operands(R, N, K, Rk)
        x, res = bf(torch.randn(R, K, device=dev())), torch.randn(R, N, device=dev())
        for ldrop, odrop in ((None, None), (ops.Dropout(seed, 4, 0.05), ops.Dropout(seed, 8, 0.1))):
            def run():
                u, out = torch.zeros(R, 64, dtype=torch.bfloat16, device=dev()), torch.full((R, N), 7.0, device=dev())
                ops.dec_proj(x, w, acat, wext, u, out, K, residual=res, in_drop=ldrop, out_drop=odrop)
                return [u, out]
            both(run)
    for R, N, K, Rk, f32out in ((8, 2048, 6144, 24, True), (14, 2048, 10240, 16, True), (16, 5120, 2048, 8, False), (8, 2048, 2048, 8, False)):   # backward form
        dy, wt = bf(torch.randn(R, K, device=dev())), bf(torch.randn(N, K, device=dev()) * 0.03)
        bblk = bf(torch.randn(Rk, K, device=dev()) * 0.05)
        acatt = torch.zeros(N, 64, dtype=torch.bfloat16, device=dev())
        acatt[:, :Rk] = bf(torch.randn(N, Rk, device=dev()) * 0.05)
        res = torch.randn(R, N, device=dev()) if f32out else None
        for ldrop in (None, ops.Dropout(seed, 4, 0.05)):
            def run():
                g, dx = torch.zeros(R, 64, dtype=torch.bfloat16, device=dev()), torch.zeros(R, N, dtype=torch.float32 if f32out else torch.bfloat16, device=dev())
                touts = [torch.full((1, 32, 64, 32), 7.5, dtype=torch.bfloat16, device=dev())] if (not f32out and N == 2048) else None
                ops.dec_proj(dy, wt, bblk, acatt, g, dx, K, residual=res, ext_drop=ldrop, tout=touts, t_rows=R)
                return [g, dx] + (touts or [])
            both(run)


# ------------------------------------------------------------------------------------------------ IEEE fp16 operands (the fp16-operand ViT, round 4)
def test_fp16_operand_gemm_four_wave_kernel(ops):
    """mrblip_gemm_f16: the 4-wave 256x256 kernel on v_mfma_f32_32x32x16_f16 — every epilogue the frozen ViT uses (bias -> fp16, bias + GELU
    -> fp16, fp32 out + bias, fp32 residual in place) against fp32 torch on the SAME fp16-rounded operands, and the precision argument:
    on fp32 data the fp16-operand product is ~8x closer to the fp32 product than the bf16-operand one (3 more mantissa bits)."""
    torch.manual_seed(5)
    M, N, K = 700, 520, 320  # 3 x 3 tiles with ragged edges, 5 K-tiles
    a32, w32 = torch.randn(M, K, device=dev()), torch.randn(N, K, device=dev()) * 0.1
    a, w = a32.half(), w32.half()
    bias = torch.randn(N, device=dev())
    res = torch.randn(M, N, device=dev())
    base = a.float() @ w.float().t() + bias
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev())
    ops.gemm(a, w, out, bias=bias)
    assert rel(out.float(), base) < 4e-4            # fp16 output rounding only (2^-11 / sqrt(3))
    ops.gemm(a, w, out, bias=bias, act=1)
    assert rel(out.float(), torch.nn.functional.gelu(base)) < 6e-4
    x = torch.empty(M, N, device=dev())
    ops.gemm(a, w, x, bias=bias)
    assert rel(x, base) < 2e-6
    x = res.clone()
    ops.gemm(a, w, x, bias=bias, residual=x)
    assert rel(x, res + base) < 2e-6
    # fp32 data: operand rounding error of the product, fp16 vs bf16
    full = a32 @ w32.t()
    y16, yb = torch.empty(M, N, device=dev()), torch.empty(M, N, device=dev())
    ops.gemm(a, w, y16)
    ops.gemm(bf(a32), bf(w32), yb, tile_cfg=13)
    e16, eb = rel(y16, full), rel(yb, full)
    from util import record
    record("gemm 700x520x320: fp16-operand product vs fp32 product", e16, 6e-4)
    record("gemm 700x520x320: bf16-operand product vs fp32 product", eb, 5e-3)
    assert e16 < 6e-4 and eb > 5 * e16
    for bad in (dict(gated=True), dict(tile_cfg=2), dict(out2=torch.empty(M, N, dtype=torch.bfloat16, device=dev()))):
        with pytest.raises(ops.MrblipError):
            ops.gemm(a, w, out, **bad)
    with pytest.raises(ops.MrblipError):       # a 16-bit output takes the operands' format
        ops.gemm(a, w, torch.empty(M, N, dtype=torch.bfloat16, device=dev()))


def test_fp16_layernorm_patchify_and_vit_attention(ops):
    """the other producers / consumers of the fp16-operand ViT: LayerNorm with an fp16 output, fp16 patch rows (fp32 and uint8 frames), and
    the row-major-V attention (head_dim 88, 257 tokens) on fp16 q / k / v against fp32 torch"""
    torch.manual_seed(9)
    M, D = 77, 1408
    x = torch.randn(M, D, device=dev()) * 2 + 0.5
    g, b = torch.randn(D, device=dev()) * 0.1 + 1, torch.randn(D, device=dev()) * 0.1
    want = torch.nn.functional.layer_norm(x, (D,), g, b, 1e-6)
    o16 = torch.empty(M, D, dtype=torch.float16, device=dev())
    ops.layernorm_fwd(x, g, b, 1e-6, out_bf16=o16)
    assert rel(o16.float(), want) < 4e-4 and (o16.float() - want.half().float()).abs().max() <= 2e-3
    F_, IMG, P = 3, 56, 14
    G = IMG // P
    video = torch.randn(F_, 3, IMG, IMG, device=dev())
    out = torch.full((F_ * G * G, 640), 7.0, dtype=torch.float16, device=dev())
    ops.patchify(video, out, P)
    ref = video.reshape(F_, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(F_ * G * G, 3 * P * P)
    assert torch.equal(out[:, :588], ref.half()) and out[:, 588:].abs().max() == 0
    u8 = torch.randint(0, 256, (F_, 3, IMG, IMG), device=dev(), dtype=torch.uint8)
    # the processor's arithmetic (blip_processors.py:63-66) with correctly rounded fp32 divisions: on the CPU (the device's torch.div by a
    # scalar multiplies by the reciprocal, 1 ulp off now and then — invisible in bf16, visible in a handful of fp16 roundings)
    mean = torch.tensor(ops.CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ops.CLIP_STD).view(1, 3, 1, 1)
    norm = ((u8.cpu().float() / 255.0 - mean) / std).to(dev())
    o_u8, o_f = torch.empty_like(out), torch.empty_like(out)
    ops.patchify(u8, o_u8, P)
    ops.patchify(norm.contiguous(), o_f, P)
    assert torch.equal(o_u8, o_f)
    B, H, S, Dh = 2, 3, 257, 88
    qkv = (torch.randn(B * S, 3 * H * Dh, device=dev())).half()
    view = lambda c0: torch.as_strided(qkv, (B, S, H, Dh), (S * 3 * H * Dh, 3 * H * Dh, Dh, 1), c0)  # noqa: E731
    q, k, v = view(0), view(H * Dh), view(2 * H * Dh)
    scale = Dh ** -0.5
    o = torch.full((B, S, H, Dh), float("nan"), dtype=torch.float16, device=dev())
    lse = torch.zeros(B, H, ops.rup32(S), device=dev())
    ops.attention_fwd_rowv(q, k, v, o, lse, scale=scale)
    want = torch.nn.functional.scaled_dot_product_attention(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3),
                                                            v.float().permute(0, 2, 1, 3), scale=scale).permute(0, 2, 1, 3)
    e16 = rel(o.float(), want)
    ob = torch.empty(B, S, H, Dh, dtype=torch.bfloat16, device=dev())
    qb = qkv.bfloat16()
    vb = lambda c0: torch.as_strided(qb, (B, S, H, Dh), (S * 3 * H * Dh, 3 * H * Dh, Dh, 1), c0)  # noqa: E731
    ops.attention_fwd_rowv(vb(0), vb(H * Dh), vb(2 * H * Dh), ob, None, scale=scale)
    from util import record
    record("ViT attention 257 x 88: fp16 operands vs fp32 torch", e16, 8e-4)
    record("ViT attention 257 x 88: bf16 operands vs fp32 torch (same data rounded to bf16)", rel(ob.float(), want), 8e-3)
    assert e16 < 8e-4
    s = (q.float().permute(0, 2, 1, 3) @ k.float().permute(0, 2, 3, 1)) * scale
    assert rel(lse[:, :, :S], torch.logsumexp(s, -1)) < 1e-5


# ---- round 6: the C ABI keeps no process-global state -------------------------------------------------------------------------------
def test_ordered_reductions_on_two_streams_use_their_own_workspaces(ops):
    """mrblip_cross_entropy, mrblip_colsum and mrblip_layernorm_bwd(dgamma) add their partial sums in a fixed order through a CALLER-provided
    workspace (mrblip_set_reduce_workspace; mrblip/ops.py keeps one per stream).  Until round 5 scratch and tickets were library-owned
    __device__ arrays: two streams running these kernels at the same time corrupted each other silently.  Two different problem sets run
    concurrently on two streams, in opposite order, many times — every result must be the serial result, bit for bit."""
    torch.manual_seed(77)
    sets = []
    for k in range(2):
        R, V = 14, 32128
        logits = torch.randn(R, V, device=dev()) * 3
        labels = torch.randint(0, V, (R,), device=dev(), dtype=torch.int32)
        labels[3 + k] = -100
        M, D = 7710, 1408
        x, dy = torch.randn(M, D, device=dev()), torch.randn(M, D, device=dev())
        gamma = torch.randn(D, device=dev()) * 0.1 + 1
        cx = torch.randn(1920, 2048, device=dev())
        sets.append(dict(logits=logits, labels=labels, x=x, dy=dy, gamma=gamma, cx=cx))

    def run(s, order):
        out = dict(loss=torch.zeros(1, device=dev()), dg=torch.zeros(1408, device=dev()), db=torch.zeros(1408, device=dev()),
                   cs=torch.zeros(2048, device=dev()), dx=torch.empty_like(s["x"]))
        jobs = dict(ce=lambda: ops.cross_entropy(s["logits"], s["labels"], 1.0 / 13, out["loss"]),
                    ln=lambda: ops.layernorm_bwd(s["dy"], s["x"], s["gamma"], 1e-5, out["dx"], dgamma=out["dg"], dbeta=out["db"]),
                    cs=lambda: ops.colsum(s["cx"], out["cs"]))
        for name in order:
            jobs[name]()
        return out

    want = [run(sets[0], ("ce", "ln", "cs")), run(sets[1], ("ce", "ln", "cs"))]
    torch.cuda.synchronize()
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    orders = [("ln", "ce", "cs", ), ("cs", "ce", "ln")]
    for rep in range(12):
        got = [None, None]
        for k in range(2):
            with torch.cuda.stream(st[k]):
                got[k] = run(sets[k], orders[(k + rep) % 2])
        torch.cuda.synchronize()
        for k in range(2):
            for name in ("loss", "dg", "db", "cs"):
                assert torch.equal(got[k][name], want[k][name]), (rep, k, name)
    # no workspace registered: the atomic fall-back (arrival order) still gives the sums, to rounding
    assert ops._set_reduce_ws(None, 0) == 0
    loss = torch.zeros(1, device=dev())
    assert ops._ce(sets[0]["logits"].data_ptr(), V, sets[0]["labels"].data_ptr(), 14, V, 1.0 / 13, loss.data_ptr(), None, 0, ops._stream()) == 0
    assert abs(loss.item() - want[0]["loss"].item()) < 1e-5 * abs(want[0]["loss"].item())
    with pytest.raises(ops.MrblipError):
        small = torch.zeros(1024, dtype=torch.uint8, device=dev())
        if ops._set_reduce_ws(small.data_ptr(), small.numel()) != 0:
            raise ops.MrblipError(ops._lib.mrblip_last_error().decode())


def test_one_shot_gemm_extras_do_not_outlive_a_failed_call(ops):
    """a prefetch range / head-transposed copies / thin role set for a GEMM that then fails in Python (bad operand) must not ride on the
    thread's next launch: mrblip_gemm_ksplit refuses to launch with pending one-shots, so it is the detector."""
    a = bf(torch.randn(512, 256, device=dev()))
    w = bf(torch.randn(256, 256, device=dev()))
    parts = torch.zeros(2, 512, 256, device=dev())
    ops.gemm_prefetch(w)
    with pytest.raises(ops.MrblipError):
        ops.gemm(a.float(), w, torch.empty(512, 256, device=dev()))          # fp32 operand: rejected before the launch
    ops.gemm_ksplit(a, w, parts, 256, 2)                                      # would fail with "belong to the generic tile kernel" if the range were still pending
    tout = torch.zeros(1, 4, 64, 512, dtype=torch.bfloat16, device=dev())
    with pytest.raises(ops.MrblipError):
        ops.lora_dx(a.float(), w, a[:, :64], w[:, :64], torch.empty(512, 256, dtype=torch.bfloat16, device=dev()), 256, tout=(tout,), t_rows=512)
    ops.gemm_ksplit(a, w, parts, 256, 2)
    ref = a[:, :128].float() @ w[:, :128].float().t()
    assert rel(parts[0], ref) < 5e-6


# ---- round 6: the gated-GELU form of the hand-pipelined 4-wave kernel -----------------------------------------------------------------
@pytest.mark.parametrize("M,Nh,K,p", [(2012, 5120, 2048, 0.1), (300, 136, 128, 0.1), (517, 1032, 320, 0.0), (2012, 640, 768, 0.1)])
def test_gated_gelu_on_the_four_wave_kernel_equals_the_generic_tile(ops, M, Nh, K, p):
    """T5 wi_0 / wi_1 (modeling_t5.py:323-329) with the LoRA term: ONE plain product [x | u] x [W | B]^T over K + 64 on gemm_w4_kernel<GATED>
    (tile_cfg 13, gated) against the generic tile's gated epilogue with its K extension — the same K order, GELU and dropout hash, so y and
    the pre-activations [h0 | h1] must be the same bits; rows >= M / columns >= Nh of the last tiles must not be written (canary)."""
    torch.manual_seed(61)
    x = bf(torch.randn(M, K, device=dev()))
    w = bf(torch.randn(2 * Nh, K, device=dev()) * 0.05)
    u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev())
    u[:, :16] = bf(torch.randn(M, 16, device=dev()))
    wext = torch.zeros(2 * Nh, 64, dtype=torch.bfloat16, device=dev())
    wext[:, :16] = bf(torch.randn(2 * Nh, 16, device=dev()) * 0.05)
    seed = torch.tensor([1234], dtype=torch.int32, device=dev())
    drop = ops.Dropout(seed, 31, p) if p > 0 else None
    y0 = torch.zeros(M, Nh + 64, dtype=torch.bfloat16, device=dev())[:, :Nh]
    h0 = torch.zeros(M, 2 * Nh, dtype=torch.bfloat16, device=dev())
    ops.gemm(x, w, y0, aext=u, wext=wext, out2=h0, gated=True, drop=drop, tile_cfg=2)
    xu = torch.cat([x, u], 1).contiguous()
    wc = torch.cat([w, wext], 1).contiguous()
    ybuf = torch.full((M + 5, Nh + 64), 7.0, dtype=torch.bfloat16, device=dev())
    hbuf = torch.full((M + 5, 2 * Nh + 8), 7.0, dtype=torch.bfloat16, device=dev())
    y1, h1 = ybuf[:M, :Nh], hbuf[:M, :2 * Nh]
    ops.gemm(xu, wc, y1, out2=h1, gated=True, drop=drop, tile_cfg=13, K=K + 64)
    torch.cuda.synchronize()
    assert torch.equal(h1, h0), rel(h1.float(), h0.float())
    assert torch.equal(y1, y0), rel(y1.float(), y0.float())
    assert torch.all(ybuf[M:] == 7.0) and torch.all(ybuf[:, Nh:] == 7.0) and torch.all(hbuf[M:] == 7.0) and torch.all(hbuf[:, 2 * Nh:] == 7.0)
    if p > 0:
        assert 0.85 < (y1 != 0).float().mean().item() < 0.95
    ref = x.float() @ w.float().t() + u.float() @ wext.float().t()
    from util import check
    check(f"gated 4-wave kernel M={M} Nh={Nh} K={K}: pre-activations vs fp32 torch", rel(h1.float(), ref), 4e-3)
    with pytest.raises(ops.MrblipError):      # no K extension in this kernel: the LoRA term must come concatenated
        ops.gemm(x, w, y1, aext=u, wext=wext, out2=h1, gated=True, tile_cfg=13)
