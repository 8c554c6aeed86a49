"""torch.ops.mrblip.* (mrblip/torch_ops.py) on the GPU: forward values and autograd gradients of each operator against the plain
PyTorch fp32 op it stands for, plus torch.library.opcheck (schema / fake-tensor / autograd-registration consistency)."""
import pytest
import torch

from util import check

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def bf(t):
    return t.to(torch.bfloat16)


def test_linear_forward_and_gradients():
    from mrblip import torch_ops  # noqa: F401
    torch.manual_seed(0)
    M, K, N = 200, 256, 320
    x = bf(torch.randn(M, K, device=dev())).requires_grad_(True)
    w = bf(torch.randn(N, K, device=dev()) * 0.05).requires_grad_(True)
    b = (torch.randn(N, device=dev()) * 0.1).requires_grad_(True)
    dy = bf(torch.randn(M, N, device=dev()))
    y = torch.ops.mrblip.linear(x, w, b)
    y.backward(dy)
    xr, wr, br = x.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.linear(xr, wr, br)
    ref.backward(dy.float())
    check("torch_ops.linear: y vs fp32 torch", rel(y, ref), 4e-3)
    check("torch_ops.linear: dx vs fp32 autograd", rel(x.grad, xr.grad), 4e-3)
    check("torch_ops.linear: dw vs fp32 autograd", rel(w.grad, wr.grad), 4e-3)
    check("torch_ops.linear: dbias vs fp32 autograd", rel(b.grad, br.grad), 1e-5)


def test_norms_forward_and_gradients():
    from mrblip import torch_ops  # noqa: F401
    torch.manual_seed(1)
    M, D = 77, 1408
    x = (torch.randn(M, D, device=dev()) * 2 + 0.5).requires_grad_(True)
    g = (torch.randn(D, device=dev()) * 0.1 + 1).requires_grad_(True)
    b = (torch.randn(D, device=dev()) * 0.1).requires_grad_(True)
    dy = torch.randn(M, D, device=dev())
    y = torch.ops.mrblip.layer_norm(x, g, b, 1e-6)
    y.backward(bf(dy))
    xr, gr, br = (t.detach().clone().requires_grad_(True) for t in (x, g, b))
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    ref.backward(bf(dy).float())
    check("torch_ops.layer_norm: y vs fp32 torch", rel(y, ref), 4e-3)
    check("torch_ops.layer_norm: dx vs fp32 autograd", rel(x.grad, xr.grad), 1e-5)
    check("torch_ops.layer_norm: dgamma vs fp32 autograd", rel(g.grad, gr.grad), 1e-5)
    check("torch_ops.layer_norm: dbeta vs fp32 autograd", rel(b.grad, br.grad), 1e-5)
    x2 = x.detach().clone().requires_grad_(True)
    y = torch.ops.mrblip.rms_norm(x2, g.detach(), 1e-6)
    y.backward(bf(dy))
    xr = x.detach().clone().requires_grad_(True)
    ref = g.detach() * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    ref.backward(bf(dy).float())
    check("torch_ops.rms_norm: y vs fp32 torch", rel(y, ref), 4e-3)
    check("torch_ops.rms_norm: dx vs fp32 autograd", rel(x2.grad, xr.grad), 1e-5)
    with pytest.raises(RuntimeError):       # frozen on this path: asking for the weight gradient is an error, not a silent zero
        torch.ops.mrblip.rms_norm(x.detach(), g, 1e-6).sum().backward()


def test_attention_forward_and_gradients():
    from mrblip import ops, torch_ops
    torch.manual_seed(2)
    B, S, H, D = 2, 150, 4, 64
    q, k, v = (bf(torch.randn(B, S, H, D, device=dev()) * 0.5).requires_grad_(True) for _ in range(3))
    do = bf(torch.randn(B, S, H, D, device=dev()))
    lut = torch.randn(H, 257, device=dev())
    kmask = torch.zeros(B, ops.rup32(S), dtype=torch.int32, device=dev())
    kmask[:, :S] = 1
    kmask[1, S - 9:] = 0
    o = torch_ops.attention(q, k, v, 1.0, lut, kmask, False)
    o.backward(do)
    qr, kr, vr = (t.detach().float().permute(0, 2, 1, 3).clone().requires_grad_(True) for t in (q, k, v))
    relpos = (torch.arange(S, device=dev())[None, :] - torch.arange(S, device=dev())[:, None]).clamp(-128, 128) + 128
    s = qr @ kr.transpose(-1, -2) + lut[:, relpos][None]
    s = s.masked_fill(~kmask[:, :S].bool()[:, None, None, :], -1e30)
    ref = torch.softmax(s, -1) @ vr
    ref.backward(do.float().permute(0, 2, 1, 3))
    check("torch_ops.attention: o vs fp32 torch", rel(o.permute(0, 2, 1, 3), ref), 6e-3)
    check("torch_ops.attention: dq vs fp32 autograd", rel(q.grad.permute(0, 2, 1, 3), qr.grad), 1.2e-2)
    check("torch_ops.attention: dk vs fp32 autograd", rel(k.grad.permute(0, 2, 1, 3), kr.grad), 1.2e-2)
    check("torch_ops.attention: dv vs fp32 autograd", rel(v.grad.permute(0, 2, 1, 3), vr.grad), 1.2e-2)


def test_cross_entropy_and_adamw():
    from mrblip import torch_ops
    torch.manual_seed(3)
    R, V = 24, 32128
    logits = (torch.randn(R, V, device=dev()) * 2).requires_grad_(True)
    labels = torch.randint(0, V, (R,), device=dev())
    labels[::5] = -100
    loss = torch_ops.cross_entropy(logits, labels)
    (loss * 3.0).sum().backward()
    lr_ = logits.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr_, labels, ignore_index=-100)
    (ref * 3.0).backward()
    check("torch_ops.cross_entropy: loss vs fp32 torch (rel)", abs(loss.item() - ref.item()) / ref.item(), 1e-6)
    check("torch_ops.cross_entropy: dlogits vs fp32 autograd", rel(logits.grad, lr_.grad), 4e-3)     # the kernel hands dlogits on as bf16
    n = 10007
    p = torch.randn(n, device=dev())
    g = torch.randn(n, device=dev())
    m, v = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    for t in range(1, 4):
        hyper = torch.tensor([3e-4, 1 / (1 - 0.9 ** t), 1 / (1 - 0.999 ** t) ** 0.5, 1.0], device=dev())
        torch.ops.mrblip.adamw_(p, g, m, v, hyper, 0.9, 0.999, 1e-8, 0.05)
        pr.grad = g.clone()
        opt.step()
    check("torch_ops.adamw_: 3 steps vs torch.optim.AdamW", rel(p, pr.detach()), 1e-6)


def test_opcheck():
    from mrblip import torch_ops  # noqa: F401
    torch.manual_seed(4)
    x = bf(torch.randn(64, 128, device=dev())).requires_grad_(True)
    w = bf(torch.randn(192, 128, device=dev()) * 0.1).requires_grad_(True)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.mrblip.linear.default, (x, w, None), test_utils=tests)
    xf = torch.randn(32, 256, device=dev(), requires_grad=True)
    torch.library.opcheck(torch.ops.mrblip.rms_norm.default, (xf, torch.ones(256, device=dev()), 1e-6), test_utils=tests)
    torch.library.opcheck(torch.ops.mrblip.layer_norm.default, (xf, torch.ones(256, device=dev(), requires_grad=True), torch.zeros(256, device=dev(), requires_grad=True), 1e-6), test_utils=tests)
    q = bf(torch.randn(1, 40, 2, 64, device=dev())).requires_grad_(True)
    torch.library.opcheck(torch.ops.mrblip.attention_forward.default, (q, q, q, 1.0, None, None, False), test_utils=tests)
    out = torch.empty(64, 192, dtype=torch.bfloat16, device=dev())
    torch.library.opcheck(torch.ops.mrblip.gemm_.default, (x.detach(), w.detach(), out, None, None, 0, 0), test_utils=("test_schema", "test_faketensor"))
