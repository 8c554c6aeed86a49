"""torch.library registration of the kernels (mrblip/torch_ops.py) without a GPU: the operators exist under torch.ops.mrblip with the
documented schemas, their fake implementations propagate shapes / dtypes under FakeTensorMode, and a CPU tensor is refused (there is
no CPU fallback behind these operators)."""
import pytest
import torch


def test_operators_are_registered_with_schemas():
    from mrblip import torch_ops

    for name in torch_ops.OPS:
        op = getattr(torch.ops.mrblip, name)
        assert op.default._schema.name == "mrblip::" + name
    s = str(torch.ops.mrblip.gemm_.default._schema)
    assert "Tensor(a2!) out" in s or "Tensor(a!) out" in s, s                     # `out` is declared as mutated
    s = str(torch.ops.mrblip.adamw_.default._schema)
    assert s.count("!") == 3, s                                                   # p, m, v
    assert "Tensor? bias" in str(torch.ops.mrblip.linear.default._schema)


def test_fake_implementations_propagate_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    from mrblip import torch_ops

    with FakeTensorMode():
        x = torch.empty(12, 128, dtype=torch.bfloat16)
        w = torch.empty(256, 128, dtype=torch.bfloat16)
        y = torch.ops.mrblip.linear(x, w, None)
        assert y.shape == (12, 256) and y.dtype == torch.bfloat16
        h = torch.ops.mrblip.rms_norm(torch.empty(12, 128), torch.empty(128), 1e-6)
        assert h.shape == (12, 128) and h.dtype == torch.bfloat16
        h = torch.ops.mrblip.layer_norm(torch.empty(12, 128), torch.empty(128), torch.empty(128), 1e-6)
        assert h.dtype == torch.bfloat16
        q = torch.empty(2, 40, 4, 64, dtype=torch.bfloat16)
        o, lse = torch.ops.mrblip.attention_forward(q, q, q, 1.0, None, None, False)
        assert o.shape == q.shape and lse.shape == (2, 4, 64) and lse.dtype == torch.float32
        loss, dl = torch.ops.mrblip.cross_entropy_forward(torch.empty(12, 1000), torch.empty(12, dtype=torch.int32))
        assert loss.shape == (1,) and dl.shape == (12, 1000) and dl.dtype == torch.bfloat16
        assert torch_ops.attention(q, q, q, 1.0).shape == q.shape


def test_cpu_tensors_are_refused():
    from mrblip import torch_ops  # noqa: F401

    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.mrblip.rms_norm(torch.zeros(4, 64), torch.ones(64), 1e-6)
