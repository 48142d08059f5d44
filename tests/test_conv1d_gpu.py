"""GPU parity for the short depthwise conv1d (reference tests/test_conv1d.py re-stated)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def make(b, d, l, k, in_dtype, w_dtype, is_bhl):
    from flashfftconv import FlashDepthWiseConv1d
    torch.manual_seed(42)
    pad = k // 2
    ref = nn.Conv1d(d, d, k, groups=d, padding=pad).to("cuda")
    x = torch.randn(b, d, l, device="cuda")
    m = FlashDepthWiseConv1d(d, k, pad, ref.weight.detach(), ref.bias.detach(), is_bhl=is_bhl, device="cuda", dtype=w_dtype)
    return ref, m, x


@pytest.mark.parametrize("in_dtype,w_dtype", [(torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16),
                                              (torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                              (torch.float16, torch.float32), (torch.float16, torch.bfloat16)])
@pytest.mark.parametrize("k", [3, 5, 7])
@pytest.mark.parametrize("b,d,l", [(2, 768, 1024), (4, 1024, 2048), (1, 2048, 8192), (3, 72, 1000)])
@pytest.mark.parametrize("is_bhl", [True, False])
def test_conv1d_fwd(b, d, l, k, in_dtype, w_dtype, is_bhl):
    ref, m, x = make(b, d, l, k, in_dtype, w_dtype, is_bhl)
    with torch.no_grad():
        y_ref = ref(x.to(in_dtype).float())
        xin = x.to(in_dtype)
        y = m(xin if is_bhl else xin.transpose(1, 2).contiguous())
        if not is_bhl:
            y = y.transpose(1, 2)
    assert y.dtype == in_dtype
    assert torch.allclose(y.float(), y_ref, atol=1e-1)             # reference tolerance (test_conv1d.py:53-55)
    tol = {torch.float32: 1e-5, torch.float16: 3e-3, torch.bfloat16: 2e-2}[in_dtype]
    wtol = {torch.float32: 0, torch.float16: 1e-3, torch.bfloat16: 8e-3}[w_dtype]
    assert ((y.float() - y_ref).norm() / y_ref.norm()).item() < tol + wtol


@pytest.mark.parametrize("in_dtype,w_dtype", [(torch.float16, torch.float16), (torch.float16, torch.float32),
                                              (torch.float32, torch.float32), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("k", [3, 7])
@pytest.mark.parametrize("b,d,l", [(2, 768, 1024), (3, 72, 1000)])
@pytest.mark.parametrize("is_bhl", [True, False])
def test_conv1d_bwd(b, d, l, k, in_dtype, w_dtype, is_bhl):
    ref, m, x = make(b, d, l, k, in_dtype, w_dtype, is_bhl)
    xq = x.to(in_dtype).detach().clone()
    xr = xq.float().detach().clone().requires_grad_(True)
    y_ref = ref(xr)
    dout = torch.randn_like(y_ref)
    y_ref.backward(dout)
    xin = (xq if is_bhl else xq.transpose(1, 2)).contiguous().detach().clone().requires_grad_(True)
    y = m(xin)
    y.backward((dout if is_bhl else dout.transpose(1, 2).contiguous()).to(in_dtype))
    dx = xin.grad if is_bhl else xin.grad.transpose(1, 2)
    dw = m.weights.grad if is_bhl else m.weights.grad.transpose(0, 1)
    rel = lambda a, bb: ((a.float() - bb.float()).norm() / bb.float().norm()).item()
    tol = {torch.float32: 1e-4, torch.float16: 4e-3, torch.bfloat16: 3e-2}[in_dtype]
    assert rel(dx, xr.grad) < tol
    assert rel(dw, ref.weight.grad.squeeze(1)) < tol + 2e-3
    assert rel(m.bias.grad, ref.bias.grad) < tol + 2e-3
    assert torch.allclose(dx.float(), xr.grad, atol=1)             # reference tolerance (test_conv1d.py:161-163)
