"""project_in / project_out (flashfftconv/hyena.py): pure-torch GEMM forms of the projections around the Hyena operator; CPU check
of values and gradients against the reference callers' forms (hyenadna_flashfftconv.py:269-270, :286-288)."""
import os, sys
import pytest
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-fft-conv_amd"))


@pytest.mark.parametrize("B", [1, 3, 20])      # 20 > the per-row loop limit: the one-GEMM + copy form
def test_projections_match_the_reference_forms(B):
    from flashfftconv.hyena import project_in, project_out
    torch.manual_seed(B)
    L, D = 40, 16
    W = torch.randn(3 * D, D, dtype=torch.float64, requires_grad=True); u = torch.randn(B, L, D, dtype=torch.float64, requires_grad=True)
    Wo = torch.randn(D, D, dtype=torch.float64, requires_grad=True); bo = torch.randn(D, dtype=torch.float64, requires_grad=True)
    y = torch.randn(B, D, L, dtype=torch.float64, requires_grad=True)
    a, ar = project_in(W, u), W @ u.transpose(-1, -2)
    o, orf = project_out(Wo, bo, y), torch.nn.functional.linear(y.transpose(-1, -2), Wo, bo)
    assert a.shape == (B, 3 * D, L) and a.is_contiguous() and o.shape == (B, L, D) and o.is_contiguous()
    assert torch.allclose(a, ar, atol=1e-12) and torch.allclose(o, orf, atol=1e-12)
    ga, go = torch.randn_like(a), torch.randn_like(o)
    g = torch.autograd.grad([a, o], [W, u, Wo, bo, y], [ga, go])
    gr = torch.autograd.grad([ar, orf], [W, u, Wo, bo, y], [ga, go])
    for x, z in zip(g, gr):
        assert torch.allclose(x, z, atol=1e-10)
    assert torch.allclose(project_out(Wo, None, y), y.transpose(-1, -2) @ Wo.t(), atol=1e-12)


@pytest.mark.parametrize("B", [2, 20])      # 20 > the per-row loop limit (that branch lacked the cast)
def test_projections_take_fp32_master_weights(B):
    """bf16 activations with fp32 weights (mixed-precision training): cast inside, gradients come back in the weights' dtype"""
    from flashfftconv.hyena import project_in, project_out
    torch.manual_seed(0)
    L, D = 24, 8
    W = torch.randn(3 * D, D, requires_grad=True); Wo = torch.randn(D, D, requires_grad=True); bo = torch.randn(D, requires_grad=True)
    u = torch.randn(B, L, D).bfloat16().requires_grad_(True); y = torch.randn(B, D, L).bfloat16().requires_grad_(True)
    a = project_in(W, u); o = project_out(Wo, bo, y)
    assert a.dtype == torch.bfloat16 and o.dtype == torch.bfloat16
    g = torch.autograd.grad([a, o], [W, u, Wo, bo, y], [torch.ones_like(a), torch.ones_like(o)])
    assert [t.dtype for t in g] == [torch.float32, torch.bfloat16, torch.float32, torch.float32, torch.bfloat16]
    ar = (W.bfloat16() @ u.transpose(-1, -2)).float()
    assert ((a.float() - ar).norm() / ar.norm()).item() < 2e-2
