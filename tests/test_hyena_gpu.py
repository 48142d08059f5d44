"""FlashHyenaOp (flashfftconv/hyena.py) against the composition the reference's callers run
(examples/hyena-dna/hyenadna_flashfftconv.py:274-284): short depthwise conv -> split -> x1*v -> FFT conv -> *x2.
Oracle: nn.Conv1d (fp32) + the torch.fft oracle with autograd; also the same composition on this package's own unfused modules
(the gated kernel reading the slices in place must agree with it to rounding)."""
import pytest
import torch
import torch.nn as nn

from oracle.torch_ref import ref_fft_conv

pytestmark = pytest.mark.gpu


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,D,L,fft,dtype", [(2, 64, 1024, 2048, torch.bfloat16), (3, 96, 2048, 4096, torch.bfloat16),
                                             (2, 128, 8192, 16384, torch.float16), (4, 256, 16384, 32768, torch.bfloat16),
                                             (2, 32, 32768, 65536, torch.bfloat16), (2, 16, 65536, 131072, torch.bfloat16),
                                             (1, 16, 131072, 262144, torch.bfloat16), (2, 40, 512, 1024, torch.bfloat16),
                                             (2, 24, 1000, 4096, torch.bfloat16),
                                             # an operator built for a longer sequence than it is run on: the fft size fitted to the rows
                                             # (FlashFFTConv._fit_seqlen: 16384 / 65536 points, the slices still read in place)
                                             (2, 16, 8192, 131072, torch.bfloat16), (1, 8, 20000, 1048576, torch.bfloat16)])
def test_hyena_op_matches_reference_composition(B, D, L, fft, dtype):
    from flashfftconv import FlashHyenaOp, FlashFFTConv, FlashDepthWiseConv1d
    torch.manual_seed(11)
    sf = nn.Conv1d(3 * D, 3 * D, 3, padding=1, groups=3 * D).cuda()
    with torch.no_grad():       # the modules hold the short filter in `dtype`
        sf.weight.copy_(sf.weight.to(dtype).float()); sf.bias.copy_(sf.bias.to(dtype).float())
    op = FlashHyenaOp(D, fft, sf.weight.detach(), sf.bias.detach(), dtype=dtype, device="cuda").cuda()
    u = torch.randn(B, 3 * D, L, device="cuda").to(dtype)
    k = torch.randn(D, L, device="cuda") * 0.05
    dy = torch.randn(B, D, L, device="cuda").to(dtype)

    # ---- fused operator
    uf, kf_ = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
    y = op(uf, kf_)
    gy = torch.autograd.grad(y, [uf, kf_, op.short_filter.weights, op.short_filter.bias], dy)

    # ---- fp32 oracle of the reference composition
    uo, ko = u.float().requires_grad_(True), k.clone().requires_grad_(True)
    uc = sf(uo)[..., :L].to(dtype).float() if False else sf(uo)[..., :L]
    x1, x2, v = uc.split(D, dim=1)
    yo = ref_fft_conv((x1 * v).to(dtype), ko, n=fft).float() * x2
    go = torch.autograd.grad(yo, [uo, ko, sf.weight, sf.bias], dy.float())
    f = 2.0 if fft >= 262144 else 1.0
    tol = {torch.bfloat16: 3e-2, torch.float16: 8e-3}[dtype] * f     # conv1d output rounding + the gated conv's gates
    assert rel(y, yo) < tol, f"y {rel(y, yo):.3e}"
    assert rel(gy[0], go[0]) < tol, f"du {rel(gy[0], go[0]):.3e}"
    assert rel(gy[1], go[1]) < tol, f"dk {rel(gy[1], go[1]):.3e}"
    assert rel(gy[2], go[2].squeeze(1)) < tol, f"dw {rel(gy[2], go[2].squeeze(1)):.3e}"
    assert rel(gy[3], go[3]) < tol, f"dbias {rel(gy[3], go[3]):.3e}"

    # ---- the same composition on the unfused modules of this package: agrees to rounding of the x1*v product
    conv = FlashFFTConv(fft, dtype=dtype).cuda()
    short = FlashDepthWiseConv1d(3 * D, 3, 1, sf.weight.detach(), sf.bias.detach(), device="cuda", dtype=dtype)
    uu, ku = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
    ucu = short(uu)[..., :L]
    a1, a2, av = ucu.split(D, dim=1)
    yu = conv((a1 * av).contiguous(), ku) * a2
    gu = torch.autograd.grad(yu, [uu, ku], dy)
    assert rel(y, yu) < tol / 2 and rel(gy[0], gu[0]) < tol / 2 and rel(gy[1], gu[1]) < tol / 2


def test_hyena_op_eval_and_errors():
    from flashfftconv import FlashHyenaOp, gated_conv_from_slices, FlashFFTConv
    D, L = 32, 2048
    w = torch.randn(3 * D, 3, device="cuda"); b = torch.randn(3 * D, device="cuda")
    op = FlashHyenaOp(D, 4096, w, b, dtype=torch.bfloat16, device="cuda").cuda().eval()
    u = torch.randn(2, 3 * D, L, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(D, L, device="cuda") * 0.1
    with torch.no_grad():
        y = op(u, k)
    assert y.shape == (2, D, L) and torch.isfinite(y).all()
    with pytest.raises(RuntimeError):
        gated_conv_from_slices(FlashFFTConv(4096, dtype=torch.bfloat16).cuda(), u[:, :64], k)    # not 3*D channels


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_mixer_equals_the_reference_operator(dtype):
    """FlashHyenaMixer (projections as batched GEMMs on transposed views) == the reference callers' HyenaOperator.forward
    (hyenadna_flashfftconv.py:266-289) written with this package's drop-in modules, forward and gradients"""
    from flashfftconv import FlashHyenaMixer, FlashFFTConv, FlashDepthWiseConv1d
    torch.manual_seed(5)
    B, L, D, fft = 2, 2048, 64, 4096
    inp = torch.nn.Linear(D, 3 * D).cuda().to(dtype); outp = torch.nn.Linear(D, D).cuda().to(dtype)
    sf = torch.nn.Conv1d(3 * D, 3 * D, 3, padding=2, groups=3 * D).cuda()
    k = (torch.randn(D, L, device="cuda") * 0.02).requires_grad_(True)
    u = (torch.randn(B, L, D, device="cuda") * 0.5).to(dtype).requires_grad_(True)
    mixer = FlashHyenaMixer(D, fft, inp, outp, sf.weight.detach(), sf.bias.detach(), dtype=dtype, device="cuda").cuda()
    short = FlashDepthWiseConv1d(3 * D, 3, padding=1, weights=sf.weight.detach(), bias=sf.bias.detach(), dtype=dtype).cuda()
    conv = FlashFFTConv(fft, dtype=dtype).cuda()

    def reference(u, k):
        x = inp.weight @ u.transpose(-1, -2)
        uc = short(x)[..., :L]
        x1, x2, v = uc.split(D, dim=1)
        y = conv((x1 * v).contiguous(), k) * x2
        return outp(y.transpose(-1, -2))
    y = mixer(u, k); yr = reference(u, k)
    assert y.shape == (B, L, D) and y.is_contiguous()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    assert rel(y, yr) < tol, rel(y, yr)
    dy = torch.randn_like(y) * 0.1
    g = torch.autograd.grad(y, [u, k, inp.weight, outp.weight, outp.bias], dy)
    gr = torch.autograd.grad(yr, [u, k, inp.weight, outp.weight, outp.bias], dy)
    for a, b, n in zip(g, gr, ("du", "dk", "d in_proj.weight", "d out_proj.weight", "d out_proj.bias")):
        assert rel(a, b) < 2 * tol, f"{n}: {rel(a, b):.3e}"
