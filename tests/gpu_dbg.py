import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv
from oracle.torch_ref import ref_fft_conv
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
for scale in (1.0, 0.02):
  for dtype in (torch.float16, torch.bfloat16):
    for N in (256, 4096):
        torch.manual_seed(0)
        B, H, L = 2, 8, N // 2
        u, pre, post = ((torch.randn(B, H, L, device="cuda").to(dtype) * scale).requires_grad_(True) for _ in range(3))
        k = (torch.randn(H, L, device="cuda") * scale).requires_grad_(True)
        c = [t.detach().clone().requires_grad_(True) for t in (u, k, pre, post)]
        out = FlashFFTConv(N, dtype=dtype).to("cuda")(u, k, pre, post)
        ref = ref_fft_conv(c[0] * c[2], c[1], n=N) * c[3]
        dout = torch.randn_like(out) * scale
        out.backward(dout); ref.backward(dout.clone())
        print(scale, dtype, N, "out %.2e" % rel(out, ref), " ".join("%s %.2e" % (n, rel(a.grad, b.grad)) for n, a, b in zip(("du","dk","dpre","dpost"), (u,k,pre,post), c)),
              "maxabs out", out.abs().max().item(), "allclose", torch.allclose(out, ref, atol=1e-2), flush=True)
