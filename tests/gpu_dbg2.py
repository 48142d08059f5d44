import sys, os
sys.path.insert(0, "/root/repo/flash-fft-conv_amd"); sys.path.insert(0, "/root/repo")
import torch
from flashfftconv import FlashFFTConv
from oracle.torch_ref import ref_fft_conv
torch.manual_seed(0)
for dtype in (torch.float16, torch.bfloat16):
  for (B, H, N) in ((1, 768, 4096), (2, 16, 4096), (4, 8, 32768), (1, 8, 8192)):
    L = N
    u = (torch.randn(B, H, L, device="cuda") * 0.1).to(dtype); k = torch.randn(H, L, device="cuda") * 0.05
    u[..., L // 2:] = 0; 
    uc, kc = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
    u.requires_grad_(True); k.requires_grad_(True)
    conv = FlashFFTConv(N, dtype=dtype).cuda()
    out = conv(u, k); ref = ref_fft_conv(uc, kc, n=N)
    dout = torch.randn_like(out) * 0.02
    ref.backward(dout.clone()); out.backward(dout)
    g, gr = u.grad.float(), uc.grad.float()
    e = (g - gr).abs()
    print(dtype, B, H, N, "out rel", ((out.float() - ref.float()).norm() / ref.float().norm()).item(),
          "du rel", (e.norm() / gr.norm()).item(), "du zero frac", (g == 0).float().mean().item(),
          "dk rel", ((k.grad - kc.grad).norm() / kc.grad.norm()).item())
    if (e.norm() / gr.norm()) > 0.05:
        bad = (e > 0.05 * gr.abs().max())
        print("  bad per b:", bad.sum((1, 2)).tolist()[:8], "bad per h (first 16):", bad.sum((0, 2)).tolist()[:16])
        pos = bad[0, 0].nonzero().flatten()
        print("  h0 bad positions:", pos[:20].tolist(), "count", pos.numel())
