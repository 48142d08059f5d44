import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
N, B, H = 2048, 4, 16
for L in (1024, 1016, 1000, 2040):
    torch.manual_seed(0)
    u = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda") / 8
    plan = FlashFFTConv(N, dtype=torch.bfloat16).cuda()._get_plan(u.device); kf = C._kernel_fft(plan, k)
    z = torch.zeros(lib.ffc_spectrum_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    ys = []
    for rep in range(2):
        y0, y1 = torch.full_like(u, 7.0), torch.full_like(u, 9.0)
        lib.ffc_conv_fwd(plan.handle, P(u), P(kf), None, None, P(y0), B, H, L, 0, sp())
        lib.ffc_conv_fwd_z(plan.handle, P(u), P(kf), None, None, P(y1), P(z), None, B, H, L, 0, 0, 0, 0, sp())
        torch.cuda.synchronize(); ys.append((y0, y1))
    ref = torch.fft.irfft(torch.fft.rfft(u.float(), n=N) * torch.fft.rfft(k, n=N), n=N)[..., :L]
    y0, y1 = ys[0]
    d = (y0.float() - y1.float()).abs()
    print(L, "equal", torch.equal(y0, y1), "rerun equal", torch.equal(ys[0][0], ys[1][0]), torch.equal(ys[0][1], ys[1][1]),
          "err0", (y0.float() - ref).abs().max().item(), "err1", (y1.float() - ref).abs().max().item(), "ndiff", int((d > 0).sum()),
          "where", d.nonzero()[:4].tolist())
