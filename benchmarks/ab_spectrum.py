"""Saved-spectrum forward/backward (ffc_conv_fwd_z / ffc_conv_bwd_z) against the recomputing pair: times and bitwise comparison."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib = _lib.lib(); sp = _lib.stream_ptr; P = _lib.ptr
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
cases = ([tuple(int(x) for x in a.split(',')[:4]) + (a.endswith('g'),) for a in sys.argv[1:]] if len(sys.argv) > 1 else None) or [(65536, 16, 768, 32768, False), (131072, 16, 768, 65536, False), (65536, 8, 256, 32768, True), (32768, 16, 768, 16384, False), (32768, 16, 768, 32768, False), (16384, 8, 1024, 8192, True), (8192, 16, 768, 4096, False), (4096, 16, 768, 2048, False),
         (16384, 5, 111, 8000, True), (32768, 3, 24, 16380, True)]
for (N, B, H, L, gated) in cases:
    torch.manual_seed(0)
    u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda") / 30
    pre = torch.randn_like(u) if gated else None; post = torch.randn_like(u) if gated else None
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
    kf = C._kernel_fft(plan, k)
    ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    ws2 = torch.empty_like(ws)
    z = torch.empty(lib.ffc_spectrum_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    y0, y1 = torch.empty_like(u), torch.empty_like(u)
    yraw = torch.empty_like(u) if gated else None
    outs0 = [torch.empty_like(u) for _ in range(3)]; outs1 = [torch.empty_like(u) for _ in range(3)]
    g = lambda t: P(t) if gated else None
    f0 = lambda: _lib.check(lib.ffc_conv_fwd(plan.handle, P(u), P(kf), P(pre), P(post), P(y0), B, H, L, 0, sp()), "fwd")
    f1 = lambda: _lib.check(lib.ffc_conv_fwd_z(plan.handle, P(u), P(kf), P(pre), P(post), P(y1), P(z), P(yraw), B, H, L, 0, 0, 0, 0, sp()), "fwd_z")
    b0 = lambda: _lib.check(lib.ffc_conv_bwd_gated(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(outs0[0]), g(outs0[1]), g(outs0[2]), P(ws), B, H, L, sp()), "bwd")
    def b1():
        if gated: torch.mul(dout, yraw, out=outs1[2])      # dpostgate from the saved pre-postgate output
        _lib.check(lib.ffc_conv_bwd_z(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(outs1[0]), g(outs1[1]), None, P(ws2), P(z), B, H, L, 0, 0, 0, 0, 0, 0, 0, sp()), "bwd_z")
    f0(); f1(); b0(); b1(); torch.cuda.synchronize()
    dk0 = torch.empty(H, L, device="cuda"); dk1 = torch.empty(H, L, device="cuda")
    _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, P(ws), B, H, L, P(dk0), sp()), "dk")
    _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, P(ws2), B, H, L, P(dk1), sp()), "dk")
    same = [torch.equal(y0, y1), torch.equal(outs0[0], outs1[0]), torch.equal(dk0, dk1)] + ([torch.equal(outs0[1], outs1[1]), torch.equal(outs0[2], outs1[2])] if gated else [])
    t = [1e9] * 4
    for rep in range(3):
        for i, fn in enumerate((f0, f1, b0, b1)): t[i] = min(t[i], ev(fn))
    print(f"fft {N} B{B} H{H} L{L} gated={gated}: fwd {t[0]:.4f} -> fwd_z {t[1]:.4f}   bwd {t[2]:.4f} -> bwd_z {t[3]:.4f}   sum {t[0]+t[2]:.4f} -> {t[1]+t[3]:.4f}   bitwise equal (y, du, dk[, dpre, dpost]): {same}", flush=True)
