import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
from oracle.torch_ref import ref_fft_conv
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
dtype = torch.bfloat16
for N, B, H in ((4096, 2, 8), (4096, 16, 8), (8192, 2, 3)):
    torch.manual_seed(0)
    L = N // 2
    u = torch.randn(B, H, L, device="cuda").to(dtype)
    k = (torch.randn(H, L, device="cuda") * 0.1)
    dout = torch.randn(B, H, L, device="cuda").to(dtype)
    mod = FlashFFTConv(N, dtype=dtype).to("cuda"); plan = mod._get_plan(u.device); lib = _lib.lib()
    kf = C._kernel_fft(plan, k)
    ref = torch.fft.ifft((torch.fft.fft(dout.float(), n=N) * torch.fft.fft(u.float(), n=N).conj()).sum(0)).real[:, :L]
    def dk_from(ws):
        dk = torch.empty(H, L, dtype=torch.float32, device="cuda")
        _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, L, _lib.ptr(dk), None), "ifft"); return dk
    nb = lib.ffc_dkf_workspace_bytes(plan.handle, B, H)
    ws1 = torch.zeros(nb, dtype=torch.uint8, device="cuda"); ws2 = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ffc_conv_bwd_dkf(plan.handle, _lib.ptr(dout), _lib.ptr(u), None, None, _lib.ptr(ws1), B, H, L, None), "dkf")
    du = torch.empty_like(u)
    _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(du), None, _lib.ptr(ws2), B, H, L, None), "bwd")
    d1, d2 = dk_from(ws1), dk_from(ws2)
    nslab_f = (nb // 4)
    w1 = ws1.view(torch.float32); w2 = ws2.view(torch.float32)
    print(N, B, H, "dk(dkf) %.2e dk(fused) %.2e  slabs equal: %s  maxdiff %.3e" % (rel(d1, ref), rel(d2, ref), torch.equal(w1[:H*N*2], w2[:H*N*2]), (w1[:H*N*2*8]-w2[:H*N*2*8]).abs().max().item()))
    print("   per-head dkf:", ["%.1e" % rel(d1[h], ref[h]) for h in range(H)])
    e = (d1 - ref).abs(); print("   worst positions head0:", e[0].topk(5).indices.tolist(), " ref scale %.2f" % ref.abs().mean().item())
print("--- determinism of the dk inverse (same workspace, repeated)")
for N, B, H in ((4096, 2, 8), (8192, 2, 3), (16384, 2, 3)):
    torch.manual_seed(0)
    L = N // 2
    u = torch.randn(B, H, L, device="cuda").to(dtype); dout = torch.randn(B, H, L, device="cuda").to(dtype)
    mod = FlashFFTConv(N, dtype=dtype).to("cuda"); plan = mod._get_plan(u.device); lib = _lib.lib()
    nb = lib.ffc_dkf_workspace_bytes(plan.handle, B, H)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ffc_conv_bwd_dkf(plan.handle, _lib.ptr(dout), _lib.ptr(u), None, None, _lib.ptr(ws), B, H, L, None), "dkf")
    ref = torch.fft.ifft((torch.fft.fft(dout.float(), n=N) * torch.fft.fft(u.float(), n=N).conj()).sum(0)).real[:, :L]
    outs = []
    for i in range(4):
        dk = torch.full((H, L), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, L, _lib.ptr(dk), None), "ifft")
        torch.cuda.synchronize(); outs.append(dk)
    print(N, [("%.2e" % rel(o, ref)) for o in outs], "identical:", [torch.equal(outs[0], o) for o in outs[1:]], "nan:", [int(torch.isnan(o).sum()) for o in outs])
