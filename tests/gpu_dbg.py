import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
dtype = torch.bfloat16
for N, B, H in ((4096, 2, 8), (4096, 2, 1), (8192, 2, 3)):
    torch.manual_seed(0)
    L = N // 2
    u = torch.randn(B, H, L, device="cuda").to(dtype); dout = torch.randn(B, H, L, device="cuda").to(dtype)
    mod = FlashFFTConv(N, dtype=dtype).to("cuda"); plan = mod._get_plan(u.device); lib = _lib.lib()
    nb = lib.ffc_dkf_workspace_bytes(plan.handle, B, H)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ffc_conv_bwd_dkf(plan.handle, _lib.ptr(dout), _lib.ptr(u), None, None, _lib.ptr(ws), B, H, L, None), "dkf")
    ref = torch.fft.ifft((torch.fft.fft(dout.float(), n=N) * torch.fft.fft(u.float(), n=N).conj()).sum(0)).real[:, :L]
    outs = []
    for i in range(6):
        dk = torch.full((H, L), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, L, _lib.ptr(dk), None), "ifft")
        torch.cuda.synchronize(); outs.append(dk)
    S = torch.stack(outs)                      # (6,H,L)
    var = (S.max(0).values != S.min(0).values)  # positions that vary
    print(N, H, "varying positions per head:", var.sum(1).tolist())
    if var.any():
        h = int(var.sum(1).argmax()); idx = var[h].nonzero().flatten()
        print("   head", h, "first varying idx:", idx[:24].tolist(), "... count", len(idx))
        Mi = N // (16 if N == 4096 else 32)
        print("   rows n1 =", sorted(set((idx // Mi).tolist()))[:20], " cols mod 8:", sorted(set((idx % 8).tolist())), " cols%Mi range", int((idx % Mi).min()), int((idx % Mi).max()))
        err = (S - ref).abs()   # which runs are bad where
        print("   mean abs err per run:", ["%.3f" % err[i, h].mean().item() for i in range(6)], "ref mean abs %.2f" % ref[h].abs().mean().item())
