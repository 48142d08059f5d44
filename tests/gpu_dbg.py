import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
dtype = torch.bfloat16
lib = _lib.lib()
def vary(outs):
    S = torch.stack([o.float() for o in outs]); return int((S.max(0).values != S.min(0).values).sum())
for N, B, H in ((4096, 2, 8), (8192, 2, 3), (16384, 2, 2)):
    torch.manual_seed(0)
    L = N // 2
    u = torch.randn(B, H, L, device="cuda").to(dtype); dout = torch.randn(B, H, L, device="cuda").to(dtype)
    k = torch.randn(H, L, device="cuda") * 0.1
    mod = FlashFFTConv(N, dtype=dtype).to("cuda"); plan = mod._get_plan(u.device)
    kfs = [C._kernel_fft(plan, k).clone() for _ in range(4)]
    print(N, "kfft varying:", vary(kfs))
    ys = [C._conv(plan, u, kfs[0], None, None, False).clone() for _ in range(4)]
    print(N, "conv fwd varying:", vary(ys))
    nb = lib.ffc_dkf_workspace_bytes(plan.handle, B, H)
    wss = []
    for i in range(3):
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        _lib.check(lib.ffc_conv_bwd_dkf(plan.handle, _lib.ptr(dout), _lib.ptr(u), None, None, _lib.ptr(ws), B, H, L, None), "dkf")
        wss.append(ws.view(torch.float32)[: 8 * H * N * 2 if N == 4096 else 4 * H * N * 2].clone())
    print(N, "dkf slabs varying:", vary(wss))
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    _lib.check(lib.ffc_conv_bwd_dkf(plan.handle, _lib.ptr(dout), _lib.ptr(u), None, None, _lib.ptr(ws), B, H, L, None), "dkf")
    for Lk in (L, L - 1):
        outs = []
        for i in range(5):
            dk = torch.zeros(H, Lk, dtype=torch.float32, device="cuda")
            _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, Lk, _lib.ptr(dk), None), "ifft")
            torch.cuda.synchronize(); outs.append(dk)
        print(N, "dk inverse Lk=%d varying:" % Lk, vary(outs))
