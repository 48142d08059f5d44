"""A/B: the levels read the fp32 filter / write the fp32 dk themselves (BigArgs::lf32, default) against cast kernels around them
(FFC_BIG_LONG_F32=0 form), config 4 (fft 4M, B1 H16 L=1M) and fft 1M at B2 H48: forward / backward ms per call, HIP events."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C


def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / n)
    return sorted(best)[1]


for (N, B, H, L, dt) in ((4194304, 1, 16, 1048576, torch.bfloat16), (1048576, 2, 48, 524288, torch.float16)):
    u = torch.randn(B, H, L, device="cuda").to(dt).requires_grad_(True); dout = torch.randn(B, H, L, device="cuda").to(dt)
    k = (torch.randn(H, L, device="cuda") * 0.05).requires_grad_(True)
    mod = FlashFFTConv(N, dtype=dt).cuda()
    row = {"fft": N, "B": B, "H": H, "L": L, "dtype": str(dt)}
    for flag in (False, True, False, True):
        C._TorchOps.LONG_F32 = flag
        y = mod(u, k)
        f = timeit(lambda: mod(u, k))
        def step():
            y = mod(u, k); torch.autograd.grad(y, (u, k), dout)
        s = timeit(step)
        row.setdefault("lf32" if flag else "cast", []).append([round(f, 4), round(s - f, 4)])
    print(json.dumps(row), flush=True)
