"""A/B of FlashFFTConv.fit_fft (the fft size fitted to the rows, conv.py _fit_seqlen) on one box: the same module, the same tensors,
fit_fft on / off interleaved.  Rows: BASELINE config 4 and two "module built for the longest sequence, called with a shorter one" shapes.
Prints one line per (row, setting): training forward / backward ms (HIP events, median of 3 x iters), inference forward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT, os.path.dirname(os.path.abspath(__file__))]
import torch
from flashfftconv import FlashFFTConv
from sweep import ev_time

ROWS = [("cfg4 FlashFFTConv(4194304) B1 H16 L1048576", 4194304, 1, 16, 1048576, 5),
        ("FlashFFTConv(2097152) B1 H16 L=131072 (HyenaDNA-1M module, 128K prompt)", 2097152, 1, 16, 131072, 10),
        ("FlashFFTConv(131072) B16 H768 L=16384", 131072, 16, 768, 16384, 10)]
for name, N, B, H, L, iters in ROWS:
    u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True)
    k = torch.randn(H, L, device="cuda").requires_grad_(True)
    dout = torch.randn(B, H, L, device="cuda").bfloat16()
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda()
    for rep in range(2):
        for fit in (False, True):
            mod.fit_fft = fit
            mod.train()
            tf, _ = ev_time(lambda: mod(u, k), iters)
            y = mod(u, k)

            def bwd():
                u.grad = None; k.grad = None
                y.backward(dout, retain_graph=True)
            tb, _ = ev_time(bwd, iters)
            del y
            with torch.no_grad():
                mod.eval()
                ti, _ = ev_time(lambda: mod(u, k), iters)
            print(f"{name}: fit_fft={fit!s:5} fft run {mod._fit_seqlen(L, L):8d}: fwd {tf:.4f} ms  bwd {tb:.4f} ms  inference fwd {ti:.4f} ms", flush=True)
    del mod, u, k, dout
    torch.cuda.empty_cache()
