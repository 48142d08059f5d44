"""Few-heads shapes (a head-sharded rank of B=16 H=768 on 8 GPUs has 96 heads): fwd / bwd through the module."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (N, B, H, L) in ((32768, 16, 96, 16384), (32768, 16, 192, 16384), (32768, 16, 384, 16384), (32768, 16, 768, 16384), (16384, 16, 96, 8192), (4096, 16, 96, 2048), (32768, 64, 48, 16384)):
    u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True); k = torch.randn(H, L, device="cuda").requires_grad_(True)
    dout = torch.randn(B, H, L, device="cuda").bfloat16()
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda()
    y = mod(u, k)
    def bwd():
        u.grad = None; k.grad = None
        y.backward(dout, retain_graph=True)
    for mode in (None, "2"):
        if mode is None: os.environ.pop("FFC_WG_MULT", None)
        else: os.environ["FFC_WG_MULT"] = mode
        with torch.no_grad():
            tf = ev(lambda: mod(u, k))
        tb = ev(bwd)
        print(f"fft={N} B={B} H={H} L={L} {'cost-based chunks' if mode is None else 'former rule (mult 2)'}: fwd {tf:.4f} bwd {tb:.4f}", flush=True)
    os.environ.pop("FFC_WG_MULT", None); __import__("flashfftconv.conv").conv.reload_env()
