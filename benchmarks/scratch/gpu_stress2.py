"""Stress under GPU sharing: run several copies of this script at once.  Each case is computed once, then repeated;
any bitwise difference between repeats is reported with the positions that differ."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, _lib
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def run(mod, u, k, dout, gates):
    u = u.detach().requires_grad_(True); k = k.detach().requires_grad_(True)
    g = [x.detach().requires_grad_(True) for x in gates]
    y = mod(u, k, *g)
    y.backward(dout)
    torch.cuda.synchronize()
    return [y.detach(), u.grad, k.grad] + [x.grad for x in g]


bad = 0
cases = [(1024, 4, 111, 512), (1024, 1, 768, 512), (256, 4, 111, 256), (512, 4, 111, 256), (4096, 4, 111, 2048), (8192, 4, 111, 4096), (16384, 2, 111, 8192),
         (32768, 4, 64, 16384), (65536, 2, 32, 32768), (524288, 2, 32, 262144)]
for rnd in range(3):
    for dtype in (torch.float16, torch.bfloat16):
        for (N, B, H, L) in cases:
            torch.manual_seed(N + B)
            u = torch.randn(B, H, L, device="cuda").to(dtype) * 0.02; dout = torch.randn(B, H, L, device="cuda").to(dtype) * 0.02
            k = torch.randn(H, L, device="cuda") * 0.02
            mod = FlashFFTConv(N, dtype=dtype).cuda()
            ref = run(mod, u, k, dout, [])
            for rep in range(reps):
                got = run(mod, u, k, dout, [])
                for n, a, b in zip(["y", "du", "dk"], ref, got):
                    d = (a != b)
                    nm = int(d.sum())
                    if nm:
                        bad += 1
                        idx = d.nonzero()
                        lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
                        err = ((a.float() - b.float()).norm() / a.float().norm()).item()
                        print(f"[{tag}] MISMATCH N={N} {str(dtype)[6:]} B={B} H={H} L={L} rep={rep} {n}: {nm} differ, rel {err:.3e}, idx range {lo}..{hi}", flush=True)
print(f"[{tag}] stress done, mismatching tensors: {bad}")
