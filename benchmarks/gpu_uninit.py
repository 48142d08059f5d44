"""Uninitialised-global-memory hunt: every buffer the Python layer allocates with torch.empty/empty_like is pre-filled
with NaN bit patterns (0xFF bytes); results must equal a run whose buffers were pre-filled with zeros."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, _lib

_empty, _empty_like = torch.empty, torch.empty_like
FILL = [0]


def _fill(t):
    t.view(torch.uint8).fill_(FILL[0]) if t.is_contiguous() else None
    return t


torch.empty = lambda *a, **k: _fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _fill(_empty_like(*a, **k))


def run(mod, u, k, dout, gates):
    u = u.detach().requires_grad_(True); k = k.detach().requires_grad_(True)
    g = [x.detach().requires_grad_(True) for x in gates]
    y = mod(u, k, *g)
    y.backward(dout)
    torch.cuda.synchronize()
    return [y.detach(), u.grad, k.grad] + [x.grad for x in g]


bad = 0
cases = [(N, B, H, L) for N in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768) for (B, H) in ((4, 111), (1, 16), (5, 3), (2, 768)) for L in (N // 2, N)]
cases += [(32768, 2, 3, 20000), (1024, 3, 5, 1002), (65536, 2, 32, 32768), (65536, 3, 2, 65536), (524288, 2, 32, 262144), (1048576, 1, 3, 524288), (4194304, 1, 16, 1048576)]
for dtype in (torch.float16, torch.bfloat16):
    for (N, B, H, L) in cases:
        for gated in (False, True):
            torch.manual_seed(N + B)
            u = torch.randn(B, H, L, device="cuda").to(dtype); dout = torch.randn(B, H, L, device="cuda").to(dtype)
            k = torch.randn(H, L, device="cuda") * 0.1
            gates = [torch.randn(B, H, L, device="cuda").to(dtype) for _ in range(2)] if gated else []
            mod = FlashFFTConv(N, dtype=dtype).cuda()
            FILL[0] = 0
            ref = run(mod, u, k, dout, gates)
            FILL[0] = 255
            got = run(mod, u, k, dout, gates)
            for n, a, b in zip(["y", "du", "dk", "dpre", "dpost"], ref, got):
                nm = int((a != b).sum()) - int((a.isnan() & b.isnan()).sum())
                if nm:
                    bad += 1
                    idx = (a != b).nonzero()
                    print(f"UNINIT N={N} {str(dtype)[6:]} B={B} H={H} L={L} gated={gated} {n}: {nm} differ, nan={int(b.isnan().sum())}, idx {idx.min(0).values.tolist()}..{idx.max(0).values.tolist()}", flush=True)
print("uninit hunt done, mismatching tensors:", bad)
