"""Stress: every kernel must give bitwise identical results after the CUs' LDS / register files were poisoned
(ffc_debug_poison) as after a clean run.  Prints mismatching element counts per (fft size, dtype, kernel)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, _lib
lib = _lib.lib()


def run(mod, u, k, dout, gates):
    u = u.detach().requires_grad_(True); k = k.detach().requires_grad_(True)
    g = [x.detach().requires_grad_(True) for x in gates]
    y = mod(u, k, *g)
    y.backward(dout)
    torch.cuda.synchronize()
    return [y.detach(), u.grad, k.grad] + [x.grad for x in g]


def poison():
    _lib.check(lib.ffc_debug_poison(_lib.stream_ptr()), "poison")


bad = 0
cases = [(N, B, H, N // 2) for N in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768) for (B, H) in ((4, 111), (1, 16), (5, 3))]
cases += [(1024, 4, 111, 1024), (4096, 3, 5, 4096), (32768, 2, 3, 20000), (65536, 2, 32, 32768), (524288, 2, 32, 262144), (4194304, 1, 16, 1048576)]
for dtype in (torch.float16, torch.bfloat16):
    for (N, B, H, L) in cases:
        for gated in (False, True):
            torch.manual_seed(N + B)
            u = torch.randn(B, H, L, device="cuda").to(dtype); dout = torch.randn(B, H, L, device="cuda").to(dtype)
            k = torch.randn(H, L, device="cuda") * 0.1
            gates = [torch.randn(B, H, L, device="cuda").to(dtype) for _ in range(2)] if gated else []
            mod = FlashFFTConv(N, dtype=dtype).cuda()
            ref = run(mod, u, k, dout, gates)
            for rep in range(3):
                poison()
                got = run(mod, u, k, dout, gates)
                names = ["y", "du", "dk", "dpre", "dpost"]
                for n, a, b in zip(names, ref, got):
                    nm = int((a != b).sum()) + int((a.isnan() != b.isnan()).sum())
                    if nm:
                        bad += 1
                        err = ((a.float() - b.float()).norm() / a.float().norm()).item()
                        print(f"MISMATCH N={N} {dtype} B={B} H={H} L={L} gated={gated} rep={rep} {n}: {nm} elements differ, rel {err:.3e}", flush=True)
print("poison stress done, mismatching tensors:", bad)
