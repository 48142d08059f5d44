"""HBM-level outer passes alone (csrc/ffc_big.h, ffc_outer_pass / ffc_outer_pass_all): ms per launch and GB/s of the bytes a pass has
to move (long side + short side, once each), forward and inverse, for the level shapes the module uses.  Run once per library build
(FFC_LIB=flash-fft-conv_amd/lib/variants/<name>/libflashfftconv_hip.so) to compare knock-out / geometry variants."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib, bigfft as BG


def ev(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / it)
    return min(ts)


print("library:", _lib.LIB_PATH)
dt = torch.bfloat16
# (fft, n0, long rows B, heads, long length): one level of 16 x 32768 (fft 512K), 32 x 32768 (1M), 16 x 16384, 128 x 32768 (4M, L = N/4)
for (N, n0, B, H, L) in ((524288, 16, 4, 48, 262144), (524288, 16, 4, 48, 524288), (1048576, 32, 4, 24, 524288), (1048576, 32, 4, 24, 1048576),
                         (262144, 16, 4, 96, 131072), (4194304, 128, 2, 24, 1048576)):
    mod = FlashFFTConv(N, dtype=dt).cuda()
    ops = C._TorchOps(mod, torch.device("cuda", 0))
    mi = N // n0
    npair = (B + 1) // 2
    x = torch.randn(B, H, L, device="cuda").to(dt)
    short = torch.empty(2 * npair, H * n0, mi, dtype=dt, device="cuda")
    out = torch.empty_like(x)
    sc = BG.level_scale(n0)
    tf = ev(lambda: ops.outer(dt, n0, True, x, short, None, B, npair, H, mi, L, sc))
    ti = ev(lambda: ops.outer(dt, n0, False, short, out, None, B, npair, H, mi, L, 1.0 / (n0 * sc)))
    nbytes = x.numel() * 2 + short.numel() * 2
    print(json.dumps({"fft": N, "n0": n0, "B": B, "H": H, "L": L, "MB": round(nbytes / 1e6, 1), "fwd_ms": round(tf, 4), "inv_ms": round(ti, 4),
                      "fwd_GBs": round(nbytes / tf / 1e6), "inv_GBs": round(nbytes / ti / 1e6)}), flush=True)
