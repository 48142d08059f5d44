"""A/B of FFC_FLAGS values (argv) on the forward / fused backward kernels of config 2 and fft 16384, same process, interleaved repeats.
Bits: 2 = k_f streamed, 4 = scratch streamed, bits 4..7 = start-up stagger s (workgroup b waits ((b/8) mod 8) * s * 512 cycles)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib = _lib.lib(); sp = _lib.stream_ptr
flags = sys.argv[1:] or ["0", "16", "32", "64", "128", "240"]
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (N, B, H, L) in ((32768, 16, 768, 16384), (16384, 16, 768, 8192), (32768, 16, 768, 32768)):
    u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
    kf = C._kernel_fft(plan, k)
    ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda"); du = torch.empty_like(u)
    res = {}
    for rep in range(3):
        for fl in flags:
            os.environ["FFC_FLAGS"] = fl; C.reload_env()
            tf = ev(lambda: C._conv(plan, u, kf, None, None, False))
            tb = ev(lambda: _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(du), None, _lib.ptr(ws), B, H, L, sp()), "bwd"))
            res.setdefault(fl, []).append((tf, tb))
    for fl in flags:
        r = res[fl]
        print(f"N={N} L={L} FFC_FLAGS={fl:>4s}: conv_fwd min {min(x[0] for x in r):.4f} med {sorted(x[0] for x in r)[1]:.4f}   bwd_fused min {min(x[1] for x in r):.4f} med {sorted(x[1] for x in r)[1]:.4f}", flush=True)
    os.environ.pop("FFC_FLAGS"); C.reload_env()
