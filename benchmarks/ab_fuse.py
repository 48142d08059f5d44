"""A/B of the round-4 launch fusions at config 2 (and neighbours): k -> k_f inside the forward launch (tuning flag 64 turns it off),
dk out of the backward launch (flag 32 off), LDS-DMA dout rows (flag 8 off).  Module step (fwd + bwd) and the two C-ABI calls,
same process, interleaved repeats."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib = _lib.lib(); sp = _lib.stream_ptr; P = _lib.ptr
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
flags = sys.argv[1:] or ["0", "32", "64", "96"]
for (N, B, H, L) in ((32768, 16, 768, 16384), (16384, 16, 768, 8192), (8192, 16, 768, 4096), (4096, 16, 768, 2048), (32768, 4, 768, 16384)):
    torch.manual_seed(0)
    u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True); dout = torch.randn(B, H, L, device="cuda").bfloat16()
    k = (torch.randn(H, L, device="cuda") / 30).requires_grad_(True)
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
    ud, kd = u.detach(), k.detach()
    kf = torch.empty(H, plan.kf_elems, 2, dtype=torch.bfloat16, device="cuda")
    z = torch.empty(lib.ffc_spectrum_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    y, du = torch.empty_like(ud), torch.empty_like(ud); dk = torch.empty(H, L, device="cuda")
    def step():
        u.grad = None; k.grad = None
        mod(u, k).backward(dout)
    fk = lambda: _lib.check(lib.ffc_conv_fwd_k(plan.handle, P(kd), L, P(kf), P(ud), None, None, P(y), P(z), None, B, H, L, sp()), "fwd_k")
    bk = lambda: _lib.check(lib.ffc_conv_bwd_k(plan.handle, P(dout), P(ud), P(kf), None, None, P(du), None, None, P(ws), P(z), None, P(dk), L, B, H, L, sp()), "bwd_k")
    res = {}
    for rep in range(3):
        for fl in flags:
            os.environ["FFC_FLAGS"] = fl; C.reload_env()
            res.setdefault(fl, []).append((ev(step), ev(fk), ev(bk)))
    for fl in flags:
        r = res[fl]
        print(f"fft {N} B{B} H{H} L{L} FFC_FLAGS={fl:>4s}: step min {min(x[0] for x in r):.4f} med {sorted(x[0] for x in r)[1]:.4f}   fwd_k min {min(x[1] for x in r):.4f}   bwd_k min {min(x[2] for x in r):.4f} ms", flush=True)
    os.environ.pop("FFC_FLAGS"); C.reload_env()
