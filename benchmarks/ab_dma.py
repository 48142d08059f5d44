"""A/B of the LDS-DMA input rows of the saved-spectra backward (Body::rows_dma; FFC_FLAGS bit 8 = register path), same process,
interleaved repeats, bitwise comparison of du and the dk_f sums.  argv: FFC_FLAGS values to compare (default 0 8)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib = _lib.lib(); sp = _lib.stream_ptr; P = _lib.ptr
flags = sys.argv[1:] or ["0", "8"]
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for (N, B, H, L, dt) in ((32768, 16, 768, 16384, torch.bfloat16), (32768, 16, 768, 16384, torch.float16), (8192, 16, 768, 4096, torch.bfloat16),
                         (32768, 16, 768, 8192, torch.bfloat16), (32768, 4, 96, 16384, torch.bfloat16)):
    torch.manual_seed(0)
    u = torch.randn(B, H, L, device="cuda").to(dt); dout = torch.randn(B, H, L, device="cuda").to(dt); k = torch.randn(H, L, device="cuda") / 30
    mod = FlashFFTConv(N, dtype=dt).cuda(); plan = mod._get_plan(u.device)
    kf = C._kernel_fft(plan, k)
    z = torch.empty(lib.ffc_spectrum_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    y = torch.empty_like(u)
    _lib.check(lib.ffc_conv_fwd_z(plan.handle, P(u), P(kf), None, None, P(y), P(z), None, B, H, L, 0, 0, 0, 0, sp()), "fwd_z")
    res, outs = {}, {}
    for rep in range(3):
        for fl in flags:
            os.environ["FFC_FLAGS"] = fl; C.reload_env()
            ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda"); du = torch.empty_like(u)
            fn = lambda: _lib.check(lib.ffc_conv_bwd_z(plan.handle, P(dout), P(u), P(kf), None, None, P(du), None, None, P(ws), P(z), B, H, L,
                                                       0, 0, 0, 0, 0, 0, 0, sp()), "bwd_z")
            res.setdefault(fl, []).append(ev(fn))
            n = lib.ffc_dkf_slab_count(plan.handle, B, H) * H * plan.kf_elems * 8
            outs[fl] = (du.clone(), ws[:n].clone())
    base = outs[flags[0]]
    for fl in flags:
        same = torch.equal(outs[fl][0], base[0]) and torch.equal(outs[fl][1], base[1])
        print(f"fft {N} B{B} H{H} L{L} {str(dt).split('.')[-1]} FFC_FLAGS={fl:>3s}: bwd_z min {min(res[fl]):.4f} med {sorted(res[fl])[1]:.4f} ms   bitwise == flags {flags[0]}: {same}", flush=True)
    os.environ.pop("FFC_FLAGS"); C.reload_env()
