"""Times of the forward / spectrum-saving forward / backward-on-spectra kernels for a list of shapes, with a digest of the results:
run once per library build (FFC_LIB=flash-fft-conv_amd/lib/variants/<name>/libflashfftconv_hip.so) and compare the lines.
argv: fft,B,H,L[,g] ... (default: config 2, fft 16384, fft 65536, fft 8192, config 3 gated)."""
import hashlib, os, sys, torch
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
dig = lambda t: hashlib.sha256(t.view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:10]
cases = ([tuple(int(x) for x in a.split(',')[:4]) + (a.endswith('g'),) for a in sys.argv[1:]] if len(sys.argv) > 1 else None) or [
    (32768, 16, 768, 16384, False), (32768, 16, 768, 32768, False), (16384, 16, 768, 8192, False), (65536, 16, 768, 32768, False),
    (8192, 16, 768, 4096, False), (4096, 16, 768, 2048, False), (16384, 8, 1024, 8192, True)]
print("library:", _lib.LIB_PATH)
for (N, B, H, L, gated) in cases:
    torch.manual_seed(0)
    u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda") / 30
    pre = torch.randn_like(u) if gated else None; post = torch.randn_like(u) if gated else None
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda(); plan = mod._get_plan(u.device)
    kf = C._kernel_fft(plan, k)
    ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    z = torch.empty(lib.ffc_spectrum_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
    y0, y1, du, dpre, dpost = (torch.empty_like(u) for _ in range(5))
    yraw = torch.empty_like(u) if gated else None
    g = lambda t: P(t) if gated else None
    f0 = lambda: _lib.check(lib.ffc_conv_fwd(plan.handle, P(u), P(kf), P(pre), P(post), P(y0), B, H, L, 0, sp()), "fwd")
    f1 = lambda: _lib.check(lib.ffc_conv_fwd_z(plan.handle, P(u), P(kf), P(pre), P(post), P(y1), P(z), P(yraw), B, H, L, 0, 0, 0, 0, sp()), "fwd_z")
    if gated:
        b1 = lambda: _lib.check(lib.ffc_conv_bwd_zy(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(du), P(dpre), P(dpost), P(ws), P(z), P(yraw), B, H, L, 0, 0, 0, 0, 0, 0, 0, sp()), "bwd_zy")
    else:
        b1 = lambda: _lib.check(lib.ffc_conv_bwd_z(plan.handle, P(dout), P(u), P(kf), None, None, P(du), None, None, P(ws), P(z), B, H, L, 0, 0, 0, 0, 0, 0, 0, sp()), "bwd_z")
    t = [1e9] * 3
    for rep in range(3):
        for i, fn in enumerate((f0, f1, b1)): t[i] = min(t[i], ev(fn))
    n = lib.ffc_dkf_slab_count(plan.handle, B, H) * H * plan.kf_elems * 8
    print(f"fft {N} B{B} H{H} L{L} gated={gated}: fwd {t[0]:.4f}  fwd_z {t[1]:.4f}  bwd_z {t[2]:.4f} ms   digests y {dig(y0)} y_z {dig(y1)} du {dig(du)} dkf {dig(ws[:n])}", flush=True)
