"""Manual hardware probe (run through gpurun): primitive self-test vs simulator, forward parity, timing."""
import ctypes, os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-fft-conv_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import simlib
from flashfftconv import _lib

L = _lib.lib()
rng = np.random.default_rng(0)
# ---- primitives
vals = rng.standard_normal((64, 8, 2)).astype(np.float32)
inp = (simlib.f32_to_bf16_bits(vals[..., 0]).astype(np.uint32) | (simlib.f32_to_bf16_bits(vals[..., 1]).astype(np.uint32) << 16))
inp = np.ascontiguousarray(inp, dtype=np.uint32)
out_d = np.zeros((64, 40), np.uint32); out_s = np.zeros((64, 40), np.uint32)
_lib.check(L.ffc_selftest_primitives(inp.ctypes.data, out_d.ctypes.data), "selftest")
simlib.lib().ffcsim_selftest_primitives(inp.ctypes.data_as(ctypes.c_void_p), out_s.ctypes.data_as(ctypes.c_void_p))
mf_d, mf_s = out_d[:, :16].view(np.float32), out_s[:, :16].view(np.float32)
print("mfma bf16 max abs diff", np.abs(mf_d - mf_s).max(), "ref max", np.abs(mf_s).max())
print("tr16 equal:", np.array_equal(out_d[:, 32:34], out_s[:, 32:34]))
print("pack bf16 equal:", np.array_equal(out_d[:, 34], out_s[:, 34]), "pack f16 equal:", np.array_equal(out_d[:, 35], out_s[:, 35]))
print("unpack equal:", np.array_equal(out_d[:, 36:40], out_s[:, 36:40]))
if not np.array_equal(out_d[:, 32:34], out_s[:, 32:34]):
    print("device tr16 row0-3:", out_d[:4, 32:34]); print("sim    tr16 row0-3:", out_s[:4, 32:34])

dev = torch.device("cuda")
def run(N, dtype, B, H, Lx, gated=False, iters=0):
    dt = {torch.bfloat16: 0, torch.float16: 1}[dtype]
    plan = ctypes.c_void_p()
    _lib.check(L.ffc_plan_create(N, dt, ctypes.byref(plan)), "plan")
    torch.manual_seed(0)
    u = torch.randn(B, H, Lx, device=dev).to(dtype)
    k = torch.randn(H, Lx, device=dev) * 0.1
    kf_nat = torch.fft.fft(k, n=N).contiguous()
    kf = torch.empty(H, L.ffc_plan_kf_elems(plan), 2, dtype=dtype, device=dev)
    _lib.check(L.ffc_kf_pack(plan, _lib.ptr(kf_nat), H, _lib.ptr(kf), None), "pack")
    g1 = torch.randn_like(u) if gated else None
    g2 = torch.randn_like(u) if gated else None
    y = torch.empty_like(u)
    def call():
        _lib.check(L.ffc_conv_fwd(plan, _lib.ptr(u), _lib.ptr(kf), _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(y), B, H, Lx, 0, None), "conv")
    call(); torch.cuda.synchronize()
    v = (u * g1) if gated else u
    ref = torch.fft.ifft(torch.fft.fft(v.float(), n=N) * kf_nat).real[..., :Lx]
    if gated: ref = ref * g2.float()
    err = ((y.float() - ref).norm() / ref.norm()).item()
    ms = None
    if iters:
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
    L.ffc_plan_destroy(plan)
    return err, ms

for N in (256, 512, 1024, 4096, 8192, 16384, 32768):
    for dtype in (torch.bfloat16, torch.float16):
        for Lx in (N, N // 2):
            try:
                err, _ = run(N, dtype, 5, 24, Lx, gated=(Lx == N // 2))
                print(f"N={N} {dtype} L={Lx} relL2={err:.3e}", flush=True)
            except Exception as ex:
                print(f"N={N} {dtype} L={Lx} FAILED {ex}", flush=True)
print("unaligned L (slow path):", run(1024, torch.bfloat16, 3, 8, 1002)[0], run(4096, torch.bfloat16, 3, 8, 2050)[0])
for (N, B, H, Lx) in ((32768, 16, 768, 16384), (32768, 16, 768, 32768), (16384, 8, 1024, 8192), (4096, 16, 768, 2048), (1024, 16, 768, 512), (8192, 16, 768, 4096)):
    err, ms = run(N, torch.bfloat16, B, H, Lx, iters=10)
    gb = B * H * Lx * 2 * 2 / 1e9
    print(f"TIMING N={N} B={B} H={H} L={Lx}: {ms:.3f} ms  relL2={err:.2e}  io={gb/ms*1e3:.0f} GB/s", flush=True)
