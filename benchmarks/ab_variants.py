"""A/B of library builds (flash-fft-conv_amd/build.py --variant NAME -D...): every variant is timed in its own process
(FFC_LIB=<variant .so>) on the same box, interleaved and repeated, fwd / fused-bwd kernels of a few shapes.
usage: python benchmarks/ab_variants.py base nopk prio ...   ("base" = the product library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(32768, 16, 768, 16384), (16384, 16, 768, 8192), (4096, 16, 768, 2048), (32768, 8, 768, 32768)]
if os.environ.get("AB_SHAPES"):      # e.g. AB_SHAPES="32768,8,768,32768;8192,16,768,8192"
    SHAPES = [tuple(int(x) for x in t.split(",")) for t in os.environ["AB_SHAPES"].split(";")]

CHILD = r'''
import os, sys, json, torch
ROOT = sys.argv[1]
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv, conv as C, _lib
lib = _lib.lib(); sp = _lib.stream_ptr
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
out = {}
for (N, B, H, L) in json.loads(sys.argv[2]):
    u = torch.randn(B, H, L, device="cuda").bfloat16(); dout = torch.randn(B, H, L, device="cuda").bfloat16(); k = torch.randn(H, L, device="cuda")
    plan = C.get_plan(N, torch.bfloat16, u.device)
    kf = C._kernel_fft(plan, k)
    ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda"); du = torch.empty_like(u)
    tf = ev(lambda: C._conv(plan, u, kf, None, None, False))
    tb = ev(lambda: _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(du), None, _lib.ptr(ws), B, H, L, sp()), "bwd"))
    out[f"N{N}_B{B}_L{L}"] = (round(tf, 4), round(tb, 4))
print("RESULT " + json.dumps(out))
'''

def run(variant):
    env = dict(os.environ)
    if variant != "base":
        env["FFC_LIB"] = os.path.join(ROOT, "flash-fft-conv_amd", "lib", "variants", variant, "libflashfftconv_hip.so")
        assert os.path.exists(env["FFC_LIB"]), env["FFC_LIB"]
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, json.dumps(SHAPES)], env=env, capture_output=True, text=True, timeout=600)
    for ln in r.stdout.splitlines():
        if ln.startswith("RESULT "):
            return json.loads(ln[7:])
    return {"error": (r.stdout + r.stderr)[-800:]}

if __name__ == "__main__":
    variants = sys.argv[1:] or ["base"]
    for rep in range(2):
        for v in variants:
            print(json.dumps({"variant": v, "rep": rep, "ms_fwd_bwd": run(v)}), flush=True)
