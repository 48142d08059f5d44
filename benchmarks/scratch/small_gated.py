"""single-tile sizes, B64 H768 (GPU-bound): saved spectra on / off, gated and not"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from benchmarks.sweep import conv_row
for N in (256, 1024, 2048):
    for gated in (False, True):
        r = conv_row(f"N={N}", N, 64, 768, N // 2, gated=gated)
        print(os.environ.get("FFC_SAVE_SPECTRUM", "1"), N, "gated" if gated else "plain", r["fwd_ms"], r["bwd_ms"], round(r["fwd_ms"] + r["bwd_ms"], 4), flush=True)
