"""fft 4M / 2M one-level sizes: all passes of the 128- / 64-point level in one launch (default) vs one launch per pass"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from benchmarks.sweep import conv_row
tag = os.environ.get("FFC_BIG_ONE_LAUNCH", "1")
for (N, B, H, L, Hrun, gated) in ((4194304, 1, 16, 1048576, None, False), (4194304, 2, 16, 1048576, None, True), (2097152, 16, 768, 1048576, 48, False)):
    r = conv_row("x", N, B, H, L, gated=gated, Hrun=Hrun)
    print("one_launch=" + tag, N, B, H, L, "gated" if gated else "plain", r["fwd_ms"], r["bwd_ms"], r["fwd_infer_ms"], flush=True)
