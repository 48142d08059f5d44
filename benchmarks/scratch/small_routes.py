"""fft 2048 routing A/B (2 passes of the 1024 kernel vs the 4096 plan with k periodised) next to its neighbours."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from benchmarks.sweep import conv_row
for N in (512, 1024, 2048, 4096):
    r = conv_row(f"N={N}", N, 16, 768, N // 2)
    print(os.environ.get("FFC_MULTIPASS", "default"), N, r["fwd_ms"], r["fwd_infer_ms"], r["bwd_ms"], flush=True)
