"""sweep rows for chosen L: python benchmarks/sweep_some.py 32768 65536 ..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from benchmarks import sweep as SW
for a in sys.argv[1:]:
    L = int(a); N = 2 * L
    Hrun = 768 if N <= 131072 else max(16, 768 * 131072 // N)
    r = SW.conv_row(f"sweep L={L}", N, 16, 768, L, Hrun=Hrun)
    print(json.dumps({k: r[k] for k in ("row", "fft", "fwd_ms", "bwd_ms", "fwd_hbm_frac", "bwd_hbm_frac")}), flush=True)
