import sys, json
sys.path[:0]=["flash-fft-conv_amd","."]
from benchmarks import sweep as SW
import torch
for N,L in ((256,128),(512,256),(1024,512),(2048,1024)):
    r = SW.conv_row(f"N={N}", N, 64, 768, L)
    print(json.dumps({k:r[k] for k in ("row","fwd_ms","bwd_ms","fwd_infer_ms")}), flush=True)
for r in SW.readme_rows([256, 1024]): print(json.dumps({k:r[k] for k in ("row","fwd_ms_scaled_to_B64_H768")}))
