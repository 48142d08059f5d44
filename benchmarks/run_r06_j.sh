# round-6 GPU call J: validation of the final library: smoke, the full GPU suite (the driver's round-end command), conv1d BLH backward before / after the launch-bound fix is not
# separable any more (one library): its timing beside the BHL form for the record
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_j; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os, torch
sys.path[:0] = [os.path.join(os.environ["GRAFT_REPO_ROOT"], "flash-fft-conv_amd")]
from flashfftconv import FlashDepthWiseConv1d
import torch.nn as nn
def ev(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
B, D, L, K = 64, 2048, 8192, 3
ref = nn.Conv1d(D, D, K, groups=D, padding=1).cuda()
for bhl in (True, False):
    m = FlashDepthWiseConv1d(D, K, 1, ref.weight.detach(), ref.bias.detach(), is_bhl=bhl, device="cuda", dtype=torch.bfloat16)
    x = torch.randn((B, D, L) if bhl else (B, L, D), device="cuda", dtype=torch.bfloat16, requires_grad=True)
    y = m(x); dout = torch.randn_like(y)
    def bwd():
        x.grad = None
        for p in m.parameters(): p.grad = None
        y.backward(dout, retain_graph=True)
    with torch.no_grad():
        tf = ev(lambda: m(x))
    print(f"conv1d k=3 B{B} D{D} L{L} bf16 {'BHL' if bhl else 'BLH'}: fwd {tf:.4f} ms  bwd {ev(bwd):.4f} ms")
PY
