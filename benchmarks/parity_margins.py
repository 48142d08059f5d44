"""Measured parity margins: relative L2 error of out / du / dk (/ dpregate / dpostgate) of the HIP path against the torch.fft fp32
oracle (oracle/torch_ref.py = reference tests/test_flashfftconv.py:5-13), per fft size, dtype and gated or not, next to the gate the
GPU tests apply (tests/test_flashfftconv_gpu.py REL x the size / gating factors).  VERDICT r03 weak #1: the margins were nowhere on
record.  Inputs as in the reference tests (0.02-scaled randn, exp-decaying k), padded case L = N/2.  Prints one row per case."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
from flashfftconv import FlashFFTConv
from oracle.torch_ref import ref_fft_conv

REL = {torch.bfloat16: 2e-2, torch.float16: 5e-3}
from tests.test_flashfftconv_gpu import rel      # the tests' measure: fp16 tensors get a floor of one subnormal step per element
# (the reference's 0.02-scaled gates put fp16 gated outputs at |out| ~ 5e-7, on fp16's fixed 6e-8 grid: without the floor the
# figure is the quantisation of the RESULT, 2e-2 .. 3e-2, not an error of the transform; the second pass below uses unit gates)
SIZES = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304]
print(f"{'fft':>8} {'dtype':>9} {'gated':>5} {'B':>2} {'H':>3} | {'out':>9} {'du':>9} {'dk':>9} {'dpre':>9} {'dpost':>9} | gate(out,du) gate(dk)")
for gscale in (0.02, 1.0):
  if gscale == 1.0:
      print("---- gated cases again with unit-scale gates (randn), plain relative L2 without the fp16 floor")
      rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
  for N in SIZES:
    for dt in (torch.bfloat16, torch.float16):
        for gated in ((False, True) if gscale == 0.02 else (True,)):
            torch.manual_seed(N % 9973 + gated)
            B, H = (4, 16) if N <= 131072 else (2, 4)
            L = N // 2
            u = (torch.randn(B, H, L, device="cuda") * 0.02).to(dt)
            k = torch.randn(H, L, device="cuda") * 0.02 * torch.exp(-0.1 * torch.arange(L, device="cuda"))
            g = [(torch.randn(B, H, L, device="cuda") * gscale).to(dt) for _ in range(2)] if gated else []
            dout = (torch.randn(B, H, L, device="cuda") * 0.02).to(dt)
            lv = [t.clone().requires_grad_(True) for t in [u, k] + g]
            lr = [t.clone().requires_grad_(True) for t in [u, k] + g]
            out = FlashFFTConv(N, dtype=dt).cuda()(*lv)
            ref = ref_fft_conv(lr[0] * lr[2], lr[1], n=N) * lr[3] if gated else ref_fft_conv(lr[0], lr[1], n=N)
            gv = torch.autograd.grad(out, lv, dout)
            gr = torch.autograd.grad(ref, lr, dout)
            e = [rel(out, ref)] + [rel(a, b) for a, b in zip(gv, gr)]
            f = (2.0 if N >= 65536 else 1.0) * (1.5 if gated else 1.0)
            gate, gate_dk = REL[dt] * f, max(REL[dt] * f, 1e-2 * f)
            cells = " ".join(f"{x:9.2e}" for x in e) + " " * (20 if not gated else 0)
            worst = max(e[0], e[1], *(e[3:] if gated else [0])) / gate
            print(f"{N:>8} {str(dt).split('.')[-1]:>9} {str(gated):>5} {B:>2} {H:>3} | {cells} | {gate:8.1e} {gate_dk:8.1e}   worst/gate {max(worst, e[2] / gate_dk):.2f}", flush=True)
            del lv, lr, out, ref, gv, gr
    torch.cuda.empty_cache()
