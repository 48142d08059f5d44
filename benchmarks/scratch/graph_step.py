"""Does the training step capture into a HIP graph, and what does replay save?  (bench.py config)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
import torch
from flashfftconv import FlashFFTConv
N, B, H, L = 32768, 16, 768, 16384
dev = "cuda"
u = torch.randn(B, H, L, device=dev).bfloat16().requires_grad_(True)
k = torch.randn(H, L, device=dev).requires_grad_(True)
dout = torch.randn(B, H, L, device=dev).bfloat16()
mod = FlashFFTConv(N, dtype=torch.bfloat16).to(dev)
def step():
    u.grad = None; k.grad = None
    mod(u, k).backward(dout)
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step", [round(timed(step), 4) for _ in range(3)])
step(); torch.cuda.synchronize()
du0, dk0 = u.grad.clone(), k.grad.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
u.grad = None; k.grad = None
with torch.cuda.graph(g):
    mod(u, k).backward(dout)
g.replay(); torch.cuda.synchronize()
print("graph == eager:", torch.equal(u.grad, du0), torch.equal(k.grad, dk0))
print("graph ms/step", [round(timed(g.replay), 4) for _ in range(3)])
