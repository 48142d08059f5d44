"""Where do the two 277 us direct_copy kernels per Hyena layer come from, and which GEMM call forms avoid them?"""
import torch, torch.nn as nn, time
B, L, D = 2, 32768, 256
dev, dt = "cuda", torch.bfloat16
u = torch.randn(B, L, D, device=dev, dtype=dt)
inp = nn.Linear(D, 3 * D).to(dev, dt); outp = nn.Linear(D, D).to(dev, dt)
y = torch.randn(B, D, L, device=dev, dtype=dt)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def kern(fn):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        fn(); torch.cuda.synchronize()
    return [(e.key[:60], round(e.device_time_total)) for e in p.key_averages() if e.device_time_total > 5]
cases = {
    "in: W @ u^T": lambda: inp.weight @ u.transpose(-1, -2),
    "in: (W @ u^T).contiguous()": lambda: (inp.weight @ u.transpose(-1, -2)).contiguous(),
    "in: linear(u)^T.contiguous()": lambda: inp(u).transpose(-1, -2).contiguous(),
    "in: bmm(W.expand, u^T)": lambda: torch.bmm(inp.weight.unsqueeze(0).expand(B, -1, -1), u.transpose(-1, -2)),
    "in: baddbmm(bias, W.expand, u^T)": lambda: torch.baddbmm(inp.bias.view(1, -1, 1), inp.weight.unsqueeze(0).expand(B, -1, -1), u.transpose(-1, -2)),
    "out: linear(y^T)": lambda: outp(y.transpose(-1, -2)),
    "out: matmul(y^T, W^T) + b": lambda: torch.matmul(y.transpose(-1, -2), outp.weight.t()) + outp.bias,
    "out: baddbmm(b, y^T, W^T.expand)": lambda: torch.baddbmm(outp.bias.view(1, 1, -1), y.transpose(-1, -2), outp.weight.t().unsqueeze(0).expand(B, -1, -1)),
    "out: (W @ y)^T view": lambda: (outp.weight @ y).transpose(-1, -2),
}
for n, f in cases.items():
    r = f()
    print(f"{n:38s} {t(f):7.3f} ms  contiguous={r.is_contiguous()} shape={tuple(r.shape)}  {kern(f)}")
