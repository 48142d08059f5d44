"""Host-side cost of one module call: a shape so small that the GPU is never the bottleneck (fft 256, B=2, H=4), many calls,
wall time per call + the cProfile split.  usage: host_overhead.py [profile]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
import torch
from flashfftconv import FlashFFTConv

N, B, H = 256, 2, 4
L = N // 2
u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True)
k = torch.randn(H, L, device="cuda").requires_grad_(True)
g = [torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True) for _ in range(2)]
dout = torch.randn(B, H, L, device="cuda").bfloat16()
mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda()


def wall(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


def fwd_infer():
    with torch.no_grad():
        mod(u, k)


def fwd_train(): mod(u, k)
def fwd_train_gated(): mod(u, k, *g)


def step():
    u.grad = None; k.grad = None
    mod(u, k).backward(dout)


def step_gated():
    for t in (u, k, *g): t.grad = None
    mod(u, k, *g).backward(dout)


def torch_ref():       # the torch.fft form of the same op, for scale
    with torch.no_grad():
        torch.fft.irfft(torch.fft.rfft(u.float(), n=N) * torch.fft.rfft(k, n=N), n=N)[..., :L]


for f in (fwd_infer, fwd_train, fwd_train_gated, step, step_gated, torch_ref):
    print(f"{f.__name__:18s} {wall(f):7.1f} us per call", flush=True)
if len(sys.argv) > 1:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): step()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)

# ---- the C-ABI calls alone (host cost of one launch through ctypes) and the autograd round trip of an empty Function
from flashfftconv import conv as C, _lib
lib, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
plan = mod._get_plan(u.device)
ud, kd = u.detach(), k.detach()
kf = C._kernel_fft(plan, kd)
ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
du = torch.empty_like(ud); dk = torch.empty(H, L, device="cuda"); y = torch.empty_like(ud)
calls = {
    "ffc_kernel_fft": lambda: lib.ffc_kernel_fft(plan.handle, P(kd), H, L, P(kf), sp()),
    "ffc_conv_fwd": lambda: lib.ffc_conv_fwd(plan.handle, P(ud), P(kf), None, None, P(y), B, H, L, 0, sp()),
    "ffc_conv_bwd": lambda: lib.ffc_conv_bwd(plan.handle, P(dout), P(ud), P(kf), None, None, P(du), None, P(ws), B, H, L, sp()),
    "ffc_kernel_ifft_grad": lambda: lib.ffc_kernel_ifft_grad(plan.handle, P(ws), B, H, L, P(dk), sp()),
    "torch.empty_like": lambda: torch.empty_like(ud),
    "stream_ptr": lambda: sp(),
}
for n, f in calls.items():
    print(f"{n:22s} {wall(f):7.1f} us per call", flush=True)


class _Nop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return a.view_as(a)

    @staticmethod
    def backward(ctx, g):
        return g, None


def nop_step():
    u.grad = None
    _Nop.apply(u, k).backward(dout)


print(f"{'empty autograd.Function fwd+bwd':22s} {wall(nop_step):7.1f} us per call")
