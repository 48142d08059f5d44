"""Spectrum-saving forward / backward pair (ffc_conv_fwd_z / ffc_conv_bwd_z; include/flashfftconv_hip.h): the training forward
keeps FFT(u * pregate) and the backward kernel reads it instead of transforming u a second time (the reference recomputes:
kernels_bf16/monarch_cuda_32_32_32_bwd_kernel_bf16.h).  Checked against the recomputing pair through the C-ABI (y, du,
dpregate bit for bit; dk, dpostgate to the rounding of the spectrum) and against the torch.fft oracle through the module."""
import pytest
import torch

from oracle.torch_ref import ref_fft_conv
from tests.test_flashfftconv_gpu import rel, make_inputs, stable, REL

pytestmark = pytest.mark.gpu
SIZES = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072]      # 2048: 2 passes of the 1024 kernel; the last two: 2 / 4 passes of the 32768 kernel
SHAPES = [(4, 16), (3, 7), (1, 5), (16, 24)]      # (B, H): even, odd batch, single row, several chunks
SHAPES_BIG = [(4, 8), (3, 5)]


def _call_pair(N, B, H, L, dtype, gated):
    from flashfftconv import FlashFFTConv, conv as C, _lib
    lib, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
    torch.manual_seed(N + 7 * B + H)
    u = torch.randn(B, H, L, device="cuda").to(dtype); dout = torch.randn(B, H, L, device="cuda").to(dtype)
    k = torch.randn(H, L, device="cuda") * torch.exp(-0.05 * torch.arange(L, device="cuda")) / 4
    pre = torch.randn_like(u) if gated else None; post = torch.randn_like(u) if gated else None
    plan = FlashFFTConv(N, dtype=dtype).cuda()._get_plan(u.device)
    kf = C._kernel_fft(plan, k)
    zb = lib.ffc_spectrum_bytes(plan.handle, B, H)
    if N >= 4096:
        assert zb == ((B + 1) // 2) * H * N * 4
    else:       # single-tile sizes: 4 KB slots per tile of G pairs (and pass)
        G = {256: 4, 512: 2, 1024: 1, 2048: 1}[N]
        assert zb == H * -(-((B + 1) // 2) // G) * (2 if N == 2048 else 1) * 4096
    z = torch.empty(zb, dtype=torch.uint8, device="cuda")
    ws0 = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda"); ws1 = torch.empty_like(ws0)
    y0, y1 = torch.full_like(u, 7.0), torch.full_like(u, 9.0)
    yraw = torch.full_like(u, 11.0) if gated else None
    o0 = [torch.full_like(u, 3.0) for _ in range(3)]; o1 = [torch.full_like(u, 5.0) for _ in range(3)]
    g = (lambda t: P(t)) if gated else (lambda t: None)
    _lib.check(lib.ffc_conv_fwd(plan.handle, P(u), P(kf), P(pre), P(post), P(y0), B, H, L, 0, sp()), "fwd")
    _lib.check(lib.ffc_conv_fwd_z(plan.handle, P(u), P(kf), P(pre), P(post), P(y1), P(z), P(yraw), B, H, L, 0, 0, 0, 0, sp()), "fwd_z")
    _lib.check(lib.ffc_conv_bwd_gated(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(o0[0]), g(o0[1]), g(o0[2]), P(ws0), B, H, L, sp()), "bwd")
    # saved spectra + saved pre-postgate output: dpostgate = dout * y_raw, no dpost from the kernel
    _lib.check(lib.ffc_conv_bwd_z(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(o1[0]), g(o1[1]), None, P(ws1), P(z), B, H, L,
                                  0, 0, 0, 0, 0, 0, 0, sp()), "bwd_z")
    if gated:
        o1[2] = dout * yraw
        # the kernel can still produce dpost from the saved spectrum alone (one more inverse transform)
        o2 = torch.full_like(u, 13.0)
        _lib.check(lib.ffc_conv_bwd_z(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(o1[0]), g(o1[1]), P(o2), P(ws1), P(z), B, H, L,
                                      0, 0, 0, 0, 0, 0, 0, sp()), "bwd_z")
        assert rel(o2, o0[2]) < (1e-2 if dtype == torch.bfloat16 else 2e-3)
        # ffc_conv_bwd_zy: dpost = dout * y_raw written by the kernel's dout row load -- bit for bit the elementwise product
        # (fp32 product, rounded once), du / dpregate / dk_f unchanged by the side product
        o3 = [torch.full_like(u, 17.0) for _ in range(3)]; ws3 = torch.empty_like(ws0)
        _lib.check(lib.ffc_conv_bwd_zy(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(o3[0]), P(o3[1]), P(o3[2]), P(ws3), P(z), P(yraw),
                                       B, H, L, 0, 0, 0, 0, 0, 0, 0, sp()), "bwd_zy")
        assert torch.equal(o3[2], o1[2]), "fused dpostgate != dout * y_raw"
        assert torch.equal(o3[0], o1[0]) and torch.equal(o3[1], o1[1]), "du / dpregate changed by the fused dpostgate"
        nsl = lib.ffc_dkf_slab_count(plan.handle, B, H) * H * plan.kf_elems * 2 * 4
        assert torch.equal(ws3[:nsl], ws1[:nsl]), "dk_f sums changed by the fused dpostgate"
        if N <= 2048:
            # round 6, single-tile sizes: y_raw kept WITHOUT the spectra (the module's gated form at fft <= 1024) -- the backward transforms
            # u * pregate again, exactly as the recomputing kernel does, and takes dpostgate from y_raw
            y4, yraw4 = torch.full_like(u, 19.0), torch.full_like(u, 23.0)
            _lib.check(lib.ffc_conv_fwd_z(plan.handle, P(u), P(kf), P(pre), P(post), P(y4), None, P(yraw4), B, H, L, 0, 0, 0, 0, sp()), "fwd_z (y_raw alone)")
            assert torch.equal(y4, y1) and torch.equal(yraw4, yraw), "forward output changed by leaving the spectrum store out"
            o4 = [torch.full_like(u, 29.0) for _ in range(3)]; ws4 = torch.empty_like(ws0)
            _lib.check(lib.ffc_conv_bwd_zy(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(o4[0]), P(o4[1]), P(o4[2]), P(ws4), None, P(yraw4),
                                           B, H, L, 0, 0, 0, 0, 0, 0, 0, sp()), "bwd_zy (y_raw alone)")
            assert torch.equal(o4[0], o0[0]) and torch.equal(o4[1], o0[1]), "du / dpregate (y_raw alone) != the recomputing kernel's"
            assert torch.equal(o4[2], o1[2]), "dpostgate (y_raw alone) != dout * y_raw"
            assert torch.equal(ws4[:nsl], ws0[:nsl]), "dk_f sums (y_raw alone) != the recomputing kernel's"
    dk0 = torch.empty(H, L, device="cuda"); dk1 = torch.empty(H, L, device="cuda")
    _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, P(ws0), B, H, L, P(dk0), sp()), "dk")
    _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, P(ws1), B, H, L, P(dk1), sp()), "dk")
    torch.cuda.synchronize()
    return (y0, o0, dk0), (y1, o1, dk1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("N", SIZES)
def test_saved_spectrum_equals_recompute(N, gated, dtype):
    for (B, H) in (SHAPES if N <= 32768 else SHAPES_BIG):
        for L in (N // 2, N, N // 2 - 8, N - 3):      # padded, full, ragged fast path, ragged slow path (L % 8 != 0)
            (y0, o0, dk0), (y1, o1, dk1) = _call_pair(N, B, H, L, dtype, gated)
            tag = f"fft {N} B{B} H{H} L{L} gated={gated} {dtype}"
            step = 1e-2 if dtype == torch.bfloat16 else 2e-3
            if N >= 4096:
                assert torch.equal(y0, y1), f"{tag}: forward output changed by the spectrum store"
            else:       # single-tile kernels: the two instantiations contract the fp32 k_f product differently (last-bit steps)
                assert rel(y1, y0) < step / 4, f"{tag}: forward output rel {rel(y1, y0):.2e}"
            assert torch.equal(o0[0], o1[0]), f"{tag}: du"
            # same spectrum up to the last bit of its rounding to the plan dtype (the fp32 schedule of the two kernels differs):
            # outputs that are rounded once more (dpostgate) move by single steps of the dtype
            step = 1e-2 if dtype == torch.bfloat16 else 2e-3
            dk_gate = 2e-3
            if N == 16384:
                # round 5: the FORWARD of fft 16384 folds the outer twiddle into its stage matrices while the
                # recomputing backward transforms u with the chains: the saved spectrum and the recomputed one are two roundings of the
                # same values (each ~7e-3 / 8e-4 from the oracle), no longer the same arithmetic
                dk_gate, step = (1.2e-2, 2e-2) if dtype == torch.bfloat16 else (4e-3, 4e-3)
            assert rel(dk1, dk0) < dk_gate, f"{tag}: dk rel {rel(dk1, dk0):.2e}"
            if gated:
                assert torch.equal(o0[1], o1[1]), f"{tag}: dpregate"
                assert rel(o1[2], o0[2]) < step, f"{tag}: dpostgate rel {rel(o1[2], o0[2]):.2e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("N", SIZES)
def test_module_gradients_with_saved_spectrum(N, gated, dtype):
    """module level: save_spectrum on (default) against the torch.fft oracle and against save_spectrum off."""
    from flashfftconv import FlashFFTConv
    torch.manual_seed(1)
    B, H, L = 6, 24, N // 2
    u, k = make_inputs(B, H, L, N, dtype, False)
    gates = [torch.randn_like(u) * 0.5 for _ in range(2)] if gated else []
    dout = torch.randn_like(u) * 0.02
    grads = {}
    from flashfftconv import conv as C
    for save in (True, False):
        conv = FlashFFTConv(N, dtype=dtype).cuda()
        conv.save_spectrum = save
        leaves = [u.clone().requires_grad_(True), k.clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in gates]
        out = conv(*leaves)
        g = torch.autograd.grad(out, leaves, dout, retain_graph=True)
        g2 = torch.autograd.grad(out, leaves, dout)
        assert all(torch.equal(a, b) for a, b in zip(g, g2)), "second backward through a retained graph differs"
        grads[save] = (out.detach(), g)
    y_only_max = C._Y_ONLY_MAX
    if gated and N <= 2048:
        # the gated single-tile sizes keeping y_raw WITHOUT the spectra (round 6, opt-in: FFC_Y_ONLY_MAX) against the default (both kept)
        try:
            C._Y_ONLY_MAX = 0 if N <= y_only_max else 2048
            conv = FlashFFTConv(N, dtype=dtype).cuda()
            leaves = [u.clone().requires_grad_(True), k.clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in gates]
            out = conv(*leaves)
            grads["both"] = (out.detach(), torch.autograd.grad(out, leaves, dout))
        finally:
            C._Y_ONLY_MAX = y_only_max
        # same forward instantiation, du / dpregate from the same arithmetic, dpostgate = dout * the same y_raw: bit for bit
        assert torch.equal(grads["both"][0], grads[True][0])
        assert all(torch.equal(grads["both"][1][i], grads[True][1][i]) for i in (0, 2, 3))
    if N >= 4096:
        assert torch.equal(grads[True][0], grads[False][0]) and torch.equal(grads[True][1][0], grads[False][1][0])
    else:       # single-tile kernels: equal to last-bit steps (see test_saved_spectrum_equals_recompute)
        assert rel(grads[True][0], grads[False][0]) < REL[dtype] / 4 and rel(grads[True][1][0], grads[False][1][0]) < REL[dtype] / 4
    lc = [u.clone().requires_grad_(True), k.clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in gates]
    fwd = (lambda: (ref_fft_conv(lc[0] * lc[2], lc[1], n=N) * lc[3],)) if gated else (lambda: (ref_fft_conv(lc[0], lc[1], n=N),))
    (ref,) = stable(fwd, "forward")
    gref = stable(lambda: torch.autograd.grad(ref, lc, dout.clone(), retain_graph=True), "backward")
    tol = REL[dtype] * (1.5 if gated else 1.0)
    for save in grads:
        out, g = grads[save]
        assert rel(out, ref) < tol and rel(g[0], gref[0]) < tol
        assert rel(g[1], gref[1]) < max(tol, 1e-2 * (1.5 if gated else 1.0)), f"dk rel-L2 {rel(g[1], gref[1]):.3e}"
        if gated:
            assert rel(g[2], gref[2]) < tol and rel(g[3], gref[3]) < tol, f"gate gradients (save_spectrum={save})"


def test_spectrum_buffer_is_refused_when_misaligned_or_missing():
    from flashfftconv import FlashFFTConv, _lib
    lib = _lib.lib()
    plan = FlashFFTConv(1024, dtype=torch.bfloat16).cuda()._get_plan(torch.device("cuda", 0))
    u = torch.zeros(2, 8, 512, device="cuda", dtype=torch.bfloat16); kf = torch.zeros(8, plan.kf_elems, 2, device="cuda", dtype=torch.bfloat16)
    z = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    rc = lib.ffc_conv_fwd_z(plan.handle, _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(u), z.data_ptr() + 4, None, 2, 8, 512, 0, 0, 0, 0, _lib.stream_ptr())
    assert rc != 0 and b"spectrum" in lib.ffc_last_error()
    rc = lib.ffc_conv_fwd_z(plan.handle, _lib.ptr(u), _lib.ptr(kf), None, None, _lib.ptr(u), None, None, 2, 8, 512, 0, 0, 0, 0, _lib.stream_ptr())
    assert rc != 0
    # y_raw without the spectra is a single-tile form (fft <= 2048): refused by the fused sizes in either direction
    plan4 = FlashFFTConv(4096, dtype=torch.bfloat16).cuda()._get_plan(torch.device("cuda", 0))
    kf4 = torch.zeros(8, plan4.kf_elems, 2, device="cuda", dtype=torch.bfloat16)
    g, yr, ws = torch.ones_like(u), torch.zeros_like(u), torch.zeros(lib.ffc_dkf_workspace_bytes(plan4.handle, 2, 8), dtype=torch.uint8, device="cuda")
    outs = [torch.zeros_like(u) for _ in range(4)]
    P = _lib.ptr
    rc = lib.ffc_conv_fwd_z(plan4.handle, P(u), P(kf4), P(g), P(g), P(outs[0]), None, P(yr), 2, 8, 512, 0, 0, 0, 0, _lib.stream_ptr())
    assert rc != 0 and b"single-tile" in lib.ffc_last_error()
    rc = lib.ffc_conv_bwd_zy(plan4.handle, P(u), P(u), P(kf4), P(g), P(g), P(outs[1]), P(outs[2]), P(outs[3]), P(ws), None, P(yr), 2, 8, 512,
                             0, 0, 0, 0, 0, 0, 0, _lib.stream_ptr())
    assert rc != 0 and b"single-tile" in lib.ffc_last_error()


@pytest.mark.gpu
def test_training_step_captures_into_a_hip_graph():
    """forward (spectra saved) + backward launch on torch's current stream with no host synchronisation and no allocation
    outside torch's allocator, so the whole step can be captured and replayed as one HIP graph; replay == eager, bitwise"""
    from flashfftconv import FlashFFTConv
    N, B, H, L = 8192, 4, 16, 4096
    torch.manual_seed(3)
    u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True)
    k = torch.randn(H, L, device="cuda").requires_grad_(True)
    dout = torch.randn(B, H, L, device="cuda").bfloat16()
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda()

    def step():
        u.grad = None; k.grad = None
        mod(u, k).backward(dout)
    step(); torch.cuda.synchronize()
    du0, dk0 = u.grad.clone(), k.grad.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    u.grad = None; k.grad = None
    with torch.cuda.graph(g):
        mod(u, k).backward(dout)
    u.grad.zero_(); k.grad.zero_()
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(u.grad, du0) and torch.equal(k.grad, dk0)


@pytest.mark.gpu
def test_no_spectra_are_stored_without_a_graph():
    """autograd.Function.forward sees grad mode off and needs_input_grad = the inputs' requires_grad flags either way: the module
    notes the caller's grad mode, so a forward under torch.no_grad() (training mode, inputs that require grad) allocates no
    spectrum buffer"""
    from flashfftconv import FlashFFTConv, conv as C
    N, B, H, L = 8192, 4, 16, 4096
    u = torch.randn(B, H, L, device="cuda").bfloat16().requires_grad_(True)
    k = torch.randn(H, L, device="cuda").requires_grad_(True)
    mod = FlashFFTConv(N, dtype=torch.bfloat16).cuda()
    calls = []
    orig = C._spectrum_buffer
    C._spectrum_buffer = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
    try:
        with torch.no_grad():
            y0 = mod(u, k)
        assert not calls
        y1 = mod(u, k)
        assert calls and torch.equal(y0, y1)
        y1.sum().backward()
        assert u.grad is not None and k.grad is not None
    finally:
        C._spectrum_buffer = orig


@pytest.mark.gpu
@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("N,B,H,L", [(32768, 4, 512, 16384), (32768, 3, 600, 16376), (32768, 2, 512, 32768),
                                     (16384, 6, 512, 8192), (16384, 3, 768, 16384), (8192, 10, 512, 4096), (8192, 5, 600, 4090)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_one_launch_per_direction_equals_the_separate_kernels(N, B, H, L, gated, dt):
    """ffc_conv_fwd_k / ffc_conv_bwd_k with a workgroup per head (H >= 512 on 256 CUs): k -> k_f runs inside the forward
    launch (Modes::kfft_head, fft 8192 .. 32768) and dk comes out of the backward launch (Modes::dk_tail / dk_tail_multi, fft 8192 ..
    32768) -- against the same calls with the tuning flags 32 | 64 (k -> k_f and dk_f -> dk as kernels of their own): k_f, y, du,
    gate gradients bit for bit (where k_f is), dk to rounding."""
    import os
    from flashfftconv import FlashFFTConv, conv as C, _lib
    lib, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
    torch.manual_seed(B + H)
    u = torch.randn(B, H, L, device="cuda").to(dt); dout = torch.randn(B, H, L, device="cuda").to(dt)
    k = torch.randn(H, L, device="cuda") * torch.exp(-0.01 * torch.arange(L, device="cuda")) / 4
    pre = torch.randn_like(u) if gated else None; post = torch.randn_like(u) if gated else None
    plan = FlashFFTConv(N, dtype=dt).cuda()._get_plan(u.device)
    res = {}
    for fl in ("128", "96"):      # 128: the dk tail also at fft 8192 (off by default there); 96: both fusions off
        os.environ["FFC_FLAGS"] = fl; C.reload_env()
        try:
            kf = torch.full((H, plan.kf_elems, 2), float("nan"), dtype=dt, device="cuda")
            z = torch.empty(lib.ffc_spectrum_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
            y = torch.full_like(u, 7.0); yraw = torch.full_like(u, 9.0) if gated else None
            _lib.check(lib.ffc_conv_fwd_k(plan.handle, P(k), L, P(kf), P(u), P(pre), P(post), P(y), P(z), P(yraw), B, H, L, sp()), "fwd_k")
            ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device="cuda")
            du, dpre, dpost = (torch.full_like(u, 3.0) for _ in range(3))
            dk = torch.full((H, L), float("nan"), device="cuda")
            g = (lambda t: P(t)) if gated else (lambda t: None)
            _lib.check(lib.ffc_conv_bwd_k(plan.handle, P(dout), P(u), P(kf), P(pre), P(post), P(du), g(dpre), g(dpost), P(ws), P(z), P(yraw),
                                          P(dk), L, B, H, L, sp()), "bwd_k")
            torch.cuda.synchronize()
            res[fl] = (kf, y, du, dpre, dpost, dk)
        finally:
            os.environ.pop("FFC_FLAGS"); C.reload_env()
    a, b = res["128"], res["96"]
    assert not torch.isnan(a[0].float()).any() and not torch.isnan(a[5]).any()
    # k_f comes from the same source compiled into two kernels (kfft_kernel / conv_kernel): the compiler may contract the fp32
    # twiddle products differently, so single values can land on the neighbouring bf16 -- equal to rounding, usually bit for bit
    ndiff = int((a[0] != b[0]).sum())
    print(f"k_f: {ndiff} of {a[0].numel()} values differ between the two kernels")
    assert rel(a[0], b[0]) < (2e-3 if dt == torch.bfloat16 else 3e-4)
    names = ("k_f", "y", "du", "dpre", "dpost")
    for i in (1, 2) + ((3, 4) if gated else ()):
        if ndiff == 0:
            assert torch.equal(a[i], b[i]), names[i]
        else:
            assert rel(a[i], b[i]) < 4e-3, f"{names[i]} rel {rel(a[i], b[i]):.2e}"
    assert rel(a[5], b[5]) < 2e-3, f"dk rel {rel(a[5], b[5]):.2e}"
