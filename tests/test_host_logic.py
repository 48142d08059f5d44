"""Host-side logic that needs no GPU: the memory budget of the saved spectra (flashfftconv/conv.py _spectrum_budget_ok), the head
groups of the pipelined B-shard, the frequency map of the HBM-level sizes against a brute-force statement of the level algebra."""
import os, sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-fft-conv_amd"))


def test_spectrum_budget_is_self_limiting(monkeypatch):
    from flashfftconv import conv as C
    free = {"v": 80 << 30}
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda idx=None: (free["v"], 288 << 30))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda idx=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda idx=None: 0)
    C._free_cache.clear()
    dev = torch.device("cuda", 0)
    assert C._spectrum_budget_ok(1 << 20, dev)                     # small requests never ask the driver
    assert C._spectrum_budget_ok(9 << 30, dev)                     # 9 GB of 80 GB free: within 1/8
    assert not C._spectrum_budget_ok(11 << 30, dev)                # more than 1/8 of what is left (71 GB)
    assert C._spectrum_budget_ok(11 << 30, dev, "always")          # "always" skips the test
    # layer after layer (cached figure, requests subtracted): the total stays below the memory that was free at the start
    C._free_cache.clear()
    granted, left = 0, 80 << 30
    for _ in range(200):
        n = 2 << 30
        if C._spectrum_budget_ok(n, dev):
            granted += n
    assert granted < (80 << 30) and granted >= 60 << 30
    assert not C._spectrum_budget_ok(2 << 30, dev)


@pytest.mark.parametrize("H,world,ng", [(768, 8, 2), (111, 8, 2), (5, 2, 2), (16, 8, 4), (3, 8, 2)])
def test_head_groups_partition_the_heads(H, world, ng):
    from flashfftconv.sharding import head_groups, head_range
    groups = head_groups(H, world, ng)
    assert groups[0][0] == 0 and groups[-1][1] == H and all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
    owned = []
    for (g0, g1) in groups:
        for r in range(world):
            s, e = head_range(g1 - g0, r, world)
            owned += list(range(g0 + s, g0 + e))
    assert owned == list(range(H))


@pytest.mark.parametrize("N,fac", [(65536, ((16,), 4096)), (4194304, ((16, 16), 16384)), (4194304, ((128,), 32768)), (2097152, ((64,), 32768))])
def test_row_freq_is_a_bijection_onto_the_spectrum(N, fac):
    """every natural frequency of the N-point spectrum sits in exactly one (inner row, inner frequency)"""
    from flashfftconv import bigfft as BG
    offs, stride = BG.row_freq(N, fac)
    M = fac[1]
    assert len(offs) * M == N and stride * M == N
    f = (np.asarray(offs, np.int64)[:, None] + stride * np.arange(M, dtype=np.int64)[None, :]) % N
    assert np.array_equal(np.sort(f.ravel()), np.arange(N))


def test_loads_in_flight_audit_of_the_build():
    """build.py check_load_runs: the level kernels must keep a block's 16 row loads in flight together (round 4: with run-time switches
    inside the row loop every load was followed by its own s_waitcnt and nothing failed -- 2.95 instead of 4.45 TB/s)"""
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-fft-conv_amd"))
    import build
    head = "_Z10big_kernelILi32ELi0ELb1EEvN3ffc7BigArgsE:\n"
    good = head + "\tglobal_load_dwordx4 v[0:3], v[4:5], off\n\tv_add_u32 v1, v2, v3\n" * 16 + "\ts_waitcnt vmcnt(0)\n\ts_endpgm\n"
    bad = head + "\tglobal_load_dwordx4 v[0:3], v[4:5], off\n\ts_waitcnt vmcnt(0)\n" * 16 + "\ts_endpgm\n"
    other = "_Z3fooPv:\n\ts_endpgm\n"
    rules = build.LOAD_RUNS["ffc_k_big.hip"][:1]
    for txt, ok in ((good, True), (bad, False), (other, False)):      # (no kernel matching the rules is a failure too)
        with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
            f.write(txt)
        try:
            if ok:
                build.check_load_runs(f.name, rules)
            else:
                with pytest.raises(RuntimeError):
                    build.check_load_runs(f.name, rules)
        finally:
            os.unlink(f.name)


# ---------------------------------------------------------------- round 5: ADVICE r04
class _StubLib:
    def __init__(self):
        self.reloaded = []

    def ffc_plan_reload_env(self, handle):
        self.reloaded.append(handle)

    def ffc_spectrum_bytes(self, handle, B, H):
        return 64 << 20


def test_reload_env_drops_the_cached_workspace_sizes(monkeypatch):
    """the dk_f workspace size depends on FFC_WG_MULT (chunks per head): a stale cache after reload_env would hand the backward kernel
    a workspace that is too small"""
    from flashfftconv import conv as C, _lib
    stub = _StubLib()
    monkeypatch.setattr(_lib, "lib", lambda: stub)

    class P:
        handle = 7
    p = P()
    p._ws_cache = {(16, 768): 123}
    monkeypatch.setitem(C._PLANS, ("test", 0, 0), p)
    C.reload_env()
    assert stub.reloaded == [7] and not hasattr(p, "_ws_cache")


def test_spectrum_fallbacks_are_counted_and_the_budget_is_charged_once(monkeypatch):
    """a refused spectrum buffer is counted (benchmarks can assert which backward ran); the inner buffers of the HBM-level path
    pass mode "always" because the module-level test already charged them"""
    from flashfftconv import conv as C, _lib
    monkeypatch.setattr(_lib, "lib", lambda: _StubLib())
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda idx=None: (128 << 20, 288 << 30))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda idx=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda idx=None: 0)
    C._free_cache.clear()

    class Plan:
        seqlen, handle = 32768, 1
    before = dict(C.SPECTRUM_FALLBACKS)
    dev = torch.device("cuda", 0)
    assert C._spectrum_buffer(Plan(), 16, 768, dev, True, True) is None          # 64 MB of 128 MB free: over 1/8 -> refused, counted
    assert C.SPECTRUM_FALLBACKS["budget"] == before["budget"] + 1
    free_after = C._free_cache[0][0]
    alloc = []
    monkeypatch.setattr(torch, "empty", lambda n, **kw: alloc.append(n) or "buf")
    assert C._spectrum_buffer(Plan(), 16, 768, dev, True, "always") == "buf" and alloc == [64 << 20]
    assert C._free_cache[0][0] == free_after                                     # "always": nothing charged a second time


def test_fft_131072_route_is_shared_by_the_wrappers():
    """fft 131072 takes the HBM-level form for rows longer than N/2; the B-shard and the Hyena operator must pick their kernels by
    the same rule as the single-rank module"""
    from flashfftconv import FlashFFTConv
    import inspect
    from flashfftconv import sharding, hyena
    m = FlashFFTConv(131072, dtype=torch.bfloat16)
    assert not m._big and not m._route_big(65536) and m._route_big(65537)
    assert "_route_big" in inspect.getsource(sharding.BatchShardedFFTConv.forward)
    assert "_route_big" in inspect.getsource(hyena.gated_conv_from_slices)
    assert "_fit_seqlen" in inspect.getsource(hyena.gated_conv_from_slices)      # and fits the fft size to the rows like the module


def test_fft_size_fits_the_rows():
    """FlashFFTConv._fit_seqlen: modules above the single-launch sizes run the smallest fft size that holds the linear convolution of
    the rows they are handed; full-size calls, the fused sizes and the frequency-sparse modules keep their size."""
    from flashfftconv import FlashFFTConv
    m = FlashFFTConv(4194304, dtype=torch.bfloat16)
    assert m._fit_seqlen(1048576, 1048576) == 2097152            # BASELINE config 4
    assert m._fit_seqlen(1048576, 1048577) == 2097152 and m._fit_seqlen(1048576, 1048578) == 4194304
    assert m._fit_seqlen(2097152, 2097152) == 4194304 and m._fit_seqlen(4194304, 4194304) == 4194304
    assert m._fit_seqlen(1000, 24) == 1024 and m._fit_seqlen(1, 1) == 256
    assert FlashFFTConv(131072, dtype=torch.float16)._fit_seqlen(16384, 16384) == 32768
    assert FlashFFTConv(65536, dtype=torch.float16)._fit_seqlen(32768, 32768) == 65536
    assert FlashFFTConv(32768, dtype=torch.float16)._fit_seqlen(100, 100) == 32768      # fused sizes: implicit padding already
    m.fit_fft = False
    assert m._fit_seqlen(1000, 24) == 4194304
    m.fit_fft, m._kf_keep = True, 100
    assert m._fit_seqlen(1000, 24) == 4194304                    # frequency-sparse: the mask lives on the seqlen-point spectrum
    m._kf_keep = None
    # the fitted module follows the caller's mode switches and never fits again
    m.eval(); m.save_spectrum = False; m.cache_kf = True
    s = m._fitted_module(1024)
    assert s is m._fitted_module(1024) and s.seqlen == 1024 and s.dtype == m.dtype
    assert (s.training, s.save_spectrum, s.cache_kf, s.fit_fft) == (False, False, True, False)
    assert list(m.state_dict()) == [] and list(m.children()) == []
    import copy, pickle
    assert copy.deepcopy(m)._fitted[1024].seqlen == 1024 and pickle.loads(pickle.dumps(m)).seqlen == 4194304


@pytest.mark.parametrize("N,Lu,Lk", [(4096, 1024, 1024), (4096, 700, 1349), (2048, 1, 300), (1024, 257, 256)])
def test_fitted_fft_size_is_the_same_convolution(N, Lu, Lk):
    """the identity _fit_seqlen rests on, on the oracle: while Lu + Lk - 1 <= n the n-point circular convolution, its du and its dk
    equal the N-point ones (to the rounding of the oracle's fp32 transforms); half of that size wraps"""
    from oracle.torch_ref import ref_fft_conv
    torch.manual_seed(N + Lu)
    u, k, dout = torch.randn(2, 3, Lu, dtype=torch.float64), torch.randn(3, Lk, dtype=torch.float64), torch.randn(2, 3, Lu, dtype=torch.float64)
    n = N
    while n // 2 >= Lu + Lk - 1:
        n //= 2
    assert n < N
    res = []
    sizes = (N, n) + ((n // 2,) if n // 2 >= max(Lu, Lk) else ())
    for size in sizes:
        uu, kk = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
        y = ref_fft_conv(uu, kk, n=size)
        res.append((y.detach(),) + torch.autograd.grad(y, (uu, kk), dout))
    for a, b in zip(res[0], res[1]):
        assert (a - b).norm() < 1e-5 * b.norm()
    if len(res) == 3:
        assert (res[0][0] - res[2][0]).norm() > 1e-2 * res[0][0].norm()


def test_level_routing_of_the_2m_and_4m_sizes(monkeypatch):
    """bigfft.choose: fft 4194304 = 128 x 32768 / 2097152 = 64 x 32768 in ONE level while the rows fit the first 32 long-side rows (L <= N / 4 resp. N / 2);
    beyond that two levels / the 2-pass inner size -- or, opt-in (FFC_BIG_WIDE=1, round 6), the level's wide form, which stores all rows (no half-row form)."""
    from flashfftconv import bigfft as BG

    class Ops:
        HAS_128 = True
        HAS_WIDE = True
    monkeypatch.setattr(BG, "WIDE", False)
    assert BG.choose(4194304, 1048576, Ops) == ((128,), 32768) and BG.choose(4194304, 1048577, Ops) == ((16, 16), 16384)
    assert BG.choose(2097152, 1048576, Ops) == ((64,), 32768) and BG.choose(2097152, 2097152, Ops) == ((32,), 65536)
    assert BG.half_ok(4194304, 1, 1048576, Ops) and not BG.half_ok(4194304, 1, 2000000, Ops) and BG.half_ok(2097152, 1, 2097152, Ops)
    monkeypatch.setattr(BG, "WIDE", True)
    assert BG.choose(4194304, 4194304, Ops) == ((128,), 32768) and BG.choose(2097152, 1500001, Ops) == ((64,), 32768)
    assert BG.is_wide(128, 32768, 1048577) and not BG.is_wide(128, 32768, 1048576) and not BG.is_wide(32, 65536, 2097152)
    assert BG.half_ok(4194304, 1, 1048576, Ops) and not BG.half_ok(4194304, 1, 1048577, Ops) and not BG.half_ok(2097152, 1, 2097152, Ops)
    assert BG.choose(4194304, 4194304, type("NoWide", (), {"HAS_128": True})) == ((16, 16), 16384)
