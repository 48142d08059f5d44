"""Drop-in boundary of the module: things a caller of the reference module relies on besides forward().
CPU part: copy / pickle / state_dict compatibility; GPU part: the same after a forward, device placement checks."""
import copy, io, pickle
import pytest
import torch


def reference_shaped_state_dict(prefix=""):
    """Key set of a checkpoint saved from the REFERENCE FlashFFTConv(32768) / FlashFFTConv(4194304): its __init__
    registers these as persistent buffers (/root/reference/flashfftconv/conv.py:222-246 for 32768, :497-551 for 4M)."""
    names = ["f_32_fft", "f_32_ifft", "twiddle_factors_fft_32_32", "twiddle_factors_ifft_32_32", "twiddle_factors_fft_32_1K",
             "twiddle_factors_ifft_32_1K", "f_128_fft", "f_128_ifft", "f_128_fft_real", "f_128_fft_imag", "f_128_ifft_real",
             "f_128_ifft_imag", "twiddle_factors_fft_real", "twiddle_factors_fft_imag", "twiddle_factors_ifft_real",
             "twiddle_factors_ifft_imag", "f_sqrt_N_fft", "f_sqrt_N_ifft", "twiddle_factors_fft", "twiddle_factors_ifft", "twid",
             "f_16_fft", "f_64_ifft_imag", "twiddle_factors_fft_16_256", "twiddle_factors_ifft_16_1K"]
    return {prefix + n: torch.zeros(2, 2) for n in names}


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        from flashfftconv import FlashFFTConv
        self.lin = torch.nn.Linear(4, 4)
        self.flashfftconv = FlashFFTConv(32768, dtype=torch.bfloat16)


def test_reference_checkpoint_loads_strictly():
    m = Tiny()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert not any(k.startswith("flashfftconv.") for k in sd)           # this module registers no tables
    sd.update(reference_shaped_state_dict("flashfftconv."))
    res = m.load_state_dict(sd, strict=True)                            # the reference's buffer keys are swallowed
    assert not res.unexpected_keys and not res.missing_keys
    sd["flashfftconv.not_a_reference_buffer"] = torch.zeros(1)          # anything else is still an error
    with pytest.raises(RuntimeError):
        Tiny().load_state_dict(sd, strict=True)


def test_module_copies_and_pickles_cpu():
    from flashfftconv import FlashFFTConv, PartialFFTConv
    for m in (FlashFFTConv(4096, dtype=torch.float16), PartialFFTConv(128), Tiny()):
        c = copy.deepcopy(m)
        assert type(c) is type(m)
        m2 = pickle.loads(pickle.dumps(m))
        assert type(m2) is type(m)
        buf = io.BytesIO(); torch.save(m, buf); buf.seek(0)
        assert type(torch.load(buf, weights_only=False)) is type(m)


@pytest.mark.gpu
def test_module_copies_and_pickles_after_forward():
    """advisor finding (round 1): after the first forward the module held a ctypes plan handle and deepcopy / torch.save
    raised; plans now live in a process-wide cache, the module only holds tensors."""
    from flashfftconv import FlashFFTConv, FrequencySparseFFTConv
    u = torch.randn(2, 4, 2048, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(4, 2048, device="cuda") * 0.1
    m = FlashFFTConv(4096, dtype=torch.bfloat16).cuda()
    m.cache_kf = True
    y = m(u, k)
    for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert torch.equal(c(u, k), y)
    buf = io.BytesIO(); torch.save(m, buf); buf.seek(0)
    assert torch.equal(torch.load(buf, weights_only=False)(u, k), y)
    s = FrequencySparseFFTConv(512)
    ys = s(u, k)
    assert torch.equal(copy.deepcopy(s)(u, k), ys)
    # an averaged (EMA / SWA) copy of a model that contains the module
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__(); self.conv = m; self.k = torch.nn.Parameter(k.clone())
        def forward(self, x): return self.conv(x, self.k)
    net = Net()
    ema = torch.optim.swa_utils.AveragedModel(net)
    ema.update_parameters(net)
    assert torch.equal(ema(u), net(u))


@pytest.mark.gpu
def test_device_placement_is_checked():
    from flashfftconv import FlashFFTConv
    m = FlashFFTConv(1024, dtype=torch.bfloat16).cuda()
    u = torch.randn(2, 4, 512, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        m(u, torch.randn(4, 512))                                       # k on the CPU: was used as a device pointer
    with pytest.raises(RuntimeError):
        m(u, torch.randn(4, 512, device="cuda"), u.cpu(), u)            # gate on the CPU
    # launches follow the tensors' device and its current stream, not whatever device is current
    s = torch.cuda.Stream()
    k = torch.randn(4, 512, device="cuda")
    y0 = m(u, k)
    with torch.cuda.stream(s):
        y1 = m(u, k)
    s.synchronize()
    assert torch.equal(y0, y1)
