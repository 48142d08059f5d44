"""The C-ABI library loads on a machine without a GPU and exports every symbol the header declares."""
import ctypes, os, re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "flashfftconv_hip.h")
SO = os.path.join(ROOT, "flash-fft-conv_amd", "lib", "libflashfftconv_hip.so")


def declared():
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ffc_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    names = declared()
    for must in ("ffc_plan_create", "ffc_kernel_fft", "ffc_conv_fwd", "ffc_conv_bwd_dkf", "ffc_kernel_ifft_grad",
                 "ffc_conv1d_fwd", "ffc_conv1d_bwd", "ffc_last_error"):
        assert must in names


def test_library_exports_all_declared_symbols():
    if not os.path.exists(SO):
        import importlib.util
        spec = importlib.util.spec_from_file_location("ffc_build", os.path.join(ROOT, "flash-fft-conv_amd", "build.py"))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        m.build_hip()
    lib = ctypes.CDLL(SO)
    missing = [n for n in declared() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.ffc_version() >= 100


def test_module_refuses_cpu_tensors():
    """No CPU fallback in the product path: CPU tensors raise instead of silently running torch.fft."""
    import torch
    from flashfftconv import FlashFFTConv
    conv = FlashFFTConv(1024, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        conv(torch.zeros(1, 2, 512, dtype=torch.bfloat16), torch.zeros(2, 512))
    with pytest.raises(NotImplementedError):
        FlashFFTConv(3000)
