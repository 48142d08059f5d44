"""ctypes binding of libflashfftconv_hip.so (C-ABI in include/flashfftconv_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, a
RuntimeError is raised (the reference raises RuntimeError from TORCH_CHECK the same way,
csrc/flashfftconv/monarch_cuda/monarch_fwd.h:196-528)."""
import ctypes, os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FFC_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libflashfftconv_hip.so")   # FFC_LIB: tuning builds
_lib = None

c_vp, c_i64, c_int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"flashfftconv: HIP library not built ({LIB_PATH}); run `python flash-fft-conv_amd/build.py`")
        L = ctypes.CDLL(LIB_PATH)
        L.ffc_last_error.restype = ctypes.c_char_p
        L.ffc_plan_create.argtypes = [c_i64, c_int, ctypes.POINTER(c_vp)]
        L.ffc_plan_destroy.argtypes = [c_vp]
        L.ffc_plan_reload_env.argtypes = [c_vp]; L.ffc_plan_reload_env.restype = None
        L.ffc_plan_kf_elems.argtypes = [c_vp]; L.ffc_plan_kf_elems.restype = c_i64
        L.ffc_plan_kf_scale.argtypes = [c_vp]; L.ffc_plan_kf_scale.restype = ctypes.c_double
        L.ffc_plan_kf_index.argtypes = [c_vp, c_vp]
        L.ffc_kernel_fft.argtypes = [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]
        L.ffc_kf_pack.argtypes = [c_vp, c_vp, c_i64, c_vp, c_vp]
        L.ffc_conv_fwd.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]
        L.ffc_conv_fwd_sparse.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp]
        L.ffc_conv_fwd_strided.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_i64, c_i64, c_i64, c_i64, c_vp]
        L.ffc_conv_bwd_gated_strided.argtypes = [c_vp] * 10 + [c_i64] * 10 + [c_vp]
        L.ffc_spectrum_bytes.argtypes = [c_vp, c_i64, c_i64]; L.ffc_spectrum_bytes.restype = c_i64
        L.ffc_conv_fwd_z.argtypes = [c_vp] * 8 + [c_i64] * 7 + [c_vp]
        L.ffc_conv_bwd_z.argtypes = [c_vp] * 11 + [c_i64] * 10 + [c_vp]
        L.ffc_conv_bwd_zy.argtypes = [c_vp] * 12 + [c_i64] * 10 + [c_vp]
        L.ffc_conv_fwd_k.argtypes = [c_vp, c_vp, c_i64] + [c_vp] * 7 + [c_i64] * 3 + [c_vp]
        L.ffc_conv_bwd_k.argtypes = [c_vp] * 13 + [c_i64] * 4 + [c_vp]
        L.ffc_conv_fwd_kx.argtypes = [c_vp, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]
        L.ffc_conv_bwd_kx.argtypes = [c_vp] * 8 + [ctypes.c_float, c_i64, c_i64, c_i64, c_vp]
        L.ffc_dkf_workspace_bytes.argtypes = [c_vp, c_i64, c_i64]; L.ffc_dkf_workspace_bytes.restype = c_i64
        L.ffc_dkf_slab_count.argtypes = [c_vp, c_i64, c_i64]; L.ffc_dkf_slab_count.restype = c_i64
        L.ffc_conv_bwd_dkf.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]
        L.ffc_conv_bwd.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]
        L.ffc_conv_bwd_gated.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]
        L.ffc_kernel_ifft_grad.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]
        L.ffc_kernel_ifft_grad_slabs.argtypes = [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]
        c_f = ctypes.c_float
        L.ffc_outer_pass.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_vp]
        L.ffc_outer_pass_r.argtypes = [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_vp]
        L.ffc_outer_pass_all.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_vp]
        L.ffc_kernel_fft_c.argtypes = [c_vp, c_vp, c_i64, c_vp, c_f, c_vp]
        L.ffc_kernel_ifft_grad_c.argtypes = [c_vp, c_vp, c_i64, c_i64, c_vp, c_f, c_vp]
        L.ffc_kernel_ifft_grad_c_slabs.argtypes = [c_vp, c_vp, c_i64, c_i64, c_vp, c_f, c_vp]
        L.ffc_conv1d_fwd.argtypes = [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp]
        L.ffc_conv1d_bwd.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp]
        L.ffc_debug_poison.argtypes = [c_vp]
        L.ffc_debug_peaks.argtypes = [c_vp, c_vp]
        L.ffc_selftest_primitives.argtypes = [c_vp, c_vp]
        L.ffc_conv_fwd_prof.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"flashfftconv: {what} failed: {lib().ffc_last_error().decode()}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = None


def stream_ptr(device=None):
    """current stream of `device` (default: the current device; the autograd functions run under
    torch.cuda.device(u.device), so that is the device of the tensors) as a raw hipStream_t.
    torch._C._cuda_getCurrentRawStream is the accessor kernel launchers use (0.3 us); torch.cuda.current_stream() builds a
    Stream object and costs 5-10 us per call, four times per forward + backward of a short-sequence step."""
    import torch
    global _raw_stream
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        idx = None if device is None else (device if isinstance(device, int) else torch.device(device).index)
        return _raw_stream(torch._C._cuda_getDevice() if idx is None else idx)
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
