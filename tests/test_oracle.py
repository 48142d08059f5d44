"""The CPU oracle is pinned against golden vectors generated from the REFERENCE's own oracle source
(oracle/make_golden.py), and against the structure-faithful restatement of the reference's Monarch
factorisation (oracle/monarch_ref.py)."""
import glob, os
import numpy as np
import pytest

from oracle import ref_fft_conv as O
from oracle import monarch_ref as M

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "conv_*.npz")))
DT = {"bfloat16": "bf16", "float16": "fp16"}


def rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[5:-4] for p in GOLD])
def test_oracle_matches_reference_golden(path):
    g = np.load(path)
    N, dt = int(g["N"]), DT[str(g["dtype"])]
    if int(g["gated"]):
        out = O.ref_fft_conv_gated(g["u"], g["k"], g["pre"], g["post"], N, dtype=dt)
        du, dk, dpre, dpost = O.ref_grads(g["u"], g["k"], g["dout"], N, g["pre"], g["post"])
    else:
        out = O.ref_fft_conv(g["u"], g["k"], N)
        du, dk = O.ref_grads(g["u"], g["k"], g["dout"], N)
    # golden outputs were rounded to bf16/fp16 by the reference oracle (".to(u.dtype)")
    tol = 6e-3 if dt == "bf16" else 8e-4
    assert rel(out, g["out"].astype(np.float64)) < tol
    assert rel(du, g["du"].astype(np.float64)) < 2 * tol
    assert rel(dk, g["dk"].astype(np.float64)) < 2 * tol     # fp32 in the reference
    if int(g["gated"]):
        assert rel(dpre, g["dpre"].astype(np.float64)) < 2 * tol
        assert rel(dpost, g["dpost"].astype(np.float64)) < 2 * tol


@pytest.mark.parametrize("N", [256, 1024])
def test_reference_monarch_2stage_equals_fft(N):
    rng = np.random.default_rng(N)
    u = rng.standard_normal((2, 3, N // 2)); k = rng.standard_normal((3, N // 2))
    assert rel(M.monarch_conv_2stage(u, k, N), O.ref_fft_conv(u, k, N)) < 1e-10


@pytest.mark.parametrize("N,n1,n2", [(4096, 16, 16), (8192, 32, 16), (16384, 16, 32), (32768, 32, 32)])
def test_reference_monarch_3stage_equals_fft(N, n1, n2):
    rng = np.random.default_rng(N)
    u = rng.standard_normal((1, 2, N // 2)); k = rng.standard_normal((2, N // 2))
    assert rel(M.monarch_conv_3stage(u, k, N, n1, n2), O.ref_fft_conv(u, k, N)) < 1e-10


@pytest.mark.parametrize("L,Lk", [(2048, 2048), (1024, 700), (2048, 1)])
def test_fft2048_fold_identity(L, Lk):
    """flashfftconv/conv.py FOLDED_SEQLENS: the 2048-point circular convolution equals the 4096-point one with
    k periodised ([k_2048 | k_2048]); dk folds back as dk'[:2048] + dk'[2048:]."""
    rng = np.random.default_rng(L + Lk)
    u = rng.standard_normal((2, 3, L)); k = rng.standard_normal((3, Lk)); dout = rng.standard_normal((2, 3, L))
    kp = np.zeros((3, 2048)); kp[:, :Lk] = k
    k2 = np.concatenate([kp, kp], -1)
    assert rel(O.ref_fft_conv(u, k2, 4096), O.ref_fft_conv(u, k, 2048)) < 1e-12
    du, dk = O.ref_grads(u, k, dout, 2048)
    du2, dk2 = O.ref_grads(u, k2, dout, 4096)
    assert rel(du2, du) < 1e-12
    assert rel((dk2[:, :2048] + dk2[:, 2048:])[:, :Lk], dk) < 1e-12


SPARSE = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sparse_*.npz")))


@pytest.mark.parametrize("path", SPARSE, ids=[os.path.basename(p)[7:-4] for p in SPARSE])
def test_oracle_matches_reference_sparse_golden(path):
    """oracle restatement of the reference's PartialFFTConv / FrequencySparseFFTConv vs vectors produced by the
    reference's own module (oracle/make_golden.py: sparse_golden)."""
    g = np.load(path)
    dt, Np = DT[str(g["dtype"])], int(g["N_partial"])
    tol = 6e-3 if dt == "bf16" else 8e-4
    if str(g["kind"]) == "partial":
        out = O.ref_partial_conv(g["x"], g["k"], Np)
        kt = g["k"].copy(); kt[..., Np:] = 0
        dx, dk = O.ref_grads(g["x"], kt, g["dout"], 2 * int(g["L"]))
        dk[..., Np:] = 0
    else:
        out, dx, dk = O.ref_freq_sparse_conv(g["x"], g["k"], Np, g["dout"])
    assert rel(out, g["out"].astype(np.float64)) < tol
    assert rel(dx, g["dx"].astype(np.float64)) < 2 * tol
    assert rel(dk, g["dk"].astype(np.float64)) < 2 * tol
