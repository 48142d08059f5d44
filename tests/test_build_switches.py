"""The build switches that select between two forms of the same arithmetic must not change a single bit (round 4's load-scheduling work:
FFC_GATE_BATCH = 0 is the one-load-at-a-time form of the gated rows, FFC_RP_HOIST = 0 the multi-pass rows with the access-width switch
inside every load).  One simulator build with every switch flipped, compared with the
default simulator on forward, backward and the spectrum-saving pair of single-tile, fused and multi-pass sizes, gated and ragged."""
import ctypes, hashlib, os, subprocess, sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "flash-fft-conv_amd")
# round 5: FFC_OUTER_QUAD = 0 is the tile-pair form of phases A / C (4-byte LDS accesses), FFC_RP_FASTK = 0 the multi-pass backward with the
# run-time access-width switch in every row access (the default launches a 16-byte-only instantiation on aligned tensors)
# FFC_PK_GATE = 0: fp16 gate multiplies through fp32 (the packed fp16 multiply rounds the exact product once: the same bits)
# round 6: FFC_IP_MERGE = 0 is the fft-2048 forward with one row load / one read-modify-write of the output per PASS (the default keeps the pair's rows
# and the passes' sum in registers: one load, one store)
ALT_FLAGS = ["-DFFC_GATE_BATCH=0", "-DFFC_RP_HOIST=0", "-DFFC_OUTER_QUAD=0", "-DFFC_RP_FASTK=0", "-DFFC_PK_GATE=0", "-DFFC_IP_MERGE=0"]


def _alt_sim():
    """built in-tree under lib/variants/sim_alt/ and reused while no source is newer"""
    d = os.path.join(PKG, "lib", "variants", "sim_alt")
    so = os.path.join(d, "libffcsim.so")
    csrc = os.path.join(PKG, "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        os.makedirs(d, exist_ok=True)
        subprocess.check_call(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-pthread"] + ALT_FLAGS + ["-o", so,
                               os.path.join(csrc, "ffc_sim.cpp"), os.path.join(csrc, "ffc_plan.cpp")])
    return so


CASES = [(256, 200, 5, 2, 1, True), (1024, 1024, 3, 2, 0, True), (2048, 1024, 4, 1, 0, True), (2048, 1000, 3, 2, 1, False), (2048, 2048, 2, 1, 0, True), (4096, 2048, 5, 1, 1, True),
         (32768, 16384, 3, 1, 0, False), (32768, 9000, 2, 1, 0, True), (65536, 32768, 3, 1, 0, True), (65536, 40004, 1, 1, 1, False),
         (65536, 65536, 2, 1, 0, True), (131072, 65536, 1, 1, 0, False), (8192, 4096, 3, 1, 1, True), (16384, 16384, 2, 1, 0, False)]
_SCRIPT = r'''
import sys, hashlib, numpy as np
sys.path[:0] = [%r, %r, %r]
import simlib as S
dig = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
for (N, L, B, H, dt, gated) in %r:
    rng = np.random.default_rng(N + L)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    pre = S.to_bits(g1, dt) if gated else None; post = S.to_bits(g2, dt) if gated else None
    out = [dig(kf), dig(S.sim_conv_fwd(N, dt, S.to_bits(u, dt), kf, pre, post))]
    du, dpre, dk = S.sim_bwd(N, dt, S.to_bits(d, dt), S.to_bits(u, dt), kf, L, pre, post, 1)
    out += [dig(du), dig(dk)] + ([dig(dpre)] if gated else [])
    out += [dig(x) for x in S.sim_fwd_bwd_z(N, dt, S.to_bits(u, dt), S.to_bits(d, dt), kf, pre, post) if x is not None]
    print(N, L, B, H, dt, gated, " ".join(out), flush=True)
'''


def _digests(sim_lib):
    env = dict(os.environ)
    if sim_lib:
        env["FFC_SIM_LIB"] = sim_lib
    else:
        env.pop("FFC_SIM_LIB", None)
    src = _SCRIPT % (HERE, PKG, os.path.dirname(HERE), CASES)
    r = subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()


def test_build_switches_do_not_change_a_bit():
    base = _digests(None)
    alt = _digests(_alt_sim())
    assert len(base) == len(CASES) == len(alt)
    for a, b in zip(base, alt):
        assert a == b, (a, b)


# ---------------------------------------------------------------- FFC_FOLD_TW (round 5; measured, not adopted: DESIGN.md section 8)
_FOLD_SCRIPT = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import simlib as S
from oracle import ref_fft_conv as O
rel = lambda a, b: np.linalg.norm(np.asarray(a).astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30)
q = lambda x, dt: S.from_bits(S.to_bits(x, dt), dt).astype(np.float64)
for (N, L, B, H, gated, dt) in [(32768, 16384, 3, 1, False, 0), (32768, 32768, 2, 1, False, 0), (32768, 9000, 2, 1, True, 0), (32768, 16384, 2, 1, True, 1)]:
    rng = np.random.default_rng(N + L + B)
    u, d, g1, g2 = (rng.standard_normal((B, H, L)).astype(np.float32) for _ in range(4))
    k = (rng.standard_normal((H, L)) * 0.1).astype(np.float32)
    kf = S.sim_kernel_fft(N, dt, k)
    pre = S.to_bits(g1, dt) if gated else None; post = S.to_bits(g2, dt) if gated else None
    y, du, dpre, dpost, ws = S.sim_fwd_bwd_z(N, dt, S.to_bits(u, dt), S.to_bits(d, dt), kf, pre, post, 1, flags=0)
    nt, _, _, _ = S.plan_info(N, dt)
    dk = np.full((H, L), np.nan, np.float32)
    assert S.lib().ffcsim_kernel_ifft_grad(N, dt, S.p(ws), ws.size // (H * nt * 2048), H, L, S.p(dk)) == 0
    ref = O.ref_fft_conv_gated(q(u, dt), k, q(g1, dt), q(g2, dt), N, dtype=("bf16", "fp16")[dt]) if gated else O.ref_fft_conv(q(u, dt), k, N)
    r = O.ref_grads(q(u, dt), k, q(d, dt), N, q(g1, dt), q(g2, dt)) if gated else O.ref_grads(q(u, dt), k, q(d, dt), N)
    print(rel(S.from_bits(y, dt), ref), rel(S.from_bits(du, dt), r[0]), rel(dk, r[1]), dt, flush=True)
'''


def test_folded_outer_twiddle_variant_matches_the_oracle():
    """-DFFC_FOLD_TW=2: the outer twiddle folded into per-tile inner DFT matrices also at fft 32768 (forward kernels + the saved-spectra backward;
    the product folds the forward of fft 16384 only).  The variant simulator takes ~6 min to compile, so this runs only where it has been built
    (g++ -O0 -std=c++17 -fPIC -shared -pthread -DFFC_FOLD_TW=2 -o lib/variants/sim_fold/libffcsim2.so csrc/ffc_sim.cpp csrc/ffc_plan.cpp)."""
    so = os.path.join(PKG, "lib", "variants", "sim_fold", "libffcsim2.so")
    csrc = os.path.join(PKG, "csrc")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(csrc, f)) for f in ("ffc_body.h", "ffc_modes.h", "ffc_plan.cpp")):
        pytest.skip("variant simulator not built (or older than the kernel sources)")
    env = dict(os.environ, FFC_SIM_LIB=so)
    r = subprocess.run([sys.executable, "-c", _FOLD_SCRIPT % (HERE, PKG, os.path.dirname(HERE))], env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(rows) == 4
    for y, du, dk, dt in rows:
        tol = 1.2e-2 if dt == "0" else 1.5e-3
        assert float(y) < tol and float(du) < tol and float(dk) < 1.8e-2, (y, du, dk, dt)
