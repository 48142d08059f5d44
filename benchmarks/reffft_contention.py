"""Which side flickers under GPU time-slicing: the torch.fft (rocFFT) reference or the HIP module?  

Cases of the reference's own test file (reference tests/test_flashfftconv.py:48-324) are re-computed here the way that file computes them
(same seed, same input construction, forward + backward through autograd) by NPROC processes that time-slice ONE GPU -- the situation of the
pytest-xdist run in tests/test_reference_verbatim_gpu.py.  Every iteration evaluates the torch.fft reference TWICE and the HIP module TWICE and
compares each evaluation bit for bit with the process's first one; the reference test's own allclose gates are applied to every
(reference, module) pairing, so a flicker large enough to fail the test is counted and attributed to the side that moved.

Default cases: the two that failed once in the round-5 full run and passed alone -- test_flash_fft_conv[262144-dtype1-111-8] (bf16, B -> 4,
H 111) and [32768-dtype0-111-64] (fp16, B -> 16, H 111).
usage: python benchmarks/reffft_contention.py [iterations=200] [nproc=4] [kind:seqlen:dtype:H:B ...]     kind = plain | padded | gating | gating_padded
exit status 1 when the HIP module ever changed (a race in the product), 0 otherwise; the last line is a JSON summary."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["plain:262144:bfloat16:111:4", "plain:32768:float16:111:16"]


def worker(iters, cases):
    import torch
    sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
    from flashfftconv import FlashFFTConv

    def ref_fft_conv(uu, kk, n):      # the formula of the reference test's own oracle (reference tests/test_flashfftconv.py:5-13)
        out = torch.fft.ifft(torch.fft.fft(uu.to(torch.float32), n=n) * torch.fft.fft(kk.to(torch.float32), n=n), n=n)
        return out.real.to(uu.dtype)[..., :uu.size(-1)]
    res = {}
    for case in cases:
        kind, seqlen, dt, H, B = case.split(":")
        seqlen, H, B = int(seqlen), int(H), int(B)
        dtype = getattr(torch, dt)
        padded, gated = kind.endswith("padded"), kind.startswith("gating")
        L = seqlen // 2 if padded else seqlen
        torch.manual_seed(0)
        mk = lambda: torch.randn(B, H, L, device="cuda").to(dtype) * 0.02
        u = mk()
        gates = [mk(), mk()] if gated else []
        k = torch.randn(H, L, device="cuda") * 0.02
        if not padded:
            for t in [u] + gates: t[:, :, seqlen // 2:] = 0.
            k[:, seqlen // 2:] = 0.
        k = k * torch.exp(-0.1 * torch.arange(0, seqlen, device="cuda"))[:L]
        dout = torch.randn(B, H, L, device="cuda").to(dtype) * 0.02
        conv = FlashFFTConv(seqlen, dtype=dtype).to("cuda")
        ktol = 1e-1 if seqlen < 16 * 32768 else 1 if seqlen < 128 * 32768 else 2

        def ev(side):
            leaves = [t.clone().requires_grad_(True) for t in [u, k] + gates]
            if side == "ref":
                out = ref_fft_conv(leaves[0] * leaves[2], leaves[1], seqlen) * leaves[3] if gated else ref_fft_conv(leaves[0], leaves[1], seqlen)
            else:
                out = conv(*([leaves[0], leaves[1]] + leaves[2:]))
            out.backward(dout)
            return [out.detach()] + [t.grad for t in leaves]
        ref0, hip0 = ev("ref"), ev("hip")
        c = {"iters": iters, "ref_changed": 0, "hip_changed": 0, "ref_worst_rel": 0.0, "hip_worst_rel": 0.0, "gate_fail_ref_flicker": 0,
             "gate_fail_hip_flicker": 0, "gate_fail_both_stable": 0}
        tol = [1e-2, 1e-2, ktol, 1e-2, 1e-2]
        gate = lambda a, b: all(torch.allclose(x, y, atol=t) for x, y, t in zip(a, b, tol))
        same = lambda a, b: all(torch.equal(x, y) for x, y in zip(a, b))
        relmax = lambda a, b: max(((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-30)).item() for x, y in zip(a, b))
        for it in range(iters):
            for rep in range(2):
                r, h = ev("ref"), ev("hip")
                rs, hs = same(r, ref0), same(h, hip0)
                if not rs:
                    c["ref_changed"] += 1; c["ref_worst_rel"] = max(c["ref_worst_rel"], relmax(r, ref0))
                if not hs:
                    c["hip_changed"] += 1; c["hip_worst_rel"] = max(c["hip_worst_rel"], relmax(h, hip0))
                if not gate(h, r):           # the reference test's asserts on this pairing
                    c["gate_fail_ref_flicker" if not rs else "gate_fail_hip_flicker" if not hs else "gate_fail_both_stable"] += 1
        res[case] = c
        del conv, u, k, dout, gates, ref0, hip0
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(res), flush=True)


def run(iters=200, nproc=4, cases=None, out=sys.stdout):
    """-> {case: totals}; raises nothing: the caller decides (tests/test_reference_verbatim_gpu.py asserts hip_changed == 0)"""
    cases = list(cases or DEFAULT)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(iters)] + cases, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for _ in range(nproc)]
    print(f"# {nproc} processes time-slicing one GPU, {iters} iterations x 2 evaluations of either side per case; every evaluation compared bitwise "
          "with the process's first one", file=out)
    tot = {}
    for i, p in enumerate(procs):
        o = p.communicate()[0]
        got = [l for l in o.splitlines() if l.startswith("RESULT ")]
        if not got:
            print(f"process {i}: no result: {o[-600:]}", file=out)
            tot.setdefault("_errors", []).append(o[-600:])
            continue
        r = json.loads(got[0][7:])
        print(f"process {i}: {json.dumps(r)}", file=out)
        for case, c in r.items():
            t = tot.setdefault(case, {})
            for k2, v in c.items():
                t[k2] = max(t.get(k2, 0), v) if k2.endswith("_rel") else t.get(k2, 0) + v
    for case, t in tot.items():
        if case == "_errors":
            continue
        print(f"TOTAL {case}: {t['iters'] * 2} evaluations per side: torch.fft changed {t['ref_changed']}x (worst rel {t['ref_worst_rel']:.2e}), HIP changed "
              f"{t['hip_changed']}x (worst rel {t['hip_worst_rel']:.2e}); reference-test gates failed {t['gate_fail_ref_flicker']}x with a flickering "
              f"reference, {t['gate_fail_hip_flicker']}x with a flickering module, {t['gate_fail_both_stable']}x with both stable", file=out)
    return tot


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), sys.argv[3:])
        sys.exit(0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    nproc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tot = run(iters, nproc, sys.argv[3:] or None)
    print(json.dumps(tot))
    sys.exit(1 if any(isinstance(t, dict) and t.get("hip_changed") for t in tot.values()) or "_errors" in tot else 0)
