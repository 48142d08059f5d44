"""The multi-GPU wrappers (flashfftconv/sharding.py) around the PRODUCT: two ranks, one process each, both on cuda:0 of
the 1-GPU test box (the same hook bench.py has: FFC_BENCH_SAME_GPU), gloo as the transport because RCCL refuses two ranks
on one device.  Each rank compares its sharded HIP result with the single-rank HIP result of the same inputs
(forward bitwise: rows are independent; dk to 1e-3: the order of the fp32 partial sums differs) and with the torch.fft
oracle.  On a real node the only difference is backend="nccl" (RCCL over xGMI) and one device per rank."""
import os, socket
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    try:
        _body(rank, world, port, q)
    except Exception:
        import traceback
        q.put((rank, {"exception: " + traceback.format_exc()[-2000:]: False}))


def _body(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "flash-fft-conv_amd"), root]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flashfftconv import FlashFFTConv
    from flashfftconv.sharding import HeadShardedFFTConv, BatchShardedFFTConv, head_range
    from oracle.torch_ref import ref_fft_conv
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    ok = {}
    dev = torch.device("cuda", 0)
    cases = ((4096, 4, 16, 2048, False, True), (32768, 4, 10, 16384, True, True), (1024, 8, 7, 1024, False, True),
             (65536, 4, 6, 32768, False, True), (262144, 4, 6, 131072, False, True), (524288, 4, 4, 262144, True, True),
             # one level of 64 x 32768 (L <= N/2), also in the B-shard (round 4); B = 4 so that a rank's rows are whole pairs of the full
             # batch (a row packed with a zero partner differs from the same row packed with its neighbour by the bf16 rounding of the
             # shared spectrum, ~3e-3).  Round 5: these rows fit 1048576 points (FlashFFTConv._fit_seqlen); fit_fft off keeps the
             # 2097152-point form under test ...
             (2097152, 4, 2, 524288, False, False),
             # ... and this one runs fitted: 65536 points hold the rows of a 262144-point module, in the module, the H-shard and the B-shard alike
             (262144, 4, 4, 32768, False, True),
             # fft 131072 with rows longer than N/2: the single-rank module routes them to the HBM-level form (FlashFFTConv._route_big), and so
             # must the B-shard
             (131072, 4, 4, 131072, False, True))
    for (N, B, H, L, gated, fit) in cases[-int(os.environ.get("FFC_SHARD_TEST_LAST", len(cases))):]:
        torch.manual_seed(7)                      # same inputs on both ranks
        dt = torch.bfloat16
        mk = lambda: torch.randn(B, H, L, device=dev).to(dt)
        u, dout = mk(), mk()
        k = torch.randn(H, L, device=dev) * 0.1
        gates = [mk(), mk()] if gated else []
        mod = FlashFFTConv(N, dtype=dt).to(dev)
        mod.fit_fft = fit
        # single-rank HIP result + oracle
        lv = [t.clone().requires_grad_(True) for t in [u, k] + gates]
        full = mod(*lv)
        gfull = torch.autograd.grad(full, lv, dout)
        lo = [t.clone().requires_grad_(True) for t in [u, k] + gates]
        oref = ref_fft_conv(lo[0] * lo[2], lo[1], N) * lo[3] if gated else ref_fft_conv(lo[0], lo[1], N)
        goref = torch.autograd.grad(oref, lo, dout)
        tag = f"N{N}_L{L}"
        if fit and N > 32768 and 2 * L - 1 <= N // 2:
            ok[tag + "_runs_fitted"] = set(mod._fitted) == {mod._fit_seqlen(L, L)} and mod._fit_seqlen(L, L) < N
        # ---- H-shard with the differentiable gather
        hv = [t.clone().requires_grad_(True) for t in [u, k] + gates]
        y = HeadShardedFFTConv(mod, gather=True)(*hv)
        ok[tag + "_hshard_out_bitwise"] = torch.equal(y, full)
        s, e = head_range(H, rank, world)
        gy = torch.autograd.grad(y, hv, dout)     # both ranks push the same dout: shard grads are world x single-rank
        ok[tag + "_hshard_du"] = rel(gy[0][:, s:e], world * gfull[0][:, s:e]) < 1e-2 and float(gy[0][:, :s].abs().sum()) == 0.0
        ok[tag + "_hshard_dk"] = rel(gy[1][s:e], world * gfull[1][s:e]) < 1e-2
        # ---- B-shard: k_f all-gather / dk_f reduce-scatter (fused sizes), recompute + all-reduce (big sizes)
        b0, b1 = rank * B // world, (rank + 1) * B // world
        bv = [u[b0:b1].clone().requires_grad_(True), k.clone().requires_grad_(True)] + [g[b0:b1].clone().requires_grad_(True) for g in gates]
        # "/2g": the head-group pipeline (two groups; every launch on an in-place head range, collectives in flight between them)
        for mode in (("allgather_kf", "allgather_kf/2g", "recompute") if N <= 32768 else ("allgather_kf", "allgather_kf/2g") if N <= 131072 else ("allgather_kf",)):
            bs = BatchShardedFFTConv(mod, mode=mode.split("/")[0], groups=2 if mode.endswith("/2g") else 1)
            yl = bs(*bv)
            # bitwise where a pair meets the same kernel code alone as in the full batch; the inner sizes 8192 / 16384 run two
            # pairs of a head in lock-step when they have them (Body::inner_tile2x) and one pair alone otherwise: same
            # arithmetic, the compiler contracts it differently, results move by single steps of the dtype
            # round 4: where a workgroup owns its head the module transforms the filter inside its forward launch (Modes::kfft_head),
            # elsewhere -- other batch split, the exchange mode's ops.kernel_fft -- in the kernel of its own: the same source in two
            # kernels, whose fp32 twiddle products the compiler may contract differently (~1e-5 of the k_f values land on the
            # neighbouring bf16, tests/test_spectrum_gpu.py); then "equal" means equal to that rounding
            same = (lambda a, b: torch.equal(a, b) or rel(a, b) < 2e-3) if N < 262144 else (lambda a, b: rel(a, b) < 4e-3)
            ok[f"{tag}_bshard_{mode}_out_bitwise"] = same(yl, full[b0:b1])
            ok[f"{tag}_bshard_{mode}_out_vs_oracle"] = rel(yl, oref[b0:b1]) < 2e-2
            gl = torch.autograd.grad(yl, bv, dout[b0:b1])
            ok[f"{tag}_bshard_{mode}_du_bitwise"] = same(gl[0], gfull[0][b0:b1])
            # FULL dk on every rank.  allgather_kf: one inverse of the reduced fp32 sums, like the single-rank run;
            # recompute: each rank inverts ITS partial sums (bf16 operands, 2^-9) and the results are all-reduced
            dk_tol = 2e-3 if (mode == "allgather_kf" and N <= 32768) else 8e-3 if N <= 131072 else 1.6e-2
            ok[f"{tag}_bshard_{mode}_dk_vs_single"] = rel(gl[1], gfull[1]) < dk_tol
            ok[f"{tag}_bshard_{mode}_dk_vs_oracle"] = rel(gl[1], goref[1]) < 3e-2
            if gated:
                ok[f"{tag}_bshard_{mode}_dgates_bitwise"] = same(gl[2], gfull[2][b0:b1]) and same(gl[3], gfull[3][b0:b1])
        if N >= 262144:      # fft sizes with HBM levels exchange the k_f / dk_f rows of their inner size too (no silent recompute)
            from flashfftconv import sharding as SH
            ok[f"{tag}_bshard_exchanges_rows"] = isinstance((SH._BigOps if mod._big else SH._HipOps)(mod, dev), SH._BigOps)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_product_two_ranks_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps: p.join(60)
    for rank, ok in res:
        bad = [k for k, v in ok.items() if not v]
        assert not bad, (rank, bad)


def test_bench_two_ranks_through_the_driver_launch_line():
    """bench.py under the driver's N > 1 launch (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2),
    both ranks on cuda:0 (FFC_BENCH_SAME_GPU) over gloo: one JSON line from rank 0; `value` is the FIXED problem of
    BASELINE's metric (B16 x H768 in total, "scaling": "strong"), the weak-scaled job of the same run rides in `weak`."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFC_BENCH_SAME_GPU="1", FFC_BENCH_BACKEND="gloo", FFC_BENCH_STRONG_ROWS="sweep L=1024,sweep L=131072,cfg4 H-sharded")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and not l.startswith('{"table"')]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-2000:]
    d = json.loads(lines[0])
    assert len(lines[0]) < 4096
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong" and d["unit"] == "seq/s"
    assert d["value"] > 0 and abs(d["value"] - 16 * 768 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    assert d["strong"]["heads_per_rank"] == 384 and abs(d["strong"]["value"] - d["value"]) < 1e-2 * d["value"]
    assert d["weak"]["heads_per_rank"] == 768 and abs(d["weak"]["value"] - 2 * 16 * 768 / (d["weak"]["ms_per_step"] * 1e-3)) < 1e-2 * d["weak"]["value"]
    assert "cpu_baseline" not in d and "sweep_fwd_bwd_ms" not in d          # rank 0 at N = 1 only
    # round 6 (VERDICT r05 missing #4): the N > 1 run also carries the metric's other rows, strong-scaled -- the fixed problem's heads split
    # over the ranks, fwd+bwd ms (max over ranks) and heads per rank on the contract line, the full rows as table lines before it
    sr = d["strong_rows_step_ms"]
    assert set(sr) == {"sweep L=1024", "sweep L=131072", "cfg4 H-sharded"}
    assert sr["sweep L=1024"][1] == 384 and sr["cfg4 H-sharded"][1] == 8 and all(v[0] > 0 for v in sr.values())
    tables = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"table"')]
    rows = {t["row"]: t for t in tables if t["table"] == "strong_rows"}
    assert rows["cfg4 H-sharded"]["fft_run"] == 2097152 and rows["cfg4 H-sharded"]["n_gpus"] == 2 and rows["sweep L=131072"]["scaling"] == "strong"
    assert d["tflops_fft_equiv"] > 0 and d["tflops_dense_monarch"] > 0
