"""Benchmark of the FlashFFTConv hot path on MI355X (driver contract: see task statement).

Step = one forward + backward of FlashFFTConv(32768) on BASELINE.json configs[1]
(B=16, H=768, L=16384, bf16 activations, fp32 k), synthetic randn inputs resident in HBM:
   k -> k_f (kfft kernel), conv forward, fused backward (du + fp32 dk_f partial sums), dk inverse.
Nothing is cached between steps (the reference recomputes k_f every forward, conv.py:572-575).
N > 1 GPUs: one process per GPU, heads sharded (weak scaling: every rank runs the full per-GPU shape
on its own heads, no data-path collective); barrier + max-over-ranks timing.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "flash-fft-conv_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

CFG = dict(N=32768, B=16, H=768, L=16384, dtype=torch.bfloat16)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16


def flops_dense_fwd_per_row(N):
    # reference factorisation 32x32x32, r2c first / c2r last stage
    n = [32, 32, 32]; p = 3
    return 2 * (sum(8 * N * ni for ni in n) - 4 * N * n[0]) + 6 * N * (2 * (p - 1) + 1)


def flops_fft_equiv(N):
    import math
    lg = math.log2(N)
    return 2 * 5 * N * lg + 6 * N, 3 * 5 * N * lg + 14 * N


def time_kernel(fn, iters=20):
    """HIP-event timing on torch's current stream (the stream the library launches on)."""
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def cpu_baseline(seconds_target=12.0):
    """Reference CPU path (torch.fft oracle, oracle/torch_ref.py) fwd+bwd on an H-slice of the workload."""
    from oracle.torch_ref import ref_fft_conv
    torch.set_num_threads(os.cpu_count() or 1)
    B, L, N, Hs = CFG["B"], CFG["L"], CFG["N"], 48
    g = torch.Generator().manual_seed(0)
    u = torch.randn(B, Hs, L, generator=g).to(CFG["dtype"]).requires_grad_(True)
    k = torch.randn(Hs, L, generator=g).requires_grad_(True)
    dout = torch.randn(B, Hs, L, generator=g).to(CFG["dtype"])

    def step():
        u.grad = None; k.grad = None
        ref_fft_conv(u, k, n=N).backward(dout)

    step()
    t0 = time.perf_counter(); reps = 0
    while True:
        step(); reps += 1
        el = time.perf_counter() - t0
        if el > seconds_target or reps >= 50:
            break
    return {"value": B * Hs * reps / el, "unit": "seq/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"fwd+bwd, B={B} H={Hs} (1/16 of H=768) L={L} fft={N}, {reps} reps in {el:.1f}s, torch.fft oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if int(os.environ.get("FFC_BENCH_SAME_GPU", "0")):     # test hook: all ranks on cuda:0
            local = 0
        torch.cuda.set_device(local)
        # RCCL ("nccl" on ROCm) over xGMI; FFC_BENCH_BACKEND=gloo only exists to exercise this path on a 1-GPU box
        dist.init_process_group(os.environ.get("FFC_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)

    from flashfftconv import FlashFFTConv
    from flashfftconv import conv as C, _lib

    N, B, H, L, dtype = CFG["N"], CFG["B"], CFG["H"], CFG["L"], CFG["dtype"]
    torch.manual_seed(rank)
    u = torch.randn(B, H, L, device=dev).to(dtype).requires_grad_(True)
    k = torch.randn(H, L, device=dev).requires_grad_(True)
    dout = torch.randn(B, H, L, device=dev).to(dtype)
    mod = FlashFFTConv(N, dtype=dtype).to(dev)

    def step():
        u.grad = None; k.grad = None
        mod(u, k).backward(dout)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()

    # ---- per-kernel timing (rank 0 reports): the four launches of one step
    plan = mod._get_plan(dev)
    ud, kd = u.detach(), k.detach()
    kf = C._kernel_fft(plan, kd)
    lib = _lib.lib()
    ws = torch.empty(lib.ffc_dkf_workspace_bytes(plan.handle, B, H), dtype=torch.uint8, device=dev)
    dk = torch.empty(H, L, dtype=torch.float32, device=dev)
    dk_du = torch.empty_like(ud)
    sp = _lib.stream_ptr
    kt = {
        "kfft": time_kernel(lambda: C._kernel_fft(plan, kd)),
        "conv_fwd": time_kernel(lambda: C._conv(plan, ud, kf, None, None, False)),
        "conv_dx": time_kernel(lambda: C._conv(plan, dout, kf, None, None, True)),
        "bwd_fused": time_kernel(lambda: _lib.check(lib.ffc_conv_bwd(plan.handle, _lib.ptr(dout), _lib.ptr(ud), _lib.ptr(kf), None, None, _lib.ptr(dk_du), None, _lib.ptr(ws), B, H, L, sp()), "bwd")),
        "dkf_only": time_kernel(lambda: _lib.check(lib.ffc_conv_bwd_dkf(plan.handle, _lib.ptr(dout), _lib.ptr(ud), None, None, _lib.ptr(ws), B, H, L, sp()), "dkf")),
        "dk_ifft": time_kernel(lambda: _lib.check(lib.ffc_kernel_ifft_grad(plan.handle, _lib.ptr(ws), B, H, L, _lib.ptr(dk), sp()), "dk")),
    }
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    rows = B * H
    sec_per_step = el / args.steps
    seq_s = world * rows / sec_per_step
    dense_fwd = flops_dense_fwd_per_row(N)
    fft_fwd, fft_bwd = flops_fft_equiv(N)
    # roofline objects: `roofline` = the dominant kernel of the step (bwd_kernel, ~60 % of the step time),
    # `roofline_fwd` = the forward conv_kernel.  Flop basis = SURVEY 8(d): dense Monarch count of the reference's 32x32x32
    # factorisation (forward 42.9 MFLOP/row; backward = 1.5 x the forward matmul flops + 14 N pointwise = 63.4 MFLOP/row),
    # x rows per launch; our pair-packed kernels execute about half of these.
    matmul_fwd = dense_fwd - 30 * N
    dense_bwd = 1.5 * matmul_fwd + 14 * N
    t_conv, t_bwd = kt["conv_fwd"], kt["bwd_fused"]
    fwd_bytes = B * H * L * 2 * 2 + H * N * 4                      # read u, write y, read k_f once
    bwd_bytes = B * H * L * 2 * 3 + H * N * 4 + H * N * 8          # read u + dout, write du, read k_f, write fp32 dk_f

    def roof(name, flops_row, alg_bytes, t, traffic, src):
        fl = flops_row * rows
        return {"kernel": name, "bound": "mfma", "achieved": fl / t / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / t / 1e12 / MFMA_PEAK_TFLOPS,
                # L2<->fabric bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE);
                # measured by tests/measure_r01_end.sh / tests/pmc_bwd.sh, not re-measured inside bench.py
                "traffic": traffic, "traffic_source": src, "launch_ms": t * 1e3,
                "alg_bytes": alg_bytes, "hbm_GBs": alg_bytes / t / 1e9, "hbm_frac": alg_bytes / t / 1e9 / HBM_PEAK_GBS}

    roof_bwd = roof("bwd_kernel<Geo<32,32,32>,bf16,HALF> (fused backward: du + fp32 dk_f)", dense_bwd, bwd_bytes, t_bwd,
                    3826.4e6, "profiles/r01_end_pmc_bwd_kernel.txt")
    roof_fwd = roof("conv_kernel<Geo<32,32,32>,bf16,HALF> (forward)", dense_fwd, fwd_bytes, t_conv,
                    1611.7e6, "profiles/r01_end_pmc_conv_kernel.txt")
    roof_bwd["basis"] = roof_fwd["basis"] = "SURVEY 8(d) dense-Monarch flops of the reference factorisation x rows per launch"
    out = {
        "metric": "FFT-conv fwd+bwd seq/s, B=16 H=768 L=16384 fft=32768 bf16",
        "value": seq_s, "unit": "seq/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "FlashFFTConv(32768) B=16 H=768 L=16384 bf16, fwd+bwd incl. k->k_f and dk (BASELINE configs[1])",
                   "per_gpu_rows": rows, "parallelism": f"head-shard x{world} (no collective)"},
        "tflops_dense_monarch": world * rows * dense_fwd * 2.5 / sec_per_step / 1e12,
        "tflops_fft_equiv": world * rows * (fft_fwd + fft_bwd) / sec_per_step / 1e12,
        "kernel_ms": {n: v * 1e3 for n, v in kt.items()},
        "roofline": roof_bwd,
        "roofline_fwd": roof_fwd,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
