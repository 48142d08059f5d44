"""The first RCCL execution of the multi-GPU code: backend "nccl" (= RCCL on ROCm) with world_size = 1 on
cuda:0.  Every box this repo is built and tested on has ONE GPU, and RCCL refuses two ranks on one device, so the two-rank tests
(tests/test_sharding_gpu.py, tests/test_sharding_cpu.py) run over gloo; here the SAME wrappers run over the backend the product names.
With one rank every collective is an identity, so the sharded modules must reproduce the single-rank module -- what this exercises is the
RCCL call path itself: the capability probe (sharding._caps: all_gather_into_tensor / reduce_scatter_tensor really executed by RCCL), async
work handles waited for on torch's stream (_Pending), the padded tensor collectives, fp32 reduce-scatter of the dk_f sums, bf16 all-gather of
k_f, the head-group pipeline, and the H-shard's differentiable gather.  No scaling claim follows from it: no multi-GPU run has been measured."""
import os, socket
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(q):
    import sys, traceback
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path[:0] = [os.path.join(root, "flash-fft-conv_amd"), root]
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        from flashfftconv import FlashFFTConv, sharding as SH
        from oracle.torch_ref import ref_fft_conv
        rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
        ok = {"backend_is_nccl": dist.get_backend() == "nccl"}
        dev = torch.device("cuda", 0)
        # the probe must find RCCL's tensor collectives (a backend without them falls back to list collectives / all-reduce)
        ag, rs = SH._caps(None, torch.zeros(4, device=dev))
        ok["caps_tensor_collectives"] = bool(ag) and bool(rs)
        # raw collectives of the wrappers, uneven head counts -> the padded forms
        x = torch.randn(7, 5, device=dev)
        p = SH._all_gather_heads_async(x, 7); ok["all_gather_heads"] = torch.equal(p.wait(), x)
        y = torch.randn(7, 5, device=dev)
        p = SH._reduce_scatter_heads_async(y.clone()); ok["reduce_scatter_heads"] = torch.equal(p.wait(), y)
        for (N, B, H, L, gated) in ((32768, 4, 10, 16384, True), (4096, 4, 16, 2048, False), (65536, 2, 6, 32768, False), (262144, 2, 4, 131072, False)):
            torch.manual_seed(3)
            dt = torch.bfloat16
            mk = lambda: torch.randn(B, H, L, device=dev).to(dt)
            u, dout = mk(), mk()
            k = torch.randn(H, L, device=dev) * 0.1
            gates = [mk(), mk()] if gated else []
            mod = FlashFFTConv(N, dtype=dt).to(dev)
            lv = [t.clone().requires_grad_(True) for t in [u, k] + gates]
            full = mod(*lv)
            gfull = torch.autograd.grad(full, lv, dout)
            lo = [t.clone().requires_grad_(True) for t in [u, k] + gates]
            oref = ref_fft_conv(lo[0] * lo[2], lo[1], N) * lo[3] if gated else ref_fft_conv(lo[0], lo[1], N)
            tag = f"N{N}"
            ok[tag + "_single_vs_oracle"] = rel(full, oref) < 2e-2
            hv = [t.clone().requires_grad_(True) for t in [u, k] + gates]
            yh = SH.HeadShardedFFTConv(mod, gather=True)(*hv)
            gh = torch.autograd.grad(yh, hv, dout)
            ok[tag + "_hshard_bitwise"] = torch.equal(yh, full) and all(torch.equal(a, b) for a, b in zip(gh, gfull))
            for mode, groups in (("allgather_kf", 1), ("allgather_kf", 2)) if N <= 131072 else (("allgather_kf", 1),):
                bv = [t.clone().requires_grad_(True) for t in [u, k] + gates]
                yb = SH.BatchShardedFFTConv(mod, mode=mode, groups=groups)(*bv)
                gb = torch.autograd.grad(yb, bv, dout)
                # (k_f made by the exchange path's kernel, the module's inside its forward launch: equal to the k_f rounding, see test_sharding_gpu)
                ok[f"{tag}_bshard_g{groups}_out"] = torch.equal(yb, full) or rel(yb, full) < 4e-3
                ok[f"{tag}_bshard_g{groups}_du"] = torch.equal(gb[0], gfull[0]) or rel(gb[0], gfull[0]) < 4e-3
                ok[f"{tag}_bshard_g{groups}_dk"] = rel(gb[1], gfull[1]) < 1.6e-2
                if gated:
                    ok[f"{tag}_bshard_g{groups}_dgates"] = rel(gb[2], gfull[2]) < 4e-3 and rel(gb[3], gfull[3]) < 4e-3
        torch.cuda.synchronize()
        dist.destroy_process_group()
        q.put(ok)
    except Exception:
        q.put({"exception: " + traceback.format_exc()[-2500:]: False})


def test_sharded_modules_over_rccl_world_size_one():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_run, args=(q,))
    p.start()
    ok = q.get(timeout=600)
    p.join(60)
    bad = [k for k, v in ok.items() if not v]
    assert not bad, bad
    assert ok.get("backend_is_nccl") and ok.get("caps_tensor_collectives")
