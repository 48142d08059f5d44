"""world_size-2 gloo tests of the multi-GPU partitions (flashfftconv/sharding.py) on CPU: the collective / partition logic
with a torch.fft stand-in as the compute (the HIP compute under the same wrappers is tests/test_sharding_gpu.py)."""
import os, socket
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class OracleOps:
    """CPU stand-in for sharding._HipOps: the same four operations written with torch.fft + autograd."""
    GENERIC_SLICES = True        # head groups of the pipelined exchange run on copies of the head range

    def __init__(self, N):
        self.N = N

    def kernel_fft(self, k):
        return torch.view_as_real(torch.fft.fft(k.double(), n=self.N)).contiguous()

    def _conv(self, u, kf, pre, post):
        x = u.double() if pre is None else u.double() * pre.double()
        y = torch.fft.ifft(torch.fft.fft(x, n=self.N) * torch.view_as_complex(kf), n=self.N).real[..., : u.shape[-1]]
        return y if post is None else y * post.double()

    def conv(self, u, kf, pre, post):
        return self._conv(u, kf, pre, post).to(u.dtype)

    @torch.enable_grad()
    def backward(self, dout, u, kf, pre, post):
        leaves = [t.detach().clone().requires_grad_(True) for t in ((u, kf) if pre is None else (u, kf, pre, post))]
        y = self._conv(leaves[0], leaves[1], *(leaves[2:] if pre is not None else (None, None)))
        g = torch.autograd.grad(y, leaves, dout.double())
        return g[0].to(u.dtype), (g[2].to(u.dtype) if pre is not None else None), (g[3].to(u.dtype) if pre is not None else None), g[1]

    @torch.enable_grad()
    def dk_from_dkf(self, dkf, Lk):
        k0 = torch.zeros(dkf.shape[0], Lk, dtype=torch.float64, requires_grad=True)
        (dk,) = torch.autograd.grad(self.kernel_fft(k0), k0, dkf)
        return dk.float()


def _worker(rank, world, port, H, q):
    try:
        _worker_body(rank, world, port, H, q)
    except Exception as ex:            # report instead of leaving the parent waiting for the queue
        import traceback
        q.put((rank, {"exception: " + traceback.format_exc()[-1500:]: False}, (0, 0)))


def _worker_body(rank, world, port, H, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "flash-fft-conv_amd"), root]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flashfftconv.sharding import HeadShardedFFTConv, BatchShardedFFTConv, head_range
    from oracle.torch_ref import ref_fft_conv
    torch.manual_seed(0)
    B, L, N = 4, 64, 128
    u = torch.randn(B, H, L); k = torch.randn(H, L); dout = torch.randn(B, H, L)
    pre, post = torch.randn(B, H, L), torch.randn(B, H, L)
    ok = {}
    # ---- H-shard: local result, gathered result, and gradients THROUGH the gather (it used to detach)
    uc, kc = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
    full = ref_fft_conv(uc, kc, N)
    full.backward(dout)
    s, e = head_range(H, rank, world)
    local = HeadShardedFFTConv(lambda a, b: ref_fft_conv(a, b, N))(u, k)
    ok["hshard_local"] = torch.allclose(local, full[:, s:e].detach(), atol=1e-5)
    ug, kg = u.clone().requires_grad_(True), k.clone().requires_grad_(True)
    y = HeadShardedFFTConv(lambda a, b: ref_fft_conv(a, b, N), gather=True)(ug, kg)
    ok["hshard_gather"] = torch.allclose(y, full.detach(), atol=1e-5)
    # every rank applies the same dout to its full copy: the shard's gradient is world x the single-process one
    y.backward(dout)
    ok["hshard_gather_grad"] = (torch.allclose(ug.grad[:, s:e], world * uc.grad[:, s:e], atol=1e-4)
                                and torch.allclose(kg.grad[s:e], world * kc.grad[s:e], atol=1e-3)
                                and float(ug.grad[:, :s].abs().sum() + ug.grad[:, e:].abs().sum()) == 0.0)
    # ---- B-shard: batch rows split, k replicated; k_f all-gathered, dk_f reduce-scattered, dk all-gathered
    b0, b1 = rank * B // world, (rank + 1) * B // world
    # spy on the collectives: the exchange mode must issue every one of them with async_op=True (waited for where the result is
    # first needed, sharding._Pending), never synchronously in the middle of the compute
    issued = []
    for name in ("all_gather_into_tensor", "reduce_scatter_tensor", "all_gather", "all_reduce"):
        def make(fn, name):
            def spy(*a, **kw):
                issued.append((name, bool(kw.get("async_op", False))))
                return fn(*a, **kw)
            return spy
        setattr(dist, name, make(getattr(dist, name), name))
    for gated in (False, True):
        for mode in ("allgather_kf", "allgather_kf/2groups", "recompute"):
            ngroups = 2 if mode.endswith("2groups") else 1
            tagmode, mode = mode.replace("/", "_"), mode.split("/")[0]
            issued.clear()
            leaves = [t.clone().requires_grad_(True) for t in ((u, k, pre, post) if gated else (u, k))]
            ref = ref_fft_conv(leaves[0] * leaves[2], leaves[1], N) * leaves[3] if gated else ref_fft_conv(leaves[0], leaves[1], N)
            ref.backward(dout)
            ul = u[b0:b1].clone().requires_grad_(True); kl = k.clone().requires_grad_(True)
            gl = [t[b0:b1].clone().requires_grad_(True) for t in (pre, post)] if gated else []
            ops = OracleOps(N)
            conv = BatchShardedFFTConv((lambda a, b, p=None, q=None: ops.conv(a, ops.kernel_fft(b), p, q)) if mode == "recompute" else None,
                                       mode=mode, ops=ops if mode == "allgather_kf" else None, groups=ngroups)
            if mode == "recompute":
                # differentiable stand-in for the module: oracle conv through autograd
                conv.conv = (lambda a, b, p=None, q=None: (ref_fft_conv(a * p, b, N) * q) if p is not None else ref_fft_conv(a, b, N))
            yl = conv(ul, kl, *gl)
            yl.backward(dout[b0:b1])
            tag = f"bshard_{tagmode}_{'gated' if gated else 'plain'}"
            if mode == "allgather_kf":
                # per head group: k_f all-gather, dk_f reduce(-scatter), dk all-gather -- all asynchronous
                ok[tag + "_async"] = len(issued) == 3 * ngroups and all(a for _, a in issued)
            ok[tag + "_out"] = torch.allclose(yl, ref[b0:b1].detach(), atol=1e-4)
            ok[tag + "_du"] = torch.allclose(ul.grad, leaves[0].grad[b0:b1], atol=1e-4)
            ok[tag + "_dk"] = torch.allclose(kl.grad, leaves[1].grad, atol=1e-3)          # FULL gradient on every rank
            if gated:
                ok[tag + "_dgates"] = (torch.allclose(gl[0].grad, leaves[2].grad[b0:b1], atol=1e-4)
                                       and torch.allclose(gl[1].grad, leaves[3].grad[b0:b1], atol=1e-4))
    q.put((rank, ok, (s, e)))
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [8, 5])
def test_sharding_gloo(H):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, H, q)) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps: p.join(60)
    for rank, ok, _ in res:
        bad = [k for k, v in ok.items() if not v]
        assert not bad, (rank, bad)
    ranges = sorted(r for _, _, r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == H and ranges[0][1] == ranges[1][0]
