"""Multi-GPU execution of the FFT-conv path.  One process per GPU, torch.distributed over
RCCL/xGMI (backend "nccl"); every (b, h) row is independent given k[h], so there are two ways to split:

  H-shard (HeadShardedFFTConv)   rank r owns heads [r*H/W, (r+1)*H/W) of u, k, the gates, y and dk.
                                 NO data-path collective; optional all-gather of y for callers that need it replicated.
  B-shard (BatchShardedFFTConv)  data parallel: rank r owns batch rows, k is replicated (a DDP-style parameter).
                                 forward : every rank transforms H/W heads of k on the device and the ranks ALL-GATHER
                                           k_f (cfg2: 100.7 MB total) instead of each recomputing all of FFT(k);
                                 backward: each rank's fp32 dk_f partial sums (over ITS batch rows, all heads) are
                                           REDUCE-SCATTERed, each rank inverts its H/W heads, dk is all-gathered so the
                                           replicated parameter gets the full gradient (what DDP's all-reduce would give).
                                 mode="recompute" skips the k_f exchange (every rank runs the whole FFT(k), dk is
                                 all-reduced): cheaper than the collective's latency for short fft sizes.  The exchange mode
                                 covers every fft size: the multi-pass plans (65536, 131072) are fused plans like the
                                 smaller ones, the HBM-level sizes (>= 262144) exchange the rows of their inner size
                                 (_BigOps); only the folded and the frequency-sparse forms fall back to recompute.

The compute goes through an `ops` object (GPU: _HipOps over the C-ABI; CPU tests: a torch.fft stand-in) so the collective
logic is exercised by world_size-2 gloo tests without a GPU.  The reference has no distributed code; this is new."""
import torch
import torch.distributed as dist


def head_range(H, rank, world):
    """Contiguous, balanced head partition (first H % world ranks get one extra head)."""
    base, rem = divmod(H, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_heads(x, dim, rank=None, world=None):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    s, e = head_range(x.shape[dim], rank, world)
    return x.narrow(dim, s, e - s).contiguous()


def _all_gather_uneven(x_local, H, dim, group=None):
    """All-gather head shards back to the full tensor (uneven shards supported via padding)."""
    world = dist.get_world_size(group)
    sizes = [head_range(H, r, world)[1] - head_range(H, r, world)[0] for r in range(world)]
    m = max(sizes)
    if all(s == m for s in sizes) and dim == 0 and x_local.is_contiguous():
        out = x_local.new_empty((H,) + tuple(x_local.shape[1:]))
        try:        # one tensor collective (RCCL; gloo has it for CPU tensors)
            dist.all_gather_into_tensor(out, x_local, group=group)
        except (RuntimeError, NotImplementedError):
            dist.all_gather(list(out.chunk(world, 0)), x_local, group=group)
        return out
    pad_shape = list(x_local.shape); pad_shape[dim] = m
    buf = x_local.new_zeros(pad_shape)
    buf.narrow(dim, 0, x_local.shape[dim]).copy_(x_local)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return torch.cat([o.narrow(dim, 0, s) for o, s in zip(outs, sizes)], dim=dim)


def _reduce_scatter_heads(x_full, dim, group=None):
    """sum over ranks of x_full, this rank's head range of the result."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H = x_full.shape[dim]
    s, e = head_range(H, rank, world)
    if H % world == 0 and dim == 0 and x_full.is_contiguous():
        out = x_full.new_empty((H // world,) + tuple(x_full.shape[1:]))
        try:        # one tensor collective (RCCL; gloo has it for CPU tensors)
            dist.reduce_scatter_tensor(out, x_full, group=group)
            return out
        except (RuntimeError, NotImplementedError):
            pass
    x = x_full.clone()
    dist.all_reduce(x, group=group)          # gloo (CPU tests / same-GPU tests) and uneven shards
    return x.narrow(dim, s, e - s).contiguous()


def _all_gather_rows(x_local, H, rows, group=None):
    """all-gather of head shards whose dim 0 is (head, row) with `rows` rows per head"""
    if rows == 1:
        return _all_gather_uneven(x_local, H, 0, group)
    full = _all_gather_uneven(x_local.reshape((-1, rows) + tuple(x_local.shape[1:])), H, 0, group)
    return full.reshape((H * rows,) + tuple(x_local.shape[1:]))


class _GatherHeads(torch.autograd.Function):
    """all-gather along `dim`, differentiable: every rank holds (and may use) the whole gathered tensor, so the gradient
    of a rank's shard is the SUM over ranks of that shard's slice of their gradients (reduce-scatter)."""

    @staticmethod
    def forward(ctx, x_local, H, dim, group):
        ctx.dim, ctx.group = dim, group
        return _all_gather_uneven(x_local.contiguous(), H, dim, group)

    @staticmethod
    def backward(ctx, g):
        return _reduce_scatter_heads(g.contiguous(), ctx.dim, ctx.group), None, None, None


def gather_heads(x_local, H, dim, group=None):
    """Differentiable all-gather of head shards (round 1 used dist.all_gather directly, which detached the result)."""
    return _GatherHeads.apply(x_local, H, dim, group)


class HeadShardedFFTConv(torch.nn.Module):
    """Wraps a conv callable `conv(u, k, pregate, postgate)`; each rank computes its head shard.
    `gather=True` returns the full (B,H,L) output on every rank (differentiable all-gather over xGMI)."""

    def __init__(self, conv, gather=False, group=None):
        super().__init__()
        self.conv, self.gather, self.group = conv, gather, group

    def forward(self, u, k, pregate=None, postgate=None):
        H = u.shape[1]
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        ul, kl = shard_heads(u, 1, rank, world), shard_heads(k, 0, rank, world)
        pl = None if pregate is None else shard_heads(pregate, 1, rank, world)
        ql = None if postgate is None else shard_heads(postgate, 1, rank, world)
        y = self.conv(ul, kl, pl, ql) if pl is not None else self.conv(ul, kl)
        return gather_heads(y, H, 1, self.group) if self.gather else y


# ------------------------------------------------------------------------------------------------ B-shard
class _HipOps:
    """GPU backend of the B-shard: the C-ABI entry points of one fused plan."""

    def __init__(self, mod, device):
        from . import conv as C, _lib
        self.C, self.L, self.mod = C, _lib, mod
        self.plan = mod._get_plan(device, mod._plan_seqlen)

    def kernel_fft(self, k):                       # (h, Lk) fp32 -> (h, kf_elems, 2) dtype, internal order
        if k.shape[0] == 0:                        # more ranks than heads: an empty shard
            return torch.empty(0, self.plan.kf_elems, 2, dtype=self.plan.dtype, device=k.device)
        return self.C._kernel_fft(self.plan, k)

    def conv(self, u, kf, pre, post):
        return self.C._conv(self.plan, u, kf, pre, post, False)

    def conv_keep(self, u, kf, pre, post):
        """training forward: (out, kept) with kept = (spectra, output before the postgate) or None (FlashFFTConv.save_spectrum)"""
        z = self.C._spectrum_buffer(self.plan, u.shape[0], u.shape[1], u.device, pre is not None) if self.mod.save_spectrum else None
        if z is None:
            return self.conv(u, kf, pre, post), None
        yraw = torch.empty_like(u) if pre is not None else None
        return self.C._conv_save(self.plan, u, kf, pre, post, z, yraw), (z, yraw)

    def backward(self, dout, u, kf, pre, post, kept=None):
        """-> du, dpre, dpost, dk_f (H, kf_elems, 2) fp32 summed over the local batch"""
        lib, L = self.L.lib(), self.L
        B, H, Lu = u.shape
        ws = torch.empty(lib.ffc_dkf_workspace_bytes(self.plan.handle, B, H), dtype=torch.uint8, device=u.device)
        du = torch.empty_like(u)
        dpre = torch.empty_like(u) if pre is not None else None
        if kept is not None:
            z, yraw = kept
            dpost = dout * yraw if pre is not None else None
            L.check(lib.ffc_conv_bwd_z(self.plan.handle, L.ptr(dout), L.ptr(u), L.ptr(kf), L.ptr(pre), L.ptr(post), L.ptr(du), L.ptr(dpre),
                                       None, L.ptr(ws), L.ptr(z), B, H, Lu, 0, 0, 0, 0, 0, 0, 0, L.stream_ptr()), "ffc_conv_bwd_z")
        else:
            dpost = torch.empty_like(u) if pre is not None else None
            L.check(lib.ffc_conv_bwd_gated(self.plan.handle, L.ptr(dout), L.ptr(u), L.ptr(kf), L.ptr(pre), L.ptr(post), L.ptr(du),
                                           L.ptr(dpre), L.ptr(dpost), L.ptr(ws), B, H, Lu, L.stream_ptr()), "ffc_conv_bwd_gated")
        nslab = lib.ffc_dkf_slab_count(self.plan.handle, B, H)
        nfl = H * self.plan.kf_elems * 2
        slabs = ws[: nslab * nfl * 4].view(torch.float32).view(nslab, H, self.plan.kf_elems, 2)
        return du, dpre, dpost, (slabs[0] if nslab == 1 else slabs.sum(0))

    def dk_from_dkf(self, dkf, Lk):                # (h, kf_elems, 2) fp32 -> (h, Lk) fp32
        L = self.L
        h = dkf.shape[0]
        dk = torch.empty(h, Lk, dtype=torch.float32, device=dkf.device)
        if h == 0:
            return dk
        L.check(L.lib().ffc_kernel_ifft_grad_slabs(self.plan.handle, L.ptr(dkf.contiguous()), 1, h, Lk, L.ptr(dk), L.stream_ptr()),
                "ffc_kernel_ifft_grad_slabs")
        return dk


class _BigOps:
    """GPU backend of the B-shard for the HBM-level sizes (fft >= 262144): the exchanged tensors are the k_f / dk_f rows of
    the inner size, `rows` per head (head-major), so the head partition carries over as a row partition."""

    def __init__(self, mod, device):
        from . import conv as C, bigfft
        self.C, self.mod = C, mod
        factors, self.M = bigfft.BIG_FACTORS[mod.seqlen]
        self.rows = 1
        for n0 in factors:
            self.rows *= n0

    def kernel_fft(self, k):
        if k.shape[0] == 0:                        # more ranks than heads: an empty shard
            plan = self.mod._get_plan(k.device, self.M)
            return torch.empty(0, plan.kf_elems, 2, dtype=self.mod.dtype, device=k.device)
        return self.C._big_kernel_fft(self.mod, k)

    def conv(self, u, kf, pre, post):
        return self.C._big_forward(self.mod, u, None, pre, post, False, kf)[0]

    def conv_keep(self, u, kf, pre, post):
        out, _, kept = self.C._big_forward(self.mod, u, None, pre, post, bool(self.mod.save_spectrum), kf)
        return out, kept

    def backward(self, dout, u, kf, pre, post, kept=None):
        du, dkf, dpre, dpost = self.C._big_backward(self.mod, dout, u, kf, pre, post, 0, kept, True)
        return du, dpre, dpost, dkf

    def dk_from_dkf(self, dkf, Lk):
        if dkf.shape[0] == 0:
            return torch.empty(0, Lk, dtype=torch.float32, device=dkf.device)
        return self.C._big_dk_from_dkf(self.mod, dkf, Lk)


class _BShardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, k, pre, post, ops, group, training, keep=True):
        H, Lk = k.shape
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        s, e = head_range(H, rank, world)
        rows = getattr(ops, "rows", 1)          # exchanged rows per head (1 for the fused plans)
        kf_local = ops.kernel_fft(k.detach()[s:e].to(torch.float32).contiguous())      # (an empty shard when H < world)
        kf = _all_gather_rows(kf_local, H, rows, group)                # the path's one forward collective
        u = u.contiguous()
        pre = None if pre is None else pre.contiguous()
        post = None if post is None else post.contiguous()
        kept = None
        if training and keep and hasattr(ops, "conv_keep"):
            out, kept = ops.conv_keep(u, kf, pre, post)
        else:
            out = ops.conv(u, kf, pre, post)
        ctx.kept_layout = None if kept is None else tuple(t is not None for t in kept)
        ctx.ops, ctx.group, ctx.Lk, ctx.k_dtype, ctx.gated, ctx.H, ctx.rows = ops, group, Lk, k.dtype, pre is not None, H, rows
        if training:
            extra = () if kept is None else tuple(t for t in kept if t is not None)
            ctx.save_for_backward(*(((u, kf, pre, post) if pre is not None else (u, kf)) + extra))
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.saved_tensors:
            raise RuntimeError("BatchShardedFFTConv: backward needs module.training=True at forward time")
        nb = 4 if ctx.gated else 2
        if ctx.gated:
            u, kf, pre, post = ctx.saved_tensors[:4]
        else:
            (u, kf), pre, post = ctx.saved_tensors[:2], None, None
        kept = None
        if ctx.kept_layout is not None:
            it = iter(ctx.saved_tensors[nb:])
            kept = tuple(next(it) if present else None for present in ctx.kept_layout)
        import contextlib
        with (torch.cuda.device(u.device) if u.is_cuda else contextlib.nullcontext()):
            if kept is not None:
                du, dpre, dpost, dkf = ctx.ops.backward(dout.contiguous(), u, kf, pre, post, kept)
            else:
                du, dpre, dpost, dkf = ctx.ops.backward(dout.contiguous(), u, kf, pre, post)
            H, rows = ctx.H, ctx.rows
            # fp32 sums over every rank's batch rows; (H, rows, ...) view so that the partition is the head partition
            dkf_local = _reduce_scatter_heads(dkf.contiguous().view((H, rows) + tuple(dkf.shape[1:])), 0, ctx.group)
            dk_local = ctx.ops.dk_from_dkf(dkf_local.reshape((-1,) + tuple(dkf.shape[1:])), ctx.Lk)
            dk = _all_gather_uneven(dk_local, H, 0, ctx.group)        # replicated parameter -> full gradient everywhere
        return du, dk.to(ctx.k_dtype), dpre, dpost, None, None, None, None


class _AllReduceGrad(torch.autograd.Function):
    """identity whose gradient is summed over the ranks (a replicated parameter used on every rank's batch shard)"""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, group=ctx.group)
        return g, None


class BatchShardedFFTConv(torch.nn.Module):
    """Data-parallel FlashFFTConv: `forward(u_local, k, pregate_local, postgate_local)` with u/gates the rank's batch
    rows and k the full replicated (H, Lk) filter; returns the local rows of y.  k.grad is the gradient summed over ALL
    ranks' batch rows (identical on every rank).  See the module docstring for the two modes."""

    def __init__(self, conv, mode="allgather_kf", group=None, ops=None):
        super().__init__()
        assert mode in ("allgather_kf", "recompute")
        self.conv, self.mode, self.group, self._ops = conv, mode, group, ops

    def forward(self, u, k, pregate=None, postgate=None):
        if pregate is not None or postgate is not None:
            assert pregate is not None and postgate is not None
        mode = self.mode
        if self._ops is None and mode == "allgather_kf":
            if self.conv._folded or self.conv._kf_keep is not None:
                mode = "recompute"          # (periodised k / masked k_f: not worth an exchange path of their own)
        if mode == "recompute":
            kk = _AllReduceGrad.apply(k, self.group)
            return self.conv(u, kk, pregate, postgate) if pregate is not None else self.conv(u, kk)
        ops = self._ops if self._ops is not None else (_BigOps if self.conv._big else _HipOps)(self.conv, u.device)
        training = self.conv.training if hasattr(self.conv, "training") else True
        keep = training and torch.is_grad_enabled()      # spectra are only worth storing when a graph is being recorded
        if self._ops is None:
            from .conv import _check_inputs
            _check_inputs(self.conv, u, k, (pregate, postgate))
            with torch.cuda.device(u.device):
                return _BShardFn.apply(u, k, pregate, postgate, ops, self.group, training, keep)
        return _BShardFn.apply(u, k, pregate, postgate, ops, self.group, training, keep)
