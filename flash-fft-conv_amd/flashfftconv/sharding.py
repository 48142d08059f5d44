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

Collectives are issued with async_op=True and waited for where their result is first needed (_Pending): RCCL runs them on the
process group's stream, so the kernels launched in between overlap them; the B-shard cuts the heads into groups to have such
kernels (_BShardFn).  Which collective a backend supports is probed once per group (_caps), never decided by catching an error
around a live collective.

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


def _sizes(H, world):
    return [head_range(H, r, world)[1] - head_range(H, r, world)[0] for r in range(world)]


# Which collectives a process group has is decided ONCE per (group, device type), by the same tiny probe on every rank, and
# cached (a try / except around a live collective lets one rank that fails locally -- out of memory, an async RCCL
# error -- fall into a different collective than its peers: a hang instead of an error).  A backend either has the tensor
# collectives or raises before it communicates, so the ranks agree.
_CAPS = {}


def _caps(group, like):
    # keyed by the backend + the group's ranks, not id(group): a destroyed group's id can be handed to a new one
    backend = dist.get_backend(group)
    key = (backend, tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)), like.device.type)
    c = _CAPS.get(key)
    if c is None:
        if backend == "nccl" and like.device.type == "cuda":          # RCCL on device tensors: both tensor collectives
            c = (True, True)
        else:
            world = dist.get_world_size(group)
            one, many = like.new_zeros(1, dtype=torch.float32), like.new_zeros(world, dtype=torch.float32)
            ag = rs = True
            try:
                dist.all_gather_into_tensor(many, one, group=group)
            except (RuntimeError, NotImplementedError):
                ag = False
            try:
                dist.reduce_scatter_tensor(one, many, group=group)
            except (RuntimeError, NotImplementedError):
                rs = False
            c = (ag, rs)
        _CAPS[key] = c
    return c


class _Pending:
    """a collective in flight: issued with async_op=True (RCCL runs it on the process group's own stream, behind an event
    recorded on the compute stream at issue time, so kernels launched afterwards overlap it); wait() makes the compute stream
    wait for it -- no host synchronisation with RCCL -- and returns the result"""

    def __init__(self, work, finish):
        self.work, self.finish = work, finish

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.finish()


def _all_gather_heads_async(x_local, H, group=None):
    """all-gather of head shards along dim 0 (uneven shards: padded to the largest one) -> _Pending of the (H, ...) tensor"""
    world = dist.get_world_size(group)
    sizes = _sizes(H, world)
    m = max(sizes)
    x_local = x_local.contiguous()
    tail = tuple(x_local.shape[1:])
    ag_tensor, _ = _caps(group, x_local)
    if all(sz == m for sz in sizes):
        out = x_local.new_empty((H,) + tail)
        if ag_tensor:
            w = dist.all_gather_into_tensor(out, x_local, group=group, async_op=True)
        else:
            w = dist.all_gather(list(out.chunk(world, 0)), x_local, group=group, async_op=True)
        return _Pending(w, lambda: out)
    buf = x_local.new_zeros((m,) + tail)
    buf[: x_local.shape[0]].copy_(x_local)
    padded = x_local.new_empty((world * m,) + tail)
    if ag_tensor:
        w = dist.all_gather_into_tensor(padded, buf, group=group, async_op=True)
    else:
        w = dist.all_gather(list(padded.chunk(world, 0)), buf, group=group, async_op=True)

    def strip():
        out = x_local.new_empty((H,) + tail)
        at = 0
        for r, sz in enumerate(sizes):
            out[at:at + sz].copy_(padded[r * m:r * m + sz])
            at += sz
        return out
    return _Pending(w, strip)


def _reduce_scatter_heads_async(x_full, group=None):
    """sum over the ranks of x_full (H, ...), this rank's head range of the result -> _Pending.  Uneven shards go through a padded
    reduce_scatter_tensor (round 3 all-reduced a clone of the whole tensor: 201 MB per rank at config 2 for H = 111-style shapes);
    a backend without the tensor collective (gloo) all-reduces."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H = x_full.shape[0]
    sizes = _sizes(H, world)
    m = max(sizes)
    tail = tuple(x_full.shape[1:])
    x_full = x_full.contiguous()
    _, rs_tensor = _caps(group, x_full)
    s, e = head_range(H, rank, world)
    if not rs_tensor:
        x = x_full.clone()
        w = dist.all_reduce(x, group=group, async_op=True)
        return _Pending(w, lambda: x[s:e].contiguous())
    if all(sz == m for sz in sizes):
        out = x_full.new_empty((m,) + tail)
        w = dist.reduce_scatter_tensor(out, x_full, group=group, async_op=True)
        return _Pending(w, lambda: out)
    padded = x_full.new_zeros((world * m,) + tail)
    at = 0
    for r, sz in enumerate(sizes):
        padded[r * m:r * m + sz].copy_(x_full[at:at + sz])
        at += sz
    out = x_full.new_empty((m,) + tail)
    w = dist.reduce_scatter_tensor(out, padded, group=group, async_op=True)
    return _Pending(w, lambda: out[: e - s])


def _all_gather_uneven(x_local, H, dim, group=None):
    """All-gather head shards back to the full tensor along `dim` (uneven shards supported via padding)."""
    if dim == 0:
        return _all_gather_heads_async(x_local, H, group).wait()
    world = dist.get_world_size(group)
    sizes = _sizes(H, world)
    m = max(sizes)
    pad_shape = list(x_local.shape); pad_shape[dim] = m
    buf = x_local.new_zeros(pad_shape)
    buf.narrow(dim, 0, x_local.shape[dim]).copy_(x_local)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return torch.cat([o.narrow(dim, 0, sz) for o, sz in zip(outs, sizes)], dim=dim)


def _reduce_scatter_heads(x_full, dim, group=None):
    """sum over ranks of x_full, this rank's head range of the result."""
    if dim == 0:
        return _reduce_scatter_heads_async(x_full, group).wait()
    return _reduce_scatter_heads_async(x_full.movedim(dim, 0).contiguous(), group).wait().movedim(0, dim).contiguous()


def _all_gather_rows_async(x_local, H, rows, group=None):
    """all-gather of head shards whose dim 0 is (head, row) with `rows` rows per head"""
    if rows == 1:
        return _all_gather_heads_async(x_local, H, group)
    tail = tuple(x_local.shape[1:])
    p = _all_gather_heads_async(x_local.reshape((-1, rows) + tail), H, group)
    return _Pending(None, lambda: p.wait().reshape((H * rows,) + tail))


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
    """GPU backend of the B-shard: the C-ABI entry points of one fused plan.  Every call works on a HEAD RANGE [g0, g1) of
    (B, H, L) tensors in place (batch-strided entry points): the head groups of the pipelined B-shard (_BShardFn) need no copies."""
    SLICES = True

    def __init__(self, mod, device):
        from . import conv as C, _lib
        self.C, self.L, self.mod = C, _lib, mod
        self.plan = mod._get_plan(device, mod._plan_seqlen)

    def kernel_fft(self, k):                       # (h, Lk) fp32 -> (h, kf_elems, 2) dtype, internal order
        if k.shape[0] == 0:                        # more ranks than heads: an empty shard
            return torch.empty(0, self.plan.kf_elems, 2, dtype=self.plan.dtype, device=k.device)
        return self.C._kernel_fft(self.plan, k)

    def _sp(self, t, g0):
        """pointer to head g0 of a contiguous (B, H, L) tensor (None stays None)"""
        import ctypes
        return None if t is None else ctypes.c_void_p(t.data_ptr() + g0 * t.shape[-1] * t.element_size())

    def conv_slice(self, u, kf, pre, post, out, g0, g1, keep):
        """out[:, g0:g1] = conv of heads [g0, g1) with their k_f rows `kf`; keep: also store the spectra (+ the output before the
        postgate) for backward_slice -> (z, yraw) or None"""
        L, lib = self.L, self.L.lib()
        B, H, Lu = u.shape
        hg, sb = g1 - g0, H * Lu
        if hg == 0:
            return None
        z = yraw = None
        if keep and self.mod.save_spectrum:
            z = self.C._spectrum_buffer(self.plan, B, hg, u.device, pre is not None, self.mod.save_spectrum)
            if z is not None and pre is not None:
                try:
                    yraw = torch.empty(B, hg, Lu, dtype=u.dtype, device=u.device)
                except torch.cuda.OutOfMemoryError:      # same fallback as the single-rank module
                    z = None
        if z is None:
            L.check(lib.ffc_conv_fwd_strided(self.plan.handle, self._sp(u, g0), L.ptr(kf), self._sp(pre, g0), self._sp(post, g0),
                                             self._sp(out, g0), B, hg, Lu, 0, sb, sb, sb, sb, L.stream_ptr()), "ffc_conv_fwd_strided")
            return None
        L.check(lib.ffc_conv_fwd_z(self.plan.handle, self._sp(u, g0), L.ptr(kf), self._sp(pre, g0), self._sp(post, g0), self._sp(out, g0),
                                   L.ptr(z), L.ptr(yraw), B, hg, Lu, sb, sb, sb, sb, L.stream_ptr()), "ffc_conv_fwd_z")
        return (z, yraw)

    def backward_slice(self, dout, u, kf, pre, post, du, dpre, dpost, g0, g1, kept=None):
        """gradients of heads [g0, g1) written into their slices of du / dpre / dpost -> dk_f (hg, kf_elems, 2) fp32, summed over
        the local batch"""
        lib, L = self.L.lib(), self.L
        B, H, Lu = u.shape
        hg, sb = g1 - g0, H * Lu
        if hg == 0:
            return torch.empty(0, self.plan.kf_elems, 2, dtype=torch.float32, device=u.device)
        ws = torch.empty(lib.ffc_dkf_workspace_bytes(self.plan.handle, B, hg), dtype=torch.uint8, device=u.device)
        if kept is not None and pre is not None:
            z, yraw = kept
            L.check(lib.ffc_conv_bwd_zy(self.plan.handle, self._sp(dout, g0), self._sp(u, g0), L.ptr(kf), self._sp(pre, g0), self._sp(post, g0),
                                        self._sp(du, g0), self._sp(dpre, g0), self._sp(dpost, g0), L.ptr(ws), L.ptr(z), L.ptr(yraw),
                                        B, hg, Lu, sb, sb, sb, sb, sb, sb, sb, L.stream_ptr()), "ffc_conv_bwd_zy")
        elif kept is not None:
            L.check(lib.ffc_conv_bwd_z(self.plan.handle, self._sp(dout, g0), self._sp(u, g0), L.ptr(kf), None, None, self._sp(du, g0), None,
                                       None, L.ptr(ws), L.ptr(kept[0]), B, hg, Lu, sb, sb, sb, sb, sb, sb, sb, L.stream_ptr()), "ffc_conv_bwd_z")
        else:
            L.check(lib.ffc_conv_bwd_gated_strided(self.plan.handle, self._sp(dout, g0), self._sp(u, g0), L.ptr(kf), self._sp(pre, g0),
                                                   self._sp(post, g0), self._sp(du, g0), self._sp(dpre, g0), self._sp(dpost, g0), L.ptr(ws),
                                                   B, hg, Lu, sb, sb, sb, sb, sb, sb, sb, L.stream_ptr()), "ffc_conv_bwd_gated_strided")
        nslab = lib.ffc_dkf_slab_count(self.plan.handle, B, hg)
        nfl = hg * self.plan.kf_elems * 2
        slabs = ws[: nslab * nfl * 4].view(torch.float32).view(nslab, hg, self.plan.kf_elems, 2)
        return slabs[0] if nslab == 1 else slabs.sum(0)

    # whole-tensor forms (one head group)
    def conv(self, u, kf, pre, post):
        out = torch.empty_like(u)
        self.conv_slice(u, kf, pre, post, out, 0, u.shape[1], False)
        return out

    def conv_keep(self, u, kf, pre, post):
        out = torch.empty_like(u)
        return out, self.conv_slice(u, kf, pre, post, out, 0, u.shape[1], True)

    def backward(self, dout, u, kf, pre, post, kept=None):
        """-> du, dpre, dpost, dk_f (H, kf_elems, 2) fp32 summed over the local batch"""
        du = torch.empty_like(u)
        dpre = torch.empty_like(u) if pre is not None else None
        dpost = torch.empty_like(u) if pre is not None else None
        dkf = self.backward_slice(dout, u, kf, pre, post, du, dpre, dpost, 0, u.shape[1], kept)
        return du, dpre, dpost, dkf

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

    def __init__(self, mod, device, Lmax=None):
        from . import conv as C, bigfft
        self.C, self.mod = C, mod
        # the factorisation the single-rank module would pick for these lengths (fft 2M / 4M take the one-level
        # 64 x / 128 x 32768 form when every row fits N/2 / N/4); every rank sees the same lengths, so the ranks agree
        self.fac = bigfft.choose(mod.seqlen, Lmax, C._TorchOps) if Lmax is not None else bigfft.BIG_FACTORS[mod.seqlen]
        factors, self.M = self.fac
        self.rows = 1
        for n0 in factors:
            self.rows *= n0

    def kernel_fft(self, k):
        if k.shape[0] == 0:                        # more ranks than heads: an empty shard
            plan = self.mod._get_plan(k.device, self.M)
            return torch.empty(0, plan.kf_elems, 2, dtype=self.mod.dtype, device=k.device)
        return self.C._big_kernel_fft(self.mod, k, self.fac)

    def conv(self, u, kf, pre, post):
        return self.C._big_forward(self.mod, u, None, pre, post, False, kf, self.fac)[0]

    def conv_keep(self, u, kf, pre, post):
        want = bool(self.mod.save_spectrum)
        keep = want and self.C._spectrum_budget_ok(
            ((u.shape[0] + 1) // 2) * u.shape[1] * self.mod.seqlen * (8 if pre is not None else 4), u.device, self.mod.save_spectrum)
        if want and not keep:
            self.C.SPECTRUM_FALLBACKS["budget"] += 1      # counted like the single-rank module's
        try:
            out, _, kept = self.C._big_forward(self.mod, u, None, pre, post, keep, kf, self.fac)
        except torch.cuda.OutOfMemoryError:      # same retry as the single-rank module
            if not keep:
                raise
            self.C.SPECTRUM_FALLBACKS["oom"] += 1
            out, _, kept = self.C._big_forward(self.mod, u, None, pre, post, False, kf, self.fac)
        return out, kept

    def backward(self, dout, u, kf, pre, post, kept=None):
        du, dkf, dpre, dpost = self.C._big_backward(self.mod, dout, u, kf, pre, post, 0, kept, True, self.fac)
        return du, dpre, dpost, dkf

    def dk_from_dkf(self, dkf, Lk):
        if dkf.shape[0] == 0:
            return torch.empty(0, Lk, dtype=torch.float32, device=dkf.device)
        return self.C._big_dk_from_dkf(self.mod, dkf, Lk, self.fac)


def head_groups(H, world, ngroups):
    """the pipelined B-shard's head groups: `ngroups` contiguous ranges; inside a group rank r owns head_range(size, r, world)"""
    ngroups = max(1, min(int(ngroups), max(H // max(world, 1), 1)))
    return [head_range(H, g, ngroups) for g in range(ngroups)]


def _default_groups(H, world, ops):
    import os
    if not getattr(ops, "SLICES", False) and not getattr(ops, "GENERIC_SLICES", False):
        return 1                 # HBM-level sizes: no in-place head ranges
    e = os.environ.get("FFC_SHARD_GROUPS")
    if e:
        return max(1, int(e))
    return 2 if H >= 16 * world else 1


class _BShardFn(torch.autograd.Function):
    """Head-group pipeline (SURVEY 8(e) notes).  The heads are cut into G groups; inside a group every rank owns
    1/W of the heads.  Forward: each rank transforms its heads of EVERY group first and starts the G k_f all-gathers
    (async_op: RCCL runs them on its own stream); group g's convolution waits for gather g only, so gathers g+1 .. run under it.
    Backward: group g's fused backward kernel, then its fp32 dk_f reduce-scatter in flight while group g+1 computes; the
    per-group dk inverses and dk all-gathers are chained the same way.  G = 1 is the unpipelined form (HBM-level sizes)."""

    @staticmethod
    def forward(ctx, u, k, pre, post, ops, group, training, keep=True, ngroups=1):
        H, Lk = k.shape
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        rows = getattr(ops, "rows", 1)          # exchanged rows per head (1 for the fused plans)
        groups = head_groups(H, world, ngroups)
        k32 = k.detach().to(torch.float32)
        pend = []
        for (g0, g1) in groups:                 # every group's local transform + its all-gather, issued up front
            s, e = head_range(g1 - g0, rank, world)
            kf_local = ops.kernel_fft(k32[g0 + s:g0 + e].contiguous())      # (an empty shard when the group has < world heads)
            pend.append(_all_gather_rows_async(kf_local, g1 - g0, rows, group))
        u = u.contiguous()
        pre = None if pre is None else pre.contiguous()
        post = None if post is None else post.contiguous()
        do_keep = training and keep
        kfs, kepts = [], []
        if len(groups) == 1:
            kf = pend[0].wait()
            kfs.append(kf)
            if do_keep and hasattr(ops, "conv_keep"):
                out, kept = ops.conv_keep(u, kf, pre, post)
            else:
                out, kept = ops.conv(u, kf, pre, post), None
            kepts.append(kept)
        else:
            out = torch.empty_like(u)
            for (g0, g1), p in zip(groups, pend):
                kf = p.wait()
                kfs.append(kf)
                kepts.append(_conv_group(ops, u, kf, pre, post, out, g0, g1, do_keep))
        ctx.ops, ctx.group, ctx.Lk, ctx.k_dtype, ctx.gated, ctx.H, ctx.rows, ctx.groups = ops, group, Lk, k.dtype, pre is not None, H, rows, groups
        if training:
            flat, layout = [], []
            for kept in kepts:
                layout.append(None if kept is None else tuple(t is not None for t in kept))
                if kept is not None:
                    flat += [t for t in kept if t is not None]
            ctx.kept_layout = layout
            ctx.save_for_backward(*(((u, pre, post) if pre is not None else (u,)) + tuple(kfs) + tuple(flat)))
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.saved_tensors:
            raise RuntimeError("BatchShardedFFTConv: backward needs module.training=True at forward time")
        sv = list(ctx.saved_tensors)
        if ctx.gated:
            u, pre, post = sv[:3]; sv = sv[3:]
        else:
            u, pre, post = sv[0], None, None; sv = sv[1:]
        groups, ops, rows = ctx.groups, ctx.ops, ctx.rows
        kfs, sv = sv[:len(groups)], sv[len(groups):]
        it = iter(sv)
        kepts = [None if lay is None else tuple(next(it) if present else None for present in lay) for lay in ctx.kept_layout]
        import contextlib
        with (torch.cuda.device(u.device) if u.is_cuda else contextlib.nullcontext()):
            dout = dout.contiguous()
            rs = []
            if len(groups) == 1:
                args = (dout, u, kfs[0], pre, post) + ((kepts[0],) if kepts[0] is not None else ())
                du, dpre, dpost, dkf = ops.backward(*args)
                rs.append(_reduce_scatter_heads_async(dkf.contiguous().view((ctx.H, rows) + tuple(dkf.shape[1:])), ctx.group))
            else:
                du = torch.empty_like(u)
                dpre = torch.empty_like(u) if ctx.gated else None
                dpost = torch.empty_like(u) if ctx.gated else None
                for (g0, g1), kf, kept in zip(groups, kfs, kepts):
                    dkf = _backward_group(ops, dout, u, kf, pre, post, du, dpre, dpost, g0, g1, kept)
                    # fp32 sums over every rank's batch rows of this group, in flight while the next group computes
                    rs.append(_reduce_scatter_heads_async(dkf.contiguous().view((g1 - g0, rows) + tuple(dkf.shape[1:])), ctx.group))
            ag = []
            for (g0, g1), p in zip(groups, rs):
                dkf_local = p.wait()
                dk_local = ops.dk_from_dkf(dkf_local.reshape((-1,) + tuple(dkf_local.shape[2:])), ctx.Lk)
                ag.append(_all_gather_heads_async(dk_local, g1 - g0, ctx.group))      # replicated parameter -> full gradient everywhere
            parts = [p.wait() for p in ag]
            dk = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        return du, dk.to(ctx.k_dtype), dpre, dpost, None, None, None, None, None


def _conv_group(ops, u, kf, pre, post, out, g0, g1, keep):
    if getattr(ops, "SLICES", False):
        return ops.conv_slice(u, kf, pre, post, out, g0, g1, keep)
    sl = lambda t: None if t is None else t[:, g0:g1].contiguous()      # generic ops (CPU stand-in): on copies of the head range
    if keep and hasattr(ops, "conv_keep"):
        y, kept = ops.conv_keep(sl(u), kf, sl(pre), sl(post))
    else:
        y, kept = ops.conv(sl(u), kf, sl(pre), sl(post)), None
    out[:, g0:g1] = y
    return kept


def _backward_group(ops, dout, u, kf, pre, post, du, dpre, dpost, g0, g1, kept):
    if getattr(ops, "SLICES", False):
        return ops.backward_slice(dout, u, kf, pre, post, du, dpre, dpost, g0, g1, kept)
    sl = lambda t: None if t is None else t[:, g0:g1].contiguous()
    args = (sl(dout), sl(u), kf, sl(pre), sl(post)) + ((kept,) if kept is not None else ())
    a, b, c, dkf = ops.backward(*args)
    du[:, g0:g1] = a
    if pre is not None:
        dpre[:, g0:g1] = b; dpost[:, g0:g1] = c
    return dkf


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

    def __init__(self, conv, mode="allgather_kf", group=None, ops=None, groups=None):
        super().__init__()
        assert mode in ("allgather_kf", "recompute")
        self.conv, self.mode, self.group, self._ops = conv, mode, group, ops
        self.groups = groups        # head groups of the pipelined exchange (None: 2 when H >= 16 x world, FFC_SHARD_GROUPS overrides)

    def forward(self, u, k, pregate=None, postgate=None):
        if pregate is not None or postgate is not None:
            assert pregate is not None and postgate is not None
        mode = self.mode
        conv = self.conv
        if self._ops is None and hasattr(conv, "_fit_seqlen") and u.dim() == 3 and k.dim() == 2:
            # rows much shorter than the fft size: the smallest size that holds them, as the single-rank module (FlashFFTConv._fit_seqlen;
            # every rank sees the same row lengths, so every rank picks the same size)
            n = conv._fit_seqlen(u.shape[-1], k.shape[-1])
            if n != conv.seqlen:
                conv = conv._fitted_module(n)
        if self._ops is None and mode == "allgather_kf":
            if conv._folded or conv._kf_keep is not None:
                mode = "recompute"          # (periodised k / masked k_f: not worth an exchange path of their own)
        if mode == "recompute":
            kk = _AllReduceGrad.apply(k, self.group)
            return conv(u, kk, pregate, postgate) if pregate is not None else conv(u, kk)
        if self._ops is not None:
            ops = self._ops
        elif conv._big or conv._route_big(max(u.shape[-1], k.shape[-1])):
            # (fft 131072 with rows longer than N/2 takes the HBM-level form in the single-rank module: the same route here)
            ops = _BigOps(conv, u.device, max(u.shape[-1], k.shape[-1]))
        else:
            ops = _HipOps(conv, u.device)
        training = conv.training if hasattr(conv, "training") else True
        keep = training and torch.is_grad_enabled()      # spectra are only worth storing when a graph is being recorded
        world = dist.get_world_size(self.group)
        ng = self.groups if self.groups is not None else _default_groups(k.shape[0], world, ops)
        if not (getattr(ops, "SLICES", False) or getattr(ops, "GENERIC_SLICES", False)):
            ng = 1
        if self._ops is None:
            from .conv import _check_inputs
            _check_inputs(conv, u, k, (pregate, postgate))
            with torch.cuda.device(u.device):
                return _BShardFn.apply(u, k, pregate, postgate, ops, self.group, training, keep, ng)
        return _BShardFn.apply(u, k, pregate, postgate, ops, self.group, training, keep, ng)
