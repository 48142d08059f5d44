"""Multi-GPU execution of the FFT-conv path: every (b, h) row is independent given k[h], so the
head axis shards across ranks with NO data-path collective: rank r owns heads
[r*H/W, (r+1)*H/W) of u, k, the gates, y and dk.  One process per GPU, torch.distributed over
RCCL/xGMI (backend "nccl") only for the optional gather of results / dk.
The reference has no distributed code in its core; this is new, not a port."""
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


def gather_heads(x_local, H, dim, group=None):
    """All-gather head shards back to the full tensor (uneven shards supported via padding)."""
    world = dist.get_world_size(group)
    sizes = [head_range(H, r, world)[1] - head_range(H, r, world)[0] for r in range(world)]
    m = max(sizes)
    pad_shape = list(x_local.shape); pad_shape[dim] = m
    buf = x_local.new_zeros(pad_shape)
    buf.narrow(dim, 0, x_local.shape[dim]).copy_(x_local)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return torch.cat([o.narrow(dim, 0, s) for o, s in zip(outs, sizes)], dim=dim)


class HeadShardedFFTConv(torch.nn.Module):
    """Wraps a conv callable `conv(u, k, pregate, postgate)`; each rank computes its head shard.
    `gather=True` returns the full (B,H,L) output on every rank (all-gather over xGMI)."""

    def __init__(self, conv, gather=False):
        super().__init__()
        self.conv, self.gather = conv, gather

    def forward(self, u, k, pregate=None, postgate=None):
        H = u.shape[1]
        ul, kl = shard_heads(u, 1), shard_heads(k, 0)
        pl = None if pregate is None else shard_heads(pregate, 1)
        ql = None if postgate is None else shard_heads(postgate, 1)
        y = self.conv(ul, kl, pl, ql) if pl is not None else self.conv(ul, kl)
        return gather_heads(y, H, 1) if self.gather else y
