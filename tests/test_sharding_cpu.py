"""world_size-2 gloo test of the head-sharded multi-GPU path (compute = oracle on CPU)."""
import os, socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, H, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "flash-fft-conv_amd"), root]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flashfftconv.sharding import HeadShardedFFTConv, head_range
    from oracle.torch_ref import ref_fft_conv
    torch.manual_seed(0)
    B, L, N = 2, 64, 128
    u = torch.randn(B, H, L); k = torch.randn(H, L)
    conv = HeadShardedFFTConv(lambda a, b: ref_fft_conv(a, b, N), gather=True)
    y = conv(u, k)
    full = ref_fft_conv(u, k, N)
    ok = torch.allclose(y, full, atol=1e-5)
    s, e = head_range(H, rank, world)
    local = HeadShardedFFTConv(lambda a, b: ref_fft_conv(a, b, N))(u, k)
    ok = ok and torch.allclose(local, full[:, s:e], atol=1e-5)
    q.put((rank, bool(ok), (s, e)))
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [8, 5])
def test_head_sharding_gloo(H):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, H, q)) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps: p.join(60)
    assert all(ok for _, ok, _ in res), res
    ranges = sorted(r for _, _, r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == H and ranges[0][1] == ranges[1][0]
