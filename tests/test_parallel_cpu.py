"""CPU, world_size 2 over gloo: clip-level sharding and the single all-gather of the N > 1 path
(fdm_amd/parallel.py).  The arithmetic of a shard is the single-GPU path (GPU tests prove per-clip results
do not depend on batch composition and that Philox noise is keyed by the global clip index), so the
multi-process logic to verify here is: contiguous shard ranges (even + ragged), rank-ordered gather,
init-time broadcast."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fdm_amd.parallel import broadcast_state, gather_clips, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_clip_result(global_clip, L=6, d=8):
    """Stands in for 'the latent of clip i': a function of the GLOBAL clip index only."""
    g = torch.Generator().manual_seed(1000 + global_clip)
    return torch.randn(L, d, generator=g)


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_clips, rank, world)
        local = torch.stack([_fake_clip_result(i) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 6, 8)
        full = gather_clips(local, dist)
        ref = torch.stack([_fake_clip_result(i) for i in range(n_clips)])
        ok = torch.equal(full, ref)
        state = {"w": torch.full((4, 2), float(rank + 1)), "b": torch.arange(3.0) * (rank + 1), "idx": torch.arange(5) + 10 * rank}
        broadcast_state(state, dist, src=0)
        ok = ok and torch.equal(state["w"], torch.ones(4, 2)) and torch.equal(state["b"], torch.arange(3.0))
        ok = ok and torch.equal(state["idx"], torch.arange(5))             # one flattened collective per dtype
        q.put((rank, lo, hi, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [8, 5])
def test_two_rank_shard_and_gather(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[3] for r in res)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_clips      # contiguous cover


def _vertex_worker(rank, world, port, q):
    """cfg5's collective at its benched size: every rank contributes [4, 498, 15069] fp32 vertices (120 MB)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, L, V3 = 4, 498, 15069
        ramp = torch.arange(L * V3, dtype=torch.float32).reshape(L, V3) * 1e-6
        local = torch.stack([ramp + float(rank * B + b) for b in range(B)])          # a function of the GLOBAL clip index
        full = gather_clips(local, dist, sizes=[B] * world)                            # equal shards: no size exchange (bench.py)
        ok = full.shape == (B * world, L, V3) and full.numel() * 4 == 120_069_792 * world
        for i in range(B * world):
            ok = ok and torch.equal(full[i], ramp + float(i))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_of_cfg5_sized_vertices():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_vertex_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)


def test_shard_range_properties():
    for n in (1, 4, 7, 32):
        for w in (1, 2, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    assert gather_clips(torch.ones(2, 3), None) is not None
