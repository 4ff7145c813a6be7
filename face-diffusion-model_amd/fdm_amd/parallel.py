"""Clip-level batch sharding across the GPUs of one node.

Clips are fully independent (attention is per clip, LayerNorm per token, InstanceNorm per clip-channel;
the reference processes one clip at a time, samples/sample_diffusion_vocaset.py:51), so there is no
exchange step inside the T-step loop.  One process per GPU; rank r owns clips [r*B/W, (r+1)*B/W); every
rank holds the full weights (optionally broadcast once from rank 0 as one flattened arena: broadcast_state); the only
collective on the path is one all-gather of the finished outputs (RCCL over xGMI when the backend is "nccl"; gloo in the
CPU tests).  Per-clip noise streams are keyed by
the global clip index, so results are bit-identical for any world size."""
import torch


def shard_range(n_clips, rank, world):
    """Contiguous clip range of `rank`; the first n_clips % world ranks take one extra clip."""
    base, extra = divmod(n_clips, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_clips(local, dist=None, sizes=None):
    """All-gather per-rank clip tensors [b_r, ...] into [sum b_r, ...] in rank order.

    dist: the torch.distributed module (initialised) or None for a single process.  Ragged shards are
    padded to the largest shard for the collective and trimmed afterwards."""
    if dist is None or not dist.is_initialized():
        return local
    world = dist.get_world_size()
    if dist.get_backend() != "nccl" and local.is_cuda:      # gloo dry runs: stage through host memory
        return gather_clips(local.cpu(), dist, sizes).to(local.device)
    if sizes is None:
        n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
        alln = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(alln, n)
        sizes = [int(x[0]) for x in alln]
    mx = max(sizes)
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        local = torch.cat([local, pad])
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local.contiguous())
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


def broadcast_state(state, dist=None, src=0, device=None):
    """Init-time broadcast of a state dict (weights / conditioning) from rank `src`: ONE collective per dtype over a
    flattened arena (xGMI links are point to point: few large transfers, not hundreds of small ones), unflattened in place.
    Every rank passes tensors of the same names / shapes (e.g. freshly constructed modules); rank `src`'s values win.
    device: where the arena lives for the collective (a cuda device for the RCCL backend; None = the tensors' own)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return state
    by_dtype = {}
    for k in sorted(state):
        by_dtype.setdefault(state[k].dtype, []).append(k)
    for dt, keys in by_dtype.items():
        dev = device if device is not None else state[keys[0]].device
        flat = torch.cat([state[k].detach().reshape(-1).to(dev) for k in keys]) if keys else None
        dist.broadcast(flat, src=src)
        off = 0
        for k in keys:
            n = state[k].numel()
            state[k].detach().copy_(flat[off:off + n].reshape(state[k].shape).to(state[k].device))
            off += n
    return state
