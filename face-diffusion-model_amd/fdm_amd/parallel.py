"""Clip-level batch sharding across the GPUs of one node.

Clips are fully independent (attention is per clip, LayerNorm per token, InstanceNorm per clip-channel;
the reference processes one clip at a time, samples/sample_diffusion_vocaset.py:51), so there is no
exchange step inside the T-step loop.  One process per GPU; rank r owns clips [r*B/W, (r+1)*B/W); every
rank holds the full weights; the only collective on the path is one all-gather of the finished outputs
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  Per-clip noise streams are keyed by
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
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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


def broadcast_state(state, dist=None, src=0):
    """Optional init-time broadcast of a state dict (weights / conditioning) from rank `src`."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return state
    for k in sorted(state):
        dist.broadcast(state[k], src=src)
    return state
