"""Multi-GPU inference: similar_anime / similar_users all-pairs and model_recs batched prediction
shard BY QUERY ROWS across ranks: independent units, the normalised table is
replicated, no collective inside the loop — only the final gather of the [n/G, k] blocks."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) slice of n query rows for `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local, n_total, world, rank):
    """All-gather variable-length row blocks (padded to the largest shard); every rank gets [n_total, ...]."""
    if world == 1:
        return local
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)


def sharded_cosine_topk(What, k, exclude_self=True, keep=None, topk_fn=None):
    """All-pairs neighbours of every row of ``What`` (replicated on each rank): rank r scores the
    queries of its slice against all keys; returns the gathered (idx [n,k], score [n,k])."""
    if topk_fn is None:
        from .ops import cosine_topk_mfma as topk_fn
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    n = What.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    q = torch.arange(lo, hi, dtype=torch.int32, device=What.device)
    res = topk_fn(What, q, k, exclude_self=exclude_self, keep=keep)
    idx, sc = res[0], res[1]
    return gather_rows(idx, n, world, rank), gather_rows(sc, n, world, rank)


def sharded_predict_topk(U, A, head, users, k, watched_bits=None, predict_fn=None):
    """Top-k unwatched anime for every listed user, users sharded across ranks."""
    if predict_fn is None:
        from .ops import predict_topk_mfma as predict_fn     # matrix cores; exact kernels for flagged users
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    users = torch.as_tensor(users, device=U.device)
    n = users.numel()
    lo, hi = shard_bounds(n, rank, world)
    wb = None if watched_bits is None else torch.as_tensor(watched_bits, device=U.device)[lo:hi]
    res = predict_fn(U, A, head, users[lo:hi], k, wb)
    idx, p = res[0], res[1]
    return gather_rows(idx, n, world, rank), gather_rows(p, n, world, rank)
