"""Query-sharded multi-GPU execution: one process per GPU, index replica per rank, no data-path collective.

Queries are independent and the index is read-only, so a global batch is cut into contiguous
slices, each rank searches its slice on its own replica, and the only exchange is the optional gather of the
k*(u32 id, f64 score) result rows (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).
"""
import numpy as np


def shard_bounds(n_queries, world_size, rank):
    """Contiguous slice [lo, hi) of a batch of n_queries owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_queries, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_queries(blob, offs, world_size, rank):
    """-> (blob_slice, offs_slice) of this rank's queries, offsets re-based to 0."""
    lo, hi = shard_bounds(len(offs) - 1, world_size, rank)
    o = np.asarray(offs, dtype=np.uint64)
    b0, b1 = int(o[lo]), int(o[hi])
    return np.asarray(blob)[b0:b1], (o[lo:hi + 1] - o[lo]).astype(np.uint64)


def gather_results(ids, scores, counts, n_queries, group=None):
    """All-gather per-rank result rows into global order on every rank.  Tensors are torch tensors on the
    process group's device type ([n_local,k] ids int32/uint32-as-int32, [n_local,k] float64, [n_local] int32)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    k = ids.shape[1]
    pad = max(shard_bounds(n_queries, world, r)[1] - shard_bounds(n_queries, world, r)[0] for r in range(world))

    def padded(t, shape, dtype):
        out = torch.zeros(shape, dtype=dtype, device=t.device)
        out[: t.shape[0]] = t
        return out

    g_ids = torch.zeros((world * pad, k), dtype=ids.dtype, device=ids.device)
    g_sc = torch.zeros((world * pad, k), dtype=scores.dtype, device=scores.device)
    g_cnt = torch.zeros(world * pad, dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(g_ids, padded(ids, (pad, k), ids.dtype), group=group)
    dist.all_gather_into_tensor(g_sc, padded(scores, (pad, k), scores.dtype), group=group)
    dist.all_gather_into_tensor(g_cnt, padded(counts, (pad,), counts.dtype), group=group)
    keep = torch.cat([torch.arange(r * pad, r * pad + (shard_bounds(n_queries, world, r)[1] - shard_bounds(n_queries, world, r)[0]))
                      for r in range(world)]).to(ids.device)
    return g_ids[keep], g_sc[keep], g_cnt[keep]
