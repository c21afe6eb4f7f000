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


# ------------------------------------------------------------------------------------------------------------------------
# docID-range shards (the one-exchange design for dictionaries that do not fit — or should not be
# replicated on — every GPU): rank r indexes the documents [lo_r, hi_r) only, every rank searches the WHOLE batch on its
# shard, the per-shard top-k rows are all-gathered (RCCL over xGMI: world * n_q * k * 12 bytes) and merged with the
# reference's order (score desc, docID asc).  That order is total, so the merged top-k equals the unsharded one —
# for dictionaries without documents that repeat a term; with them the primary entries are the same but the reference's
# *secondary* duplicate rows depend on relative list lengths, which differ inside a shard.
# ------------------------------------------------------------------------------------------------------------------------
FLAG_MIN = 0xFFFFFFF0     # out_counts values from here up are SG_COUNT_* flags


def merge_topk(ids, scores, counts, k):
    """ids, scores: [W, n, k]; counts: [W, n] (int64; flags >= FLAG_MIN)  ->  (ids [n, k] int64, scores [n, k] f64, counts [n] int64).
    Torch tensors on any device.  A query flagged by a shard (the reference itself cannot answer it) stays flagged."""
    import torch

    W, n, kk = ids.shape
    flag = counts.max(dim=0).values
    flagged = flag >= FLAG_MIN
    c = torch.where(counts >= FLAG_MIN, torch.zeros_like(counts), counts).clamp(max=kk)
    valid = torch.arange(kk, device=ids.device)[None, None, :] < c[:, :, None]
    big = torch.iinfo(torch.int64).max
    flat_ids = torch.where(valid, ids.to(torch.int64), torch.full_like(ids, big, dtype=torch.int64)).permute(1, 0, 2).reshape(n, W * kk)
    flat_sc = torch.where(valid, scores, torch.full_like(scores, float("-inf"))).permute(1, 0, 2).reshape(n, W * kk)
    order = torch.sort(flat_ids, dim=1, stable=True).indices                    # docID asc ...
    flat_ids, flat_sc = torch.gather(flat_ids, 1, order), torch.gather(flat_sc, 1, order)
    order = torch.sort(flat_sc, dim=1, descending=True, stable=True).indices    # ... then score desc, stable
    flat_ids, flat_sc = torch.gather(flat_ids, 1, order)[:, :k], torch.gather(flat_sc, 1, order)[:, :k]
    total = c.sum(dim=0).clamp(max=k)
    keep = torch.arange(k, device=ids.device)[None, :] < total[:, None]
    out_ids = torch.where(keep, flat_ids, torch.zeros_like(flat_ids))
    out_sc = torch.where(keep, flat_sc, torch.zeros_like(flat_sc))
    return out_ids, out_sc, torch.where(flagged, flag, total)


def _all_gather_stacked(t, world, group=None):
    """all_gather_into_tensor (the output is the concatenation along dim 0) viewed as [world, *t.shape]"""
    import torch
    import torch.distributed as dist
    out = torch.zeros((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out.view((world,) + tuple(t.shape))


class DocShardedIndex:
    """One shard of a dictionary split by docID range.  Collective calls (all_reduce at construction, all_gather per
    batch) go through `group` (default group if None); with world size 1 it degenerates to a plain index."""

    def __init__(self, blob, offs, description, rank=None, world=None, device=0, group=None, build="host"):
        import torch
        import torch.distributed as dist
        from .index import NGramIndex

        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        offs = np.asarray(offs, dtype=np.uint64)
        n_docs = len(offs) - 1
        self.doc_lo, self.doc_hi = shard_bounds(n_docs, self.world, self.rank)
        sb = np.asarray(blob)[int(offs[self.doc_lo]):int(offs[self.doc_hi])]
        so = (offs[self.doc_lo:self.doc_hi + 1] - offs[self.doc_lo]).astype(np.uint64)
        index = NGramIndex(blob=sb, offs=so, description=description, device=device, upload=False, build=build)
        segs = index.stats()["n_segments"]
        if self.world > 1 and dist.is_initialized():                 # agree on the global number of cardinality segments
            t = torch.tensor([segs], dtype=torch.int64, device=self._comm_device(device))
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            if int(t.item()) != segs:
                index.close()
                index = NGramIndex(blob=sb, offs=so, description=description, device=device, upload=False, build=build, min_segments=int(t.item()))
        self.index = index.upload(device)
        self.device = device

    def _comm_device(self, device):
        import torch
        import torch.distributed as dist
        return torch.device("cuda", device) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def suggest_batch(self, blob, offs, metric, similarity, k):
        """Every rank passes the same batch; every rank gets the merged rows (numpy: ids uint32, scores f64, counts uint32)."""
        import torch
        import torch.distributed as dist

        if self.doc_hi == self.doc_lo:     # an empty shard (more ranks than documents): it has nothing to add, and it must not
            n_q = len(offs) - 1            # report the reference's panic for a window it cannot even see
            ids, sc, cnt = np.zeros((n_q, k), np.uint32), np.zeros((n_q, k), np.float64), np.zeros(n_q, np.uint32)
        else:
            ids, sc, cnt = self.index.suggest_batch(blob=blob, offs=offs, metric=metric, similarity=similarity, k=k)
        dev = self._comm_device(self.device) if (self.world > 1 and dist.is_initialized()) else torch.device("cpu")
        t_ids = torch.from_numpy(ids.astype(np.int64) + self.doc_lo).to(dev)         # local docID -> dictionary docID
        t_sc = torch.from_numpy(sc).to(dev)
        t_cnt = torch.from_numpy(cnt.astype(np.int64)).to(dev)
        if self.world > 1 and dist.is_initialized():
            g_ids, g_sc, g_cnt = (_all_gather_stacked(t, self.world, self.group) for t in (t_ids, t_sc, t_cnt))
        else:
            g_ids, g_sc, g_cnt = t_ids[None], t_sc[None], t_cnt[None]
        m_ids, m_sc, m_cnt = merge_topk(g_ids, g_sc, g_cnt, k)
        return m_ids.cpu().numpy().astype(np.uint32), m_sc.cpu().numpy(), m_cnt.cpu().numpy().astype(np.uint32)
