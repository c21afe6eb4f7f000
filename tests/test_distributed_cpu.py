"""world_size-2 gloo test of the N>1 path: query sharding + result gather (CPU; the per-rank search is stood in
for by the oracle, which is what the GPU path is checked against elsewhere)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_docs, n_q, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from suggest_amd import synth
    from suggest_amd.distributed import gather_results, shard_queries
    blob, offs = synth.make_dict(n_docs, seed=1)
    qb, qo = synth.make_queries(n_q, blob, offs, seed=2)
    ora = oracle.OracleIndex(blob=blob, offs=offs, **synth.DESCRIPTION)
    sb, so = shard_queries(qb, qo, world, rank)
    ids, sc, cnt, _ = ora.suggest_batch(sb, so, "jaccard", 0.5, 5, threads=1)
    g = gather_results(torch.from_numpy(ids.view(np.int32)), torch.from_numpy(sc), torch.from_numpy(cnt.view(np.int32)), n_q)
    if rank == 0:
        full = ora.suggest_batch(qb, qo, "jaccard", 0.5, 5, threads=1)
        ok = np.array_equal(g[0].numpy().view(np.uint32), full[0]) and np.array_equal(g[1].numpy(), full[1]) and \
            np.array_equal(g[2].numpy().view(np.uint32), full[2])
        ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    from suggest_amd.distributed import shard_bounds
    for n in (0, 1, 7, 64, 65537):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_two_rank_gloo_shard_and_gather():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 3000, 37, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def _doc_shard_worker(rank, world, port, n_docs, n_q, ret):
    """docID-range shards (SURVEY.md §8e alternative): every rank searches the whole batch on its own range of documents
    (the oracle stands in for the per-shard GPU search), the rows are all-gathered over gloo and merged."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from suggest_amd import synth
    from suggest_amd.distributed import merge_topk, shard_bounds
    blob, offs = synth.make_dict(n_docs, seed=1, families=3)        # families: several matches per query, ties
    qb, qo = synth.make_queries(n_q, blob, offs, seed=2)
    lo, hi = shard_bounds(n_docs, world, rank)
    sb, so = blob[int(offs[lo]):int(offs[hi])], (offs[lo:hi + 1] - offs[lo]).astype(np.uint64)
    shard = oracle.OracleIndex(blob=sb, offs=so, **synth.DESCRIPTION)
    k = 7
    ids, sc, cnt, _ = shard.suggest_batch(qb, qo, "jaccard", 0.4, k, threads=1)
    t_ids = torch.from_numpy(ids.astype(np.int64) + lo)
    t_sc, t_cnt = torch.from_numpy(sc), torch.from_numpy(cnt.astype(np.int64))
    from suggest_amd.distributed import _all_gather_stacked
    g_ids, g_sc, g_cnt = (_all_gather_stacked(t, world) for t in (t_ids, t_sc, t_cnt))
    m_ids, m_sc, m_cnt = merge_topk(g_ids, g_sc, g_cnt, k)
    if rank == 0:
        full = oracle.OracleIndex(blob=blob, offs=offs, **synth.DESCRIPTION).suggest_batch(qb, qo, "jaccard", 0.4, k, threads=1)
        valid = np.arange(k)[None, :] < full[2][:, None]
        ok = np.array_equal(m_cnt.numpy().astype(np.uint32), full[2]) and np.array_equal(m_ids.numpy().astype(np.uint32)[valid], full[0][valid]) \
            and np.array_equal(m_sc.numpy().view(np.uint64)[valid], full[1].view(np.uint64)[valid]) and bool((full[2] > 1).mean() > 0.3)
        ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_gloo_doc_shards_merge_to_the_unsharded_topk():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_doc_shard_worker, args=(r, 3, port, 4000, 64, ret)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def test_merge_topk_flags_and_ties():
    from suggest_amd.distributed import merge_topk
    ids = torch.tensor([[[5, 9, 0], [1, 0, 0]], [[2, 7, 8], [3, 4, 0]]])                  # [W=2, n=2, k=3]
    sc = torch.tensor([[[0.9, 0.5, 0.0], [0.7, 0.0, 0.0]], [[0.9, 0.5, 0.4], [0.7, 0.7, 0.0]]], dtype=torch.float64)
    cnt = torch.tensor([[2, 1], [3, 2]])
    m_ids, m_sc, m_cnt = merge_topk(ids, sc, cnt, 3)
    assert m_ids.tolist() == [[2, 5, 7], [1, 3, 4]] and m_cnt.tolist() == [3, 3]          # ties broken by the lower docID
    assert m_sc.tolist() == [[0.9, 0.9, 0.5], [0.7, 0.7, 0.7]]
    cnt[1, 1] = 0xFFFFFFFE                                                                  # a shard reports the reference's deadlock
    assert merge_topk(ids, sc, cnt, 3)[2].tolist() == [3, 0xFFFFFFFE]
