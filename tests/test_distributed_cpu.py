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
