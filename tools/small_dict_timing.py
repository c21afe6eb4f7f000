#!/usr/bin/env python3
"""Throughput on the reference's own small dictionaries (cars: 5 066 entries, words: ~100 k) — the per-query fixed cost
(tokenise, seg_off rows, group setup, top-k) dominates there, not posting traffic.  GPU box only."""
import lzma, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from suggest_amd import IndexDescription, NGramIndex
import oracle
from conftest import CARS_DESC, WORDS_DESC

def run(name, lines, desc, metric, sim, k, n_q=65536):
    rnd = np.random.RandomState(1)
    qs = []
    for i in rnd.randint(0, len(lines), size=n_q):
        w = bytearray(lines[int(i)])
        if len(w) > 2:
            w[int(rnd.randint(0, len(w)))] = ord("x")
        qs.append(bytes(w))
    qb, qo = oracle.pack_strings(qs)
    ix = NGramIndex(lines, IndexDescription(**desc))
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.view(np.int64)).to(dev)
    d_ids = torch.zeros((n_q, k), dtype=torch.int32, device=dev); d_sc = torch.zeros((n_q, k), dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(12):      # (the launch-to-launch choices — tightening, queue size — settle within ~8 launches)
        ix.suggest_batch_device(d_q.data_ptr(), d_o.data_ptr(), n_q, metric, sim, k, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ix.suggest_batch_device(d_q.data_ptr(), d_o.data_ptr(), n_q, metric, sim, k, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), stream=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    ora = oracle.OracleIndex(lines, **desc)
    t0 = time.perf_counter()
    ora.suggest_batch(qb[:int(qo[8192])], qo[:8193], metric, sim, k, threads=os.cpu_count())
    t_cpu = time.perf_counter() - t0
    print("%-6s %6d entries, %s>=%.1f k=%d: %.3f ms per 65 536 queries = %.1f M q/s   (oracle, %d threads: %.2f M q/s; %.2f results per query)"
          % (name, len(lines), metric, sim, k, dt * 1e3, n_q / dt / 1e6, os.cpu_count(), 8192 / t_cpu / 1e6, float(np.minimum(d_cnt.cpu().numpy(), k).mean())))

G = os.path.join(ROOT, "tests", "golden")
cars = open(os.path.join(G, "cars.dict"), "rb").read().splitlines()
words = lzma.open(os.path.join(G, "words.dict.xz")).read().splitlines()
run("cars", cars, CARS_DESC, "cosine", 0.5, 5)
run("words", words, WORDS_DESC, "cosine", 0.5, 5)
run("words", words, WORDS_DESC, "jaccard", 0.5, 10)
