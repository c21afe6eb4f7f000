#!/usr/bin/env python3
"""Per-call latency of the host-buffer entry points (what a Service.Suggest call per request costs): one query per call
and small batches, cars dictionary + 1M synthetic.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from suggest_amd import IndexDescription, NGramIndex, synth
import oracle

def run(ix, qb, qo, n, reps=200):
    sb, so = qb[:int(qo[n])], qo[:n + 1]
    for _ in range(10):
        ix.suggest_batch(blob=sb, offs=so, metric="jaccard", similarity=0.5, k=10)
    t0 = time.perf_counter()
    for _ in range(reps):
        ix.suggest_batch(blob=sb, offs=so, metric="jaccard", similarity=0.5, k=10)
    return (time.perf_counter() - t0) / reps * 1e6

blob, offs = synth.make_dict(1_000_000, seed=1)
qb, qo = synth.make_queries(4096, blob, offs, seed=2)
ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION))
for n in (1, 16, 256, 4096):
    print("1M dict, %4d queries per call: %8.1f us per call" % (n, run(ix, qb, qo, n)))

# SpellChecker.Predict per call (reference fixture model)
from suggest_amd import LanguageModel, SpellChecker
lm = LanguageModel(os.path.join(ROOT, "tests", "golden", "lm"), 3)
sc = SpellChecker(lm)
for _ in range(10):
    sc.Predict("i am sa", 5, 0.3)
t0 = time.perf_counter()
for _ in range(200):
    sc.Predict("i am sa", 5, 0.3)
print("SpellChecker.Predict, one query per call: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
