#!/usr/bin/env python3
"""SpellChecker.Predict throughput: synthetic vocabulary + n-gram counts, a batch of queries
("context words + partial last word") through sg_spell_predict_batch (host buffers in and out: tokenise, Next, two GPU
launches, merge/re-rank) and the CPU oracle beside it on all host threads.  GPU box only."""
import argparse, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import oracle
from suggest_amd import LanguageModel, SpellChecker, synth
from test_spell import _write_lm, SPELL_INDEX

ap = argparse.ArgumentParser()
ap.add_argument("--vocab", type=int, default=200000)
ap.add_argument("--sentences", type=int, default=300000)
ap.add_argument("--queries", type=int, default=65536)
ap.add_argument("--topk", type=int, default=5)
args = ap.parse_args()
blob, offs = synth.make_dict(args.vocab, seed=61, families=3)
vocab = sorted(set(w.decode() for w in synth.unpack(blob, offs)))
rnd = np.random.RandomState(9)
sent = [[vocab[int(i)] for i in rnd.zipf(1.2, size=int(rnd.randint(3, 9))) % len(vocab)] for _ in range(args.sentences)]
tmp = tempfile.mkdtemp()
_write_lm(tmp, vocab, 3, sent)
alpha = ("english", "numbers")
lm = LanguageModel(tmp, 3, alphabet=alpha)
sc = SpellChecker(lm)
queries = []
for i in range(args.queries):
    s = sent[int(rnd.randint(0, len(sent)))]
    cut = int(rnd.randint(1, len(s) + 1))
    w = s[cut - 1]
    if i % 2:
        p = int(rnd.randint(0, len(w))); w = w[:p] + "x" + w[p + 1:]
    else:
        w = w[:max(3, len(w) // 2)]
    queries.append(" ".join(s[max(0, cut - 3):cut - 1] + [w]).encode())
qb, qo = oracle.pack_strings(queries)
sc.predict_batch(blob=qb, offs=qo, top_k=args.topk, similarity=0.5)
t0 = time.perf_counter()
for _ in range(3):
    ids, cnt = sc.predict_batch(blob=qb, offs=qo, top_k=args.topk, similarity=0.5)
t_gpu = (time.perf_counter() - t0) / 3
ora_lm = oracle.OracleLM(tmp, 3, alphabet=alpha)
ora_ix = oracle.OracleIndex(ora_lm.words(), **SPELL_INDEX)
n_s = min(args.queries, 8192)
t0 = time.perf_counter()
oi, oc = ora_lm.predict_batch(ora_ix, qb[:int(qo[n_s])], qo[:n_s + 1], args.topk, 0.5)
t_cpu = time.perf_counter() - t0
valid = (np.arange(args.topk + 1)[None, :] < np.minimum(oc, args.topk + 1)[:, None]) & (oc < 0xFFFFFFF0)[:, None]
same = np.array_equal(cnt[:n_s], oc) and np.array_equal(ids[:n_s][valid], oi[valid])
print("vocabulary %d words, %d sentences, %d queries, topK %d" % (len(vocab), len(sent), len(queries), args.topk))
print("predictions per query: %.2f; queries needing the fuzzy top-up: %.0f%%" % (np.minimum(cnt, args.topk + 1).mean(), 100 * (cnt < args.topk).mean()))
print("GPU  sg_spell_predict_batch: %.1f ms per batch = %.2f M predictions/s (host buffers in/out, host steps included)" % (t_gpu * 1e3, len(queries) / t_gpu / 1e6))
print("CPU  oracle, %d threads, first %d queries: %.0f predictions/s; identical predictions: %s" % (os.cpu_count(), n_s, n_s / t_cpu, same))
