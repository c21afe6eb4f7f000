#!/usr/bin/env python3
"""Randomised SpellChecker.Predict parity hunt (GPU vs oracle): random vocabularies (many shared prefixes), random corpora
with hot contexts (continuation ranges from 1 to thousands of words), random orders / topK / similarity.  GPU box only."""
import argparse, os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import oracle
from suggest_amd import LanguageModel, SpellChecker
from test_spell import _write_lm, SPELL_INDEX

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
t_end, trial, bad = time.time() + args.seconds, 0, 0
while time.time() < t_end:
    seed = args.seed * 100000 + trial
    trial += 1
    rng = random.Random(seed)
    order = rng.choice([1, 2, 3, 3, 4])
    n_vocab = rng.choice([5, 60, 800, 6000])
    syll = ["ab", "ra", "ca", "da", "bra", "x", "qu", "ing", "on", "e", "st", "pre", "un", "1", "2"]
    vocab = sorted({"".join(rng.choice(syll) for _ in range(rng.randint(1, 5))) for _ in range(n_vocab)})
    hot = [rng.choice(vocab) for _ in range(4)]
    sent = []
    for _ in range(rng.choice([20, 500, 8000])):
        s = [rng.choice(vocab) if rng.random() < 0.7 else vocab[int(rng.paretovariate(1.2)) % len(vocab)] for _ in range(rng.randint(1, 7))]
        if rng.random() < 0.5:
            s[0] = rng.choice(hot)
        sent.append(s)
    tmp = tempfile.mkdtemp()
    _write_lm(tmp, vocab, order, sent)
    alpha = ("english", "numbers")
    lm = LanguageModel(tmp, order, alphabet=alpha)
    sc = SpellChecker(lm)
    ora_lm = oracle.OracleLM(tmp, order, alphabet=alpha)
    ora_ix = oracle.OracleIndex(ora_lm.words(), **SPELL_INDEX)
    queries = []
    for _ in range(200):
        s = rng.choice(sent)
        cut = rng.randint(1, len(s))
        ctx, w = s[max(0, cut - 1 - rng.randint(0, 4)):cut - 1], s[cut - 1]
        kind = rng.randint(0, 4)
        if kind == 0: w = w[:rng.randint(1, len(w))]
        elif kind == 1: w = w[:2]
        elif kind == 2 and len(w) > 2:
            p = rng.randrange(len(w)); w = w[:p] + "z" + w[p + 1:]
        elif kind == 3: ctx = ctx + ["zzzunk"]
        if rng.random() < 0.2: ctx = [rng.choice(hot)]
        queries.append(" ".join(ctx + [w]).encode())
    queries += [b"", b"   ", b"zz", rng.choice(hot).encode()]
    qb, qo = oracle.pack_strings(queries)
    for _ in range(2):
        top_k, sim = rng.choice([1, 3, 5, 20, 70]), rng.choice([0.3, 0.5, 0.8])
        ids, cnt = sc.predict_batch(blob=qb, offs=qo, top_k=top_k, similarity=sim)
        oi, oc = ora_lm.predict_batch(ora_ix, qb, qo, top_k, sim)
        valid = (np.arange(top_k + 1)[None, :] < np.minimum(oc, top_k + 1)[:, None]) & (oc < 0xFFFFFFF0)[:, None]
        if not (np.array_equal(cnt, oc) and np.array_equal(ids[valid], oi[valid])):
            bad += 1
            rows = np.nonzero((cnt != oc) | (valid & (ids != oi)).any(axis=1))[0]
            print("MISMATCH seed %d order %d vocab %d topK %d sim %.1f rows %s query %r gpu %s oracle %s" % (seed, order, len(vocab), top_k, sim, rows[:5], queries[int(rows[0])], ids[int(rows[0])][:6], oi[int(rows[0])][:6]), flush=True)
print("fuzz_spell: %d trials, %d mismatches" % (trial, bad))
sys.exit(1 if bad else 0)
