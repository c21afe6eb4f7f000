#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity hunt (GPU box only): random descriptions, dictionaries with heavy repetition (ties,
repeated terms, overflowing groups), random metrics / k / tuning knobs, suggest + autocomplete.  Prints the seed of any
mismatch.   python tools/fuzz_parity.py --seconds 300 [--seed 1]"""
import argparse, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import oracle
from suggest_amd import IndexDescription, NGramIndex

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
t_end = time.time() + args.seconds
trial, bad = 0, 0
while time.time() < t_end:
    seed = args.seed * 100000 + trial
    trial += 1
    t_trial = time.time()
    rng = random.Random(seed)
    q = rng.choice([1, 2, 2, 3, 3, 3, 4, 5])
    alpha = rng.choice([("english",), ("english", "numbers"), ("ab", "$"), ("russian", "english", "numbers", "$"), ("abc", "-")])
    wrap = rng.choice([("$", "$"), ("^", "$"), ("", ""), (" ", " "), ("$$", "")])
    pad = rng.choice(["$", "_", ""]) if q <= 4 else "$"
    if q * max(1, len(pad)) > 8:
        continue
    syms = rng.choice(["ab", "abc -", "abcdefgh 12", "абвгд ёab", "AbC.dE f", "abcdefghijklmnopqrstuvwxyz"])
    n_docs = rng.choice([1, 5, 50, 400, 3000, 20000])
    max_len = rng.choice([6, 14, 30, 60])
    base = ["".join(rng.choice(syms) for _ in range(rng.randint(0, max_len))) for _ in range(max(1, n_docs // rng.choice([1, 1, 4, 20])))]
    docs = []
    for _ in range(n_docs):
        w = list(rng.choice(base))
        for _ in range(rng.randint(0, 2)):
            if w:
                w[rng.randrange(len(w))] = rng.choice(syms)
        docs.append("".join(w))
    desc = dict(ngram_size=q, wrap=wrap, pad=pad, alphabet=alpha)
    if len(oracle.OracleIndex([docs[0]], **desc).tokenize(docs[0])) == 0:
        docs[0] = "abcabcab"
        if len(oracle.OracleIndex([docs[0]], **desc).tokenize(docs[0])) == 0:
            continue
    for name, choices in (("SG_LOG2_CNT", ["9", "10", "11", "12"]), ("SG_T_FLOOR", ["2", "4", "10", "30"]), ("SG_FILTER_LEVEL", ["0", "2", "3"]),
                          ("SG_SPLIT_CHUNKS", ["0", "1", "8", "65536"])):
        os.environ[name] = rng.choice(choices)
    try:
        gpu = NGramIndex(docs, IndexDescription(**desc), build=rng.choice(["host", "device"]) if max_len <= 60 else "host")
    except Exception as exc:  # unsupported description (key does not fit) is fine; anything else is not
        if "fit" in str(exc) or "UNSUPPORTED" in str(exc) or "-2" in str(exc):
            continue
        raise
    ora = oracle.OracleIndex(docs, **desc)
    queries = [rng.choice(docs) for _ in range(30)] + ["".join(rng.choice(syms) for _ in range(rng.randint(0, max_len + 6))) for _ in range(30)]
    queries += [d[:rng.randint(0, len(d))] + rng.choice(syms) + d[rng.randint(0, len(d)):] for d in rng.sample(docs, min(20, len(docs)))]
    qb, qo = oracle.pack_strings(queries)
    for _ in range(3):
        metric = rng.choice(["jaccard", "cosine", "dice", "overlap", "exact"])
        a = 1.0 if metric == "exact" else rng.choice([0.15, 0.3, 0.5, 0.7, 0.9, 1.0])
        k = rng.choice([1, 2, 5, 10, 64, 65, 300])
        ids, sc, cnt = gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=a, k=k)
        oi, os_, oc, _ = ora.suggest_batch(qb, qo, metric, a, k)
        valid = (np.arange(k)[None, :] < np.minimum(oc, k)[:, None]) & (oc < 0xFFFFFFF0)[:, None]
        if not (np.array_equal(cnt, oc) and np.array_equal(ids[valid], oi[valid]) and np.array_equal(sc.view(np.uint64)[valid], os_.view(np.uint64)[valid])):
            bad += 1
            rows = np.nonzero((cnt != oc) | (valid & ((ids != oi) | (sc.view(np.uint64) != os_.view(np.uint64)))).any(axis=1))[0]
            print("MISMATCH seed %d: %s %s a=%.2f k=%d env=%s rows=%s q=%r" % (seed, desc, metric, a, k, {n: os.environ[n] for n in ("SG_LOG2_CNT", "SG_T_FLOOR", "SG_FILTER_LEVEL", "SG_SPLIT_CHUNKS")}, rows[:5], queries[int(rows[0])]), flush=True)
    limit = rng.choice([1, 7, 100])
    ids, cnt = gpu.autocomplete_batch(blob=qb, offs=qo, limit=limit)
    oi, oc, _ = ora.autocomplete_batch(qb, qo, limit)
    valid = np.arange(limit)[None, :] < np.minimum(oc, limit)[:, None]
    if not (np.array_equal(cnt, oc) and np.array_equal(ids[valid], oi[valid])):
        bad += 1
        print("MISMATCH (autocomplete) seed %d: %s limit=%d" % (seed, desc, limit), flush=True)
    gpu.close()
    if time.time() - t_trial > 5:
        print("slow trial: seed %d took %.1f s: %s, %d docs, syms %r" % (seed, time.time() - t_trial, desc, n_docs, syms), flush=True)
print("fuzz: %d trials, %d mismatches" % (trial, bad))
sys.exit(1 if bad else 0)
