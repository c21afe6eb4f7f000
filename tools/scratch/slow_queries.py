import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fuzz_parity as fp, oracle, numpy as np
from suggest_amd import IndexDescription, NGramIndex
seed, scale = int(sys.argv[1]), int(sys.argv[2])
t = fp.make_trial(seed, scale)
gpu = NGramIndex(t["docs"], IndexDescription(**t["desc"]), build=t["build"])
ora = oracle.OracleIndex(t["docs"], **t["desc"])
ntok = [len(ora.tokenize(q)) for q in t["queries"]]
import collections
print("docs by tokens:", sorted(collections.Counter(len(ora.tokenize(d)) for d in t["docs"][::50]).items())[:60], flush=True)
for metric, a, k in t["searches"]:
    for q, n in zip(t["queries"], ntok):
        if n > 128: continue
        qb, qo = oracle.pack_strings([q])
        t0 = time.time(); ids, sc, cnt = gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=a, k=k); dt = time.time() - t0
        if dt > 0.05:
            t1 = time.time(); ora.suggest_batch(qb, qo, metric, a, k); do = time.time() - t1
            print(metric, a, k, "%.2f s (oracle %.3f s)" % (dt, do), n, "tokens", repr(q), "count", int(cnt[0]), flush=True)
