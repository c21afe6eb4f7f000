"""[r6] One fuzz trial's searches repeated many times in one process (a mismatch that a single replay does not show: races).
GPU box:  python tools/fuzz_repeat.py SEED [REPEATS] [NAME=VALUE ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import numpy as np
import fuzz_parity as fp, oracle
from suggest_amd import IndexDescription, NGramIndex
seed = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t = fp.make_trial(seed, 1)
t["env"].update(dict(x.split("=") for x in sys.argv[3:]))
os.environ.update(t["env"])
gpu = NGramIndex(t["docs"], IndexDescription(**t["desc"]), build=t["build"])
ora = oracle.OracleIndex(t["docs"], **t["desc"])
qb, qo = oracle.pack_strings(t["queries"])
bad = 0
for metric, a, k in t["searches"]:
    oi, os_, oc, _ = ora.suggest_batch(qb, qo, metric, a, k)
    valid = np.arange(k)[None, :] < np.minimum(oc, k)[:, None]
    for r in range(reps):
        ids, sc, cnt = gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=a, k=k)
        ok = np.array_equal(cnt, oc) and np.array_equal(ids[valid], oi[valid]) and np.array_equal(sc.view(np.uint64)[valid], os_.view(np.uint64)[valid])
        if not ok:
            bad += 1
            rows = np.nonzero((cnt != oc) | ((ids != oi) & valid).any(axis=1))[0]
            print("MISMATCH %s %.2f k=%d repeat %d rows %s" % (metric, a, k, r, rows[:8]), flush=True)
print("seed %d: %d searches x %d repeats, %d mismatching launches; env %s" % (seed, len(t["searches"]), reps, bad, t["env"]))
