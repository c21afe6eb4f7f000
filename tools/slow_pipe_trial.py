"""[r6] Device time of one `fuzz_parity.py --pipe` trial's searches, under whatever build SG_LIB_NAME names.
GPU box:  python tools/slow_pipe_trial.py SEED SCALE [NAME=VALUE ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fuzz_parity as fp, oracle, numpy as np
from suggest_amd import IndexDescription, NGramIndex
seed, scale = int(sys.argv[1]), int(sys.argv[2])
t = fp.make_trial(seed, scale)
t["env"].update(SG_PIPE="1", SG_PRETOK="1", SG_TIGHTEN="0", SG_G8="0", SG_SPLIT_CHUNKS="0", SG_LOG2_CNT="9")
t["env"].update(dict(x.split("=") for x in sys.argv[3:]))
t["searches"] = [(m_, a_, min(k_, 64)) for m_, a_, k_ in t["searches"]]
os.environ.update(t["env"])
print(os.environ.get("SG_LIB_NAME", "libsuggest_hip.so"), t["desc"], len(t["docs"]), "docs", len(t["queries"]), "queries", t["env"], flush=True)
gpu = NGramIndex(t["docs"], IndexDescription(**t["desc"]), build=t["build"])
qb, qo = oracle.pack_strings(t["queries"])
for metric, a, k in t["searches"]:
    s0 = gpu.pipe_stats()
    t0 = time.time(); ids, sc, cnt = gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=a, k=k)
    s1 = gpu.pipe_stats()
    print("  %-8s %.2f k=%-3d %8.3f s   results %d   pipeline %s" % (metric, a, k, time.time() - t0, int(np.minimum(cnt, k).sum()), {n: s1[n] - s0[n] for n in s1}), flush=True)
