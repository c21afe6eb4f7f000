"""Device time of a fuzz trial's searches, short and long queries apart.
GPU box:  python tools/slow_trial_timing.py SEED SCALE [default]   (default: without the trial's SG_* knobs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fuzz_parity as fp, oracle, numpy as np
from suggest_amd import IndexDescription, NGramIndex
seed, scale = int(sys.argv[1]), int(sys.argv[2])
t = fp.make_trial(seed, scale)
if len(sys.argv) > 3 and sys.argv[3] == "default":
    t["env"] = {}
os.environ.update(t["env"])
print(t["desc"], len(t["docs"]), t["env"], t["build"], t["searches"], flush=True)
t0 = time.time(); gpu = NGramIndex(t["docs"], IndexDescription(**t["desc"]), build=t["build"]); print("build", time.time() - t0, flush=True)
ora = oracle.OracleIndex(t["docs"], **t["desc"])
ntok = [len(ora.tokenize(q)) for q in t["queries"]]
short = [q for q, n in zip(t["queries"], ntok) if n <= 128]
longq = [q for q, n in zip(t["queries"], ntok) if n > 128]
print(len(short), "short", len(longq), "long", sorted(ntok)[-5:], flush=True)
for name, qs in (("short", short), ("short x24", short * 24), ("long", longq[:2])):
    if not qs: continue
    qb, qo = oracle.pack_strings(qs)
    for metric, a, k in t["searches"]:
        t0 = time.time(); ids, sc, cnt = gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=a, k=k)
        print(name, len(qs), metric, a, k, "%.2f s" % (time.time() - t0), "max count", int(cnt.max()), flush=True)
    t0 = time.time(); gpu.autocomplete_batch(blob=qb, offs=qo, limit=t["limit"]); print(name, "ac %.2f s" % (time.time() - t0), flush=True)
