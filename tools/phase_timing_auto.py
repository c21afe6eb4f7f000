#!/usr/bin/env python3
"""Where do a wavefront's cycles go in AUTOCOMPLETE over a vocabulary like cfg 5's (1 M random words of 3..12 letters, ids in
random order; prefixes of 2/3 of a word)?  SG_PHASE_TIMING build, GPU box:  python tools/phase_timing_auto.py [--limit 5]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from suggest_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "suggest_amd", os.environ.get("SG_PROF_LIB", "libsuggest_hip_prof.so"))
from suggest_amd import IndexDescription, NGramIndex
from suggest_amd.index import pack_strings

ap = argparse.ArgumentParser()
ap.add_argument("--vocab", type=int, default=1_000_000)
ap.add_argument("--queries", type=int, default=65536)
ap.add_argument("--limit", type=int, default=5)
args = ap.parse_args()
rng = np.random.Generator(np.random.PCG64(7))
need = int(args.vocab * 1.3) + 64
ln = rng.integers(3, 13, size=need)
chars = rng.integers(0, 26, size=(need, 12), dtype=np.uint8) + ord("a")
chars[np.arange(12)[None, :] >= ln[:, None]] = 0
raw = np.unique(np.ascontiguousarray(chars).view("S12").ravel())
raw = raw[rng.permutation(len(raw))[:args.vocab]]
words = raw.tolist()
desc = IndexDescription(ngram_size=3, wrap=("$", "$"), pad="$", alphabet=("english", "$"))
ix = NGramIndex(words, desc)
picks = rng.integers(0, len(words), size=args.queries)
qs = [words[int(i)][:max(2, (len(words[int(i)]) * 2 + 2) // 3)] for i in picks]
qb, qo = pack_strings(qs)
dev = torch.device("cuda", 0)
prof = torch.zeros(2 * 4096 * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
L.sg_debug_set_prof.argtypes = [C.c_void_p]
L.sg_debug_set_prof(prof.data_ptr())
k, n_q = args.limit, args.queries
d_q = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.view(np.int64)).to(dev)
d_ids = torch.zeros((n_q, k), dtype=torch.int32, device=dev); d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    prof.zero_(); torch.cuda.synchronize()
    ix.autocomplete_batch_device(d_q.data_ptr(), d_o.data_ptr(), n_q, k, d_ids.data_ptr(), d_cnt.data_ptr(), stream=st)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(5):
    ix.autocomplete_batch_device(d_q.data_ptr(), d_o.data_ptr(), n_q, k, d_ids.data_ptr(), d_cnt.data_ptr(), stream=st)
e1.record(); torch.cuda.synchronize()
print("autocomplete limit %d, %d words: kernel ms (instrumented build): %.3f; mean prefix length %.1f, results per query %.2f"
      % (k, len(words), e0.elapsed_time(e1) / 5, np.mean([len(q) for q in qs]), float(np.minimum(d_cnt.cpu().numpy(), k).mean())))
allp = prof.cpu().numpy().astype(np.float64).reshape(2, 4096, 8).sum(axis=1) / 6
p = allp[0]; cn = allp[1] / n_q
print('per query: groups %.1f sub-batches %.1f flagged postings %.1f queued %.2f passes %.1f emitted %.2f kept-verdicts %.0f verified %.0f' % tuple(cn))
names = ["tokenize", "tile rows + segment stats", "group setup (merge, scan, geometry)", "clear counters", "verify + emit queued candidates",
         "stream (loads + count)", "slow path (flagged)", "top-k sort + output"]
tot = p.sum()
print("cycles per query (wave-time, s_memtime): %.0f" % (tot / n_q))
for n, v in zip(names, p):
    print("  %-40s %10.0f  %5.1f%%" % (n, v / n_q, 100 * v / tot))
