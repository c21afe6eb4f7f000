#!/usr/bin/env python3
"""Where do a wavefront's cycles go?  Runs the headline batch on the SG_PHASE_TIMING build of the kernel
(s_memtime brackets per phase) and prints the share of each phase.  GPU box only:
   make -C suggest_amd/csrc prof && python tools/phase_timing.py [--dict-size N]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from suggest_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "suggest_amd", os.environ.get("SG_PROF_LIB", "libsuggest_hip_prof.so"))
from suggest_amd import IndexDescription, NGramIndex, synth

ap = argparse.ArgumentParser()
ap.add_argument("--dict-size", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=65536)
ap.add_argument("--metric", default="jaccard")
ap.add_argument("--similarity", type=float, default=0.5)
ap.add_argument("--topk", type=int, default=10)
ap.add_argument("--ngram", type=int, default=3)
ap.add_argument("--dict-variant", default="uniform")
ap.add_argument("--golden", default=None, choices=[None, "cars", "words"], help="use the reference's cars / words dictionary instead")
ap.add_argument("--autocomplete", type=int, default=0,
                help="autocomplete mode: prefixes of this many letters (+ 0..2) over a vocabulary of --dict-size random words of 3..12 letters "
                     "(the index of BASELINE config 5), limit = --topk")
args = ap.parse_args()
if args.autocomplete:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    rng = np.random.Generator(np.random.PCG64(7))
    need = int(args.dict_size * 1.3) + 64
    ln = rng.integers(3, 13, size=need)
    chars = rng.integers(0, 26, size=(need, 12), dtype=np.uint8) + ord("a")
    chars[np.arange(12)[None, :] >= ln[:, None]] = 0
    words = [bytes(w) for w in np.unique(np.ascontiguousarray(chars).view("S12").ravel())[:args.dict_size]]
    pick = rng.integers(0, len(words), size=args.queries)
    extra = rng.integers(0, 3, size=args.queries)
    qb, qo = oracle.pack_strings([words[int(i)][:args.autocomplete + int(e)] for i, e in zip(pick, extra)])
    ix = NGramIndex(words, IndexDescription(ngram_size=3, wrap=("^", "$"), pad="$", alphabet=("english", "russian", "numbers", "$^'")))
elif args.golden:
    import lzma
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from conftest import CARS_DESC, WORDS_DESC
    G = os.path.join(ROOT, "tests", "golden")
    lines = open(os.path.join(G, "cars.dict"), "rb").read().splitlines() if args.golden == "cars" else lzma.open(os.path.join(G, "words.dict.xz")).read().splitlines()
    rnd = np.random.RandomState(1)
    qs = []
    for i in rnd.randint(0, len(lines), size=args.queries):
        w = bytearray(lines[int(i)])
        if len(w) > 2:
            w[int(rnd.randint(0, len(w)))] = ord("x")
        qs.append(bytes(w))
    qb, qo = oracle.pack_strings(qs)
    ix = NGramIndex(lines, IndexDescription(**(CARS_DESC if args.golden == "cars" else WORDS_DESC)))
else:
    blob, offs = synth.make_dict(args.dict_size, seed=1, skewed="skewed" in args.dict_variant, families=3 if "families" in args.dict_variant else 0)
    qb, qo = synth.make_queries(args.queries, blob, offs, seed=2)
    ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**dict(synth.DESCRIPTION, ngram_size=args.ngram)))
dev = torch.device("cuda", 0)
prof = torch.zeros(2 * 4096 * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
L.sg_debug_set_prof.argtypes = [C.c_void_p]
L.sg_debug_set_prof(prof.data_ptr())
k, n_q = args.topk, args.queries
d_q = torch.from_numpy(qb).to(dev); d_offs = torch.from_numpy(qo.view(np.int64)).to(dev)
d_ids = torch.zeros((n_q, k), dtype=torch.int32, device=dev); d_sc = torch.zeros((n_q, k), dtype=torch.float64, device=dev)
d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
def run():
    st = torch.cuda.current_stream().cuda_stream
    if args.autocomplete:
        ix.autocomplete_batch_device(d_q.data_ptr(), d_offs.data_ptr(), n_q, k, d_ids.data_ptr(), d_cnt.data_ptr(), stream=st)
    else:
        ix.suggest_batch_device(d_q.data_ptr(), d_offs.data_ptr(), n_q, args.metric, args.similarity, k, d_ids.data_ptr(), d_sc.data_ptr(),
                                d_cnt.data_ptr(), stream=st)
for it in range(3):
    prof.zero_()
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(5):
    run()
e1.record(); torch.cuda.synchronize()
print("SG_DEBUG_SKIP=%s kernel ms (instrumented build): %.3f" % (os.environ.get("SG_DEBUG_SKIP", "0"), e0.elapsed_time(e1) / 5))
allp = prof.cpu().numpy().astype(np.float64).reshape(2, 4096, 8).sum(axis=1) / 6
p = allp[0]
cn = allp[1] / n_q
print('per query: groups %.1f batches %.1f flag_events %.1f queued %.2f passes %.1f emitted %.2f skipped_chunks %.0f of %.0f' % tuple(cn))
names = ["tokenize", "tile rows + segment stats", "group setup (merge, scan, geometry)", "clear counters", "verify + emit queued candidates",
         "stream (loads + count)", "slow path (flagged)", "top-k sort + output"]
tot = p.sum()
print("cycles per query (wave-time, s_memtime): %.0f" % (tot / n_q))
for n, v in zip(names, p):
    print("  %-40s %10.0f  %5.1f%%" % (n, v / n_q, 100 * v / tot))
