#!/usr/bin/env python3
"""Where do a wavefront's cycles go in the two search launches of SpellChecker.Predict on BASELINE config 5's real query
mix (50 M-token model, 'w1 w2 prefix' queries, a third with a typo)?  SG_PHASE_TIMING build (s_memtime brackets per phase);
the LM-ranked autocomplete launch and the fuzzy top-up launch are profiled in turn (SG_DEBUG_SKIP bits 65536 / 131072
switch the other instantiation's counters off).  GPU box:
   make -C suggest_amd/csrc prof && python tools/phase_timing_cfg5.py [--tokens N]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from suggest_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "suggest_amd", os.environ.get("SG_PROF_LIB", "libsuggest_hip_prof.so"))
import make_synthetic_lm
from suggest_amd.spell import LanguageModel, SpellChecker
from suggest_amd.index import pack_strings

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=50_000_000)
ap.add_argument("--vocab", type=int, default=1_000_000)
ap.add_argument("--queries", type=int, default=65536)
ap.add_argument("--topk", type=int, default=5)
args = ap.parse_args()
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "sg_lm_%d_%d_r0" % (args.tokens, args.vocab))
info = make_synthetic_lm.make(d, tokens=args.tokens, vocab=args.vocab, verbose=False)
lm = LanguageModel(binary=os.path.join(d, "synth.lm"), dictionary=os.path.join(d, "synth.cdb"))
sc = SpellChecker(lm, device=0)
qb, qo = pack_strings(make_synthetic_lm.make_queries(info, args.queries, 100))
dev = torch.device("cuda", 0)
n_q, k = args.queries, args.topk
prof = torch.zeros(2 * 4096 * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
L.sg_debug_set_prof.argtypes = [C.c_void_p]
L.sg_debug_set_prof(prof.data_ptr())
d_q = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.view(np.int64)).to(dev)
d_ids = torch.zeros((n_q, k + 1), dtype=torch.int32, device=dev); d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    sc.predict_batch_device(d_q.data_ptr(), d_o.data_ptr(), n_q, int(qo[-1]), k, 0.5, d_ids.data_ptr(), d_cnt.data_ptr(), stream=st)


names = ["tokenize", "tile rows + segment stats", "group setup (merge, scan, geometry)", "clear counters", "verify + emit queued candidates",
         "stream (loads + count)", "slow path (flagged)", "top-k sort + output"]
for label, bits in (("LM-ranked autocomplete launch (all queries)", 65536), ("fuzzy top-up launch (the selected subset)", 131072)):
    os.environ["SG_DEBUG_SKIP"] = str(bits)
    for it in range(3):
        prof.zero_(); torch.cuda.synchronize()
        run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    allp = prof.cpu().numpy().astype(np.float64).reshape(2, 4096, 8).sum(axis=1) / 6
    p = allp[0]
    n_run = n_q if bits == 65536 else max(1, int((d_cnt.cpu().numpy() >= 0).sum()))
    cn = allp[1] / n_q
    print("== %s; Predict step %.3f ms (instrumented build)" % (label, e0.elapsed_time(e1) / 5))
    print('per query of the batch: groups %.2f half-batches %.2f flag events %.2f queued %.2f passes %.2f emitted %.2f kept-verdicts %.2f verified %.2f' % tuple(cn))
    tot = p.sum()
    print("cycles per query of the batch (wave-time, s_memtime ticks of 10 ns): %.0f" % (tot / n_q))
    for n, v in zip(names, p):
        print("  %-40s %10.0f  %5.1f%%" % (n, v / n_q, 100 * v / tot))
