#!/usr/bin/env python3
"""cfg 5 (SpellChecker.Predict, 50 M-token model): step time under different tuning knobs of the vocabulary index, ONE model
build.  GPU box:  python tools/spell_sweep.py "SG_FILTER_LEVEL=2" "SG_FILTER_LEVEL=4,SG_T_FLOOR=4" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import make_synthetic_lm
from suggest_amd.spell import LanguageModel, SpellChecker
from suggest_amd.index import pack_strings

tokens, vocab, n_q, k = 50_000_000, 1_000_000, 65536, 5
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "sg_lm_%d_%d_r0" % (tokens, vocab))
info = make_synthetic_lm.make(d, tokens=tokens, vocab=vocab, verbose=False)
lm = LanguageModel(binary=os.path.join(d, "synth.lm"), dictionary=os.path.join(d, "synth.cdb"))
sc = SpellChecker(lm, device=0)
dev = torch.device("cuda", 0)
batches = [pack_strings(make_synthetic_lm.make_queries(info, n_q, 100 + b)) for b in range(4)]
d_q = [torch.from_numpy(qb).to(dev) for qb, _ in batches]
d_o = [torch.from_numpy(qo.view(np.int64)).to(dev) for _, qo in batches]
d_ids = torch.zeros((n_q, k + 1), dtype=torch.int32, device=dev); d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ref = None
for spec in (sys.argv[1:] or ["SG_FILTER_LEVEL=2"]):
    knobs = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",") if kv}
    sc.index.tune(**knobs)

    def step(b):
        sc.predict_batch_device(d_q[b].data_ptr(), d_o[b].data_ptr(), n_q, int(batches[b][1][-1]), k, 0.5, d_ids.data_ptr(), d_cnt.data_ptr(), stream=st)
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        step(i % 4)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    step(0); torch.cuda.synchronize()
    rows = (d_ids.cpu().numpy().copy(), d_cnt.cpu().numpy().copy())
    same = ref is None or (np.array_equal(rows[1], ref[1]) and np.array_equal(rows[0], ref[0]))
    ref = ref or rows
    print("%-44s %.3f ms per step = %.2f M predictions/s   rows equal the first setting's: %s" % (spec, ms, n_q / ms / 1e3, same), flush=True)
