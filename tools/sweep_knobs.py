#!/usr/bin/env python3
"""Knob sweep on one resident index (GPU box): kernel time of a BASELINE config under combinations of the tuning knobs.
   python tools/sweep_knobs.py --config cfg3 --levels 2,5,6,7 --floors 10,8,6 --cnt 11,12"""
import argparse, itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from suggest_amd import IndexDescription, NGramIndex, synth
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="headline")
ap.add_argument("--levels", default="2,5,6,7")
ap.add_argument("--floors", default="10,8,6")
ap.add_argument("--cnt", default="")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dict-variant", default="uniform")
args = ap.parse_args()
c = bench.CONFIGS[args.config]
desc = dict(synth.DESCRIPTION, ngram_size=c["ngram"])
blob, offs = synth.make_dict(c["dict_size"], seed=1, skewed="skewed" in args.dict_variant, families=3 if "families" in args.dict_variant else 0)
qb, qo = synth.make_queries(c["queries"], blob, offs, seed=2)
ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**desc), build="device")
alg = ix.algorithmic_bytes(qb, qo, c["metric"], c["similarity"], c["topk"])
dev = torch.device("cuda", 0)
k, n_q = c["topk"], c["queries"]
d_q = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.view(np.int64)).to(dev)
d_ids = torch.zeros((n_q, k), dtype=torch.int32, device=dev); d_sc = torch.zeros((n_q, k), dtype=torch.float64, device=dev)
d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ref = None
cnts = [int(x) for x in args.cnt.split(",")] if args.cnt else [None]
for lvl, fl, lc in itertools.product([int(x) for x in args.levels.split(",")], [int(x) for x in args.floors.split(",")], cnts):
    kn = dict(SG_FILTER_LEVEL=lvl, SG_T_FLOOR=fl)
    if lc is not None:
        kn["SG_LOG2_CNT"] = lc
    ix.tune(**kn)
    def run():
        ix.suggest_batch_device(d_q.data_ptr(), d_o.data_ptr(), n_q, c["metric"], c["similarity"], k, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), stream=st)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    res = (d_ids.cpu().numpy().tobytes(), d_sc.cpu().numpy().tobytes(), d_cnt.cpu().numpy().tobytes())
    same = True if ref is None else res == ref
    ref = ref or res
    print("%s level %d floor %2d cnt %s: %.3f ms  %.2f M q/s  alg %.0f GB/s (%.3f)  same_results=%s" %
          (args.config, lvl, fl, lc, ms, n_q / ms / 1e3, alg / ms / 1e6, alg / ms / 1e6 / 8000, same), flush=True)
