#!/usr/bin/env python3
"""Kernel time of a BASELINE config under arbitrary knob settings of ONE resident index (GPU box):
   python tools/sweep_any.py --config cfg4 [--dict-variant skewed] "SG_SPLIT_CHUNKS=32768" "SG_SPLIT_CHUNKS=65536,SG_PARTS_CNT_BONUS=1" ...
(every setting is applied on top of the previous ones through sg_index_tune; the rows are compared with the first setting's)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from suggest_amd import IndexDescription, NGramIndex, synth
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="headline")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dict-variant", default="uniform")
ap.add_argument("specs", nargs="*")
args = ap.parse_args()
c = bench.CONFIGS[args.config]
desc = dict(synth.DESCRIPTION, ngram_size=c["ngram"])
blob, offs = synth.make_dict(c["dict_size"], seed=1, skewed="skewed" in args.dict_variant, families=3 if "families" in args.dict_variant else 0)
n_b = 4
batches = [synth.make_queries(c["queries"], blob, offs, seed=2, start=b * c["queries"]) for b in range(n_b)]
ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**desc), build="device")
dev = torch.device("cuda", 0)
k, n_q = c["topk"], c["queries"]
d_q = [torch.from_numpy(qb).to(dev) for qb, _ in batches]; d_o = [torch.from_numpy(qo.view(np.int64)).to(dev) for _, qo in batches]
d_ids = torch.zeros((n_q, k), dtype=torch.int32, device=dev); d_sc = torch.zeros((n_q, k), dtype=torch.float64, device=dev)
d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ref = None
for spec in (args.specs or [""]):
    knobs = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",") if kv}
    if knobs:
        ix.tune(**knobs)

    def run(b):
        ix.suggest_batch_device(d_q[b].data_ptr(), d_o[b].data_ptr(), n_q, c["metric"], c["similarity"], k, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), stream=st)
    for i in range(10):      # (the launch-to-launch choices settle within ~8 launches)
        run(i % n_b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.steps):
        run(i % n_b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    run(0); torch.cuda.synchronize()
    rows = (d_ids.cpu().numpy().copy(), d_sc.cpu().numpy().view(np.uint64).copy(), d_cnt.cpu().numpy().copy())
    valid = np.arange(k)[None, :] < np.minimum(rows[2], k)[:, None]
    same = ref is None or (np.array_equal(rows[2], ref[2]) and np.array_equal(rows[0][valid], ref[0][valid]) and np.array_equal(rows[1][valid], ref[1][valid]))
    ref = ref or rows
    print("%s %-8s %-52s %.3f ms per step = %.3f M q/s   same rows: %s" % (args.config, args.dict_variant, spec or "(defaults)", ms, n_q / ms / 1e3, same), flush=True)
