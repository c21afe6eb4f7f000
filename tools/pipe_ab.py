#!/usr/bin/env python3
"""[r5] Same-box A/B of the three-launch pipeline (plan -> stream -> verify, pipeline.inc) against the fused kernel on one
resident index: time per call of sg_suggest_batch_device and every row compared bit for bit with the fused path's.
   python tools/pipe_ab.py --config headline --variants "nw=8,sub=4;nw=4,sub=4" """
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from suggest_amd import IndexDescription, NGramIndex, synth, _lib
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="headline")
ap.add_argument("--variants", default="nw=8,sub=4")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dict-variant", default="uniform")
ap.add_argument("--dict-size", type=int, default=0)
ap.add_argument("--queries", type=int, default=0)
ap.add_argument("--build", default="device")
ap.add_argument("--no-fused", action="store_true", help="skip the fused-kernel reference run (timing of the pipeline launches only; rows are not compared)")
args = ap.parse_args()
c = dict(bench.CONFIGS[args.config])
if args.dict_size:
    c["dict_size"] = args.dict_size
if args.queries:
    c["queries"] = args.queries
desc = dict(synth.DESCRIPTION, ngram_size=c["ngram"])
blob, offs = synth.make_dict(c["dict_size"], seed=1, skewed="skewed" in args.dict_variant, families=3 if "families" in args.dict_variant else 0)
qb, qo = synth.make_queries(c["queries"], blob, offs, seed=2)
ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**desc), build=args.build)
dev = torch.device("cuda", 0)
k, n_q = c["topk"], c["queries"]
d_q = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.view(np.int64)).to(dev)
d_ids = torch.zeros((n_q, k), dtype=torch.int32, device=dev); d_sc = torch.zeros((n_q, k), dtype=torch.float64, device=dev)
d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
NAMES = dict(dt="SG_PIPE_DT_BYTES", nw="SG_PIPE_NW", order="SG_ORDER", sub="SG_PIPE_SUB", cnt="SG_PIPE_LOG2_CNT", ccap="SG_PIPE_CAND_CAP",
             level="SG_FILTER_LEVEL", floor="SG_T_FLOOR", auto="SG_PIPE_SHAPE_AUTO")


def run():
    ix.suggest_batch_device(d_q.data_ptr(), d_o.data_ptr(), n_q, c["metric"], c["similarity"], k, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), stream=st)


def measure(label):
    d_ids.zero_(); d_sc.zero_(); d_cnt.fill_(-7)
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    cnt = d_cnt.cpu().numpy().copy(); ids = d_ids.cpu().numpy().copy(); sc = d_sc.cpu().numpy().copy()
    # rows are compared up to their count (what lies behind it is whatever an earlier launch left)
    m = np.arange(k)[None, :] < np.clip(cnt, 0, k)[:, None]
    return ms, (cnt, np.where(m, ids, 0), np.where(m, sc.view(np.uint64), 0))


ix.tune(SG_PIPE=0)
ms0, ref = (1.0, None) if args.no_fused else measure("fused")
if ref is not None: print("%s fused: %.3f ms  %.2f M q/s  results %d" % (args.config, ms0, n_q / ms0 / 1e3, int(np.clip(ref[0], 0, k).sum())), flush=True)
for var in args.variants.split(";"):
    kn = {"SG_PIPE": 1}
    for kv in var.split(","):
        if kv:
            a, b = kv.split("=")
            kn[NAMES[a]] = int(b)
    ix.tune(**kn)
    have_pv = hasattr(_lib.lib(), "sg_index_pipe_volumes")      # (an older build of the library under SG_LIB_NAME)
    ps0 = ix.pipe_stats(); ls0 = ix.launch_stats(); pv0 = ix.pipe_volumes() if have_pv else None
    ms, res = measure(var)
    ps1 = ix.pipe_stats(); ls1 = ix.launch_stats(); pv1 = ix.pipe_volumes() if have_pv else None
    if have_pv:
        ns = max(1, pv1["sampled"] - pv0["sampled"])
        print("   per sampled query: groups %.2f lists %.1f rows %.1f candidates %.2f  (stream workgroup: %s)" % (tuple((pv1[k_] - pv0[k_]) / ns for k_ in ("groups", "lists", "rows", "candidates")) + (pv1.get("stream_shape"),)), flush=True)
    print("   chunks streamed per sampled query: %.0f" % ((ls1["chunks"] - ls0["chunks"]) / max(1, ls1["sampled"] - ls0["sampled"])), flush=True)
    fb = {k_: (ps1[k_] - ps0[k_]) / (args.steps + 2.0) for k_ in ps1}
    same = ref is not None and all(np.array_equal(x, y) for x, y in zip(res, ref))
    bad = 0 if ref is None else int((res[0] != ref[0]).sum()) + int((res[1] != ref[1]).any(axis=1).sum())
    print("%s pipe %-28s: %.3f ms  %.2f M q/s  (%.3fx)  same_results=%s%s  fallback/launch: unplanned %.0f overflow %.0f repeats %.0f" %
          (args.config, var, ms, n_q / ms / 1e3, ms0 / ms, same, "" if same else "  rows differing ~%d" % bad, fb["unplanned"], fb["overflow"], fb["repeats"]), flush=True)
ix.tune(SG_PIPE=0)
