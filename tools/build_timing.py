#!/usr/bin/env python3
"""Host vs device index build on the synthetic dictionary: wall time of sg_index_build /
sg_index_build_device (host buffers in, host CSR out) and digest equality.  GPU box only."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from suggest_amd import IndexDescription, NGramIndex, synth

ap = argparse.ArgumentParser()
ap.add_argument("--dict-size", type=int, default=10_000_000)
ap.add_argument("--ngram", type=int, default=3)
args = ap.parse_args()
blob, offs = synth.make_dict(args.dict_size, seed=1)
desc = IndexDescription(**dict(synth.DESCRIPTION, ngram_size=args.ngram))
NGramIndex(blob=blob[:int(offs[1000])], offs=offs[:1001], description=desc, upload=False, build="device")   # warm up the context
t0 = time.perf_counter(); dev = NGramIndex(blob=blob, offs=offs, description=desc, upload=False, build="device"); t_dev = time.perf_counter() - t0
t0 = time.perf_counter(); host = NGramIndex(blob=blob, offs=offs, description=desc, upload=False); t_host = time.perf_counter() - t0
st = dev.stats()
print("dict %d strings q=%d: %d postings, %d terms, %d lists" % (args.dict_size, args.ngram, st["n_postings"], st["n_terms"], st["n_lists"]))
print("host build   %.2f s  (%.1f M postings/s)" % (t_host, st["n_postings"] / t_host / 1e6))
print("device build %.2f s  (%.1f M postings/s, incl. H2D of the strings and D2H of the CSR)" % (t_dev, st["n_postings"] / t_dev / 1e6))
print("identical arrays:", dev.digest() == host.digest())
