#!/usr/bin/env python3
"""A short pipelined host-buffer run (sg_suggest_submit / sg_ticket_wait, pinned buffers, N tickets in flight) for a timeline
trace:  rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d out -- python tools/async_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from suggest_amd import IndexDescription, NGramIndex, synth
from suggest_amd.index import pinned_array

n_dict = int(os.environ.get("SG_TRACE_DICT", 10_000_000))
n_q, k, depth, reps = 65536, 10, int(os.environ.get("SG_TRACE_DEPTH", 3)), int(os.environ.get("SG_TRACE_REPS", 12))
blob, offs = synth.make_dict(n_dict, seed=1)
ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION), build="device")
slots = []
for b in range(4):
    qb, qo = synth.make_queries(n_q, blob, offs, seed=2, start=b * n_q)
    pb = pinned_array((qb.size,), np.uint8); pb[:] = qb
    po = pinned_array((n_q + 1,), np.uint64); po[:] = qo
    slots.append((pb, po, pinned_array((n_q, k), np.uint32), pinned_array((n_q, k), np.float64), pinned_array((n_q,), np.uint32)))


def submit(i):
    pb, po, a, b, c = slots[i % 4]
    return ix.suggest_submit(pb, po, "jaccard", 0.5, k, a, b, c)


for i in range(4):
    submit(i).wait()
t0 = time.perf_counter()
pend, log = [], []
for i in range(reps):
    ts = time.perf_counter()
    pend.append(submit(i))
    tm = time.perf_counter()
    if len(pend) >= depth:
        pend.pop(0).wait()
    log.append((tm - ts, time.perf_counter() - tm))
for t in pend:
    t.wait()
dt = time.perf_counter() - t0
print("depth %d: %.2f M q/s; per iteration submit / wait ms: %s" % (depth, reps * n_q / dt / 1e6, ["%.2f/%.2f" % (a * 1e3, b * 1e3) for a, b in log]))
