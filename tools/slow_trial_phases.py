"""Phase shares and event counts (SG_PHASE_TIMING build) of two queries of fuzz trial 4200037 x10 — DESIGN.md §4.8.
GPU box, after `make -C suggest_amd/csrc prof`:  [SG_TIGHTEN=0|1|2] python tools/slow_trial_phases.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from suggest_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "suggest_amd", "libsuggest_hip_prof.so")
import fuzz_parity as fp, oracle
from suggest_amd import IndexDescription, NGramIndex
from suggest_amd.metric import resolve
t = fp.make_trial(4200037, 10)
gpu = NGramIndex(t["docs"], IndexDescription(**t["desc"]), build="host")
dev = torch.device("cuda", 0)
prof = torch.zeros(2 * 4096 * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
L.sg_debug_set_prof.argtypes = [C.c_void_p]
L.sg_debug_set_prof(prof.data_ptr())
names = ["tokenize", "tile rows + segment stats", "group setup", "clear counters", "verify + emit", "stream", "slow path (flagged)", "top-k sort + output"]
def run(qs, metric, a, k, reps=1):
    qb, qo = oracle.pack_strings(qs)
    n_q = len(qs)
    d_q = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.view(np.int64)).to(dev)
    d_ids = torch.zeros((n_q, k), dtype=torch.int32, device=dev); d_sc = torch.zeros((n_q, k), dtype=torch.float64, device=dev); d_cnt = torch.zeros(n_q, dtype=torch.int32, device=dev)
    prof.zero_(); torch.cuda.synchronize()
    t0 = time.time()
    gpu.suggest_batch_device(d_q.data_ptr(), d_o.data_ptr(), n_q, metric, a, k, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), stream=0)
    torch.cuda.synchronize(); dt = time.time() - t0
    allp = prof.cpu().numpy().astype(np.float64).reshape(2, 4096, 8).sum(axis=1)
    p, cn = allp[0], allp[1] / n_q
    print("%s %.2f k=%d n_q=%d: %.3f s" % (metric, a, k, n_q, dt))
    print('  per query: groups %.1f sub-batches %.1f flagged %.1f queued %.1f passes %.1f emitted %.1f kept-verdicts %.0f verified %.0f' % tuple(cn))
    tot = p.sum()
    for n, v in zip(names, p):
        print("    %-28s %14.0f  %5.1f%%" % (n, v / n_q, 100 * v / max(tot, 1)))
slow = 'ccfhcge1 dbddg1hcaecahd  1 e   dca fgfh2hedecab1gabfhb'
fast = 'ca cgdcbch12af2gacefhhheeffgbgh b2db a cgdch 112cdchbfd'
for q in (slow, fast):
    print(repr(q), flush=True)
    run([q], "dice", 0.15, 65)
    run([q] * 5000, "dice", 0.15, 65)
    run([q] * 5000, "dice", 0.15, 10)
    run([q] * 5000, "dice", 0.5, 65)
