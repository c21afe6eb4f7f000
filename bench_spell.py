"""BASELINE config 5: the spellchecker caller — SpellChecker.Predict over a 50M-token language model
(python bench.py --config cfg5 [--gpus N --steps K --warmup W]).

A "step" is one sg_spell_predict_batch_device call over one batch of 65,536 queries per GPU, queries and result rows resident
in HBM: six launches on one stream (word tokeniser + word ids, NGramModel.Next, LM-ranked autocomplete, selection, Cosine
fuzzy top-up, merge + stable re-rank).  The host-buffer entry point (sg_spell_predict_batch, PCIe-inclusive) is timed beside
it and reported as `host_buffers`, never as `value`.  The model is synthetic
(tools/make_synthetic_lm.py: Zipf words, 1M-word vocabulary, ~50M tokens incl. sentence markers), written in the
reference's production formats (<name>.lm + <name>.cdb) and loaded through RetrieveLMFromBinary's twin; per GPU: a
replica of the vocabulary's fuzzy index and of the LM arrays (weak scaling, no collective).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def main(args):
    out = run(args)
    if out is not None:
        print(json.dumps(out), flush=True)


def run(args, env=None):
    """-> the record (rank 0) or None.  `env`: bench.py's Env when called for a sub-record of the default run (N = 1: no
    process group is made or destroyed here)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import make_synthetic_lm
    from suggest_amd.spell import LanguageModel, SpellChecker
    from suggest_amd.index import pack_strings

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    if os.environ.get("SG_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("SG_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    own_group = world > 1 and env is None
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")      # control plane only (barrier, max over the ranks' clocks): no data-path collective

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    tokens = int(os.environ.get("SG_LM_TOKENS", args.dict_size or 50_000_000))
    vocab = int(os.environ.get("SG_LM_VOCAB", 1_000_000 if tokens >= 20_000_000 else max(1000, tokens // 40)))
    n_q = args.queries or 65536
    top_k = args.topk or 5
    sim = args.similarity or 0.5
    n_b = max(1, args.batches)
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "sg_lm_%d_%d_r%d" % (tokens, vocab, local_rank))
    t0 = time.time()
    info = make_synthetic_lm.make(d, tokens=tokens, vocab=vocab, verbose=rank == 0)
    log("language model written in %.1fs" % (time.time() - t0))
    t0 = time.time()
    lm = LanguageModel(binary=os.path.join(d, "synth.lm"), dictionary=os.path.join(d, "synth.cdb"))
    sc = SpellChecker(lm, device=local_rank)
    st = sc.index.stats()
    log("model loaded + vocabulary index built/uploaded in %.1fs: %d words, %s" % (time.time() - t0, len(lm), st))

    # queries: two context words of a corpus position + the next word cut to a prefix (2 of 3) or with a typo (1 of 3)
    seeds = iter(range(100 + 1000 * rank, 100 + 1000 * rank + 1000))

    def make_batch():
        return pack_strings(make_synthetic_lm.make_queries(info, n_q, next(seeds)))

    batches = [make_batch() for _ in range(n_b)]
    log("%d batches of %d queries" % (n_b, n_q))
    row = top_k + 1
    d_q = [torch.from_numpy(qb).to(dev) for qb, _ in batches]
    d_o = [torch.from_numpy(qo.view(np.int64)).to(dev) for _, qo in batches]
    d_ids = [torch.zeros((n_q, row), dtype=torch.int32, device=dev) for _ in range(n_b)]
    d_cnt = [torch.zeros(n_q, dtype=torch.int32, device=dev) for _ in range(n_b)]
    stream = torch.cuda.current_stream(dev)

    def step(b):       # device-resident: queries and result rows in HBM, the word tokeniser on the device too
        sc.predict_batch_device(d_q[b].data_ptr(), d_o[b].data_ptr(), n_q, int(batches[b][1][-1]), top_k, sim, d_ids[b].data_ptr(), d_cnt[b].data_ptr(),
                                stream=stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # (setup, untimed: calls until the clocks have settled — bench.py's measure() has the measurements behind it)
    t_settle, n_settle = time.perf_counter(), 0
    while n_settle < 64 and time.perf_counter() - t_settle < 0.25:
        step(n_settle % n_b)
        n_settle += 1
        if n_settle % 8 == 0:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i % n_b)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t_start = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        step(i % n_b)
        ev[i][1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t_start
    gpu_ms = float(np.mean([x.elapsed_time(y) for x, y in ev]))
    t = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    for b in range(n_b):
        step(b)
    torch.cuda.synchronize(dev)
    res = (d_ids[0].cpu().numpy().view(np.uint32), d_cnt[0].cpu().numpy().view(np.uint32))
    # the host-buffer entry point (what a cgo caller uses): PCIe-inclusive, never `value`; same rows
    host_rate = None
    if rank == 0 and world == 1:
        h_ids, h_cnt = sc.predict_batch(blob=batches[0][0], offs=batches[0][1], top_k=top_k, similarity=sim)
        t0 = time.perf_counter()
        for _ in range(3):
            sc.predict_batch(blob=batches[0][0], offs=batches[0][1], top_k=top_k, similarity=sim)
        host_rate = 3 * n_q / (time.perf_counter() - t0)
        if not (np.array_equal(h_cnt, res[1]) and np.array_equal(h_ids, res[0])):
            raise SystemExit("sg_spell_predict_batch (host buffers) and sg_spell_predict_batch_device disagree")

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        t0 = time.time()
        olm = oracle.OracleLM(binary=os.path.join(d, "synth.lm"), dictionary=os.path.join(d, "synth.cdb"))
        oix = oracle.OracleIndex(info["word_list"], ngram_size=sc.description.ngram_size, wrap=sc.description.wrap, pad=sc.description.pad,
                                 alphabet=sc.description.alphabet)
        log("oracle model + index in %.1fs" % (time.time() - t0))
        import bench
        quota, hw = bench.cpu_quota()
        granted = max(1, min(hw, int(round(quota)))) if quota else hw       # the cores this process can actually use
        qb, qo = batches[0]
        n_s = args.cpu_sample or min(n_q, 8192)

        def timed(n, threads):
            t0 = time.perf_counter()
            r_ = olm.predict_batch(oix, qb[:int(qo[n])], qo[:n + 1], top_k, sim, threads=threads)
            return r_, time.perf_counter() - t0

        (oi, oc), dt = timed(n_s, granted)
        legs = {"granted_cores": {"value": n_s / dt, "unit": "predictions/s", "cores": granted, "sample": "first %d queries of batch 0" % n_s}}
        if hw != granted:
            _, dt_h = timed(n_s, hw)
            legs["all_hw_threads"] = {"value": n_s / dt_h, "unit": "predictions/s", "cores": hw, "sample": "first %d queries of batch 0" % n_s,
                                      "note": "threads = every hardware thread of the host; the container's CPU quota is %s cores" % (quota,)}
        n_1 = max(64, n_s // 32)
        _, dt1 = timed(n_1, 1)
        legs["one_thread"] = {"value": n_1 / dt1, "unit": "predictions/s", "cores": 1, "sample": "first %d queries" % n_1}
        best = max(("granted_cores", "all_hw_threads"), key=lambda n_: legs.get(n_, {"value": -1.0})["value"])
        cores = legs[best]["cores"]
        cpu = {"value": legs[best]["value"], "unit": "predictions/s", "cores": cores, "cores_granted": granted, "cpu_quota_cores": quota, "hw_threads": hw,
               "best_leg": best, "kind": "port",
               "sample": "%s, same model; C++ restatement of pkg/spellchecker + pkg/lm (oracle/), OpenMP across queries" % legs[best]["sample"]}
        cpu.update(legs)
        gi, gc = res
        valid = np.arange(top_k + 1)[None, :] < np.minimum(oc, top_k + 1)[:, None]
        same = bool(np.array_equal(gc[:n_s], oc) and np.array_equal(gi[:n_s][valid], oi[valid]))
        parity = {"checked_queries": int(n_s), "bit_exact": same}
        log("cpu %.0f predictions/s on %d threads, %.0f on one; GPU == oracle on the sample: %s" % (cpu["value"], cores, cpu["one_thread"]["value"], same))

    out = None
    if rank == 0:
        # algorithmic bytes of the two searches (SURVEY.md 8d accounting): the autocomplete over all last words, the fuzzy
        # search over the queries that needed it (their count is not known on the host: upper bound = all, lower = none)
        qb, qo = batches[0]
        last = [bytes(qb[int(qo[i]):int(qo[i + 1])]).split(b" ")[-1] for i in range(min(n_q, 8192))]
        lb, lo = pack_strings(last)
        import ctypes as C
        from suggest_amd import _lib
        tot = C.c_uint64()
        _lib.check(_lib.lib().sg_autocomplete_algorithmic_bytes(sc.index._h, lb.ctypes.data, lo.ctypes.data, len(last), top_k, C.byref(tot)))
        alg_auto = tot.value / len(last) * n_q
        ms = elapsed / args.steps * 1e3
        kernels = None
        under_profiler = any(kk.startswith(("ROCPROF", "ROCP_")) for kk in os.environ)
        if world == 1 and args.traffic != "none" and not under_profiler:
            kernels = _live_kernels(args, tokens, n_q, top_k, sim, n_b, log)
        lm_k = (kernels or {}).get("lm_autocomplete")
        roof = {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "kernel": "sg_search_kernel_t<false, true, ...> (LM-ranked autocomplete: the launch with the most bytes)",
                "achieved": lm_k["gbps"] if lm_k else None, "frac": lm_k["gbps"] / 8000.0 if lm_k else None,
                "traffic": lm_k["traffic"] if lm_k else None,
                "traffic_source": lm_k["source"] if lm_k else "MISSING: no live PMC pass (N > 1, --traffic none, rocprofv3 absent or under a profiler)",
                "effective_gbps": alg_auto / (lm_k["ms"] * 1e-3) / 1e9 if lm_k else None,
                "effective_frac": alg_auto / (lm_k["ms"] * 1e-3) / 1e9 / 8000.0 if lm_k else None,
                "algorithmic_bytes_per_launch": alg_auto, "step_gpu_ms_avg": gpu_ms, "kernels": kernels,
                "note": "per kernel of one Predict step (rocprofv3 child pass of this run): ms = average dispatch duration, traffic = FETCH_SIZE x 1024 x 2 "
                        "(gfx950 correction), gbps = traffic / ms; achieved / frac = the LM-ranked autocomplete launch; effective_* = its algorithmic bytes "
                        "(sg_autocomplete_algorithmic_bytes, extrapolated from 8192 queries) over the same duration"}
        out = {
            "metric": "spellchecker predictions/sec (topK=%d, Cosine>=%.2g top-up) on a %dM-token LM, %dk-word vocabulary" % (top_k, sim, round(info["tokens"] / 1e6), len(lm) // 1000),
            "value": world * n_q * args.steps / elapsed, "unit": "predictions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_settle_calls": n_settle,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (posting / counter / count-search work)", "data": "synthetic",
            "config": {"workload": "SpellChecker.Predict: %d-token synthetic corpus -> 3-gram LM (%d words, %d bigrams, %d trigrams) in the reference's .lm/.cdb formats; "
                                   "%d queries per GPU per step ('w1 w2 prefix', one third with a typo), %d batches in rotation; queries and result rows resident in HBM "
                                   "(sg_spell_predict_batch_device: word tokeniser, word ids, Next, both searches, merge on the device)"
                                   % (info["tokens"], len(lm), info["bigrams"], info["trigrams"], n_q, n_b),
                       "baseline_config": "cfg5", "parallelism": "query-sharded x%d, vocabulary index + LM replica per GPU" % world,
                       "index": {"postings": st["n_postings"], "terms": st["n_terms"], "device_bytes": st["device_bytes"]},
                       "predictions_per_query": float(np.minimum(res[1], top_k + 1).mean())},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if host_rate:
            out["host_buffers"] = {"value": host_rate, "unit": "predictions/s",
                                   "note": "sg_spell_predict_batch: pageable host buffers in and out over PCIe, synchronous (never `value`)"}
        if parity:
            out["parity_vs_oracle"] = parity
    del sc, lm
    if own_group:
        dist.destroy_process_group()
    return out


def _live_kernels(args, tokens, n_q, top_k, sim, n_b, log):
    """Per-kernel duration and HBM traffic of a Predict step, measured now: this benchmark again, as a child under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace`.  -> {name: {ms, traffic, gbps, launches, source}} or None"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rocprof:
        return None
    tmp = tempfile.mkdtemp(prefix="sg_pmc5_", dir="/tmp")
    cmd = [rocprof, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--config", "cfg5", "--dict-size", str(tokens), "--queries", str(n_q), "--topk", str(top_k),
           "--similarity", repr(sim), "--batches", str(n_b), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--traffic", "none"]
    names = {"spell_tokenize_kernel": "tokenize", "spell_next_kernel": "lm_next", "sg_search_kernel_t<false, true,": "lm_autocomplete",
             "spell_select_kernel": "select", "sg_search_kernel_t<false, false,": "fuzzy_top_up", "spell_merge_kernel": "merge",
             "query_order_": "query_order"}
    t0 = time.time()
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            log("live PMC pass failed (%d): %s" % (r.returncode, (r.stderr or "")[-300:]))
            return None
        acc = {}
        for path in sorted(glob.glob(tmp + "/**/*counter_collection.csv", recursive=True)):
            for rowd in csv.DictReader(open(path)):
                if rowd.get("Counter_Name") != "FETCH_SIZE":
                    continue
                for pat, nm in names.items():
                    if pat in rowd["Kernel_Name"]:
                        a = acc.setdefault(nm, [0, 0.0, 0.0])
                        a[0] += 1
                        a[1] += (float(rowd["End_Timestamp"]) - float(rowd["Start_Timestamp"])) * 1e-6
                        a[2] += float(rowd["Counter_Value"]) * 1024 * 2
                        break
        out = {}
        steps = max(1, acc.get("merge", [4])[0])
        for nm, (n, ms, by) in acc.items():
            per_ms, per_b = ms / steps, by / steps
            out[nm] = {"ms": per_ms, "traffic": per_b, "gbps": per_b / (per_ms * 1e-3) / 1e9 if per_ms else None, "launches_per_step": n / steps,
                       "source": "live: rocprofv3 --pmc FETCH_SIZE --kernel-trace child pass, %d steps; per step" % steps}
        log("live per-kernel pass (%.0fs): %s" % (time.time() - t0, {k: (round(v["ms"], 3), round(v["traffic"] / 1e6, 1)) for k, v in out.items()}))
        return out
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as exc:
        log("live PMC pass failed: %r" % (exc,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
