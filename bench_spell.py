"""BASELINE config 5: the spellchecker caller — SpellChecker.Predict over a 50M-token language model
(python bench.py --config cfg5 [--gpus N --steps K --warmup W]).

A "step" is one sg_spell_predict_batch call over one batch of 65,536 queries per GPU: host word tokeniser + word ids,
then five launches on one stream (NGramModel.Next, LM-ranked autocomplete, selection, Cosine fuzzy top-up, merge +
stable re-rank).  The boundary of this path hands host buffers over (there is no device-resident entry point for
Predict), so `value` is PCIe- and host-tokeniser-inclusive — said in the JSON.  The model is synthetic
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev) if backend == "nccl" else dist.init_process_group(backend)

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
    T, words = info["corpus_sample"], info["word_list"]
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    markers = (info["start_id"], info["end_id"])

    def make_batch():
        out = []
        while len(out) < n_q:
            p = int(rng.integers(2, len(T)))
            a, b, c = int(T[p - 2]), int(T[p - 1]), int(T[p])
            if a in markers or b in markers or c in markers:
                continue
            w = words[c]
            if len(out) % 3 == 2 and len(w) > 3:
                j = int(rng.integers(1, len(w)))
                w = w[:j] + bytes([ord("a") + int(rng.integers(0, 26))]) + w[j + 1:]
            else:
                w = w[:max(2, (len(w) * 2 + 2) // 3)]
            out.append(words[a] + b" " + words[b] + b" " + w)
        return pack_strings(out)

    batches = [make_batch() for _ in range(n_b)]
    log("%d batches of %d queries" % (n_b, n_q))

    def step(b):
        return sc.predict_batch(blob=batches[b][0], offs=batches[b][1], top_k=top_k, similarity=sim)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i % n_b)
    barrier()
    t_start = time.perf_counter()
    for i in range(args.steps):
        res = step(i % n_b)
    barrier()
    elapsed = time.perf_counter() - t_start
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        t0 = time.time()
        olm = oracle.OracleLM(binary=os.path.join(d, "synth.lm"), dictionary=os.path.join(d, "synth.cdb"))
        oix = oracle.OracleIndex(words, ngram_size=sc.description.ngram_size, wrap=sc.description.wrap, pad=sc.description.pad,
                                 alphabet=sc.description.alphabet)
        log("oracle model + index in %.1fs" % (time.time() - t0))
        cores = os.cpu_count() or 1
        qb, qo = batches[0]
        n_s = args.cpu_sample or min(n_q, 8192)
        t0 = time.perf_counter()
        oi, oc = olm.predict_batch(oix, qb[:int(qo[n_s])], qo[:n_s + 1], top_k, sim, threads=cores)
        dt = time.perf_counter() - t0
        n_1 = max(64, n_s // 32)
        t0 = time.perf_counter()
        olm.predict_batch(oix, qb[:int(qo[n_1])], qo[:n_1 + 1], top_k, sim, threads=1)
        dt1 = time.perf_counter() - t0
        cpu = {"value": n_s / dt, "unit": "predictions/s", "cores": cores, "kind": "port",
               "sample": "first %d queries of batch 0, same model; C++ restatement of pkg/spellchecker + pkg/lm (oracle/), OpenMP across queries" % n_s,
               "one_thread": {"value": n_1 / dt1, "unit": "predictions/s", "cores": 1, "sample": "first %d queries" % n_1}}
        gi, gc = step(0)
        valid = np.arange(top_k + 1)[None, :] < np.minimum(oc, top_k + 1)[:, None]
        same = bool(np.array_equal(gc[:n_s], oc) and np.array_equal(gi[:n_s][valid], oi[valid]))
        parity = {"checked_queries": int(n_s), "bit_exact": same}
        log("cpu %.0f predictions/s on %d threads, %.0f on one; GPU == oracle on the sample: %s" % (cpu["value"], cores, cpu["one_thread"]["value"], same))

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
        out = {
            "metric": "spellchecker predictions/sec (topK=%d, Cosine>=%.2g top-up) on a %dM-token LM, %dk-word vocabulary" % (top_k, sim, round(info["tokens"] / 1e6), len(lm) // 1000),
            "value": world * n_q * args.steps / elapsed, "unit": "predictions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (posting / counter / count-search work)", "data": "synthetic",
            "config": {"workload": "SpellChecker.Predict: %d-token synthetic corpus -> 3-gram LM (%d words, %d bigrams, %d trigrams) in the reference's .lm/.cdb formats; "
                                   "%d queries per GPU per step ('w1 w2 prefix', one third with a typo), %d batches in rotation; host buffers in and out (PCIe + host tokeniser inside the timed region)"
                                   % (info["tokens"], len(lm), info["bigrams"], info["trigrams"], n_q, n_b),
                       "baseline_config": "cfg5", "parallelism": "query-sharded x%d, vocabulary index + LM replica per GPU" % world,
                       "index": {"postings": st["n_postings"], "terms": st["n_terms"], "device_bytes": st["device_bytes"]},
                       "predictions_per_query": float(np.minimum(res[1], top_k + 1).mean())},
            "roofline": {"bound": "hbm", "achieved": alg_auto / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg_auto / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                         "note": "algorithmic bytes of the LM-ranked autocomplete launch only (sg_autocomplete_algorithmic_bytes, extrapolated from 8192 queries) over the "
                                 "WHOLE step time incl. host tokeniser, PCIe and the four other launches: a lower bound of that kernel's rate; per-kernel times in profiles/r02_cfg5_*",
                         "algorithmic_bytes_per_launch": alg_auto},
            "cpu_baseline": cpu,
        }
        if parity:
            out["parity_vs_oracle"] = parity
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
