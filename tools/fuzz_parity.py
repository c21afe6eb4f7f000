#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity hunt (GPU box only): random descriptions, dictionaries with heavy repetition (ties,
repeated terms, overflowing groups), random metrics / k / tuning knobs / builder, suggest + autocomplete.

  python tools/fuzz_parity.py --seconds 300 [--seed 1]       hunt; prints the seed of any mismatch
  python tools/fuzz_parity.py --replay SEED [NAME=VALUE ...]  replay one trial verbosely (SG_* knobs, build=, k=, only=ROW)
tests/test_gpu_parity.py::test_fuzz_regressions replays the seeds that found bugs."""
import argparse, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

KNOBS = (("SG_LOG2_CNT", ["9", "10", "11", "12"]), ("SG_T_FLOOR", ["2", "4", "10", "30"]), ("SG_FILTER_LEVEL", ["0", "2", "3", "4", "6", "7"]),
         ("SG_SPLIT_CHUNKS", ["0", "1", "8", "65536"]), ("SG_TIGHTEN", ["0", "1", "1", "2"]), ("SG_ROOMY", ["0", "1", "2"]), ("SG_ORDER", ["0", "1", "2", "16"]))


def make_trial(seed, scale=1):
    """-> None (description not usable) or dict(desc, docs, queries, env, build, searches [(metric, a, k)], limit)"""
    import oracle
    rng = random.Random(seed)
    q = rng.choice([1, 2, 2, 3, 3, 3, 4, 5])
    alpha = rng.choice([("english",), ("english", "numbers"), ("ab", "$"), ("russian", "english", "numbers", "$"), ("abc", "-")])
    wrap = rng.choice([("$", "$"), ("^", "$"), ("", ""), (" ", " "), ("$$", "")])
    pad = rng.choice(["$", "_", ""]) if q <= 4 else "$"
    if q * max(1, len(pad)) > 8:
        return None
    syms = rng.choice(["ab", "abc -", "abcdefgh 12", "абвгд ёab", "AbC.dE f", "abcdefghijklmnopqrstuvwxyz"])
    n_docs = rng.choice([1, 5, 50, 400, 3000, 20000]) * scale
    max_len = rng.choice([6, 14, 30, 60])
    base = ["".join(rng.choice(syms) for _ in range(rng.randint(0, max_len))) for _ in range(max(1, n_docs // rng.choice([1, 1, 4, 20])))]
    docs = []
    for _ in range(n_docs):
        w = list(rng.choice(base))
        for _ in range(rng.randint(0, 2)):
            if w:
                w[rng.randrange(len(w))] = rng.choice(syms)
        docs.append("".join(w))
    desc = dict(ngram_size=q, wrap=wrap, pad=pad, alphabet=alpha)
    if len(oracle.OracleIndex([docs[0]], **desc).tokenize(docs[0])) == 0:
        docs[0] = "abcabcab"      # the reference panics if the FIRST document has no tokens (indexer_writer.go:70)
        if len(oracle.OracleIndex([docs[0]], **desc).tokenize(docs[0])) == 0:
            return None
    env = {name: rng.choice(choices) for name, choices in KNOBS}
    build = rng.choice(["host", "device"])
    queries = [rng.choice(docs) for _ in range(30)] + ["".join(rng.choice(syms) for _ in range(rng.randint(0, max_len + 6))) for _ in range(30)]
    queries += [d[:rng.randint(0, len(d))] + rng.choice(syms) + d[rng.randint(0, len(d)):] for d in rng.sample(docs, min(20, len(docs)))]
    searches = []
    for _ in range(3):
        metric = rng.choice(["jaccard", "cosine", "dice", "overlap", "exact"])
        a = 1.0 if metric == "exact" else rng.choice([0.15, 0.3, 0.5, 0.7, 0.9, 1.0])
        searches.append((metric, a, rng.choice([1, 2, 5, 10, 64, 65, 300])))
    limit = rng.choice([1, 7, 100])
    # (round 3, drawn last so that the trials of older seeds keep their content) long documents and queries — above the
    # wavefront kernel's 128 n-grams: sg_long_kernel — and k above the former 1024
    if rng.random() < 0.2:
        long_docs = ["".join(rng.choice(syms) for _ in range(rng.randint(130, 600))) for _ in range(rng.randint(1, 12))]
        docs += long_docs
        queries += [d[:rng.randint(100, len(d))] for d in long_docs] + [long_docs[0] + rng.choice(syms) * rng.randint(1, 300)]
        queries += [d[:60] + rng.choice(syms) + d[61:] for d in long_docs[:4]]
        if rng.random() < 0.5:
            searches.append((rng.choice(["jaccard", "cosine", "dice"]), rng.choice([0.2, 0.5, 0.8]), rng.choice([1500, 3000])))
    # (drawn after everything else, as above) the tokeniser as a launch of its own — sg_terms_kernel — for batches of >= n queries
    env["SG_PRETOK"] = rng.choice(["0", "1", "1", "2048"])
    # (round 4, drawn last again) 8-bit gaps for dense terms: 2 = every term (gaps above 255 all over a sparse list: chunks of one posting)
    env["SG_G8"] = rng.choice(["0", "1", "2", "2"])
    # (round 5, drawn last again) plan -> stream -> verify: forced / by policy / off, every stream workgroup, tiny candidate slots
    # (overflow -> the fused kernel), the class store (the document side of the prefix filter) in several geometries
    env["SG_PIPE"] = rng.choice(["0", "1", "1", "1", "2"])
    env["SG_PIPE_NW"] = rng.choice(["2", "4", "8"])
    env["SG_PIPE_LOG2_CNT"] = rng.choice(["9", "10", "11", "12", "13"])
    env["SG_PIPE_DT_BYTES"] = rng.choice(["1024", "2048", "4096", "8192"])
    env["SG_PIPE_SUB"] = rng.choice(["3", "4", "5"])
    env["SG_PIPE_CAND_CAP"] = rng.choice(["2", "16", "64", "64", "512"])
    env["SG_PIPE_WIDE"] = rng.choice(["0", "0", "1"])
    env["SG_PLAN2"] = rng.choice(["1", "1", "0"])
    # (round 6, drawn last again) no shape knobs: the stream workgroup is chosen per launch (capi.inc, "shape, per launch")
    env["FUZZ_SHAPE_AUTO"] = rng.choice(["0", "0", "1"])
    return dict(desc=desc, docs=docs, queries=queries, env=env, build=build, searches=searches, limit=limit, syms=syms)


def run_trial(t, verbose=False, only=None, k_override=None):
    """-> list of mismatch descriptions (empty = parity)"""
    import oracle
    from suggest_amd import IndexDescription, NGramIndex
    os.environ.update(t["env"])
    if t["env"].get("FUZZ_SHAPE_AUTO") == "1":
        for name in ("SG_PIPE_NW", "SG_PIPE_LOG2_CNT", "SG_PIPE_DT_BYTES"): os.environ.pop(name, None)
    tm = t.setdefault("timing", {"gpu": 0.0, "oracle": 0.0})
    t0 = time.time()
    try:
        gpu = NGramIndex(t["docs"], IndexDescription(**t["desc"]), build=t["build"])
    except Exception as exc:      # a description whose terms do not fit the 64-bit key is refused, not answered
        if "fit" in str(exc) or "-2" in str(exc):
            return []
        raise
    tm["gpu"] += time.time() - t0
    t0 = time.time()
    ora = oracle.OracleIndex(t["docs"], **t["desc"])
    tm["oracle"] += time.time() - t0
    queries = t["queries"] if only is None else [t["queries"][only]]
    qb, qo = oracle.pack_strings(queries)
    out = []
    for metric, a, k in t["searches"]:
        k = k_override or k
        t0 = time.time()
        ids, sc, cnt = gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=a, k=k)
        t1 = time.time()
        oi, os_, oc, _ = ora.suggest_batch(qb, qo, metric, a, k)
        tm["gpu"] += t1 - t0; tm["oracle"] += time.time() - t1
        valid = (np.arange(k)[None, :] < np.minimum(oc, k)[:, None]) & (oc < 0xFFFFFFF0)[:, None]
        rows = np.nonzero((cnt != oc) | (valid & ((ids != oi) | (sc.view(np.uint64) != os_.view(np.uint64)))).any(axis=1))[0]
        if verbose:
            print(metric, a, k, "differing rows:", rows[:10])
        if rows.size:
            r = int(rows[0])
            g = list(zip(ids[r, :int(min(cnt[r], k))].tolist(), sc[r, :int(min(cnt[r], k))].tolist()))
            o = list(zip(oi[r, :int(min(oc[r], k))].tolist(), os_[r, :int(min(oc[r], k))].tolist()))
            first = next((i for i in range(min(len(g), len(o))) if g[i] != o[i]), min(len(g), len(o)))
            out.append("%s a=%.2f k=%d rows=%s query=%r (%d tokens) counts gpu/oracle %d/%d, first difference at rank %d: gpu %s oracle %s"
                       % (metric, a, k, rows[:5].tolist(), queries[r], len(ora.tokenize(queries[r])), int(cnt[r]), int(oc[r]), first,
                          g[first:first + 2], o[first:first + 2]))
    t0 = time.time()
    ids, cnt = gpu.autocomplete_batch(blob=qb, offs=qo, limit=t["limit"])
    t1 = time.time()
    oi, oc, _ = ora.autocomplete_batch(qb, qo, t["limit"])
    tm["gpu"] += t1 - t0; tm["oracle"] += time.time() - t1
    valid = np.arange(t["limit"])[None, :] < np.minimum(oc, t["limit"])[:, None]
    if not (np.array_equal(cnt, oc) and np.array_equal(ids[valid], oi[valid])):
        out.append("autocomplete limit=%d" % t["limit"])
    try:
        t["pipe_queries"] = gpu.pipe_stats()["queries"]
    except Exception:
        t["pipe_queries"] = 0
    gpu.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--replay", type=int, default=None)
    ap.add_argument("--scale", type=int, default=1, help="multiplies the dictionary sizes (1 .. 20 000 documents)")
    ap.add_argument("--pipe", action="store_true", help="[r5] only trials the plan -> stream -> verify launches are eligible for: dictionaries above 2 048 "
                    "documents, k <= 64, the tokeniser launch on, no tightening / 8-bit gaps / split queries, SG_PIPE=1 (every other knob as drawn)")
    ap.add_argument("--force", action="append", default=[], help="NAME=VALUE forced onto every trial's knobs (bisecting a mismatch that depends on what ran before it)")
    ap.add_argument("--first", type=int, default=0, help="the first trial number of the hunt (with --seed: resume a hunt near a mismatch)")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args()
    import torch  # noqa: F401
    if args.replay is not None:
        over = dict(x.split("=") for x in args.overrides)
        t = make_trial(args.replay, args.scale)
        t["env"].update({k: v for k, v in over.items() if k.startswith("SG_")})
        t["build"] = over.get("build", t["build"])
        print(t["desc"], len(t["docs"]), "docs", t["env"], "build", t["build"])
        for m in run_trial(t, verbose=True, only=int(over["only"]) if "only" in over else None, k_override=int(over["k"]) if "k" in over else None):
            print("MISMATCH", m)
        return
    t_end, trial, bad, piped, piped_trials = time.time() + args.seconds, args.first, 0, 0, 0
    while time.time() < t_end:
        seed = args.seed * 100000 + trial
        trial += 1
        t0 = time.time()
        t = make_trial(seed, args.scale)
        if t is None:
            continue
        if args.pipe:
            if len(t["docs"]) < 3000:
                continue
            t["env"].update(SG_PIPE="1", SG_PRETOK="1", SG_TIGHTEN="0", SG_G8="0", SG_SPLIT_CHUNKS="0", SG_LOG2_CNT="9")
            t["searches"] = [(m_, a_, min(k_, 64)) for m_, a_, k_ in t["searches"]]
        t["env"].update(dict(x.split("=") for x in args.force))
        for m in run_trial(t):
            bad += 1
            print("MISMATCH seed %d: %s env=%s build=%s: %s" % (seed, t["desc"], t["env"], t["build"], m), flush=True)
        piped += t.get("pipe_queries", 0); piped_trials += 1 if t.get("pipe_queries", 0) else 0
        tm = t.get("timing", {"gpu": 0.0, "oracle": 0.0})
        if time.time() - t0 > 5 or tm["gpu"] > 1.0:
            print("slow trial: seed %d took %.1f s (device side %.2f s, oracle %.2f s): %s, %d docs, syms %r, searches %s, env %s"
                  % (seed, time.time() - t0, tm["gpu"], tm["oracle"], t["desc"], len(t["docs"]), t["syms"], t["searches"], t["env"]), flush=True)
    print("fuzz: %d trials, %d mismatches; %d trials (%d queries) went through plan -> stream -> verify" % (trial, bad, piped_trials, piped))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
