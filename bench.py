#!/usr/bin/env python3
"""bench.py — fuzzy queries/sec of the MI355X engine on BASELINE.json's workloads.

  python bench.py --gpus N --steps K --warmup W [--config headline|cfg2|cfg3|cfg4|cfg5]
  (N=1 directly; N>1 under torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path (tokenise -> posting lookup -> T-occurrence count -> score -> top-k)
over one batch of synthetic queries per GPU, inputs already resident in HBM.  The default workload is the
one BASELINE.json's `metric` is quoted on: 10M synthetic strings (len 8-32 over [a-z0-9]), q=3,
Jaccard>=0.5, k=10, 65,536 edited queries per GPU and step.  --config selects the other BASELINE.json
configs (cfg2: 1M strings; cfg3: Cosine>=0.4 k=20; cfg4: q=2 Dice>=0.5; cfg5: the spellchecker caller).
Steps rotate over --batches (default 4) distinct query batches, all resident before the timed region.
Weak scaling: every rank holds a full index replica and its own batches; the only collective is the
optional gather of the k*(u32,f64) result rows over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline      achieved = ALGORITHMIC bytes per launch (sg_suggest_algorithmic_bytes) / the search
                kernel's average launch duration (HIP events on the launch stream); traffic = HBM bytes per launch
                from a rocprofv3 --pmc FETCH_SIZE pass over this workload — live in a child process (--traffic), else the
                committed run in profiles/traffic.json — with
                wire_gbps / wire_frac = that traffic over the same duration
  cpu_baseline  the CPU oracle — a C++ restatement of the Go path, kind "port" — on this host: all hardware
                threads, and one thread (`one_thread`), each on a bounded sample of the same batch
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)

CONFIGS = {   # BASELINE.json `configs` (cfg1 is the CPU-only plumbing case: tests/test_oracle_golden.py)
    "headline": dict(dict_size=10_000_000, queries=65536, ngram=3, metric="jaccard", similarity=0.5, topk=10),
    "cfg2": dict(dict_size=1_000_000, queries=65536, ngram=3, metric="jaccard", similarity=0.5, topk=10),
    "cfg3": dict(dict_size=10_000_000, queries=65536, ngram=3, metric="cosine", similarity=0.4, topk=20),
    "cfg4": dict(dict_size=10_000_000, queries=16384, ngram=2, metric="dice", similarity=0.5, topk=10),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS) + ["cfg5"],
                    help="BASELINE.json config (default: the one the metric is quoted on); explicit flags below override it")
    ap.add_argument("--dict-size", type=int, default=None)
    ap.add_argument("--queries", type=int, default=None, help="queries per GPU per step")
    ap.add_argument("--ngram", type=int, default=None)
    ap.add_argument("--metric", default=None)
    ap.add_argument("--similarity", type=float, default=None)
    ap.add_argument("--topk", type=int, default=None)
    ap.add_argument("--batches", type=int, default=4, help="distinct query batches the steps rotate over")
    ap.add_argument("--dict-variant", default="uniform", choices=["uniform", "skewed", "families", "skewed-families"],
                    help="SURVEY.md §8d dictionary variants (headline = uniform); families = base + 3 edited copies")
    ap.add_argument("--build", default="device", choices=["device", "host"],
                    help="index build: on the GPU (sg_index_build_device, 0.4 s at 10M; the posting store stays where it was "
                         "built) or on the host (sg_index_build, then uploaded); same arrays either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries timed on the CPU oracle (0 = auto)")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: include the optional RCCL all_gather of the k*(u32,f64) result rows in every timed step "
                         "(default: results stay sharded — the path has no data-path collective; one untimed gather validates RCCL)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a rocprofv3 --pmc run, reported as roofline.traffic (default: the figure "
                         "recorded in profiles/traffic.json for this exact workload, measured with tools/pmc_run.sh)")
    ap.add_argument("--require-traffic", action="store_true", help="exit non-zero when no traffic figure could be had for the workload")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "file", "none"],
                    help="roofline.traffic: live = a rocprofv3 --pmc FETCH_SIZE pass over this same workload in a child process "
                         "(after the timed region; N=1 only); file = the committed measurement in profiles/traffic.json; "
                         "auto = live when rocprofv3 is there, else file")
    args = ap.parse_args()
    if args.config == "cfg5":
        import bench_spell
        return bench_spell.main(args)
    preset = CONFIGS[args.config]
    for key, val in preset.items():
        if getattr(args, key) is None:
            setattr(args, key, val)

    import numpy as np
    import torch
    import torch.distributed as dist
    from suggest_amd import IndexDescription, NGramIndex, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    # test hooks (1-GPU box): SG_BENCH_SINGLE_DEVICE=1 puts every rank on GPU 0, SG_BENCH_BACKEND=gloo avoids RCCL's
    # one-rank-per-GPU rule — exercises the N>1 control flow (barriers, max-over-ranks, per-rank batches), not the links
    if os.environ.get("SG_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("SG_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    # ---- workload -------------------------------------------------------------------------
    desc_kw = dict(synth.DESCRIPTION, ngram_size=args.ngram)
    k, n_q, n_b = args.topk, args.queries, max(1, args.batches)
    t0 = time.time()
    blob, offs = synth.make_dict(args.dict_size, seed=1, skewed="skewed" in args.dict_variant,
                                 families=3 if "families" in args.dict_variant else 0)
    # batch b of rank r = queries [(r * n_b + b) * n_q, ...) of one deterministic stream (seed 2)
    batches = [synth.make_queries(n_q, blob, offs, seed=2, start=(rank * n_b + b) * n_q) for b in range(n_b)]
    log("dict %d strings + %d batches of %d queries generated in %.1fs" % (args.dict_size, n_b, n_q, time.time() - t0))
    t0 = time.time()
    index = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**desc_kw), device=local_rank, build=args.build)
    st = index.stats()
    log("index built (%s) + uploaded in %.1fs: %s" % (args.build, time.time() - t0, st))
    alg = [index.algorithmic_bytes(qb, qo, args.metric, args.similarity, k) for qb, qo in batches]

    d_q = [torch.from_numpy(qb).to(dev) if qb.size else torch.zeros(1, dtype=torch.uint8, device=dev) for qb, _ in batches]
    d_offs = [torch.from_numpy(qo.view(np.int64)).to(dev) for _, qo in batches]
    d_ids = [torch.zeros((n_q, k), dtype=torch.int32, device=dev) for _ in range(n_b)]
    d_sc = [torch.zeros((n_q, k), dtype=torch.float64, device=dev) for _ in range(n_b)]
    d_cnt = [torch.zeros(n_q, dtype=torch.int32, device=dev) for _ in range(n_b)]
    if world > 1:
        g_ids = torch.zeros((world * n_q, k), dtype=torch.int32, device=dev)
        g_sc = torch.zeros((world * n_q, k), dtype=torch.float64, device=dev)
        g_cnt = torch.zeros(world * n_q, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step(b):
        index.suggest_batch_device(d_q[b].data_ptr(), d_offs[b].data_ptr(), n_q, args.metric, args.similarity, k,
                                   d_ids[b].data_ptr(), d_sc[b].data_ptr(), d_cnt[b].data_ptr(), stream=stream.cuda_stream)

    def gather(b, force=False):
        if world > 1 and (args.gather or force):   # top-k gather over RCCL/xGMI: k*(u32,f64) per query
            dist.all_gather_into_tensor(g_ids, d_ids[b])
            dist.all_gather_into_tensor(g_sc, d_sc[b])
            dist.all_gather_into_tensor(g_cnt, d_cnt[b])

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i % n_b)
        gather(i % n_b)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t_start = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        step(i % n_b)
        ev[i][1].record(stream)
        gather(i % n_b)
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    alg_timed = float(np.mean([alg[i % n_b] for i in range(args.steps)]))       # algorithmic bytes per launch, timed launches
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    gather_ok = None
    if world > 1:                       # untimed functional check of the optional result gather over RCCL
        try:
            gather(0, force=True)
            torch.cuda.synchronize(dev)
            mine = slice(rank * n_q, (rank + 1) * n_q)
            gather_ok = bool(torch.equal(g_ids[mine], d_ids[0]) and torch.equal(g_cnt[mine], d_cnt[0]))
        except Exception as exc:        # the timed region has no collective: report, do not lose the measurement
            log("result gather over RCCL failed: %r" % (exc,))
            gather_ok = False

    for b in range(min(n_b, args.steps + args.warmup), n_b):      # (batches the run never reached)
        step(b)
    torch.cuda.synchronize(dev)
    ids = [x.cpu().numpy().view(np.uint32) for x in d_ids]
    sc = [x.cpu().numpy() for x in d_sc]
    cnt = [x.cpu().numpy().view(np.uint32) for x in d_cnt]

    # ---- the host-buffer entry point (what a cgo caller uses): PCIe-inclusive, never the headline value ----
    host_rate = None
    if rank == 0 and world == 1:
        qb, qo = batches[0]
        index.suggest_batch(blob=qb, offs=qo, metric=args.metric, similarity=args.similarity, k=k)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            h_ids, h_sc, h_cnt = index.suggest_batch(blob=qb, offs=qo, metric=args.metric, similarity=args.similarity, k=k)
        host_rate = reps * n_q / (time.perf_counter() - t0)
        if not (np.array_equal(h_cnt, cnt[0]) and np.array_equal(h_ids, ids[0])):
            raise SystemExit("sg_suggest_batch (host buffers) and sg_suggest_batch_device disagree")

    # ---- CPU baseline: the oracle (restatement of the Go path) on this host, rank 0, N=1 only ----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        t0 = time.time()
        ora = oracle.OracleIndex(blob=blob, offs=offs, **desc_kw)
        log("oracle index built in %.1fs" % (time.time() - t0))
        cores = os.cpu_count() or 1
        qb, qo = batches[0]

        def timed(n, threads):
            t0 = time.perf_counter()
            res = ora.suggest_batch(qb[:int(qo[n])], qo[:n + 1], args.metric, args.similarity, k, threads=threads)
            return res, time.perf_counter() - t0

        n_s = min(args.cpu_sample or n_q, n_q)
        if not args.cpu_sample:            # calibrate so that the timed sample is ~10-15 s of wall time
            probe = min(n_q, 2048)
            _, dt = timed(probe, cores)
            n_s = int(min(n_q, max(probe, probe / max(dt, 1e-6) * 12)))
        (oi, os_, oc, used), dt = timed(n_s, cores)
        note = "C++ restatement of the Go path (oracle/), OpenMP across queries; the Go reference is not runnable here (no toolchain)"
        quota = None
        try:      # (a container's CPU quota: the threads above share this many cores)
            q_us, period = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q_us == "max" else float(q_us) / float(period)
        except (OSError, ValueError):
            pass
        cpu = {"value": n_s / dt, "unit": "queries/s", "cores": used, "cpu_quota_cores": quota, "kind": "port",
               "sample": "first %d queries of batch 0, same %d-string dictionary; %s" % (n_s, args.dict_size, note)}
        n_1 = int(max(16, min(n_s, cpu["value"] / max(used, 1) * 6)))      # ~6 s on one thread
        (_, _, _, used1), dt1 = timed(n_1, 1)
        cpu["one_thread"] = {"value": n_1 / dt1, "unit": "queries/s", "cores": used1, "sample": "first %d queries of batch 0" % n_1}
        valid = np.arange(k)[None, :] < np.minimum(oc, k)[:, None]
        same = np.array_equal(cnt[0][:n_s], oc) and np.array_equal(ids[0][:n_s][valid], oi[valid]) and \
            np.array_equal(sc[0][:n_s].view(np.uint64)[valid], os_.view(np.uint64)[valid])
        parity = {"checked_queries": int(n_s), "bit_exact": bool(same)}
        log("cpu baseline %.0f q/s on %d threads, %.0f q/s on one; GPU result bit-exact vs oracle on the sample: %s"
            % (cpu["value"], used, cpu["one_thread"]["value"], same))

    key = "%d/%d/q%d/%s/%.3g/k%d/%s" % (args.dict_size, n_q, args.ngram, args.metric, args.similarity, k, args.dict_variant)
    traffic, traffic_src = args.traffic_bytes, ("--traffic-bytes" if args.traffic_bytes else None)
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)     # (no profiler inside a profiler)
    if traffic is None and args.traffic in ("auto", "live") and rank == 0 and world == 1 and not (under_profiler and args.traffic == "auto"):
        traffic, traffic_src = _live_traffic(args, log)
        if traffic is None and args.traffic == "live":
            raise SystemExit("live PMC pass failed: " + str(traffic_src))
    if traffic is None and args.traffic != "none":      # the committed measurement of this workload
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
            if rec:
                traffic, traffic_src = rec["bytes_per_launch"], rec["source"]
        except (OSError, ValueError):
            pass
    if traffic is None:
        msg = "NO PMC TRAFFIC RECORDED for workload %r in profiles/traffic.json (run tools/pmc_run.sh): roofline.traffic is null" % key
        log("!!! " + msg)
        traffic_src = "MISSING: " + msg
        if args.require_traffic:
            raise SystemExit(msg)
    if rank == 0:
        total_q = world * n_q * args.steps
        avg_ms = float(np.mean(kernel_ms))
        achieved = alg_timed / (avg_ms * 1e-3) / 1e9
        metric_name = "fuzzy queries/sec (k=%d, %s≥%.2g) on %s-string dict" % (k, args.metric.capitalize(), args.similarity, _human(args.dict_size))
        try:      # the headline workload carries BASELINE.json's metric string verbatim (its "HBM GB/s fraction" half is `roofline.frac`)
            base_metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
            if base_metric.startswith(metric_name) and args.ngram == 3 and args.dict_variant == "uniform":
                metric_name = base_metric
        except (OSError, ValueError, KeyError):
            pass
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": "sg_search_kernel", "kernel_ms_avg": avg_ms,
                "kernel_ms_min": float(np.min(kernel_ms)), "kernel_ms_max": float(np.max(kernel_ms)),
                "algorithmic_bytes_per_launch": alg_timed, "algorithmic_bytes_per_query": alg_timed / n_q,
                "note": "achieved = algorithmic (ScanCount-volume) bytes / kernel time, SURVEY.md 8d; wire_* = PMC traffic / the same time; "
                        "kernel time = HIP events around one sg_suggest_batch_device call: the search launch, the parts launch of split "
                        "queries and the two query-ordering launches (~10 us) before them"}
        if traffic:
            roof["wire_gbps"] = traffic / (avg_ms * 1e-3) / 1e9
            roof["wire_frac"] = roof["wire_gbps"] / HBM_PEAK_GBS
            roof["traffic_over_algorithmic"] = traffic / alg_timed
        out = {
            "metric": metric_name,
            "value": total_q / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 (posting/counter work) + f64 (final score)",
            "data": "synthetic",
            "config": {"workload": "%s synthetic strings (len 8-32 over [a-z0-9]%s), q=%d, %s>=%.2g, k=%d, %d-query batch per GPU, %d distinct batches in rotation"
                                   % (_human(args.dict_size), "" if args.dict_variant == "uniform" else ", variant " + args.dict_variant,
                                      args.ngram, args.metric, args.similarity, k, n_q, n_b),
                       "baseline_config": args.config,
                       "parallelism": "query-sharded x%d, index replica per GPU%s" % (world, ", RCCL all_gather of results in every step" if world > 1 and args.gather else ""),
                       "rccl_gather_check": gather_ok,
                       "index": {"postings": st["n_postings"], "lists": st["n_lists"], "terms": st["n_terms"], "device_bytes": st["device_bytes"],
                                 "build": args.build},
                       "results_per_query": float(np.mean([np.minimum(c, k).mean() for c in cnt]))},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if host_rate:
            out["host_buffers"] = {"value": host_rate, "unit": "queries/s",
                                   "note": "sg_suggest_batch: pageable host buffers in and out over PCIe, synchronous (never `value`)"}
        if parity:
            out["parity_vs_oracle"] = parity
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _live_traffic(args, log):
    """HBM bytes per launch of the search kernel, measured now: this script again, as a child under `rocprofv3 --pmc
    FETCH_SIZE --kernel-trace` (PMC counters cannot be read from inside a process), a few steps of the same workload.
    bytes = FETCH_SIZE [KB] x 1024 x 2 — the gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section).
    -> (bytes, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rocprof:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="sg_pmc_", dir="/tmp")
    steps, warm = 4, 2
    cmd = [rocprof, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable,
           os.path.abspath(__file__), "--config", args.config, "--dict-size", str(args.dict_size), "--queries", str(args.queries),
           "--ngram", str(args.ngram), "--metric", args.metric, "--similarity", repr(args.similarity), "--topk", str(args.topk),
           "--batches", str(args.batches), "--dict-variant", args.dict_variant, "--build", args.build,
           "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--traffic", "none"]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=420)
        if r.returncode != 0:
            return None, "rocprofv3 child exited %d: %s" % (r.returncode, (r.stderr or "")[-300:])
        main_k, parts_k = [], []
        for path in sorted(glob.glob(tmp + "/**/*counter_collection.csv", recursive=True)):
            for row in csv.DictReader(open(path)):
                if row.get("Counter_Name") != "FETCH_SIZE":
                    continue
                if "sg_search_kernel_t<false, false," in row["Kernel_Name"]:
                    main_k.append(float(row["Counter_Value"]))
                elif "sg_search_kernel_t<true, false," in row["Kernel_Name"]:
                    parts_k.append(float(row["Counter_Value"]))
        if not main_k:
            return None, "no FETCH_SIZE rows for the search kernel in the child's counter CSV"
        kb = sum(main_k) / len(main_k) + (sum(parts_k) / len(main_k) if parts_k else 0.0)
        log("live PMC pass: FETCH_SIZE %.6g KB per launch over %d launches (%.0fs)" % (kb, len(main_k), time.time() - t0))
        return kb * 1024 * 2, ("live: rocprofv3 --pmc FETCH_SIZE --kernel-trace child pass of this run, %d launches of this workload "
                               "(search + parts kernels): FETCH_SIZE %.6g KB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md)" % (len(main_k), kb))
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as exc:
        return None, "live PMC pass failed: %r" % (exc,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _human(n):
    return "%dM" % (n // 1_000_000) if n % 1_000_000 == 0 else "%dk" % (n // 1000) if n % 1000 == 0 else str(n)


if __name__ == "__main__":
    main()
