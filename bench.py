#!/usr/bin/env python3
"""bench.py — fuzzy queries/sec of the MI355X engine on BASELINE.json's workloads.

  python bench.py --gpus N --steps K --warmup W [--config headline|cfg2|cfg3|cfg4|cfg5] [--mode procs|replicas]

A "step" is one pass of the hot path (tokenise -> posting lookup -> T-occurrence count -> score -> top-k)
over one batch of synthetic queries per GPU, inputs already resident in HBM.  The default workload is the
one BASELINE.json's `metric` is quoted on: 10M synthetic strings (len 8-32 over [a-z0-9]), q=3,
Jaccard>=0.5, k=10, 65,536 edited queries per GPU and step.  --config selects the other BASELINE.json
configs (cfg2: 1M strings; cfg3: Cosine>=0.4 k=20; cfg4: q=2 Dice>=0.5; cfg5: the spellchecker caller).
Steps rotate over --batches (default 4) distinct query batches, all resident before the timed region.

N > 1 (weak scaling: every GPU holds a full index replica and gets its own batches; no data-path collective):
  --mode procs (default)  one process per GPU under torch.distributed.run.  Started WITHOUT a launcher
                          (`python bench.py --gpus 8`, no WORLD_SIZE) the script re-executes itself under
                          `python -m torch.distributed.run --nproc-per-node N` — it never measures one GPU and
                          calls it N.  Rank 0 then also measures the single-process replica path below and
                          reports it as `replicas_mode`, beside the process-per-GPU number.
  --mode replicas         ONE process: sg_index_replicate over N devices + sg_suggest_batch_multi on host
                          buffers (a worker thread per replica) — what a Go host behind the C ABI runs.
                          PCIe-inclusive by construction, so its line says so (`config.pcie_inclusive`).

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline      achieved / frac = MEASURED memory traffic per call (rocprofv3 --pmc FETCH_SIZE x 1024 x 2 = the 128-byte lines
                fetched past the L2, calibrated on this box: profiles/r05_membench.txt; live child pass of the same workload,
                else the committed figure in profiles/traffic.json) / the call's average duration (HIP events on the launch
                stream) / the 8 TB/s peak — a physical fraction, always <= 1.  No counter separates Infinity-Cache hits from
                HBM reads (membench: TCC_EA0_RDREQ_DRAM = TCC_EA0_RDREQ inside the cache as outside), so it is an UPPER
                bound of the HBM fraction; frac_of_achievable = against the line traffic the memory system delivers for random
                contiguous segments on this box (7.2 TB/s, profiles/membench.json).  [r5] `kernels` = per-kernel time and
                traffic of the call (plan / stream / verify of the pipeline, or the fused kernel), `dominant` the stream
                (or fused) kernel's own fraction.
                effective_gbps / effective_frac = ALGORITHMIC bytes per launch (every posting of every
                query term in every admissible segment once, sg_suggest_algorithmic_bytes) over the same duration: the
                rate a full ScanCount scan would need to answer as fast.  List skipping and compressed postings read
                less than that volume, so this figure may exceed the peak; it is not a bandwidth.
  cpu_baseline  the CPU oracle — a C++ restatement of the Go path, kind "port" — on this host: all hardware
                threads, and one thread (`one_thread`), each on a bounded sample of the same batch
  configs       (default run only) sub-records for BASELINE.json's cfg2 / cfg3 / cfg4 measured the same way in the same
                process: value, kernel ms, both fractions, bit-exactness against the oracle on a sample
"""
import argparse
import json
import os
import socket
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
WORKLOAD_KEYS = ("dict_size", "queries", "ngram", "metric", "similarity", "topk")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a timed region of ~0.3 s at the headline, so that "
                                                            "run-to-run differences of a few per cent are above the noise; the spread is in roofline.kernel_ms_*)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS) + ["cfg5"],
                    help="BASELINE.json config (default: the one the metric is quoted on); explicit flags below override it")
    ap.add_argument("--mode", default="procs", choices=["procs", "replicas"],
                    help="N>1: a process per GPU (torch.distributed.run; self-spawned when no launcher is present) or ONE "
                         "process driving a replica per GPU through sg_suggest_batch_multi")
    ap.add_argument("--dict-size", type=int, default=None)
    ap.add_argument("--queries", type=int, default=None, help="queries per GPU per step")
    ap.add_argument("--ngram", type=int, default=None)
    ap.add_argument("--metric", default=None)
    ap.add_argument("--similarity", type=float, default=None)
    ap.add_argument("--topk", type=int, default=None)
    ap.add_argument("--batches", type=int, default=4, help="distinct query batches the steps rotate over")
    ap.add_argument("--dict-variant", default="uniform", choices=["uniform", "skewed", "families", "skewed-families", "words"],
                    help="SURVEY.md §8d dictionary variants (headline = uniform); families = base + 3 edited copies; words = the reference's "
                         "235 887-word test dictionary (tests/golden/words.dict.xz, its own index description), --dict-size ignored")
    ap.add_argument("--build", default="device", choices=["device", "host"],
                    help="index build: on the GPU (sg_index_build_device, 0.4 s at 10M; the posting store stays where it was "
                         "built) or on the host (sg_index_build, then uploaded); same arrays either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-rate", action="store_true", help="skip the host-buffer legs (the profiler child passes: every call of the run is then a device-resident one)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries timed on the CPU oracle (0 = auto)")
    ap.add_argument("--sub-configs", default="auto",
                    help="comma list of configs measured after the main one and reported under `configs`: BASELINE's cfg2..cfg5, and the "
                         "result-dense workloads families / words / skewed "
                         "(auto = cfg3,cfg4,skewed,families,cfg2,words,cfg5 for the plain default run at N=1, none otherwise; 'none' = off)")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: include the optional RCCL all_gather of the k*(u32,f64) result rows in every timed step "
                         "(default: results stay sharded — the path has no data-path collective; one untimed gather validates RCCL)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a rocprofv3 --pmc run, reported as roofline.traffic (default: measured live)")
    ap.add_argument("--require-traffic", action="store_true", help="exit non-zero when no traffic figure could be had for the workload")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "file", "none"],
                    help="roofline.traffic: live = a rocprofv3 --pmc FETCH_SIZE pass over this same workload in a child process "
                         "(after the timed region; N=1 only); file = the committed measurement in profiles/traffic.json; "
                         "auto = live when rocprofv3 is there, else file")
    args = ap.parse_args(argv)
    args.explicit_workload = any(getattr(args, k) is not None for k in WORKLOAD_KEYS)
    return args


def apply_preset(args, name):
    for key, val in CONFIGS[name].items():
        if getattr(args, key) is None:
            setattr(args, key, val)


def self_spawn(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: become `python -m torch.distributed.run ... bench.py <same args>`."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and os.environ.get("SG_BENCH_SINGLE_DEVICE") != "1":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node — refusing to measure fewer GPUs than asked for"
                         % (args.gpus, have))
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: re-executing as %s" % (args.gpus, " ".join(cmd[1:8])), file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


class Env:
    """rank / world / device of this process"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if args.mode == "procs" else 1
        if args.mode == "procs" and self.world != args.gpus:
            raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (self.world, args.gpus))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
        # test hooks (1-GPU box): SG_BENCH_SINGLE_DEVICE=1 puts every rank / replica on GPU 0, SG_BENCH_BACKEND=gloo avoids RCCL's
        # one-rank-per-GPU rule — exercises the N>1 control flow (barriers, max-over-ranks, per-rank batches), not the links
        self.single_device = os.environ.get("SG_BENCH_SINGLE_DEVICE") == "1"
        if self.single_device:
            self.local_rank = 0
        self.backend = os.environ.get("SG_BENCH_BACKEND", "nccl")
        self.nccl_group = None
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # The path has no data-path collective: barriers and the max over the ranks' clocks are control plane and go over
            # gloo (CPU) — a rank that waits for another keeps no spinning kernel on its GPU, and an RCCL problem on the node
            # cannot take the measurement with it.  RCCL is only needed for the optional gather of the result rows: a group of
            # its own, made lazily, checked once outside the timed region (config.rccl_gather_check).
            dist.init_process_group("gloo")
            self.nccl_group = None
            # what became of RCCL in this run, for the line (`config.rccl_status`): a run must not look like an RCCL run when it was not
            self.rccl_status = "not attempted (SG_BENCH_BACKEND=%s)" % self.backend
            if self.backend == "nccl":
                try:
                    self.nccl_group = dist.new_group(backend="nccl")
                    self.rccl_status = "group created"
                except Exception as exc:
                    self.rccl_status = "refused at group creation: %r" % (exc,)
                    self.log("no RCCL group (%r): the optional result gather is skipped" % (exc,))
        else:
            self.rccl_status = "n/a (one rank)"

    def log(self, *a):
        if self.rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)


_DICT_CACHE = {}
_ORACLE_CACHE = {}


WORDS_DESC = dict(ngram_size=3, wrap=("^", "$"), pad="$", alphabet=("english", "numbers", "$^"))   # (tests/conftest.py: the reference's words index)


def get_dict(size, variant):
    from suggest_amd import synth
    key = (size, variant)
    if key not in _DICT_CACHE:
        _DICT_CACHE.clear()                                   # (one 10M dictionary at a time: ~200 MB)
        if variant == "words":      # the reference's own word list: real-language n-gram lists, dozens of matches per query
            import lzma
            import numpy as np
            lines = lzma.open(os.path.join(ROOT, "tests", "golden", "words.dict.xz")).read().split(b"\n")
            if lines and lines[-1] == b"":
                lines.pop()
            offs = np.zeros(len(lines) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(x) for x in lines]).astype(np.uint64)
            _DICT_CACHE[key] = (np.frombuffer(b"".join(lines), dtype=np.uint8).copy(), offs)
        else:
            _DICT_CACHE[key] = synth.make_dict(size, seed=1, skewed="skewed" in variant, families=3 if "families" in variant else 0)
    return _DICT_CACHE[key]


def cpu_quota():
    """cores the container may use: the cgroup CPU quota, else the affinity mask -> (float or None, int)"""
    quota = None
    try:
        q_us, period = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q_us == "max" else float(q_us) / float(period)
    except (OSError, ValueError):
        pass
    try:
        hw = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        hw = os.cpu_count() or 1
    return quota, hw


def measure(env, args, w, steps, warmup, cpu_baseline=True, host_rate=True, traffic_mode="auto", replicas_leg=False, parity_rows=None):
    """One workload `w` (dict of WORKLOAD_KEYS + variant) on this rank's GPU -> the record (rank 0) or None."""
    import numpy as np
    torch, dist = env.torch, env.dist
    from suggest_amd import IndexDescription, NGramIndex, synth
    rank, world, dev, log = env.rank, env.world, env.dev, env.log
    desc_kw = dict(WORDS_DESC if w["variant"] == "words" else synth.DESCRIPTION, ngram_size=w["ngram"])
    k, n_q, n_b = w["topk"], w["queries"], max(1, args.batches)
    t0 = time.time()
    blob, offs = get_dict(w["dict_size"], w["variant"])
    w = dict(w, dict_size=len(offs) - 1)
    # batch b of rank r = queries [(r * n_b + b) * n_q, ...) of one deterministic stream (seed 2)
    batches = [synth.make_queries(n_q, blob, offs, seed=2, start=(rank * n_b + b) * n_q) for b in range(n_b)]
    log("[%s] dict %d strings + %d batches of %d queries generated in %.1fs" % (w["name"], w["dict_size"], n_b, n_q, time.time() - t0))
    t0 = time.time()
    index = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**desc_kw), device=env.local_rank, build=args.build)
    st = index.stats()
    log("[%s] index built (%s) + uploaded in %.1fs: %s" % (w["name"], args.build, time.time() - t0, st))
    alg = [index.algorithmic_bytes(qb, qo, w["metric"], w["similarity"], k) for qb, qo in batches]

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
        index.suggest_batch_device(d_q[b].data_ptr(), d_offs[b].data_ptr(), n_q, w["metric"], w["similarity"], k,
                                   d_ids[b].data_ptr(), d_sc[b].data_ptr(), d_cnt[b].data_ptr(), stream=stream.cuda_stream)

    def gather(b, force=False):
        if world > 1 and (args.gather or force):   # top-k gather over RCCL/xGMI: k*(u32,f64) per query
            dist.all_gather_into_tensor(g_ids, d_ids[b], group=env.nccl_group)
            dist.all_gather_into_tensor(g_sc, d_sc[b], group=env.nccl_group)
            dist.all_gather_into_tensor(g_cnt, d_cnt[b], group=env.nccl_group)

    # setup, untimed like the index build before it: calls until the GPU's clocks have settled (up to 64 calls or 0.25 s) — behind a
    # fresh build and the host work around it the first dozens of calls run ~10 % long, which `--steps 20 --warmup 5` would time
    # (48.6 M q/s against 50.0 M with 60 warm-up steps or 200 timed ones, same box: profiles/r06last_settle_*).  Then the W
    # warm-up steps and the K timed ones as asked for.
    t_settle, n_settle = time.perf_counter(), 0
    while n_settle < 64 and time.perf_counter() - t_settle < 0.25:
        step(n_settle % n_b)
        n_settle += 1
        if n_settle % 8 == 0:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    for i in range(warmup):
        step(i % n_b)
        gather(i % n_b)
    env.barrier()
    ls0 = index.launch_stats()
    ps0 = index.pipe_stats()
    pv0 = index.pipe_volumes()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t_start = time.perf_counter()
    for i in range(steps):
        ev[i][0].record(stream)
        step(i % n_b)
        ev[i][1].record(stream)
        gather(i % n_b)
    env.barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    ls1 = index.launch_stats()
    ps1 = index.pipe_stats()
    pv1 = index.pipe_volumes()
    pv_n = (pv1["sampled"] - pv0["sampled"]) % (1 << 32)
    pipe_vol = {k_: ((pv1[k_] - pv0[k_]) % (1 << 32)) / float(pv_n) for k_ in ("groups", "lists", "rows", "candidates")} if pv_n else None
    if pipe_vol:
        pipe_vol.update(packed_chunks=pv1["packed_chunks"], wide_descriptors=pv1["wide"], stream_workgroup=pv1.get("stream_shape"))
    pipe_fb = {k_: ((ps1[k_] - ps0[k_]) % (1 << 32)) / float(steps) for k_ in ps1 if k_ != "queries"}    # queries per call the pipeline left to the fused kernel
    pipe_q = (ps1.get("queries", 0) - ps0.get("queries", 0)) / float(steps)                              # ... and those it took
    d_s, d_c = (ls1["sampled"] - ls0["sampled"]) % (1 << 32), (ls1["chunks"] - ls0["chunks"]) % (1 << 64)
    # bytes the PACKED algorithm has to move per launch: 16 B x the chunks of the lists the kernel streams (the k longest are
    # skipped, 7 postings per chunk), from the kernel's own count over its sampled queries (one in 32), + queries in + rows out
    model_bytes = (d_c / d_s * n_q * 16 + float(np.mean([len(b[0]) for b in batches])) + 12.0 * k * n_q) if d_s else None
    alg_timed = float(np.mean([alg[i % n_b] for i in range(steps)]))       # algorithmic bytes per launch, timed launches
    t = torch.tensor([elapsed], dtype=torch.float64)
    per_rank = None
    if world > 1:
        every = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, t)                               # (gloo: the ranks' own clocks around the same K steps)
        rates = [n_q * steps / float(x.item()) for x in every]
        per_rank = {"min": min(rates), "max": max(rates), "mean": float(np.mean(rates)), "unit": "queries/s",
                    "kernel_ms_avg_rank0": float(np.mean(kernel_ms))}
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    gather_ok, gather_ms = None, None
    if world > 1:                       # functional check of the optional result gather over RCCL, and its cost per step (outside the timed region)
        try:
            gather(0, force=True)
            torch.cuda.synchronize(dev)
            mine = slice(rank * n_q, (rank + 1) * n_q)
            gather_ok = bool(torch.equal(g_ids[mine], d_ids[0]) and torch.equal(g_cnt[mine], d_cnt[0]))
            env.barrier()
            t0 = time.perf_counter()
            for i in range(5):
                gather(i % n_b, force=True)
            torch.cuda.synchronize(dev)
            tg = torch.tensor([(time.perf_counter() - t0) / 5 * 1e3], dtype=torch.float64)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            gather_ms = float(tg.item())
            if env.nccl_group is not None:
                env.rccl_status = "ok: all_gather of the result rows over RCCL, %d ranks" % world
        except Exception as exc:        # the timed region has no collective: report, do not lose the measurement
            log("result gather over RCCL failed: %r" % (exc,))
            gather_ok = False
            if env.backend == "nccl":
                env.rccl_status = "refused at the first collective: %s" % (repr(exc)[:300],)

    reached = {i % n_b for i in range(warmup)} | {i % n_b for i in range(steps)}
    for b in range(n_b):                                 # (batches the run never reached)
        if b not in reached:
            step(b)
    torch.cuda.synchronize(dev)
    ids = [x.cpu().numpy().view(np.uint32) for x in d_ids]
    sc = [x.cpu().numpy() for x in d_sc]
    cnt = [x.cpu().numpy().view(np.uint32) for x in d_cnt]

    # ---- N > 1: every rank holds a sample of ITS OWN rows (its own batch, its own GPU, its own replica of the index) against the
    #      CPU oracle, and the ranks the RCCL group really spans are listed: a run nobody can watch checks itself ----
    parity_ranks, rccl_ranks = None, None
    if world > 1:
        ok_n = torch.zeros(2, dtype=torch.float64)
        try:
            import oracle
            n_s = int(min(n_q, 512))
            ora = oracle.OracleIndex(blob=blob, offs=offs, **desc_kw)
            qb, qo = batches[0]
            oi, os_, oc, _ = ora.suggest_batch(qb[:int(qo[n_s])], qo[:n_s + 1], w["metric"], w["similarity"], k, threads=max(1, (os.cpu_count() or 1) // world))
            valid = np.arange(k)[None, :] < np.minimum(oc, k)[:, None]
            same = np.array_equal(cnt[0][:n_s], oc) and np.array_equal(ids[0][:n_s][valid], oi[valid]) and \
                np.array_equal(sc[0][:n_s].view(np.uint64)[valid], os_.view(np.uint64)[valid])
            ok_n[0], ok_n[1] = n_s, 1.0 if same else 0.0
            del ora
        except Exception as exc:
            log("rank %d: oracle check failed to run: %r" % (rank, exc))
        every = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, ok_n)                                 # (gloo)
        parity_ranks = [{"rank": r, "checked_queries": int(x[0].item()), "bit_exact": bool(x[1].item() == 1.0) if x[0].item() else None} for r, x in enumerate(every)]
        if env.nccl_group is not None:
            try:
                seen = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(world)]
                dist.all_gather(seen, torch.tensor([rank], dtype=torch.int32, device=dev), group=env.nccl_group)
                rccl_ranks = {"group_size": int(dist.get_world_size(group=env.nccl_group)), "ranks": [int(x.item()) for x in seen]}
            except Exception as exc:
                rccl_ranks = {"error": repr(exc)}

    # ---- the host-buffer entry point (what a cgo caller uses): PCIe-inclusive, never the headline value ----
    host = None
    if rank == 0 and world == 1 and host_rate:
        qb, qo = batches[0]
        index.suggest_batch(blob=qb, offs=qo, metric=w["metric"], similarity=w["similarity"], k=k)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            h_ids, h_sc, h_cnt = index.suggest_batch(blob=qb, offs=qo, metric=w["metric"], similarity=w["similarity"], k=k)
        host = reps * n_q / (time.perf_counter() - t0)
        if not (np.array_equal(h_cnt, cnt[0]) and np.array_equal(h_ids, ids[0])):
            raise SystemExit("sg_suggest_batch (host buffers) and sg_suggest_batch_device disagree")

    # ---- the same through sg_suggest_submit / sg_ticket_wait: pinned buffers, two tickets in flight from ONE host thread, so
    #      the copies of one batch run beside the kernel of another (what the Go shim's dispatcher does) ----
    piped = None
    if rank == 0 and world == 1 and host_rate:
        from suggest_amd.index import pinned_array
        slots = []
        for b in range(min(n_b, 4)):
            qb, qo = batches[b]
            pb = pinned_array((max(qb.size, 1),), np.uint8)[:qb.size]; pb[:] = qb
            po = pinned_array((n_q + 1,), np.uint64); po[:] = qo
            slots.append((pb, po, pinned_array((n_q, k), np.uint32), pinned_array((n_q, k), np.float64), pinned_array((n_q,), np.uint32)))

        def submit(i):
            pb, po, p_ids, p_sc, p_cnt = slots[i % len(slots)]
            return index.suggest_submit(pb, po, w["metric"], w["similarity"], k, p_ids, p_sc, p_cnt)

        reps = max(8, min(steps, 20))
        for i in range(len(slots)):                                  # warm-up: every slot once (allocates the engine's device blocks)
            submit(i).wait()
        piped, in_flight, piped_by = None, None, {}
        for depth in ([2, 3] if len(slots) >= 3 else [min(2, len(slots))]):
            t_sub = 0.0
            t0 = time.perf_counter()
            pending = []
            for i in range(reps):
                ts = time.perf_counter()
                pending.append(submit(i))
                t_sub += time.perf_counter() - ts
                if len(pending) >= depth:
                    pending.pop(0).wait()
            for t in pending:
                t.wait()
            rate = reps * n_q / (time.perf_counter() - t0)
            piped_by[depth] = {"queries_per_s": rate, "submit_ms_avg": t_sub / reps * 1e3}
            if piped is None or rate > piped:
                piped, in_flight = rate, depth
        for b in range(len(slots)):
            if not (np.array_equal(slots[b][4], cnt[b]) and np.array_equal(slots[b][2], ids[b]) and
                    np.array_equal(slots[b][3].view(np.uint64), sc[b].view(np.uint64))):
                raise SystemExit("sg_suggest_submit / sg_ticket_wait rows differ from sg_suggest_batch_device's (batch %d)" % b)
        log("[%s] host buffers: synchronous %.2f M q/s, pipelined (%d tickets in flight, pinned) %.2f M q/s  %s" % (w["name"], host / 1e6, in_flight, piped / 1e6, piped_by))

    # ---- one process, a replica per GPU, sg_suggest_batch_multi (rank 0, after the timed region; the other ranks wait) ----
    replicas = None
    if replicas_leg and rank == 0:
        try:
            replicas = replicas_measure(env, args, index, w, batches, ids, cnt, n_gpus=args.gpus, steps=max(4, min(steps, 10)), warmup=2)
        except Exception as exc:
            log("replicas leg failed: %r" % (exc,))
            replicas = {"error": repr(exc)}
    if replicas_leg and world > 1:
        dist.barrier()

    # ---- CPU baseline: the oracle (restatement of the Go path) on this host, rank 0, N=1 only ----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and cpu_baseline:
        import oracle
        t0 = time.time()
        okey = (w["dict_size"], w["variant"], w["ngram"])
        if okey not in _ORACLE_CACHE:                          # (cfg3 runs on the headline's dictionary: one oracle index for both)
            _ORACLE_CACHE.clear()
            _ORACLE_CACHE[okey] = oracle.OracleIndex(blob=blob, offs=offs, **desc_kw)
        ora = _ORACLE_CACHE[okey]
        log("[%s] oracle index ready in %.1fs" % (w["name"], time.time() - t0))
        cores = os.cpu_count() or 1
        qb, qo = batches[0]

        def timed(n, threads):
            t0 = time.perf_counter()
            res = ora.suggest_batch(qb[:int(qo[n])], qo[:n + 1], w["metric"], w["similarity"], k, threads=threads)
            return res, time.perf_counter() - t0

        budget = 12.0 if cpu_baseline is True else float(cpu_baseline)       # seconds of wall time per leg
        quota, hw = cpu_quota()
        granted = max(1, min(hw, int(round(quota)))) if quota else hw              # the cores this process can actually use
        note = "C++ restatement of the Go path (oracle/), OpenMP across queries; the Go reference is not runnable here (no toolchain)"

        def leg(threads, n_fixed=None):
            n = n_fixed
            if n is None:
                probe = min(n_q, 2048 if budget >= 8 else 512)
                _, dt_p = timed(probe, threads)
                n = int(min(n_q, max(probe, probe / max(dt_p, 1e-6) * budget)))
            res, dt_l = timed(n, threads)
            return res, n, dt_l

        n_fix = min(args.cpu_sample or parity_rows or 0, n_q) or None
        (oi, os_, oc, used), n_s, dt = leg(granted, n_fix)
        legs = {"granted_cores": {"value": n_s / dt, "unit": "queries/s", "cores": used, "sample": "first %d queries of batch 0" % n_s}}
        if hw != granted:      # every hardware thread the host shows: oversubscribes the container's quota (kept as a labelled record)
            n_h = int(max(16, min(n_s, legs["granted_cores"]["value"] * budget / 2)))
            (_, _, _, used_h), dt_h = timed(n_h, hw)
            legs["all_hw_threads"] = {"value": n_h / dt_h, "unit": "queries/s", "cores": used_h, "sample": "first %d queries of batch 0" % n_h,
                                      "note": "threads = every hardware thread of the host; the container's CPU quota is %s cores" % (quota,)}
        n_1 = int(max(16, min(n_s, legs["granted_cores"]["value"] / max(used, 1) * budget / 2)))      # ~budget/2 s on one thread
        (_, _, _, used1), dt1 = timed(n_1, 1)
        legs["one_thread"] = {"value": n_1 / dt1, "unit": "queries/s", "cores": used1, "sample": "first %d queries of batch 0" % n_1}
        best = max(("granted_cores", "all_hw_threads"), key=lambda n_: legs.get(n_, {"value": -1.0})["value"])
        cpu = {"value": legs[best]["value"], "unit": "queries/s", "cores": legs[best]["cores"], "cores_granted": granted, "cpu_quota_cores": quota,
               "hw_threads": hw, "best_leg": best, "kind": "port",
               "sample": "%s, same %d-string dictionary; %s" % (legs[best]["sample"], w["dict_size"], note)}
        cpu.update({n_: v for n_, v in legs.items()})
        valid = np.arange(k)[None, :] < np.minimum(oc, k)[:, None]
        same = np.array_equal(cnt[0][:n_s], oc) and np.array_equal(ids[0][:n_s][valid], oi[valid]) and \
            np.array_equal(sc[0][:n_s].view(np.uint64)[valid], os_.view(np.uint64)[valid])
        parity = {"checked_queries": int(n_s), "bit_exact": bool(same)}
        # the Go reference's threshold tightening can resolve an exact tie at the k-th place differently from run to
        # run; parity is defined against the untightened order.  How many of the checked queries are tie-sensitive (k-th and
        # (k+1)-th best scores bit-equal) says how many rows that caveat can touch at all: the oracle at k + 1 on a subsample.
        n_t = int(min(n_s, 4096))
        (_, ts_, tc_, _), _ = (ora.suggest_batch(qb[:int(qo[n_t])], qo[:n_t + 1], w["metric"], w["similarity"], k + 1, threads=granted), None)
        tie = int(np.sum((tc_ > k) & (ts_.view(np.uint64)[:, k - 1] == ts_.view(np.uint64)[:, k])))
        parity["tie_sensitive_queries"] = {"count": tie, "of": n_t, "note": "k-th and (k+1)-th best scores bit-equal (SURVEY.md A.6)"}
        log("[%s] cpu baseline %.0f q/s on %d threads (%s; %d cores granted), %.0f q/s on one; GPU result bit-exact vs oracle on %d rows: %s"
            % (w["name"], cpu["value"], cpu["cores"], best, granted, cpu["one_thread"]["value"], n_s, same))
        del ora

    key = "%d/%d/q%d/%s/%.3g/k%d/%s" % (w["dict_size"], n_q, w["ngram"], w["metric"], w["similarity"], k, w["variant"])
    traffic, traffic_src, per_kernel = args.traffic_bytes if w["name"] == args.config else None, None, None
    if traffic:
        traffic_src = "--traffic-bytes"
    under_profiler = any(kk.startswith(("ROCPROF", "ROCP_")) for kk in os.environ)     # (no profiler inside a profiler)
    if traffic is None and traffic_mode in ("auto", "live") and rank == 0 and world == 1 and not (under_profiler and traffic_mode == "auto"):
        traffic, traffic_src, per_kernel = _live_traffic(args, w, log)
        if traffic is None and traffic_mode == "live":
            raise SystemExit("live PMC pass failed: " + str(traffic_src))
        if traffic is None:
            log("live PMC pass unavailable: %s" % (traffic_src,))
    if traffic is None and traffic_mode != "none":      # the committed measurement of this workload
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
            if rec:
                traffic, traffic_src = rec["bytes_per_launch"], "file" + (" (the N=1 PMC figure of this workload, per GPU; no PMC pass at N>1)" if world > 1 else "") + ": " + rec["source"]
        except (OSError, ValueError):
            pass
    if traffic is None:
        msg = "no PMC traffic for workload %r (live pass unavailable, none recorded in profiles/traffic.json): roofline.achieved / frac are null" % key
        log("!!! " + msg)
        traffic_src = "MISSING: " + msg
        if args.require_traffic:
            raise SystemExit(msg)
    index.close()
    if rank != 0:
        return None
    total_q = world * n_q * steps
    avg_ms = float(np.mean(kernel_ms))
    effective = alg_timed / (avg_ms * 1e-3) / 1e9
    wire = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
    try:
        mb = json.load(open(os.path.join(ROOT, "profiles", "membench.json")))
        achievable = float([x for x in mb["segments"] if x["seg_bytes"] == 1024][0]["lines_GBps_1GiB"])
    except (OSError, ValueError, KeyError, IndexError):
        achievable = 6290.0              # (the guide's copy ceiling)
    dominant, kernel_timing = None, None
    if per_kernel:
        dk = per_kernel.pop("_dominant")
        kernel_timing = per_kernel.pop("_timing", "the counter pass: mean of the later half of each kernel's dispatches (counter collection does not lengthen them: profiles/r06end2_*)")
        d = per_kernel[dk]
        if d["ms_per_call_under_profiler"] > 0:
            d_gbps = d["traffic_bytes_per_call"] / (d["ms_per_call_under_profiler"] * 1e-3) / 1e9
            dominant = {"kernel": {"stream": "sg_stream_kernel", "fused": "sg_search_kernel_t"}[dk], "ms": d["ms_per_call_under_profiler"],
                        "traffic": d["traffic_bytes_per_call"], "achieved": d_gbps, "frac": d_gbps / HBM_PEAK_GBS, "frac_of_achievable": d_gbps / achievable,
                        "share_of_call_time": d["ms_per_call_under_profiler"] / max(1e-9, sum(v["ms_per_call_under_profiler"] for v in per_kernel.values()))}
    pipe_on = pipe_q > 0.5 * n_q          # (most of the timed calls' queries took plan -> stream -> verify)
    flat = {}
    if dominant:       # (flat scalars: a record that keeps only scalar keys still carries the dominant kernel's own fraction)
        flat = {"dominant_kernel": dominant["kernel"], "dominant_ms": dominant["ms"], "dominant_traffic": dominant["traffic"],
                "dominant_achieved": dominant["achieved"], "dominant_frac": dominant["frac"], "dominant_frac_of_achievable": dominant["frac_of_achievable"]}
        for kind, v in (per_kernel or {}).items():
            flat["%s_ms" % kind] = v["ms_per_call_under_profiler"]
    roof = {"bound": "hbm", "achieved": wire, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": wire / HBM_PEAK_GBS if wire else None,
            "traffic": traffic, "traffic_source": traffic_src, **flat,
            "achievable": achievable, "frac_of_achievable": wire / achievable if wire else None,
            "dominant": dominant, "kernels": per_kernel, "kernels_timed_under": kernel_timing,
            "effective_gbps": effective, "effective_frac": effective / HBM_PEAK_GBS,
            "traffic_over_algorithmic": traffic / alg_timed if traffic else None,
            "model_bytes": model_bytes, "traffic_over_model": traffic / model_bytes if traffic and model_bytes else None,
            "kernel": ("sg_plan_kernel + sg_stream_kernel + sg_verify_kernel (pipeline.inc)" if pipe_on else "sg_search_kernel_t"), "kernel_ms_avg": avg_ms,
            "kernel_ms_min": float(np.min(kernel_ms)), "kernel_ms_max": float(np.max(kernel_ms)),
            "kernel_ms_p50": float(np.median(kernel_ms)), "kernel_ms_p05": float(np.percentile(kernel_ms, 5)), "kernel_ms_p95": float(np.percentile(kernel_ms, 95)),
            "kernel_ms_stdev": float(np.std(kernel_ms)), "timed_region_s": elapsed,
            "algorithmic_bytes_per_launch": alg_timed, "algorithmic_bytes_per_query": alg_timed / n_q,
            "note": "achieved/frac = measured memory traffic per call (PMC: 128-byte lines fetched past the L2, every kernel of the call) / the call's time "
                    "(HIP events around one sg_suggest_batch_device call, no profiler): the physical fraction of the 8 TB/s HBM peak — an upper bound of the "
                    "HBM fraction, no counter separates Infinity-Cache hits from HBM reads (profiles/membench.json).  achievable = the line traffic the memory "
                    "system delivers on this box for random contiguous 1 KiB segments over 1 GiB (tools/membench.hip).  dominant = the stream (or fused) "
                    "kernel alone, time and traffic from the profiler pass.  effective_* = algorithmic (ScanCount-volume, SURVEY.md 8d) bytes / the call's time: "
                    "list skipping and the packed posting store read less than that volume, so it may exceed the peak and is not a bandwidth — superseded "
                    "by model_bytes as the numerator that describes this engine (BASELINE.md).  model_bytes = what the packed algorithm must move: 16 B x the "
                    "chunks of the streamed lists (counted by the plan / the fused kernel over its sampled queries) + queries in + 12 B x k rows out; "
                    "traffic / model_bytes = what is read on top of that (lines a list only partly fills, slot records, forward-index records of verified "
                    "candidates, seg_off rows, term table)"}
    rec = {
        "value": total_q / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "setup_settle_calls": n_settle,       # untimed calls of the setup, before the warm-up steps (clocks: see measure())
        "ms_per_step": elapsed / steps * 1e3,
        "config": {"workload": "%s synthetic strings (len 8-32 over [a-z0-9]%s), q=%d, %s>=%.2g, k=%d, %d-query batch per GPU, %d distinct batches in rotation"
                               % (_human(w["dict_size"]), "" if w["variant"] == "uniform" else ", variant " + w["variant"],
                                  w["ngram"], w["metric"], w["similarity"], k, n_q, n_b),
                   "baseline_config": w["name"],
                   "parallelism": "query-sharded x%d, index replica per GPU, one process per GPU%s"
                                  % (world, ", RCCL all_gather of results in every step" if world > 1 and args.gather else ""),
                   "rccl_gather_check": gather_ok,
                   "rccl_status": env.rccl_status,      # "ok ..." / "refused ..." (e.g. two ranks on one GPU) / "not attempted (gloo)": never silent
                   "rccl_ranks": rccl_ranks,          # the ranks the RCCL group spans, as all-gathered over it (N > 1)
                   "parity_per_rank": parity_ranks,   # every rank's own rows against the CPU oracle on a sample (N > 1)
                   "gather_ms": gather_ms,       # the optional all_gather of one step's result rows (3 collectives, k*(u32,f64)+u32 per query), max over ranks, untimed region
                   "per_rank": per_rank,
                   "index": {"postings": st["n_postings"], "lists": st["n_lists"], "terms": st["n_terms"], "device_bytes": st["device_bytes"],
                             "build": args.build},
                   "results_per_query": float(np.mean([np.minimum(c, k).mean() for c in cnt])),
                   "pipeline": {"on": pipe_on, "queries_per_call": pipe_q, "left_to_fused_kernel_per_call": pipe_fb,
                                "per_sampled_query": pipe_vol}},
        "roofline": roof,
        "cpu_baseline": cpu,
    }
    if host:
        rec["host_buffers"] = {"value": host, "unit": "queries/s",
                               "note": "sg_suggest_batch: pageable host buffers in and out over PCIe, synchronous (never `value`)"}
    if piped:
        rec["host_buffers_pipelined"] = {"value": piped, "unit": "queries/s", "frac_of_device_resident": piped / (total_q / elapsed),
                                         "tickets_in_flight": in_flight, "by_depth": piped_by,
                                         "note": "sg_suggest_submit / sg_ticket_wait from ONE host thread: pinned host buffers (sg_host_alloc), "
                                                 "PCIe-inclusive (never `value`); rows equal the device-resident run's"}
    if replicas:
        rec["replicas_mode"] = replicas
    if parity:
        rec["parity_vs_oracle"] = parity
    return rec


def replicas_measure(env, args, index, w, batches, ids0, cnt0, n_gpus, steps, warmup):
    """ONE process, a replica per GPU behind one handle, the batch sliced over them by sg_suggest_batch_multi (host buffers in
    and out, a worker thread per replica).  The step is n_gpus x the per-GPU batch; slice 0 of every step is this rank's batch
    b, so its rows must equal the device-resident run's."""
    import numpy as np
    devices = [0] * n_gpus if env.single_device else list(range(n_gpus))
    t0 = time.time()
    index.replicate(devices)
    env.log("[%s] %d replicas resident (%s) after %.1fs" % (w["name"], len(index.replicas()), index.replicas(), time.time() - t0))
    k, n_q = w["topk"], w["queries"]
    big = []
    for b in range(len(batches)):      # n_gpus copies of batch b (same work per slice as the process-per-GPU run does per rank)
        qb, qo = batches[b]
        blob = np.concatenate([qb] * n_gpus)
        offs = np.concatenate([[0]] + [qo[1:].astype(np.uint64) + np.uint64(i * int(qo[-1])) for i in range(n_gpus)]).astype(np.uint64)
        big.append((blob, offs))
    for i in range(warmup):
        index.suggest_batch(blob=big[i % len(big)][0], offs=big[i % len(big)][1], metric=w["metric"], similarity=w["similarity"], k=k, multi=True)
    t0 = time.perf_counter()
    for i in range(steps):
        r_ids, r_sc, r_cnt = index.suggest_batch(blob=big[i % len(big)][0], offs=big[i % len(big)][1], metric=w["metric"],
                                                 similarity=w["similarity"], k=k, multi=True)
    dt = time.perf_counter() - t0
    b_last = (steps - 1) % len(big)
    ok = True
    for g in range(n_gpus):
        sl = slice(g * n_q, (g + 1) * n_q)
        ok = ok and np.array_equal(r_cnt[sl], cnt0[b_last]) and np.array_equal(r_ids[sl], ids0[b_last])
    if not ok:
        raise RuntimeError("sg_suggest_batch_multi rows differ from the device-resident run's")
    rec = {"value": steps * n_gpus * n_q / dt, "unit": "queries/s", "n_gpus": n_gpus, "devices": index.replicas(), "steps": steps,
           "ms_per_step": dt / steps * 1e3, "rows_equal_device_run": bool(ok),
           "note": "ONE process: sg_index_replicate + sg_suggest_batch_multi, %d x %d queries per call from pageable host buffers, "
                   "a worker thread per replica; PCIe-inclusive (never `value` of the process-per-GPU line)" % (n_gpus, n_q)}
    # ---- the same from ONE host thread over pinned buffers: a ticket per replica (sg_suggest_submit_on), two steps in flight, so
    #      every GPU's copies run beside its own previous kernel and nothing is staged ----
    from suggest_amd.index import pinned_array
    n_b = min(len(batches), 2)
    slots = []          # [step slot][replica] -> (blob, offs, ids, scores, counts), all pinned
    for s in range(2):
        qb, qo = batches[s % n_b]
        per = []
        for g in range(n_gpus):
            pb = pinned_array((max(qb.size, 1),), np.uint8)[:qb.size]; pb[:] = qb
            po = pinned_array((n_q + 1,), np.uint64); po[:] = qo
            per.append((pb, po, pinned_array((n_q, k), np.uint32), pinned_array((n_q, k), np.float64), pinned_array((n_q,), np.uint32)))
        slots.append(per)

    def submit_step(i):
        return [index.suggest_submit(pb, po, w["metric"], w["similarity"], k, p_ids, p_sc, p_cnt, replica=g)
                for g, (pb, po, p_ids, p_sc, p_cnt) in enumerate(slots[i % 2])]
    for i in range(2):
        for t in submit_step(i):
            t.wait()
    p_steps = max(4, steps)
    t0 = time.perf_counter()
    pending = []
    for i in range(p_steps):
        pending.append(submit_step(i))
        if len(pending) >= 2:
            for t in pending.pop(0):
                t.wait()
    for ts in pending:
        for t in ts:
            t.wait()
    dt_p = time.perf_counter() - t0
    ok_p = all(np.array_equal(slots[s][g][4], cnt0[s % n_b]) and np.array_equal(slots[s][g][2], ids0[s % n_b]) for s in range(2) for g in range(n_gpus))
    if not ok_p:
        raise RuntimeError("sg_suggest_submit_on rows differ from the device-resident run's")
    rec["pipelined"] = {"value": p_steps * n_gpus * n_q / dt_p, "unit": "queries/s", "ms_per_step": dt_p / p_steps * 1e3, "steps": p_steps,
                        "rows_equal_device_run": True,
                        "note": "ONE host thread: a ticket per replica and step (sg_suggest_submit_on), two steps in flight, pinned buffers "
                                "(sg_host_alloc): PCIe-inclusive, no staging copy"}
    return rec


def workload_of(args, name):
    w = dict(CONFIGS[name], name=name, variant=args.dict_variant if name == args.config else "uniform")
    if name == args.config:
        for kk in WORKLOAD_KEYS:
            w[kk] = getattr(args, kk)
    return w


def main():
    args = parse_args()
    if args.config == "cfg5":
        import bench_spell
        return bench_spell.main(args)
    apply_preset(args, args.config)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.mode == "procs" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)            # does not return
    env = Env(args)
    import numpy as np  # noqa: F401

    w = workload_of(args, args.config)
    if args.mode == "replicas":
        return main_replicas(env, args, w)

    subs = []
    if args.sub_configs == "auto":
        if env.world == 1 and args.config == "headline" and not args.explicit_workload and args.dict_variant == "uniform":
            subs = ["cfg3", "cfg4", "skewed", "families", "cfg2", "words", "cfg5"]
    elif args.sub_configs != "none":
        subs = [s for s in args.sub_configs.split(",") if s]
        if env.world > 1:
            raise SystemExit("--sub-configs is an N=1 option")
    rec = measure(env, args, w, args.steps, args.warmup, cpu_baseline=not args.no_cpu_baseline, traffic_mode=args.traffic,
                  replicas_leg=env.world > 1, host_rate=not args.no_host_rate)
    sub_recs = {}
    for name in subs:       # cfg3 and cfg4 first: they share the headline's dictionary
        t0 = time.time()
        if name == "cfg5":      # the spellchecker caller (bench_spell.py): its own model, its own record
            import argparse
            import bench_spell
            _DICT_CACHE.clear()
            _ORACLE_CACHE.clear()
            sa = argparse.Namespace(**vars(args))
            sa.dict_size = sa.queries = sa.topk = sa.similarity = None
            sa.steps, sa.warmup, sa.cpu_sample = max(5, min(args.steps, 10)), 2, 4096
            try:
                r5 = bench_spell.run(sa, env=env)
            except (Exception, SystemExit) as exc:      # a failed sub-record must not lose the main line
                env.log("[cfg5] sub-record failed: %r" % (exc,))
                r5 = None
            if r5:
                sub_recs[name] = {"workload": r5["config"]["workload"], "metric": r5["metric"], "value": r5["value"], "unit": r5["unit"], "steps": r5["steps"],
                                  "ms_per_step": r5["ms_per_step"], "step_gpu_ms_avg": r5["roofline"]["step_gpu_ms_avg"],
                                  "frac": r5["roofline"]["frac"], "achieved_gbps": r5["roofline"]["achieved"], "traffic": r5["roofline"]["traffic"],
                                  "traffic_source": r5["roofline"]["traffic_source"], "kernels": r5["roofline"]["kernels"],
                                  "bit_exact": (r5.get("parity_vs_oracle") or {}).get("bit_exact"),
                                  "checked_queries": (r5.get("parity_vs_oracle") or {}).get("checked_queries"),
                                  "cpu_baseline": r5["cpu_baseline"], "host_buffers": r5.get("host_buffers"),
                                  "predictions_per_query": r5["config"]["predictions_per_query"]}
                env.log("[cfg5] sub-record done in %.0fs" % (time.time() - t0))
            continue
        if name in ("skewed", "families"):    # SURVEY.md 8d's variants of the headline: Zipf symbols (long lists at q = 3); families of a base
            sw = dict(CONFIGS["headline"], name=name, variant=name)      # string + 3 edited copies (several near matches per query: the top-k works)
        elif name == "words":                 # the reference's own word list, k = 10: real-language lists, dozens of matches per query
            sw = dict(CONFIGS["cfg2"], dict_size=235887, name="words", variant="words")
        else:
            sw = workload_of(args, name)
        heavy = name in ("cfg4", "skewed")
        steps = max(3, min(args.steps, 5)) if heavy else max(5, min(args.steps, 50))
        r = measure(env, args, sw, steps, 2, cpu_baseline=False if args.no_cpu_baseline else 4.0, host_rate=False,
                    traffic_mode=args.traffic, parity_rows=None if heavy else sw["queries"])
        if r:
            roof = r["roofline"]
            sub_recs[name] = {"workload": r["config"]["workload"], "value": r["value"], "unit": "queries/s", "steps": r["steps"],
                              "ms_per_step": r["ms_per_step"], "kernel_ms_avg": roof["kernel_ms_avg"],
                              "frac": roof["frac"], "achieved_gbps": roof["achieved"], "traffic": roof["traffic"],
                              "traffic_source": roof["traffic_source"],
                              "model_bytes": roof.get("model_bytes"), "traffic_over_model": roof.get("traffic_over_model"),
                              "effective_frac": roof["effective_frac"], "effective_gbps": roof["effective_gbps"],
                              "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
                              "bit_exact": (r.get("parity_vs_oracle") or {}).get("bit_exact"),
                              "checked_queries": (r.get("parity_vs_oracle") or {}).get("checked_queries"),
                              "cpu_baseline": r["cpu_baseline"], "results_per_query": r["config"]["results_per_query"],
                              "pipeline": r["config"]["pipeline"], "kernel_ms_stdev": roof["kernel_ms_stdev"],
                              "dominant_kernel": roof.get("dominant_kernel"), "dominant_ms": roof.get("dominant_ms"), "dominant_frac": roof.get("dominant_frac"),
                              "kernels_ms": {kk[:-3]: vv for kk, vv in roof.items() if kk.endswith("_ms") and kk not in ("dominant_ms",) and not kk.startswith("kernel_")}}
            env.log("[%s] sub-record done in %.0fs" % (name, time.time() - t0))
    if env.rank == 0:
        k = w["topk"]
        metric_name = "fuzzy queries/sec (k=%d, %s≥%.2g) on %s-string dict" % (k, w["metric"].capitalize(), w["similarity"], _human(w["dict_size"]))
        try:      # the headline workload carries BASELINE.json's metric string verbatim (its "HBM GB/s fraction" half is `roofline.frac`)
            base_metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
            if base_metric.startswith(metric_name) and w["ngram"] == 3 and w["variant"] == "uniform":
                metric_name = base_metric
        except (OSError, ValueError, KeyError):
            pass
        out = {"metric": metric_name, "value": rec["value"], "unit": rec["unit"], "n_gpus": rec["n_gpus"], "steps": rec["steps"],
               "warmup": rec["warmup"], "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u32 (posting/counter work) + f64 (final score)", "data": "synthetic",
               "config": rec["config"], "roofline": rec["roofline"], "cpu_baseline": rec["cpu_baseline"]}
        for extra in ("setup_settle_calls", "host_buffers", "host_buffers_pipelined", "replicas_mode", "parity_vs_oracle"):
            if extra in rec:
                out[extra] = rec[extra]
        if sub_recs:
            out["configs"] = sub_recs
        print(json.dumps(out), flush=True)
    if env.world > 1:
        env.dist.destroy_process_group()


def main_replicas(env, args, w):
    """--mode replicas: the whole run is the single-process replica path; `value` is its (PCIe-inclusive) rate."""
    import numpy as np
    from suggest_amd import IndexDescription, NGramIndex, synth
    n_gpus = args.gpus
    have = env.torch.cuda.device_count()
    if have < n_gpus and not env.single_device:
        raise SystemExit("bench.py --mode replicas --gpus %d: only %d GPU(s) visible" % (n_gpus, have))
    desc_kw = dict(synth.DESCRIPTION, ngram_size=w["ngram"])
    k, n_q, n_b = w["topk"], w["queries"], max(1, args.batches)
    blob, offs = get_dict(w["dict_size"], w["variant"])
    batches = [synth.make_queries(n_q, blob, offs, seed=2, start=b * n_q) for b in range(n_b)]
    index = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**desc_kw), device=0, build=args.build)
    st = index.stats()
    ref = [index.suggest_batch(blob=qb, offs=qo, metric=w["metric"], similarity=w["similarity"], k=k) for qb, qo in batches]
    ids0 = [r[0] for r in ref]
    cnt0 = [r[2] for r in ref]
    rep = replicas_measure(env, args, index, w, batches, ids0, cnt0, n_gpus, args.steps, args.warmup)
    out = {"metric": "fuzzy queries/sec (k=%d, %s≥%.2g) on %s-string dict; single-process replicas (sg_suggest_batch_multi), PCIe-inclusive"
                     % (k, w["metric"].capitalize(), w["similarity"], _human(w["dict_size"])),
           "value": rep["value"], "unit": "queries/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32 (posting/counter work) + f64 (final score)", "data": "synthetic",
           "config": {"workload": "%s synthetic strings, q=%d, %s>=%.2g, k=%d, %d x %d-query host batch per call"
                                  % (_human(w["dict_size"]), w["ngram"], w["metric"], w["similarity"], k, n_gpus, n_q),
                      "baseline_config": w["name"], "parallelism": "ONE process, sg_index_replicate over %s, sg_suggest_batch_multi" % (rep["devices"],),
                      "pcie_inclusive": True,
                      "index": {"postings": st["n_postings"], "lists": st["n_lists"], "terms": st["n_terms"]}},
           "roofline": None, "cpu_baseline": None, "replicas_mode": rep}
    print(json.dumps(out), flush=True)


SEARCH_KERNELS = (("stream", "sg_stream_kernel"), ("plan", "sg_plan"), ("verify", "sg_verify_kernel"), ("fused", "sg_search_kernel_t<false, false,"),
                  ("parts", "sg_search_kernel_t<true, false,"), ("tokenise", "sg_terms_kernel"), ("order", "query_order_"), ("long", "sg_long_kernel"))


def _kernel_kind(name):
    for kind, pat in SEARCH_KERNELS:
        if pat in name:
            return kind
    return None


def _live_traffic(args, w, log):
    """Memory traffic per call of sg_suggest_batch_device, measured now: this script again, as a child under `rocprofv3 --pmc
    FETCH_SIZE --kernel-trace` (PMC counters cannot be read from inside a process), a few steps of the same workload.
    bytes = FETCH_SIZE [KB] x 1024 x 2 = 128-byte lines fetched past the L2 (profiles/r05_membench.txt: equal to TCC_EA0_RDREQ x 128
    for random segments of 256 B ... 4 KiB).  Every kernel of the call is counted (plan / stream / verify of the pipeline, the fused
    kernel and what it leaves to the parts launch, tokeniser, ordering), each with its average duration.
    -> (bytes, source, per_kernel) or (None, reason, None)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rocprof:
        return None, "rocprofv3 not found", None
    steps, warm = (60, 10) if w["name"] not in ("cfg4", "skewed") else (2, 1)      # (light workloads: long enough for the clocks to settle, as in the timed region)
    tmps = []

    def child(flags):
        """this script as a child under rocprofv3 with `flags` -> (KB fetched, ns, dispatches) per kind of kernel"""
        tmp = tempfile.mkdtemp(prefix="sg_pmc_", dir="/tmp")
        tmps.append(tmp)
        cmd = [rocprof] + flags + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable,
               os.path.abspath(__file__), "--config", w["name"] if w["name"] in CONFIGS else "headline", "--dict-size", str(w["dict_size"]), "--queries", str(w["queries"]),
               "--ngram", str(w["ngram"]), "--metric", w["metric"], "--similarity", repr(w["similarity"]), "--topk", str(w["topk"]),
               "--batches", str(args.batches), "--dict-variant", w["variant"], "--build", args.build,
               "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--no-host-rate", "--traffic", "none", "--sub-configs", "none"]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=420)
        if r.returncode != 0:
            raise ValueError("rocprofv3 child exited %d: %s" % (r.returncode, (r.stderr or "")[-300:]))
        kb, ns, n, nt = {}, {}, {}, {}
        for path in sorted(glob.glob(tmp + "/**/*counter_collection.csv", recursive=True)):
            for row in csv.DictReader(open(path)):
                kind = _kernel_kind(row.get("Kernel_Name", ""))
                if kind and row.get("Counter_Name") == "FETCH_SIZE":
                    kb[kind] = kb.get(kind, 0.0) + float(row["Counter_Value"])
                    n[kind] = n.get(kind, 0) + 1
        spans = {}
        for path in sorted(glob.glob(tmp + "/**/*kernel_trace.csv", recursive=True)):
            for row in csv.DictReader(open(path)):
                kind = _kernel_kind(row.get("Kernel_Name", ""))
                if kind:
                    spans.setdefault(kind, []).append((float(row["Start_Timestamp"]), float(row["End_Timestamp"]) - float(row["Start_Timestamp"])))
        # durations: the later half of a kind's dispatches, in time order — the run's first calls (cold pages and TLBs, clocks still
        # rising, the parity pass's other batch sizes) are not what the timed region launches; ns = that mean x all dispatches
        for kind, v in spans.items():
            v.sort()
            late = [d for _, d in v[len(v) // 2:]]
            ns[kind] = sum(late) / len(late) * len(v)
            nt[kind] = len(v)
        return kb, ns, n, nt

    t0 = time.time()
    try:
        kb, ns, n, _ = child(["--pmc", "FETCH_SIZE"])
        # the kernel that carries the call: by time (a replica whose pipeline queries mostly come back — near-duplicate families — takes
        # the three launches for a few calls and the fused kernel for the next 64: both show up in one pass)
        main = max(("stream", "fused"), key=lambda kk: ns.get(kk, 0.0)) if (n.get("stream") or n.get("fused")) else None
        # a call = one sg_suggest_batch_device: it has one pair of ordering launches when the batch is ordered, else one main launch
        calls = n.get("order", 0) // 2 if n.get("order", 0) >= 2 else n.get(main, 0) if main else 0
        if not calls:
            return None, "no FETCH_SIZE rows for the search kernels in the child's counter CSV", None
        total_kb = sum(kb.values()) / calls
        per = {kind: {"launches_per_call": n[kind] / calls, "traffic_bytes_per_call": kb[kind] / calls * 2048.0,
                      "ms_per_call_under_profiler": ns.get(kind, 0.0) / calls * 1e-6} for kind in n}
        per["_dominant"] = main
        log("[%s] live PMC pass: FETCH_SIZE %.6g KB per call over %d calls (%.0fs); %s" %
            (w["name"], total_kb, calls, time.time() - t0, ", ".join("%s %.3f ms" % (k_, v["ms_per_call_under_profiler"]) for k_, v in per.items() if k_[0] != "_")))
        return total_kb * 2048.0, ("live: rocprofv3 --pmc FETCH_SIZE --kernel-trace child pass of this run, %d calls of this workload, every kernel of the call "
                                   "(%s): FETCH_SIZE %.6g KB x 1024 x 2 (128-byte lines past the L2; profiles/r05_membench.txt)" %
                                   (calls, " + ".join(k_ for k_ in per if k_[0] != "_"), total_kb)), per
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as exc:
        return None, "live PMC pass failed: %r" % (exc,), None
    finally:
        for tmp in tmps: shutil.rmtree(tmp, ignore_errors=True)


def _human(n):
    return "%dM" % (n // 1_000_000) if n % 1_000_000 == 0 else "%dk" % (n // 1000) if n % 1000 == 0 else str(n)


if __name__ == "__main__":
    main()
