"""Multi-replica dispatch, coalesced single-query callers, the N>1 control flow of bench.py — on a one-GPU box
(two replicas on GPU 0 exercise the same code as two GPUs, minus the overlap)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import oracle
from conftest import ROOT
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    from suggest_amd import NGramIndex, IndexDescription, synth
    blob, offs = synth.make_dict(200000, seed=11)
    qb, qo = synth.make_queries(5000, blob, offs, seed=12)
    gpu = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION))
    ora = oracle.OracleIndex(blob=blob, offs=offs, **synth.DESCRIPTION)
    return gpu, ora, qb, qo


def test_replicas_and_multi_dispatch(small):
    """sg_index_replicate + sg_suggest_batch_multi: contiguous slices per replica, rows in caller order"""
    gpu, ora, qb, qo = small
    assert gpu.replicas() == [0]
    gpu.replicate([0, 0, 0])
    assert gpu.replicas() == [0, 0, 0]
    gpu.upload(0)                                              # already resident: no new replica
    assert gpu.replicas() == [0, 0, 0]
    one = gpu.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.5, k=10)
    multi = gpu.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.5, k=10, multi=True)
    for a, b in zip(one, multi):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert_same(multi, ora.suggest_batch(qb, qo, "jaccard", 0.5, 10))
    a1 = gpu.autocomplete_batch(blob=qb, offs=qo, limit=5)
    a2 = gpu.autocomplete_batch(blob=qb, offs=qo, limit=5, multi=True)
    assert np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1])
    tiny = gpu.suggest_batch(blob=qb[:int(qo[3])], offs=qo[:4], metric="cosine", similarity=0.5, k=3, multi=True)   # fewer queries than replicas
    assert_same(tiny, ora.suggest_batch(qb[:int(qo[3])], qo[:4], "cosine", 0.5, 3))
    # a ticket per replica from one thread (sg_suggest_submit_on): every replica answers like the primary; no fourth one
    n_q = len(qo) - 1
    outs = [(np.zeros((n_q, 10), np.uint32), np.zeros((n_q, 10), np.float64), np.zeros(n_q, np.uint32)) for _ in range(3)]
    tickets = [gpu.suggest_submit(qb, qo, "jaccard", 0.5, 10, *outs[r], replica=r) for r in range(3)]
    for t in tickets:
        t.wait()
    for o in outs:
        for a, b in zip(one, o):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    with pytest.raises(Exception):
        gpu.suggest_submit(qb, qo, "jaccard", 0.5, 10, *outs[0], replica=3)
    # [r6] every array of every replica is resident on the replica's own device (hipPointerGetAttributes), and the replicas do not
    # share arrays: what a cross-device mix-up would look like can be checked without a second GPU
    import ctypes
    from suggest_amd import _lib
    seen = set()
    for r in range(3):
        out = (ctypes.c_int32 * 8)()
        _lib.check(_lib.lib().sg_debug_replica_devices(gpu._h, r, out))
        assert list(out)[1:] == [out[0]] * 7 and out[0] == gpu.replicas()[r], list(out)
    out = (ctypes.c_int32 * 8)()
    assert _lib.lib().sg_debug_replica_devices(gpu._h, 3, out) != 0


def test_coalesced_single_query_callers(small):
    """Suggest / Autocomplete one query per call from many threads (service_test.go:36-79): sg_suggest_one coalesces the
    concurrent callers into shared launches; every caller gets exactly the rows of the batch call"""
    from suggest_amd import synth
    gpu, ora, qb, qo = small
    queries = synth.unpack(qb, qo)[:1536]
    ids, sc, cnt = gpu.suggest_batch(queries, "jaccard", 0.5, 10)
    a_ids, a_cnt = gpu.autocomplete_batch([q[:4] for q in queries], limit=7)
    errors = []

    def worker(t, n_thr):
        try:
            for i in range(t, len(queries), n_thr):
                if i % 3 == 2:                                 # a third of the traffic uses other parameters (its own batches)
                    got = gpu.autocomplete(queries[i][:4], 7)
                    assert got == a_ids[i, :a_cnt[i]].tolist(), (i, got)
                else:
                    got = gpu.suggest(queries[i], 0.5, "jaccard", 10)
                    exp = [(int(ids[i, j]), float(sc[i, j])) for j in range(int(cnt[i]))]
                    assert got == exp, (i, got, exp)
        except Exception as exc:       # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(t, 48)) for t in range(48)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


def test_host_buffer_callers_during_index_churn(small):
    """Threads calling the host-buffer batch entry point (each on its own stream) while the main thread builds, uploads
    and frees other indexes on the same device: every call's rows are its own.  (The detector of the round-2 pool problem
    is test_cpp_mirror's stress test — under the GIL this one did not reproduce it; it covers the Python entry points.)"""
    from suggest_amd import NGramIndex, IndexDescription, synth
    gpu, ora, qb, qo = small
    queries = synth.unpack(qb, qo)[:2048]
    ids, sc, cnt = gpu.suggest_batch(queries, "jaccard", 0.5, 10)
    stop = threading.Event()
    errors, calls = [], [0]

    def worker(t):
        rng = np.random.default_rng(t)
        try:
            while not stop.is_set():
                n = int(rng.integers(1, 9)); i0 = int(rng.integers(0, len(queries) - n))
                g_ids, g_sc, g_cnt = gpu.suggest_batch(queries[i0:i0 + n], "jaccard", 0.5, 10)
                assert np.array_equal(g_cnt, cnt[i0:i0 + n]), (i0, n, g_cnt, cnt[i0:i0 + n])
                for r in range(n):
                    m = int(g_cnt[r])
                    assert np.array_equal(g_ids[r, :m], ids[i0 + r, :m]) and np.array_equal(g_sc[r, :m], sc[i0 + r, :m]), (i0, r)
                calls[0] += 1
        except Exception as exc:       # noqa: BLE001
            errors.append(exc)
            stop.set()

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    blob, offs = synth.make_dict(20000, seed=5)
    for it in range(40):
        if stop.is_set():
            break
        other = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION), build="host" if it % 2 else "device")
        other.suggest_batch(queries[:4], "jaccard", 0.5, 10)
        other.close()
    stop.set()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    assert calls[0] > 100


def test_heaviest_first_query_order_changes_nothing(small):
    """Batches of >= 8192 queries run in a counting-sort order by length (longest first; autocomplete: shortest first —
    query_order_kernels in engine.hip); rows are written by query index, so the output must be the same byte for byte,
    including empty queries and queries longer than the sort's 255-byte last bin."""
    from suggest_amd import synth
    gpu, ora, qb, qo = small
    base = synth.unpack(qb, qo)
    queries = (base * 3)[:12000]
    queries[17] = b""
    queries[4242] = b"x" * 300
    queries[9001] = base[5] * 9
    qb2, qo2 = oracle.pack_strings(queries)
    out = {}
    for order in (0, 1):
        gpu.tune(SG_ORDER=order)
        out[order] = (gpu.suggest_batch(blob=qb2, offs=qo2, metric="cosine", similarity=0.45, k=7),
                      gpu.autocomplete_batch(blob=qb2, offs=qo2, limit=6))
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    # ... and so must the tokeniser as a launch of its own (sg_terms_kernel: the default for batches of >= 2048 queries,
    # i.e. what ran above) against the search kernel tokenising each query itself
    gpu.tune(SG_PRETOK=0)
    fused = (gpu.suggest_batch(blob=qb2, offs=qo2, metric="cosine", similarity=0.45, k=7), gpu.autocomplete_batch(blob=qb2, offs=qo2, limit=6))
    gpu.tune(SG_PRETOK=2048)
    for a, b in zip(fused[0] + fused[1], out[1][0] + out[1][1]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    want = list(ora.suggest_batch(qb2, qo2, "cosine", 0.45, 7)[:3])
    assert_same(out[1][0], want)                     # (query 4242: 300 runes, past the wavefront kernel's 144 — the long-query kernel's)


def test_single_query_load_generator():
    """tools/single_query_load (C++, through the C ABI): N threads of blocking sg_suggest_one calls; checks every answer
    against the batch call and reports the rate (the number itself is not asserted here: DESIGN.md)"""
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "single_query_load")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    out = subprocess.run([exe, "100000", "64", "1.0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["mismatches"] == 0 and rec["queries"] > 0


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py's N>1 path (process per rank, barriers, max over ranks, per-rank batches, the result gather) driven by
    torch.distributed.run with both ranks on GPU 0 and gloo as the backend (RCCL wants one rank per GPU)"""
    env = dict(os.environ, SG_BENCH_SINGLE_DEVICE="1", SG_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--dict-size", "200000", "--queries", "4096", "--build", "host"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["config"]["rccl_gather_check"] is True
    assert rec["value"] > 0 and rec["roofline"]["effective_frac"] > 0
    # (no PMC pass at N > 1 and no committed traffic figure for this toy workload: the physical fraction stays null
    #  rather than being filled with the algorithmic rate)
    assert rec["roofline"]["frac"] is None or 0 < rec["roofline"]["frac"] <= 1
    rm = rec["replicas_mode"]                       # rank 0's single-process leg: sg_index_replicate + sg_suggest_batch_multi
    assert rm["n_gpus"] == 2 and rm["rows_equal_device_run"] is True and rm["value"] > 0
    # [r5] every rank checks a sample of its OWN rows against the CPU oracle (nobody watches an 8-GPU run)
    pr = rec["config"]["parity_per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["checked_queries"] > 0 and x["bit_exact"] is True for x in pr), pr


def test_bench_two_ranks_attempt_rccl_and_say_what_happened():
    """[r6] the same with the `nccl` backend ATTEMPTED: on a box with two GPUs RCCL gathers the rows (`rccl_status` "ok ..."); with both
    ranks on GPU 0 RCCL refuses (one rank per device) and the line must say so — `rccl_status` "refused ...", `rccl_gather_check`
    false — instead of quietly measuring over gloo.  Either way the timed region (no collective) and the per-rank parity stand."""
    import torch
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", NCCL_DEBUG="WARN")
    env.pop("SG_BENCH_BACKEND", None)
    if not two:
        env["SG_BENCH_SINGLE_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--dict-size", "200000", "--queries", "4096", "--build", "host"]
    import signal
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        so, se = proc.communicate(timeout=600)
    except subprocess.TimeoutExpired:          # (a wedged collective must not outlive the test: the whole process group goes)
        os.killpg(proc.pid, signal.SIGKILL)
        so, se = proc.communicate()
        raise AssertionError("bench.py --gpus 2 (nccl attempted) did not finish: " + so[-1000:] + se[-3000:])
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert lines, so[-2000:] + se[-4000:]            # (the line is printed before the process group is torn down: a teardown error after it is RCCL's, not the measurement's)
    rec = json.loads(lines[-1])
    st = rec["config"]["rccl_status"]
    assert rec["n_gpus"] == 2 and rec["value"] > 0
    if two:
        assert st.startswith("ok") and rec["config"]["rccl_gather_check"] is True, st
        assert rec["config"]["rccl_ranks"]["ranks"] == [0, 1]
    else:
        assert st.startswith("refused") and rec["config"]["rccl_gather_check"] is False, st
    pr = rec["config"]["parity_per_rank"]
    assert all(x["checked_queries"] > 0 and x["bit_exact"] is True for x in pr), pr


def test_bench_self_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver starts the scaling runs) must not measure
    one GPU and call it two: it re-executes itself under torch.distributed.run.  Both ranks on GPU 0, gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SG_BENCH_SINGLE_DEVICE="1", SG_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--dict-size", "100000", "--queries", "2048"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2
    assert rec["replicas_mode"]["n_gpus"] == 2 and rec["replicas_mode"]["rows_equal_device_run"] is True


def test_bench_cfg4_preset_two_ranks_carries_a_roofline():
    """`python bench.py --config cfg4 --gpus N` is BASELINE config 4's line (q=2, Dice, 16 384 queries per GPU, batch sharded
    over the node): preset + self-spawn, and at N > 1 — where no PMC pass can run — roofline.traffic / frac come from the
    committed N = 1 measurement of the same workload, per GPU, labelled as such; per-rank rates and the cost of the optional
    result gather are reported.  Two ranks on GPU 0, gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SG_BENCH_SINGLE_DEVICE="1", SG_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg4", "--gpus", "2", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["baseline_config"] == "cfg4" and "q=2" in rec["config"]["workload"]
    roof = rec["roofline"]
    assert roof["traffic"] and roof["frac"] and 0 < roof["frac"] < 1 and "N=1 PMC figure" in roof["traffic_source"]
    pr = rec["config"]["per_rank"]
    assert pr and pr["min"] > 0 and pr["min"] <= pr["mean"] <= pr["max"]
    assert abs(rec["value"] - 2 * 16384 * 2 / (rec["ms_per_step"] * 2 * 1e-3)) / rec["value"] < 1e-6      # whole-job rate = all ranks' queries / max time
    if rec["config"]["rccl_gather_check"]:
        assert rec["config"]["gather_ms"] and rec["config"]["gather_ms"] > 0


def test_bench_refuses_more_gpus_than_visible():
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SG_BENCH_SINGLE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_replicas_mode_single_process():
    """--mode replicas: ONE process, two replicas (both on GPU 0 here), sg_suggest_batch_multi with a worker thread each"""
    env = dict(os.environ, SG_BENCH_SINGLE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "replicas", "--steps", "3", "--warmup", "1",
                          "--dict-size", "100000", "--queries", "2048"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["pcie_inclusive"] is True and rec["replicas_mode"]["rows_equal_device_run"] is True
    # ... and the pinned leg: ONE host thread, a ticket per replica and step (sg_suggest_submit_on), two steps in flight
    piped = rec["replicas_mode"]["pipelined"]
    assert piped["rows_equal_device_run"] is True and piped["value"] > 0


def test_replica_arrays_live_on_their_own_devices():
    """[r6] with two or more GPUs: a replica per device, every array of replica r on device r (sg_debug_replica_devices)"""
    import ctypes
    import torch
    from suggest_amd import _lib, NGramIndex, IndexDescription, synth
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs two GPUs")
    blob, offs = synth.make_dict(20000, seed=3)
    gpu = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION))
    gpu.replicate(list(range(n)))
    for r, dev in enumerate(gpu.replicas()):
        out = (ctypes.c_int32 * 8)()
        _lib.check(_lib.lib().sg_debug_replica_devices(gpu._h, r, out))
        assert list(out) == [dev] * 8, list(out)


def test_multi_dispatch_over_distinct_devices():
    """sg_suggest_batch_multi over two DIFFERENT GPUs (the per-thread context of the first slice used to dangle when the
    second device's context was created): needs a box with at least two GPUs — the driver's 8-GPU node has them."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    from suggest_amd import IndexDescription, NGramIndex, synth
    blob, offs = synth.make_dict(100000, seed=21)
    qb, qo = synth.make_queries(3000, blob, offs, seed=22)
    ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION), device=0)
    ix.replicate([0, 1])
    one = ix.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.4, k=7)
    for _ in range(3):
        many = ix.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.4, k=7, multi=True)
        for a, b in zip(one, many):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_reference_built_index_with_dropped_repeats(tmp_path, golden_dir):
    """A reference-built index whose roaring-coded list (raw length > 256) dropped a document's repeated postings
    (codec.go:39-51, bitmap_posting_list.go:99): the loader keeps only the raw-length surplus for such a list — uploading
    it used to write outside the repeated-documents bitmap.  Files are laid down in the reference's format from the
    oracle's own lists."""
    import refindex
    from suggest_amd import IndexDescription, NGramIndex
    desc = dict(ngram_size=3, wrap=("$", "$"), pad="$", alphabet=("english", "$"))
    letters = "abcdefghijklmnopqrstuvwxyz"
    docs = [("ab-ab.%s%s" % (letters[i // 26], letters[i % 26])).encode() for i in range(300)]      # "$ab" and "ab$" twice each
    docs += [("xy %s%s zz" % (letters[i % 26], letters[i // 26])).encode() for i in range(120)] + [b"ab-ab", b"abab", b"ab ab ab"]
    ora = oracle.OracleIndex(docs, **desc)
    lists = ora.lists()
    assert any(raw > 256 and raw > len(post) for raw, post in lists.values())          # the case exists
    hd, dl = str(tmp_path / "t.hd"), str(tmp_path / "t.dl")
    refindex.write_index(hd, dl, ora.n_segments, lists, os.path.join(golden_dir, "db", "words_subset.hd"))
    gpu = NGramIndex.from_reference_files(hd, dl, IndexDescription(**desc))
    queries = docs[::7] + [b"ab-ab.", b"ab-ab.zz", b"abab", b"ab ab", b"xy ab zz"]
    qb, qo = oracle.pack_strings(queries)
    for metric, alpha, k in (("jaccard", 0.3, 10), ("cosine", 0.5, 400), ("dice", 0.4, 5)):
        assert_same(gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k),
                    ora.suggest_batch(qb, qo, metric, alpha, k), queries)
    from suggest_amd import NGramIndex as NI
    built = NI(docs, IndexDescription(**desc))                  # the same dictionary through the host builder
    for metric, alpha, k in (("jaccard", 0.3, 10), ("cosine", 0.5, 400)):
        assert_same(built.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k),
                    ora.suggest_batch(qb, qo, metric, alpha, k), queries)


def test_device_built_store_stays_resident_and_matches_host_build():
    """sg_index_build_device leaves the posting store in HBM and the first upload adopts it: same results as the host build"""
    from suggest_amd import NGramIndex, IndexDescription, synth
    blob, offs = synth.make_dict(300000, seed=21, families=3)
    qb, qo = synth.make_queries(3000, blob, offs, seed=22)
    d = IndexDescription(**synth.DESCRIPTION)
    dev = NGramIndex(blob=blob, offs=offs, description=d, build="device")
    host = NGramIndex(blob=blob, offs=offs, description=d, build="host")
    assert dev.digest() == host.digest()
    assert dev.stats()["device_bytes"] == host.stats()["device_bytes"]
    for metric, alpha, k in (("jaccard", 0.5, 10), ("cosine", 0.4, 20)):
        a = dev.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k)
        b = host.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k)
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8))


def test_device_pairsort_on_the_hand_traced_vectors(reference_tests):
    """the device's Go 1.14 sort.Sort (PairSort, engine.hip) against tests/golden/go_sort_small_traces.txt"""
    import ctypes as C
    from suggest_amd import _lib
    L = _lib.lib()
    for v in reference_tests["go_sort_small"]["vectors"] + reference_tests["go_sort_mid"]["vectors"]:      # <= 12, and 13 .. 40 elements
        keys = np.array(v["keys"], dtype=np.uint32)
        out = np.zeros(len(keys), dtype=np.uint32)
        _lib.check(L.sg_debug_pairsort(0, keys.ctypes.data, len(keys), out.ctypes.data))
        assert out.tolist() == v["perm"], v


def test_device_pairsort_matches_the_other_two_go_sort_restatements():
    """differential fuzz of the device's Go 1.14 sort.Sort (PairSort, engine.hip) against the oracle's and tests/gosort.py"""
    import ctypes as C
    import random
    import gosort
    from suggest_amd import _lib
    L = _lib.lib()
    rng = random.Random(2014)
    for trial in range(400):
        n = rng.choice([1, 2, 5, 12, 13, 14, 30, 41, 60, 100, 128])
        spread = rng.choice([1, 2, 3, 5, 17, 1000])
        keys = np.array([rng.randrange(spread) for _ in range(n)], dtype=np.uint32)
        if trial % 7 == 0:
            keys.sort()
        out = np.zeros(n, dtype=np.uint32)
        _lib.check(L.sg_debug_pairsort(0, keys.ctypes.data, n, out.ctypes.data))
        assert out.tolist() == gosort.go_sort(keys.tolist()) == oracle.go_sort(keys), (trial, keys.tolist())


def test_async_submit_wait_pinned_and_pageable(small):
    """sg_suggest_submit / sg_ticket_wait: several tickets in flight (copy in, launch and copy out of different batches
    overlap on three streams), pinned buffers (no staging) and pageable ones (staged), waited for in another order and
    from another thread than they were submitted in; every row equals the synchronous call's"""
    from suggest_amd.index import pinned_array
    gpu, ora, qb, qo = small
    k = 10
    cuts = [0, 700, 701, 2500, 5000]                       # four batches of different sizes, the second a single query
    ref = gpu.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.5, k=k)
    tickets, outs = [], []
    for b in range(4):
        lo, hi = cuts[b], cuts[b + 1]
        n = hi - lo
        blob = qb[int(qo[lo]):int(qo[hi])]
        offs = (qo[lo:hi + 1] - qo[lo]).astype(np.uint64)
        if b % 2 == 0:                                      # pinned: DMA straight from / to the caller's arrays
            pb = pinned_array((max(blob.size, 1),), np.uint8)[:blob.size]; pb[:] = blob
            po = pinned_array((n + 1,), np.uint64); po[:] = offs
            ids, sc, cnt = pinned_array((n, k), np.uint32), pinned_array((n, k), np.float64), pinned_array((n,), np.uint32)
            ids[:] = 0xDEAD; sc[:] = -1; cnt[:] = 77
            blob, offs = pb, po
        else:                                               # pageable, and offsets that do not start at zero (a slice of a larger batch)
            blob, offs = qb, qo[lo:hi + 1].copy()
            ids, sc, cnt = np.full((n, k), 0xDEAD, np.uint32), np.full((n, k), -1.0), np.full((n,), 77, np.uint32)
        tickets.append(gpu.suggest_submit(blob, offs, "jaccard", 0.5, k, ids, sc, cnt))
        outs.append((ids, sc, cnt))
    errs = []

    def waiter(order):
        try:
            for i in order:
                tickets[i].wait()
        except Exception as e:   # noqa
            errs.append(e)
    th = threading.Thread(target=waiter, args=([3, 1],))
    th.start()
    waiter([2, 0])
    th.join()
    assert not errs, errs
    for b in range(4):
        lo, hi = cuts[b], cuts[b + 1]
        ids, sc, cnt = outs[b]
        assert np.array_equal(cnt, ref[2][lo:hi])
        assert np.array_equal(ids, ref[0][lo:hi])
        assert np.array_equal(sc.view(np.uint64), ref[1][lo:hi].view(np.uint64))
    with pytest.raises(ValueError):
        tickets[0].wait()                                   # a ticket is waited for once
    # autocomplete through the same machinery, with a first_doc; an empty batch; more tickets than slots
    a_ref = gpu.autocomplete_batch(blob=qb[:int(qo[300])], offs=qo[:301], limit=5)
    a_ids, a_cnt = np.zeros((300, 5), np.uint32), np.zeros(300, np.uint32)
    gpu.autocomplete_submit(qb[:int(qo[300])], qo[:301].copy(), 5, a_ids, a_cnt).wait()
    assert np.array_equal(a_cnt, a_ref[1]) and np.array_equal(a_ids, a_ref[0])
    gpu.suggest_submit(np.zeros(0, np.uint8), np.zeros(1, np.uint64), "jaccard", 0.5, k, np.zeros((0, k), np.uint32), np.zeros((0, k)), np.zeros(0, np.uint32)).wait()
    from suggest_amd._lib import SuggestHipError
    many = []
    with pytest.raises(SuggestHipError, match="tickets in flight"):
        for _ in range(9):
            many.append(gpu.suggest_submit(qb[:int(qo[10])], qo[:11].copy(), "jaccard", 0.5, k, np.zeros((10, k), np.uint32), np.zeros((10, k)), np.zeros(10, np.uint32)))
    assert len(many) == 8
    for t in many:
        t.wait()
    t = gpu.suggest_submit(qb[:int(qo[10])], qo[:11].copy(), "jaccard", 0.5, k, np.zeros((10, k), np.uint32), np.zeros((10, k)), np.zeros(10, np.uint32))
    t.wait()                                                 # the slots are free again
