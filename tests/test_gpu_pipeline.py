"""[r5] The three-launch pipeline of ordinary fuzzy batches (plan -> stream -> verify, suggest_amd/csrc/pipeline.inc) against the
CPU oracle: bit-exact ids, order and scores through the C ABI, whatever its knobs, and whichever queries it hands back to the
fused kernel (pkg/suggest/suggester.go:46-131, pkg/merger/cp_merge.go:19-120 are what both must reproduce)."""
import numpy as np
import pytest

import oracle
from test_gpu_parity import METRICS, assert_same

pytestmark = pytest.mark.gpu


def _pair(n_docs, n_q, desc=None, seed=1, **variant):
    from suggest_amd import NGramIndex, IndexDescription, synth
    d = dict(synth.DESCRIPTION, **(desc or {}))
    blob, offs = synth.make_dict(n_docs, seed=seed, **variant)
    qb, qo = synth.make_queries(n_q, blob, offs, seed=seed + 1)
    gpu = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**d))
    ora = oracle.OracleIndex(blob=blob, offs=offs, **d)
    return gpu, ora, qb, qo


@pytest.fixture(scope="module")
def synth_pipe():
    gpu, ora, qb, qo = _pair(60000, 4096)
    # every eligible launch, and the stream workgroup of the large dictionaries (eight wavefronts, 2^13 counters, 8 KB of
    # descriptors): a dictionary of this size gets two wavefronts on 2^11 counters by default (tune_choice, capi.inc)
    gpu.tune(SG_PIPE=1, SG_PIPE_NW=8, SG_PIPE_LOG2_CNT=13, SG_PIPE_DT_BYTES=8192)
    return gpu, ora, qb, qo


def _delta(gpu, fn):
    s0 = gpu.pipe_stats()
    out = fn()
    s1 = gpu.pipe_stats()
    return out, {k: s1[k] - s0[k] for k in s1}


@pytest.mark.parametrize("metric,alpha", METRICS)
@pytest.mark.parametrize("k", [1, 10, 64])
def test_pipeline_parity(synth_pipe, metric, alpha, k):
    gpu, ora, qb, qo = synth_pipe
    assert_same(gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k), ora.suggest_batch(qb, qo, metric, alpha, k))


@pytest.mark.parametrize("knobs", [dict(SG_PIPE_SUB=3), dict(SG_PIPE_SUB=5), dict(SG_PIPE_NW=4), dict(SG_PIPE_NW=4, SG_PIPE_SUB=3),
                                   dict(SG_PIPE_LOG2_CNT=9), dict(SG_PIPE_LOG2_CNT=11, SG_T_FLOOR=4), dict(SG_FILTER_LEVEL=7, SG_T_FLOOR=2),
                                   dict(SG_FILTER_LEVEL=0), dict(SG_ORDER=0)])
def test_rows_do_not_depend_on_the_pipeline_knobs(synth_pipe, knobs):
    gpu, ora, qb, qo = synth_pipe
    base = dict(SG_PIPE_SUB=4, SG_PIPE_NW=8, SG_PIPE_LOG2_CNT=13, SG_T_FLOOR=8, SG_FILTER_LEVEL=4, SG_ORDER=1)
    try:
        gpu.tune(**knobs)
        for metric, alpha, k in (("jaccard", 0.5, 10), ("cosine", 0.4, 20), ("dice", 0.7, 5)):
            assert_same(gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k), ora.suggest_batch(qb, qo, metric, alpha, k))
    finally:
        gpu.tune(**base)


@pytest.mark.parametrize("shape", [(2, 11, 2048), (4, 12, 4096), (2, 10, 1024), (2, 11, 1024), (4, 9, 2048), (8, 13, 32768)])
def test_rows_do_not_depend_on_the_stream_workgroup(synth_pipe, shape):
    """the stream workgroups tune_choice picks for smaller dictionaries (and smaller ones still: more queries go back to the
    fused kernel for want of descriptors)"""
    gpu, ora, qb, qo = synth_pipe
    try:
        gpu.tune(SG_PIPE_NW=shape[0], SG_PIPE_LOG2_CNT=shape[1], SG_PIPE_DT_BYTES=shape[2])
        for metric, alpha, k in (("jaccard", 0.5, 10), ("cosine", 0.4, 20), ("overlap", 0.9, 5)):
            assert_same(gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k), ora.suggest_batch(qb, qo, metric, alpha, k))
    finally:
        gpu.tune(SG_PIPE_NW=8, SG_PIPE_LOG2_CNT=13, SG_PIPE_DT_BYTES=8192)


def test_wide_sub_row_descriptors(synth_pipe):
    """[r6] stores of 2^26 chunks (1 GiB) and more take 8-byte sub-row descriptors and 64-bit row addresses (pipeline.inc, kWide);
    SG_PIPE_WIDE forces them on a small store: the rows must not tell, and the queries still take the three launches"""
    gpu, ora, qb, qo = synth_pipe
    try:
        gpu.tune(SG_PIPE_WIDE=1)
        for nw, cnt, dt in ((8, 13, 8192), (2, 11, 2048), (4, 12, 4096)):
            gpu.tune(SG_PIPE_NW=nw, SG_PIPE_LOG2_CNT=cnt, SG_PIPE_DT_BYTES=dt)
            for metric, alpha, k in (("jaccard", 0.5, 10), ("cosine", 0.4, 20)):
                res, d = _delta(gpu, lambda: gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k))
                assert_same(res, ora.suggest_batch(qb, qo, metric, alpha, k))
                assert d["queries"] == 4096 and d["unplanned"] < 410, d
    finally:
        gpu.tune(SG_PIPE_WIDE=0, SG_PIPE_NW=8, SG_PIPE_LOG2_CNT=13, SG_PIPE_DT_BYTES=8192)


@pytest.mark.parametrize("k", [65, 100, 500])
def test_top_k_above_the_lds_rows_takes_the_pipeline(k):
    """[r6] k > 64: the verify launch keeps its top-k rows in HBM (as the fused kernel does) instead of handing the launch back —
    near-duplicate families and a low similarity, so that rows really hold more than 64 entries"""
    gpu, ora, qb, qo = _pair(40000, 2048, seed=9, families=79)
    gpu.tune(SG_PIPE=1, SG_PIPE_CAND_CAP=4096, SG_TIGHTEN=0)        # (threshold tightening is the fused kernel's: a launch with it on never takes the pipeline)
    res, d = _delta(gpu, lambda: gpu.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.2, k=k))
    want = ora.suggest_batch(qb, qo, "jaccard", 0.2, k)
    assert_same(res, want)
    assert d["queries"] == 2048, d
    assert int(np.minimum(want[2], k).max()) > 64, int(want[2].max())      # (rows that really hold more than the LDS rows' 64 entries)


def test_records_beyond_the_head_take_overflow_blocks(synth_pipe):
    """[r6] a stream record's head holds 16 groups and 174 lists; 2^9 counters and a low similarity make queries of dozens of
    groups: their tails live in overflow blocks (a pool of n / 8: some queries find it empty and go to the fused kernel)"""
    gpu, ora, qb, qo = synth_pipe
    try:
        gpu.tune(SG_PIPE_LOG2_CNT=9, SG_T_FLOOR=2, SG_FILTER_LEVEL=0)
        v0 = gpu.pipe_volumes()
        for metric, alpha, k in (("cosine", 0.3, 10), ("jaccard", 0.3, 5)):
            assert_same(gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k), ora.suggest_batch(qb, qo, metric, alpha, k))
        v1 = gpu.pipe_volumes()
        assert v1["sampled"] > v0["sampled"] and (v1["groups"] - v0["groups"]) / (v1["sampled"] - v0["sampled"]) > 8, (v0, v1)
    finally:
        gpu.tune(SG_PIPE_LOG2_CNT=13, SG_T_FLOOR=8, SG_FILTER_LEVEL=4)


def test_candidate_overflow_goes_to_the_fused_kernel(synth_pipe):
    """two candidate slots per query: nearly every query with a match overflows them and is answered by the fused kernel
    behind the three launches — the rows must not tell"""
    gpu, ora, qb, qo = synth_pipe
    try:
        gpu.tune(SG_PIPE_CAND_CAP=2)
        res, d = _delta(gpu, lambda: gpu.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.5, k=10))
    finally:
        gpu.tune(SG_PIPE_CAND_CAP=64)
    assert_same(res, ora.suggest_batch(qb, qo, "jaccard", 0.5, 10))
    assert d["overflow"] > 100, d


def test_plan_hands_back_what_it_cannot_express(synth_pipe):
    """the smallest counter array and the strictest filter table: single segments outnumber the counters — up to SG_PIPE_LOOSE
    times they are planned on the full array anyway (more flagged postings, same matches), beyond that and for queries above 64
    n-grams the plan lists the query for the fused kernel; the rows must not tell which"""
    from suggest_amd import pack_strings
    gpu, ora, qb, qo = synth_pipe
    qs = [qb[int(qo[i]):int(qo[i + 1])].tobytes() for i in range(2500)]
    long_ones = [(qs[i] + qs[i + 100] + qs[i + 200] + qs[i + 300] + qs[i + 400])[:75 + i % 25] for i in range(60)]   # 75 .. 99 bytes of distinct n-grams: above 64, below 128
    b2, o2 = pack_strings(qs + long_ones)
    try:
        gpu.tune(SG_PIPE_LOG2_CNT=9, SG_FILTER_LEVEL=0)
        res, d = _delta(gpu, lambda: gpu.suggest_batch(blob=b2, offs=o2, metric="cosine", similarity=0.3, k=10))
    finally:
        gpu.tune(SG_PIPE_LOG2_CNT=13, SG_FILTER_LEVEL=4)
    assert_same(res, ora.suggest_batch(b2, o2, "cosine", 0.3, 10))
    assert d["unplanned"] >= len(long_ones), d


def test_documents_that_repeat_a_term_and_near_duplicates():
    """an alphabet without the digits: they normalise to the pad, documents repeat terms (secondary entries — the
    fused kernel's per-list view), and families of near-duplicates flag dozens of postings per query"""
    gpu, ora, qb, qo = _pair(40000, 4096, desc=dict(alphabet=("english", "$")), seed=11, families=3)
    gpu.tune(SG_PIPE=1)
    for metric, alpha, k in (("jaccard", 0.5, 10), ("cosine", 0.4, 5), ("dice", 0.6, 64)):
        res, d = _delta(gpu, lambda: gpu.suggest_batch(blob=qb, offs=qo, metric=metric, similarity=alpha, k=k))
        assert_same(res, ora.suggest_batch(qb, qo, metric, alpha, k))
    assert d["repeats"] + d["overflow"] > 0, d


def test_long_and_odd_queries_in_a_pipeline_batch(synth_pipe):
    """queries of every kind in one batch: empty, shorter than an n-gram, above 64 n-grams (the plan hands them back), above 128
    (sg_long_kernel), non-ASCII, spaces only"""
    from suggest_amd import pack_strings
    gpu, ora, qb, qo = synth_pipe
    base = [qb[int(qo[i]):int(qo[i + 1])].tobytes() for i in range(3000)]
    odd = [b"", b"a", b"  ", b"ab", "été naïve".encode(), b"x" * 70, b"abcdefghij" * 9, b"q" * 140, b"the quick brown fox " * 8,
           bytes(range(1, 60)), b"A1B2C3D4E5F6", b"\xff\xfe\xfd abc"]
    qs = []
    for i, q in enumerate(base):
        qs.append(q)
        if i % 250 == 0:
            qs.extend(odd)
    b2, o2 = pack_strings(qs)
    for metric, alpha in (("jaccard", 0.5), ("cosine", 0.3)):
        assert_same(gpu.suggest_batch(blob=b2, offs=o2, metric=metric, similarity=alpha, k=10), ora.suggest_batch(b2, o2, metric, alpha, 10))


def test_rows_do_not_depend_on_two_queries_per_plan_wavefront(synth_pipe):
    """[r6] sg_plan2_kernel (two queries per wavefront, plan2.inc) against sg_plan_kernel (SG_PLAN2=0) and the oracle: ordinary
    batches over the metrics, an odd number of queries (the last wavefront has one), and a batch where simple queries sit next to
    ones the half-wavefront path does not take — non-ASCII, more than 32 n-grams, text to trim, empty — in either half"""
    from suggest_amd import pack_strings
    gpu, ora, qb, qo = synth_pipe
    base = [qb[int(qo[i]):int(qo[i + 1])].tobytes() for i in range(2001)]
    odd = [b"", b"a", b" lead", b"trail ", "naïve été".encode(), b"abcdefghijklmnopqrstuvwxyz0123456789", b"x" * 31, b"y" * 32, b"z" * 33, b"ab", b"abc",
           b"\xff\xfe abc", b"A1B2C3D4E5F6G7H8"]
    mixed = []
    for i, q in enumerate(base[:1500]):
        mixed.append(q)
        if i % 7 == 3:
            mixed.append(odd[(i // 7) % len(odd)])         # (an odd one now in an even slot, now in an odd one)
    cases = [pack_strings(base), pack_strings(mixed)]
    try:
        for two in (1, 0):
            gpu.tune(SG_PLAN2=two, SG_ORDER=1, SG_PRETOK=1)     # (SG_PRETOK: batches of a few thousand queries take the three launches too)
            for b2, o2 in cases:
                for metric, alpha, k in (("jaccard", 0.5, 10), ("cosine", 0.4, 20), ("dice", 0.7, 5), ("overlap", 0.9, 5), ("exact", 1.0, 3)):
                    res, d = _delta(gpu, lambda: gpu.suggest_batch(blob=b2, offs=o2, metric=metric, similarity=alpha, k=k))
                    assert_same(res, ora.suggest_batch(b2, o2, metric, alpha, k))
                    assert d["queries"] == len(o2) - 1, d
            gpu.tune(SG_ORDER=0)                            # (the batch's own order: pairs of unequal length)
            b2, o2 = cases[1]
            assert_same(gpu.suggest_batch(blob=b2, offs=o2, metric="jaccard", similarity=0.5, k=10), ora.suggest_batch(b2, o2, "jaccard", 0.5, 10))
    finally:
        gpu.tune(SG_PLAN2=1, SG_ORDER=1, SG_PRETOK=2048)


def test_pipeline_with_a_tabulated_metric(synth_pipe):
    """an opaque metric.Metric as host-built tables (pkg/metric/metric.go:7-16) goes through the same three launches"""
    gpu, ora, qb, qo = synth_pipe
    tabs = gpu.metric_tables("jaccard", 0.6, 80)
    try:
        assert_same(gpu.suggest_batch(blob=qb, offs=qo, k=10, tables=tabs), ora.suggest_batch(qb, qo, "jaccard", 0.6, 10))
    finally:
        tabs.close()


def test_default_policy():
    """the default (SG_PIPE=2, the stream workgroup by the index's expected query volume): a dictionary below the one-counter-
    per-document size keeps the fused kernel, a larger one takes the pipeline with the small workgroup; the rows do not tell"""
    for n_docs in (8000, 120000):
        gpu, ora, qb, qo = _pair(n_docs, 4096, seed=21)
        res, d = _delta(gpu, lambda: gpu.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.5, k=10))
        assert_same(res, ora.suggest_batch(qb, qo, "jaccard", 0.5, 10))
        assert d == {"unplanned": 0, "overflow": 0, "repeats": 0, "queries": 0 if n_docs == 8000 else 4096}, d


def test_a_batch_above_the_direct_ordering_limit():
    """more than 128 blocks of 1024 queries: the ordering launches keep their batch-wide histogram (atomics + memset) instead of
    summing the blocks' own (engine.hip query_order_*); ordered or not, fused kernel or pipeline, the rows are the oracle's"""
    gpu, ora, qb, qo = _pair(120000, 140000, seed=31)
    want = ora.suggest_batch(qb, qo, "jaccard", 0.5, 10)
    try:
        for order in (1, 0):
            for pipe in (1, 0):
                gpu.tune(SG_ORDER=order, SG_PIPE=pipe)
                assert_same(gpu.suggest_batch(blob=qb, offs=qo, metric="jaccard", similarity=0.5, k=10), want)
    finally:
        gpu.tune(SG_ORDER=1, SG_PIPE=2)
