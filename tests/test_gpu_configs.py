"""BASELINE.json's configs at their full size on the GPU, against the CPU oracle.

The dictionaries are the 10 M / 1 M synthetic ones of SURVEY.md §8d (built on the device in 0.4 s); the whole 65,536-query
batch of a config runs through the HIP path, and a sample the oracle finishes in seconds (every 16th query for the 10 M
configs, the whole batch for cfg 2) is compared bit for bit — ids, order, score bit patterns.  The rest of the batch is held
to size-independent properties of the path.  Workloads: pkg/suggest/ngram_index_test.go:147-158,196-206 scaled as
BASELINE.json says.
"""
import numpy as np
import pytest

import oracle
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu

N_Q = 65536


def _properties(ids, sc, cnt, k, n_docs):
    """what holds for every row whatever the inputs: best first by (score desc, id asc), ids inside the dictionary"""
    real = cnt < 0xFFFFFFF0
    c = np.where(real, np.minimum(cnt, k), 0)
    valid = np.arange(k)[None, :] < c[:, None]
    assert (ids[valid] < n_docs).all()
    assert ((sc[valid] > 0) & (sc[valid] <= 1)).all()
    both = valid[:, 1:] & valid[:, :-1]
    s0, s1, i0, i1 = sc[:, :-1][both], sc[:, 1:][both], ids[:, :-1][both], ids[:, 1:][both]
    assert ((s0 > s1) | ((s0 == s1) & (i0 < i1))).all()        # collector.go:20-26: strictly ordered, no document twice


def _subset(qb, qo, rows):
    parts = [qb[int(qo[i]):int(qo[i + 1])] for i in rows]
    offs = np.zeros(len(rows) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(p) for p in parts])
    return (np.concatenate(parts) if parts else np.zeros(0, np.uint8)), offs


class Big:
    def __init__(self, n, q):
        from suggest_amd import NGramIndex, IndexDescription, synth
        self.desc = dict(synth.DESCRIPTION, ngram_size=q)
        self.blob, self.offs = synth.make_dict(n, seed=1)
        self.n = n
        self.qb, self.qo = synth.make_queries(N_Q, self.blob, self.offs, seed=2)
        self.gpu = NGramIndex(blob=self.blob, offs=self.offs, description=IndexDescription(**self.desc), build="device")
        self.ora = oracle.OracleIndex(blob=self.blob, offs=self.offs, **self.desc)

    def check(self, metric, alpha, k, every):
        ids, sc, cnt = self.gpu.suggest_batch(blob=self.qb, offs=self.qo, metric=metric, similarity=alpha, k=k)
        _properties(ids, sc, cnt, k, self.n)
        rows = np.arange(0, N_Q, every)
        sb, so = _subset(self.qb, self.qo, rows)
        assert_same((ids[rows], sc[rows], cnt[rows]), self.ora.suggest_batch(sb, so, metric, alpha, k))
        assert int(np.minimum(cnt[rows], k).sum()) > len(rows) // 2        # the sample is not vacuous
        return ids, sc, cnt


@pytest.fixture(scope="module")
def big_q3():
    b = Big(10_000_000, 3)
    yield b
    b.gpu.close(); b.ora.close()


def test_headline_10m_jaccard_k10(big_q3):
    """the configuration BASELINE.json's metric is quoted on: 10 M strings, q=3, Jaccard >= 0.5, k=10, 65,536 queries"""
    big_q3.check("jaccard", 0.5, 10, every=1)       # every row: the oracle does 26 k q/s on that box


def test_cfg3_10m_cosine_k20(big_q3):
    """BASELINE config 3: 10 M strings, q=3, Cosine >= 0.4, k=20 (window = every segment, T about 8)"""
    ids, sc, cnt = big_q3.check("cosine", 0.4, 20, every=1)
    # the device-resident entry point gives the same rows as the host-buffer one (same launch, other plumbing)
    import torch
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(big_q3.qb).to(dev)
    d_o = torch.from_numpy(big_q3.qo.view(np.int64)).to(dev)
    d_ids = torch.zeros((N_Q, 20), dtype=torch.int32, device=dev)
    d_sc = torch.zeros((N_Q, 20), dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(N_Q, dtype=torch.int32, device=dev)
    big_q3.gpu.suggest_batch_device(d_q.data_ptr(), d_o.data_ptr(), N_Q, "cosine", 0.4, 20, d_ids.data_ptr(), d_sc.data_ptr(),
                                    d_cnt.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    assert np.array_equal(d_cnt.cpu().numpy().view(np.uint32), cnt)
    assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), ids)
    assert np.array_equal(d_sc.cpu().numpy().view(np.uint64), sc.view(np.uint64))


def test_pipeline_takes_the_headline_and_k_100(big_q3):
    """[r6] the default policy sends the headline batch — and the same batch at k = 100, top-k rows in HBM — through plan -> stream ->
    verify (every query counted by sg_index_pipe_stats, none handed back)"""
    for k in (10, 100):
        s0 = big_q3.gpu.pipe_stats()
        big_q3.check("jaccard", 0.5, k, every=16)
        s1 = big_q3.gpu.pipe_stats()
        d = {n: s1[n] - s0[n] for n in s1}
        assert d["queries"] == N_Q and d["unplanned"] + d["overflow"] + d["repeats"] == 0, (k, d)
        assert big_q3.gpu.pipe_volumes()["stream_shape"] == "4 x 2^12", big_q3.gpu.pipe_volumes()


def test_stream_workgroup_follows_the_launch_not_only_the_index(big_q3):
    """[r6] the stream workgroup is chosen per launch: on the same 10 M-string index a Jaccard >= 0.5 batch (two thirds of a query's
    lists left after skipping) streams with four wavefronts on 2^12 counters, a Cosine >= 0.4 batch (nothing skipped) with eight on
    2^13 — where the lighter shape would leave 40 % of the queries unplanned (profiles/r06final_shape_by_size.txt); none is here.
    Knobs fix the shape for every launch; SG_PIPE_SHAPE_AUTO gives it back."""
    gpu = big_q3.gpu
    s0 = gpu.pipe_stats()
    big_q3.check("cosine", 0.4, 20, every=64)
    s1 = gpu.pipe_stats()
    assert gpu.pipe_volumes()["stream_shape"] == "8 x 2^13", gpu.pipe_volumes()
    assert s1["queries"] - s0["queries"] == N_Q and s1["unplanned"] == s0["unplanned"], (s0, s1)
    big_q3.check("jaccard", 0.5, 10, every=64)
    assert gpu.pipe_volumes()["stream_shape"] == "4 x 2^12", gpu.pipe_volumes()
    gpu.tune(SG_PIPE_NW=8, SG_PIPE_LOG2_CNT=13, SG_PIPE_DT_BYTES=8192)
    try:
        big_q3.check("jaccard", 0.5, 10, every=64)
        assert gpu.pipe_volumes()["stream_shape"] == "fixed by knobs"
    finally:
        gpu.tune(SG_PIPE_SHAPE_AUTO=1)
    big_q3.check("jaccard", 0.5, 10, every=64)
    assert gpu.pipe_volumes()["stream_shape"] == "4 x 2^12"


def test_a_launch_that_starts_too_light_moves_to_the_heavier_stream_workgroup():
    """[r6] the counters behind the per-launch choice: 4 M strings under Cosine >= 0.4 want 4 wavefronts on 2^12 counters; started one
    shape too light (test hook SG_PIPE_SHAPE_BIAS) a quarter of the queries come back unplanned (profiles/r06final_shape_by_size.txt:
    15 665 of 65 536) — the replica sees it in its counters and takes the next shape from then on; rows equal the fused kernel's on
    every call.  (Calls through the host-buffer entry point: each one has drained before the next looks at the counters.)"""
    from suggest_amd import NGramIndex, IndexDescription, synth
    blob, offs = synth.make_dict(4_000_000, seed=1)
    qb, qo = synth.make_queries(N_Q, blob, offs, seed=2)
    gpu = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**dict(synth.DESCRIPTION, ngram_size=3)), build="device")
    try:
        gpu.tune(SG_PIPE=0)
        want = gpu.suggest_batch(blob=qb, offs=qo, metric="cosine", similarity=0.4, k=20)
        gpu.tune(SG_PIPE=1, SG_PIPE_SHAPE_BIAS=-1)
        seen, unplanned = [], []
        for _ in range(16):
            s0 = gpu.pipe_stats()
            got = gpu.suggest_batch(blob=qb, offs=qo, metric="cosine", similarity=0.4, k=20)
            s1 = gpu.pipe_stats()
            assert all(np.array_equal(x, y) for x, y in zip(got, want))
            seen.append(gpu.pipe_volumes()["stream_shape"]); unplanned.append(s1["unplanned"] - s0["unplanned"])
        assert seen[0] == "2 wavefronts x 2^11 counters" and unplanned[0] > N_Q // 100, (seen, unplanned)
        assert seen[-1] == "4 x 2^12" and unplanned[-1] == 0, (seen, unplanned)
    finally:
        gpu.close()


def test_25m_strings_take_the_pipeline_with_wide_descriptors():
    """[r6] 25 M strings: a packed store of ~72 M chunks (1.15 GB), above the 2^26 the stream launch's 4-byte sub-row descriptors
    address — round 5 dropped such an index to the fused kernel without a word.  It now takes the three launches with 8-byte
    descriptors and 64-bit row addresses; every 16th row of the 65 536-query batch against the oracle, at k = 10 and k = 100."""
    b = Big(25_000_000, 3)
    try:
        pv = b.gpu.pipe_volumes()
        assert pv["packed_chunks"] >= (1 << 26) and pv["wide"], pv      # the store really is beyond the 4-byte descriptors
        for k in (10, 100):
            s0 = b.gpu.pipe_stats()
            b.check("jaccard", 0.5, k, every=16)
            s1 = b.gpu.pipe_stats()
            d = {n: s1[n] - s0[n] for n in s1}
            assert d["queries"] == N_Q and d["unplanned"] + d["overflow"] < N_Q // 100, (k, d)
    finally:
        b.gpu.close(); b.ora.close()


def test_cfg4_10m_q2_dice_k10():
    """BASELINE config 4: 10 M strings, q=2 (13 MB of postings per query), Dice >= 0.5, k=10"""
    b = Big(10_000_000, 2)
    try:
        b.check("dice", 0.5, 10, every=4)
    finally:
        b.gpu.close(); b.ora.close()


def test_cfg2_1m_whole_batch():
    """BASELINE config 2: 1 M strings, q=3, Jaccard >= 0.5, k=10, the 64k-query batch — every row against the oracle"""
    b = Big(1_000_000, 3)
    try:
        b.check("jaccard", 0.5, 10, every=1)
    finally:
        b.gpu.close(); b.ora.close()


def test_cfg5_spellchecker_50m_token_model(tmp_path_factory):
    """BASELINE config 5 at its size: SpellChecker.Predict (pkg/spellchecker/spellchecker.go:40-92) over a 50 M-token
    synthetic 3-gram model in the reference's own .lm / .cdb formats (tools/make_synthetic_lm.py: ~1 M-word vocabulary, ids
    by count like `lm build-lm`), the vocabulary's fuzzy index + the LM arrays resident on the GPU.  A 65,536-query batch
    ('w1 w2 prefix', one third with a typo in the last word) goes through sg_spell_predict_batch; 16,384 sampled queries are
    compared with the oracle's Predict row by row (ids and order), the rest is held to the path's invariants."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_synthetic_lm
    from suggest_amd.index import pack_strings
    from suggest_amd.spell import LanguageModel, SpellChecker
    tokens = int(os.environ.get("SG_TEST_LM_TOKENS", 50_000_000))
    d = str(tmp_path_factory.mktemp("lm50m"))
    info = make_synthetic_lm.make(d, tokens=tokens, vocab=1_000_000 if tokens >= 20_000_000 else max(1000, tokens // 40), verbose=False)
    assert info["tokens"] >= tokens * 0.99
    lm = LanguageModel(binary=os.path.join(d, "synth.lm"), dictionary=os.path.join(d, "synth.cdb"))
    sc = SpellChecker(lm, device=0)
    top_k, sim = 5, 0.5
    queries = make_synthetic_lm.make_queries(info, N_Q, seed=4242)
    qb, qo = pack_strings(queries)
    ids, cnt = sc.predict_batch(blob=qb, offs=qo, top_k=top_k, similarity=sim)
    assert ids.shape == (N_Q, top_k + 1)
    real = cnt <= top_k + 1
    assert real.mean() > 0.99 and int(cnt[real].sum()) > N_Q            # predictions were made
    valid = np.arange(top_k + 1)[None, :] < np.where(real, cnt, 0)[:, None]
    assert (ids[valid] < len(lm)).all()
    srt = np.sort(np.where(valid, ids.astype(np.int64), -np.arange(1, top_k + 2)[None, :]), axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                            # no word twice in a row of predictions
    rows = np.arange(0, N_Q, 4)
    sb, so = _subset(qb, qo, rows)
    olm = oracle.OracleLM(binary=os.path.join(d, "synth.lm"), dictionary=os.path.join(d, "synth.cdb"))
    oix = oracle.OracleIndex(info["word_list"], ngram_size=sc.description.ngram_size, wrap=sc.description.wrap, pad=sc.description.pad,
                             alphabet=sc.description.alphabet)
    oi, oc = olm.predict_batch(oix, sb, so, top_k, sim, threads=os.cpu_count() or 1)
    assert np.array_equal(cnt[rows], oc)
    ov = np.arange(top_k + 1)[None, :] < np.minimum(oc, top_k + 1)[:, None]
    assert np.array_equal(ids[rows][ov], oi[ov])
    # the same rows through the sliced path (two slices on two streams is the default at this batch size) and in one slice
    os.environ["SG_SPELL_SLICES"] = "1"
    try:
        ids1, cnt1 = sc.predict_batch(blob=qb, offs=qo, top_k=top_k, similarity=sim)
    finally:
        del os.environ["SG_SPELL_SLICES"]
    assert np.array_equal(cnt1, cnt) and np.array_equal(ids1[valid], ids[valid])
