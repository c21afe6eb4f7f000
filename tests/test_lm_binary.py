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
