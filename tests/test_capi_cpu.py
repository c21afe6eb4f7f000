"""CPU-only checks of the product's host side: the C-ABI library loads and exports every declared symbol,
the host tokenizer and CSR builder agree with the oracle and with the reference's index files."""
import os
import re

import numpy as np
import pytest

import oracle
import refindex
from conftest import CARS_DESC, WORDS_DESC, GOLDEN, ROOT


def _desc(d):
    from suggest_amd import IndexDescription
    return IndexDescription(ngram_size=d["ngram_size"], wrap=d["wrap"], pad=d["pad"], alphabet=d["alphabet"])


def test_library_exports_every_declared_symbol():
    from suggest_amd import _lib
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "suggest_hip.h")).read()
    declared = set(re.findall(r"\b(sg_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_product_does_not_touch_the_oracle():
    # the product path must not import, link, include or call anything under oracle/
    pat = re.compile(r"import\s+oracle|from\s+oracle|liboracle|oracle/|suggest_oracle|or_suggest|or_index")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "suggest_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(text), (f, pat.search(text).group(0))


def test_search_without_upload_fails_loudly(cars_lines):
    from suggest_amd import NGramIndex, _lib
    ix = NGramIndex(cars_lines[:100], _desc(CARS_DESC), upload=False)
    with pytest.raises(_lib.SuggestHipError) as e:
        ix.suggest_batch(["nissan"], "jaccard", 0.5, 5)
    assert e.value.code == -3


def test_argument_validation(cars_lines):
    from suggest_amd import NGramIndex, IndexDescription, SearchConfig, _lib
    with pytest.raises(ValueError):
        SearchConfig("x", 0, "cosine", 0.5)          # search.go:20-22
    with pytest.raises(ValueError):
        SearchConfig("x", 1, "cosine", 0.0)          # search.go:24-26
    with pytest.raises(ValueError):
        SearchConfig("x", 1, "cosine", 1.5)
    with pytest.raises(_lib.SuggestHipError):
        NGramIndex(cars_lines[:10], IndexDescription(ngram_size=9), upload=False)


def test_host_tokenizer_matches_oracle(cars_lines, words_lines):
    from suggest_amd import NGramIndex
    for lines, desc in ((cars_lines, CARS_DESC), (words_lines[::50], WORDS_DESC)):
        ix = NGramIndex(lines[:50], _desc(desc), upload=False)
        ora = oracle.OracleIndex(lines[:50], **desc)
        probes = list(lines[::17]) + [b"", b" ", b"a", b"  x y  ", "Ёжик в тумане".encode(), b"\xff\xfeabc", "İi".encode(),
                                      b"NISSAN TITAN", b"lalala", "жи".encode()]
        for p in probes:
            for ac in (False, True):
                assert ix.tokenize(p, ac) == ora.tokenize(p, ac), (p, ac)


def test_cars_csr_matches_reference_files(cars_lines, golden_dir):
    """Host CSR == db/cars.{hd,dl} written by the reference (after de-duplicating a doc's repeated terms,
    which the CSR keeps as a multiplicity side table): same keys, raw lengths, postings."""
    from suggest_amd import NGramIndex
    ix = NGramIndex(cars_lines, _desc(CARS_DESC), upload=False)
    n_idx, ref = refindex.read_index(os.path.join(golden_dir, "db", "cars.hd"), os.path.join(golden_dir, "db", "cars.dl"))
    mine = ix.lists()
    st = ix.stats()
    assert st["n_segments"] == n_idx and st["n_lists"] == len(ref) and st["n_postings_raw"] == sum(v[0] for v in ref.values())
    assert set(mine) == set(ref)
    for key, (raw_len, post) in ref.items():
        assert mine[key] == (raw_len, sorted(set(post))), key


def test_words_csr_matches_oracle(words_lines):
    from suggest_amd import NGramIndex
    ix = NGramIndex(words_lines, _desc(WORDS_DESC), upload=False)
    ora = oracle.OracleIndex(words_lines, **WORDS_DESC).lists()
    mine = ix.lists()
    assert set(mine) == set(ora)
    assert all(mine[k] == (ora[k][0], ora[k][1]) for k in ora)


def test_threaded_host_build_equals_the_sequential_one(words_lines, cars_lines, monkeypatch):
    """build_host_index splits the docs into contiguous blocks, one thread each: term numbering (first occurrence over
    ascending docIDs), list layout and the repeated-term table must not depend on the number of blocks, nor on whether
    the blocks count into their own or into one shared set of cursors."""
    from suggest_amd import NGramIndex
    for lines, desc in ((words_lines, WORDS_DESC), (cars_lines + ["", "aaaaaa", "Škoda škoda"] * 1500, CARS_DESC)):
        got = {}
        for thr, shared in (("1", False), ("2", False), ("5", False), ("32", False), ("7", True)):
            monkeypatch.setenv("SG_BUILD_THREADS", thr)
            if shared:
                monkeypatch.setenv("SG_BUILD_SHARED_COUNTERS", "1")
            else:
                monkeypatch.delenv("SG_BUILD_SHARED_COUNTERS", raising=False)
            ix = NGramIndex(lines, _desc(desc), upload=False)
            got[(thr, shared)] = (ix.digest(), ix.stats())
            ix.close()
        first = got[("1", False)]
        assert all(v == first for v in got.values()), got


def test_algorithmic_bytes_matches_oracle_definition():
    from suggest_amd import NGramIndex, IndexDescription, synth
    blob, offs = synth.make_dict(20000, seed=1)
    qb, qo = synth.make_queries(64, blob, offs, seed=2)
    ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION), upload=False)
    ora = oracle.OracleIndex(blob=blob, offs=offs, **synth.DESCRIPTION)
    qs = synth.unpack(qb, qo)
    for metric, alpha, k in (("jaccard", 0.5, 10), ("cosine", 0.4, 20)):
        assert ix.algorithmic_bytes(qb, qo, metric, alpha, k) == sum(ora.algorithmic_bytes(q, metric, alpha, k) for q in qs)


def test_reference_built_files_load_into_the_same_csr(cars_lines, golden_dir):
    """sg_index_load_reference: the gob header + VB / skip-VB lists of db/cars.{hd,dl} give exactly the CSR
    that building from cars.dict gives (keys, raw lengths, postings), and match the independent Python decoder."""
    from suggest_amd import NGramIndex
    hd, dl = os.path.join(golden_dir, "db", "cars.hd"), os.path.join(golden_dir, "db", "cars.dl")
    loaded = NGramIndex.from_reference_files(hd, dl, _desc(CARS_DESC), upload=False)
    built = NGramIndex(cars_lines, _desc(CARS_DESC), upload=False)
    a, b = loaded.lists(), built.lists()
    assert a == b
    sa, sb = loaded.stats(), built.stats()
    for key in ("n_docs", "n_segments", "n_terms", "n_lists", "n_postings", "n_postings_raw", "posting_bytes"):
        assert sa[key] == sb[key], key
    n_idx, ref = refindex.read_index(hd, dl)
    assert {k: (v[0], sorted(set(v[1]))) for k, v in ref.items()} == a


def test_reference_built_roaring_lists_load(golden_dir):
    """db/words_subset.{hd,dl} (subset of the reference's words index, bytes verbatim) holds all three codecs incl. 40
    roaring bitmaps; the C++ loader must agree with the Python decoder list by list."""
    from suggest_amd import NGramIndex
    hd, dl = os.path.join(golden_dir, "db", "words_subset.hd"), os.path.join(golden_dir, "db", "words_subset.dl")
    loaded = NGramIndex.from_reference_files(hd, dl, _desc(WORDS_DESC), upload=False).lists()
    n_idx, ref = refindex.read_index(hd, dl)
    assert sum(1 for v in ref.values() if v[0] > 256) == 40
    assert {k: (v[0], sorted(set(v[1]))) for k, v in ref.items()} == loaded


def test_loader_rejects_bad_inputs(golden_dir, tmp_path):
    from suggest_amd import NGramIndex, IndexDescription, _lib
    hd, dl = os.path.join(golden_dir, "db", "cars.hd"), os.path.join(golden_dir, "db", "cars.dl")
    with pytest.raises(_lib.SuggestHipError):
        NGramIndex.from_reference_files(hd + ".nope", dl, _desc(CARS_DESC), upload=False)
    with pytest.raises(_lib.SuggestHipError):      # alphabet that cannot express the stored terms
        NGramIndex.from_reference_files(hd, dl, IndexDescription(alphabet=("numbers",), pad="#"), upload=False)
    bad = tmp_path / "bad.hd"
    bad.write_bytes(open(hd, "rb").read()[:1000])
    with pytest.raises(_lib.SuggestHipError):
        NGramIndex.from_reference_files(str(bad), dl, _desc(CARS_DESC), upload=False)


def test_cdb_dictionary_matches_the_source_lines(cars_lines, golden_dir):
    from suggest_amd.service import read_cdb_dictionary
    assert read_cdb_dictionary(os.path.join(golden_dir, "db", "cars.cdb")) == cars_lines


def test_stream_shapes_are_pinned():
    """[r6] The stream workgroup a launch starts from (capi.inc pipe_shape_model) for the sixteen launches it was measured on —
    1 M ... 16 M synthetic strings (q = 3, 19.99 n-grams per document, SG_T_FLOOR 8) under Jaccard >= 0.5 and Cosine >= 0.4, each
    with the three shapes side by side on one resident index (profiles/r06final_shape_by_size.txt,
    r06final_shape_auto_by_size.txt) — and the 25 M-string index of the wide-descriptor test.  0 / 1 / 2 = 2 / 4 / 8 wavefronts on
    2^11 / 2^12 / 2^13 counters; the expected query volumes are the ones the GPU box printed (SG_VERBOSE)."""
    import ctypes as C
    from suggest_amd import _lib
    L = _lib.lib()
    JACCARD, COSINE = 0, 1

    def shape(est, metric, alpha, terms=19.99, floor=8):
        out = C.c_int32(-1)
        _lib.check(L.sg_debug_pipe_shape(float(est), float(terms), floor, metric, float(alpha), C.byref(out)))
        return out.value

    est = {1: 2316, 2: 4448, 4: 8711, 6: 12975, 8: 17240, 10: 21500, 13: 27950, 16: 34299, 25: 53484}    # million strings -> chunks
    fastest = {   # (million strings, metric): the shape with the shortest call in the sweeps
        (1, JACCARD): 0, (2, JACCARD): 0, (4, JACCARD): 1, (6, JACCARD): 1, (8, JACCARD): 1, (10, JACCARD): 1, (13, JACCARD): 2, (16, JACCARD): 2,
        (25, JACCARD): 2,
        (1, COSINE): 0, (2, COSINE): 1, (4, COSINE): 1, (6, COSINE): 2, (8, COSINE): 2, (10, COSINE): 2, (13, COSINE): 2, (16, COSINE): 2,
    }
    for (m, metric), want in fastest.items():
        assert shape(est[m], metric, 0.5 if metric == JACCARD else 0.4) == want, (m, metric)
    # a similarity that skips nothing streams every list: the heavier shape earlier; a high one the lighter shape later
    assert shape(est[10], JACCARD, 0.2) == 2 and shape(est[16], JACCARD, 0.8) == 1
    assert L.sg_debug_pipe_shape(1000.0, 20.0, 8, 7, 0.5, C.byref(C.c_int32())) != 0       # (a tabulated metric has no model: the index's own choice)


def test_tuner_choices_are_pinned(cars_lines, words_lines):
    """[r5] The auto-tuner's choices (counter words, filter table, pipeline) for the dictionaries they were measured on — rounds 2-3
    shipped the wrong filter table for three regimes unnoticed.  The statistics of the large synthetic dictionaries are the ones
    the GPU box printed (SG_VERBOSE: expected query volume / longest term, in 16-byte chunks of u32 postings); the reference's own
    dictionaries and a 200 k synthetic one are built here and their statistics recomputed."""
    import ctypes as C
    from suggest_amd import NGramIndex, IndexDescription, synth, _lib
    L = _lib.lib()

    def choice(est, longest):
        out = (C.c_int32 * 6)()
        _lib.check(L.sg_debug_tune_choice(float(est), float(longest), out))
        return {"log2_cnt": out[0], "level": out[1], "pipe": out[2], "stream": tuple(out[3:6])}

    BIG, MID, SMALL = (8, 13, 8192), (4, 12, 4096), (2, 11, 2048)
    measured = {   # dictionary: (expected query volume, longest term) -> what every sweep since round 4 found best (DESIGN.md §4 knobs)
        # stream = (wavefronts, log2 counters, descriptor bytes) of a stream workgroup (profiles/r05zj_*: 60 k ... 4 M strings)
        "headline 10M q=3": ((21500, 2520), dict(log2_cnt=11, level=4, pipe=1, stream=BIG)),
        "families 10M q=3": ((21536, 2527), dict(log2_cnt=11, level=4, pipe=1, stream=BIG)),
        "4M q=3": ((9264, 1090), dict(log2_cnt=11, level=4, pipe=1, stream=MID)),
        "2M q=3": ((4632, 545), dict(log2_cnt=11, level=4, pipe=1, stream=SMALL)),
        "cfg2 1M q=3": ((2316, 272), dict(log2_cnt=11, level=4, pipe=1, stream=SMALL)),
        "60k q=3": ((311, 32), dict(log2_cnt=11, level=4, pipe=1, stream=SMALL)),
        "cfg4 10M q=2": ((826904, 78375), dict(log2_cnt=11, level=4, pipe=0, stream=BIG)),
        "skewed 10M q=3": ((583089, 443397), dict(log2_cnt=12, level=4, pipe=0, stream=BIG)),
        "cfg5 vocabulary": ((1219, 397), dict(log2_cnt=11, level=2, pipe=0, stream=SMALL)),
        "cars": ((790, 390), dict(log2_cnt=11, level=2, pipe=0, stream=SMALL)),
        "words": ((5280, 3639), dict(log2_cnt=11, level=2, pipe=0, stream=SMALL)),
    }
    for name, ((est, longest), want) in measured.items():
        assert choice(est, longest) == want, name

    def built(ix):
        st, out = (C.c_double * 2)(), (C.c_int32 * 6)()
        with ix._use() as h:
            _lib.check(L.sg_debug_tune_index(h, st, out))
        return (st[0], st[1]), {"log2_cnt": out[0], "level": out[1], "pipe": out[2], "stream": tuple(out[3:6])}

    (est, longest), got = built(NGramIndex(cars_lines, _desc(CARS_DESC), upload=False))
    assert 600 < est < 1000 and 300 < longest < 500 and got == measured["cars"][1], (est, longest, got)
    (est, longest), got = built(NGramIndex(words_lines, _desc(WORDS_DESC), upload=False))
    assert 4000 < est < 6500 and 3000 < longest < 4200 and got == measured["words"][1], (est, longest, got)
    blob, offs = synth.make_dict(200000, seed=1)      # uniform strings: the longest term a tenth of a query's volume, like the headline
    (est, longest), got = built(NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION), upload=False))
    assert 450 < est < 800 and longest < 0.25 * est and got == dict(log2_cnt=11, level=4, pipe=1, stream=SMALL), (est, longest, got)
