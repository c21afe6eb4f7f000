"""Pins the CPU oracle against every golden vector the reference's own tests hold for the hot
path and against the reference-built index files.  CPU only."""
import hashlib
import json
import os
import struct

import pytest

import oracle
import refindex
from conftest import CARS_DESC, WORDS_DESC, GOLDEN

MERGERS = ["scan_count", "cp_merge", "merge_skip", "divide_skip"]


def desc_kwargs(d):
    return dict(ngram_size=d["nGramSize"], wrap=tuple(d["wrap"]), pad=d["pad"], alphabet=tuple(d["alphabet"]))


@pytest.mark.parametrize("algo", MERGERS)
def test_merge_tables(reference_tests, algo):
    # pkg/merger/list_merger_test.go:42-175
    for case in reference_tests["merge"]["cases"]:
        got = {}
        for pos, ov in oracle.merge(algo, case["rid"], case["t"]):
            got.setdefault(str(ov), []).append(pos)
        assert got == case["expected"], (algo, case)


def test_overlap_overflow_panics(reference_tests):
    # pkg/merger/list_merger_test.go:11-17
    assert oracle.lib().or_candidate_increment(reference_tests["overlap_overflow"]["overlap"]) == -2
    assert oracle.lib().or_candidate_increment(7) == 8


def test_intersect(reference_tests):
    for case in reference_tests["intersect"]["cases"]:
        got = [p for p, _ in oracle.merge("intersector", case["rid"], 0)]
        assert got == case["expected"]
        # overlap reported by the intersector is the number of lists (list_intersector.go:57)
        assert all(o == len(case["rid"]) for _, o in oracle.merge("intersector", case["rid"], 0))


def test_ngram_tokenizer(reference_tests):
    for case in reference_tests["ngram_tokenizer"]["cases"]:
        got = [t.decode("utf-8") for t in oracle.ngram_tokenize(case["word"], case["k"])]
        assert got == case["ngrams"], case


def test_ngram_tokenizer_edge_semantics():
    # ngram_tokenizer.go:18 byte-length guard; >= q bytes but <= q runes gives the whole text once
    assert oracle.ngram_tokenize("ab", 3) == []
    assert oracle.ngram_tokenize("жи", 3) == ["жи".encode()]        # 4 bytes, 2 runes
    assert oracle.ngram_tokenize("abc", 3) == [b"abc"]
    assert oracle.ngram_tokenize("abcdefghkl123456йцукен", 3)[-1] == "кен".encode()


def test_alphabet(reference_tests):
    a = reference_tests["alphabet"]
    for ch, exp in a["russian"]:
        assert oracle.alphabet_has(["russian"], ch) == exp, ch
    for ch, exp in a["composite_russian_english_numbers"]:
        assert oracle.alphabet_has(["russian", "english", "numbers"], ch) == exp, ch
    assert oracle.alphabet_has(["$^"], "^") and oracle.alphabet_has(["$^"], "$") and not oracle.alphabet_has(["$^"], "a")


def test_to_lower_go_semantics():
    assert oracle.to_lower("NISSAN March") == b"nissan march"
    assert oracle.to_lower("ЖИГУЛИ Ё") == "жигули ё".encode()
    assert oracle.to_lower("İ") == b"i"                       # unicode.ToLower(U+0130) == 'i' (simple mapping)
    assert oracle.to_lower(b"A\xffB\xc3") == b"A\xffB\xc3".replace(b"\xff", "�".encode()).replace(
        b"\xc3", "�".encode()).lower()                  # strings.Map re-encodes invalid bytes as U+FFFD


def test_topk(reference_tests):
    t = reference_tests["topk"]
    got, lowest, can = oracle.topk(t["k"], t["inserts"], probe=t["can_take"][0])
    assert got == [tuple(x) for x in t["expected"]]
    assert lowest == t["lowest"] and can == t["can_take"][1]


def test_topk_tie_break_prefers_smaller_doc_id():
    # Candidate.Less collector.go:20-26: equal score -> larger key is "less" (worse)
    got, _, _ = oracle.topk(2, [(9, 0.5), (3, 0.5), (7, 0.5), (1, 0.25)])
    assert got == [(3, 0.5), (7, 0.5)]


def test_suggest_auto(reference_tests):
    t = reference_tests["suggest_auto"]
    ix = oracle.OracleIndex(reference_tests["small_collection"], **desc_kwargs(t["description"]))
    for algo in ["cp_merge", "scan_count", "merge_skip", "divide_skip"]:
        got = ix.suggest(t["query"], t["metric"], t["similarity"], t["topK"], algo=algo)
        assert [d for d, _ in got] == t["expected_ids"]


def test_autocomplete(reference_tests):
    t = reference_tests["autocomplete"]
    ix = oracle.OracleIndex(reference_tests["small_collection"], **desc_kwargs(t["description"]))
    assert ix.autocomplete(t["query"], t["limit"]) == t["expected_ids"]


def test_example(reference_tests):
    t = reference_tests["example"]
    coll = reference_tests["small_collection"]
    ix = oracle.OracleIndex(coll, **desc_kwargs(t["description"]))
    got = ix.suggest(t["query"], t["metric"], t["similarity"], t["topK"])
    assert [coll[d] for d, _ in got] == t["expected_values"]


@pytest.fixture(scope="module")
def cars_index(cars_lines):
    return oracle.OracleIndex(cars_lines, **CARS_DESC)


def test_service_cars(reference_tests, cars_lines, cars_index):
    t = reference_tests["service_cars"]
    for q, exp in zip(t["queries"], t["expected_values"]):
        for tighten in (False, True):
            got = cars_index.suggest(q, t["metric"], t["similarity"], t["topK"], tighten=tighten)
            assert [cars_lines[d].decode() for d, _ in got] == exp, (q, tighten)


def test_cars_index_matches_reference_files(cars_index, golden_dir):
    """Every (segment, term) list of the oracle-built cars index equals the list the reference
    wrote to db/cars.{hd,dl}: same keys, same raw lengths, same postings incl. duplicates."""
    n_idx, ref = refindex.read_index(os.path.join(golden_dir, "db", "cars.hd"), os.path.join(golden_dir, "db", "cars.dl"))
    mine = cars_index.lists()
    assert cars_index.n_segments == n_idx == 52
    assert set(mine) == set(ref) and len(ref) == 36285
    bad = [k for k in ref if mine[k] != (ref[k][0], ref[k][1])]
    assert not bad, bad[:5]
    assert mine[(12, b"an$")][1] == [1080, 1082, 1111, 2244, 2245, 3235, 3235, 3238, 3239, 3240, 3240, 3247, 3250,
                                     3251, 3252, 3253, 3254, 3259, 3265]


def test_words_index_matches_reference_digest(words_lines, golden_dir):
    """235 887-word index vs the digest of db/words.{hd,dl} (tools/make_golden.py): per-segment sha256
    over every list (VB, skip-VB and roaring classes), plus explicit samples."""
    with open(os.path.join(golden_dir, "words_index_digest.json")) as f:
        dig = json.load(f)
    ix = oracle.OracleIndex(words_lines, **WORDS_DESC)
    mine = ix.lists()
    assert ix.n_segments == dig["n_indices"] and len(mine) == dig["n_lists"]
    assert sum(v[0] for v in mine.values()) == dig["n_postings_raw"]
    per = {}
    for (seg, term) in sorted(mine):
        raw_len, post = mine[(seg, term)]
        h = per.setdefault(seg, [0, 0, hashlib.sha256()])
        h[0] += 1
        h[1] += len(post)
        h[2].update(struct.pack("<II", seg, len(term)) + term + struct.pack("<II", raw_len, len(post)))
        h[2].update(struct.pack("<%dI" % len(post), *post))
    got = {str(s): [v[0], v[1], v[2].hexdigest()] for s, v in per.items()}
    assert got == dig["segments"]
    for seg, term_hex, raw_len, post in dig["samples"]:
        assert mine[(seg, bytes.fromhex(term_hex))] == (raw_len, post)


def test_probe_predictions_cross_check(cars_index):
    # line-by-line emulation predictions (unverified against Go) — a cross-check of two
    # independent restatements, not a pin.
    got = cars_index.suggest("Nissan Mar", "jaccard", 0.5, 5)
    assert got == [(3265, 0.6923076923076923), (3230, 0.5333333333333333)]
    got = cars_index.suggest("Nissan Mar", "cosine", 0.5, 5)
    assert [d for d, _ in got] == [3265, 3230, 3254, 3231, 3256]


def test_all_mergers_agree_on_cars(reference_tests, cars_index):
    for q in reference_tests["workloads"]["cars_cosine_0.5_k5"] + reference_tests["service_cars"]["queries"]:
        base = cars_index.suggest(q, "cosine", 0.5, 5, algo="cp_merge")
        for algo in ("scan_count", "merge_skip", "divide_skip"):
            assert cars_index.suggest(q, "cosine", 0.5, 5, algo=algo) == base, (q, algo)


def test_reference_crash_inputs_are_flagged(cars_index):
    # query far longer than any entry: bMin > nSegments-1 -> the reference panics (negative channel cap)
    import random
    rng = random.Random(7)
    long_q = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz0123456789") for _ in range(400))
    assert cars_index.suggest(long_q, "jaccard", 0.5, 5) == oracle.STATUS_REFERENCE_PANICS
    # empty token list -> empty result, no error (suggester.go:49-51)
    assert cars_index.suggest("", "jaccard", 0.5, 5) == []


def test_go_sort_three_restatements_agree():
    """Go 1.14 sort.Sort orders EQUAL-length posting lists in cpMerge (cp_merge.go:24) — visible only in the secondary rows of
    documents that repeat a term.  No Go toolchain here, so it stays parity-unpinned; what can be checked is that three
    separately written restatements (oracle C++, tests/gosort.py, the device's PairSort — the latter in the GPU suite)
    produce the same unstable order on inputs full of ties, through every branch (insertion < 13, ninther > 40, the
    protected partition, heapsort on exhausted depth)."""
    import random
    import gosort
    rng = random.Random(1914)
    for trial in range(1500):
        n = rng.choice([0, 1, 2, 5, 12, 13, 14, 30, 41, 60, 100, 128])
        spread = rng.choice([1, 2, 3, 5, 17, 1000])
        keys = [rng.randrange(spread) for _ in range(n)]
        if trial % 7 == 0:
            keys.sort()
        if trial % 11 == 0:
            keys.sort(reverse=True)
        a, b = oracle.go_sort(keys), gosort.go_sort(keys)
        assert a == b, (keys, a, b)
        assert [keys[i] for i in a] == sorted(keys)


def test_lowercase_tables_of_oracle_and_product_come_from_two_sources_and_agree():
    """oracle/unicode_lower.inc is read off glibc's towlower (oracle/gen_unicode_lower_glibc.c), the product's off Python's
    unicodedata (tools/gen_unicode_lower.py): same pairs, none newer than Unicode 12.0 (Go 1.14)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tabs = []
    for rel in ("oracle/unicode_lower.inc", "suggest_amd/csrc/unicode_lower.inc"):
        txt = open(os.path.join(root, rel)).read()
        tabs.append({int(a, 16): int(b, 16) for a, b in re.findall(r"\{0x([0-9A-Fa-f]+),\s*0x([0-9A-Fa-f]+)\}", txt)})
        assert "glibc" in txt.splitlines()[0] if rel.startswith("oracle") else "gen_unicode_lower.py" in txt.splitlines()[0]
    assert tabs[0] == tabs[1] and len(tabs[0]) == 1364
    assert tabs[0][0x130] == 0x69 and tabs[0][0x410] == 0x430
    assert not any(c in tabs[0] for c in (0xA7C7, 0xA7C9, 0xA7F5, 0x2C2F, 0xA7C0, 0x10570))


def test_go_sort_hand_traced_small_inputs(reference_tests):
    """Go 1.14 sort.Sort on <= 12 elements is two straight-line steps (a ShellSort pass with gap 6, then insertionSort:
    src/sort/sort.go, quickSort) — traced compare by compare in tests/golden/go_sort_small_traces.txt by a script that
    implements nothing else (tools/gosort_hand_traces.py).  Both CPU restatements of the full algorithm must give these
    permutations; the device's PairSort is held to them in the GPU suite.  (13 .. 40 elements: the next test; the ninther pivot
    above 40 and the heapSort fallback stay pinned only by the three restatements agreeing: no Go toolchain)"""
    import numpy as np
    import gosort
    vectors = reference_tests["go_sort_small"]["vectors"]
    assert len(vectors) >= 20
    unstable = 0
    for v in vectors:
        keys = np.array(v["keys"], dtype=np.uint32)
        assert oracle.go_sort(keys) == v["perm"], v
        assert gosort.go_sort(v["keys"]) == v["perm"], v
        unstable += v["perm"] != sorted(range(len(keys)), key=lambda i: (v["keys"][i], i))
    assert unstable >= 3                    # the vectors do show the instability (a stable sort would fail them)


def test_go_sort_hand_traced_13_to_40_elements(reference_tests):
    """[r5] 13 .. 40 elements: quickSort's loop with doPivot's median-of-three branch (no ninther: hi - lo <= 40; the inputs never
    exhaust maxDepth, so no heapSort), pieces of <= 12 elements by the two straight-line steps — every Less and Swap written out
    in tests/golden/go_sort_mid_traces.txt by tools/gosort_hand_traces.py, which transcribes those functions of go1.14
    src/sort/sort.go statement by statement (quoted in its header).  Above 40 elements (ninther) and heapSort stay pinned by the
    restatements agreeing."""
    import numpy as np
    import gosort
    vectors = reference_tests["go_sort_mid"]["vectors"]
    assert len(vectors) >= 24 and {len(v["keys"]) for v in vectors} >= {13, 14, 20, 30, 37, 40}
    unstable = 0
    for v in vectors:
        assert 13 <= len(v["keys"]) <= 40
        assert oracle.go_sort(np.array(v["keys"], dtype=np.uint32)) == v["perm"], v
        assert gosort.go_sort(v["keys"]) == v["perm"], v
        unstable += v["perm"] != sorted(range(len(v["keys"])), key=lambda i: (v["keys"][i], i))
    assert unstable >= 10
