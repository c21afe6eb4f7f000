"""The spellchecker caller's language model: the CPU restatement (oracle/spell_oracle.inc) against the
reference's own goldens — pkg/lm/ngram_model_test.go, language_model_test.go over its fixtures (tests/golden/lm)."""
import os

import numpy as np
import pytest

import oracle

LM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm")


@pytest.fixture(scope="module")
def lm_golden(reference_tests):
    return reference_tests["lm"]


@pytest.fixture(scope="module")
def ora_lm(lm_golden):
    return oracle.OracleLM(LM_DIR, lm_golden["order"], lm_golden["startSymbol"], lm_golden["endSymbol"])


def test_model_score(ora_lm, lm_golden):                      # ngram_model_test.go:126-149 (testModel)
    for words, expected in lm_golden["model_score"]:
        assert abs(ora_lm.score(words) - expected) < lm_golden["tolerance"], words


def test_model_next(ora_lm, lm_golden):                       # ngram_model_test.go:28-87 (TestPredict)
    for context, word, expected in lm_golden["model_next"]:
        status, score = ora_lm.next_score(context, word, model_level=True)
        assert status == 0 and abs(score - expected) < lm_golden["tolerance"], (context, word)


def test_score_sentence(ora_lm, lm_golden):                   # language_model_test.go:52-70 (testLM)
    for words, expected in lm_golden["score_sentence"]:
        assert abs(ora_lm.score_sentence(words) - expected) < lm_golden["tolerance"], words


def test_next_rules(ora_lm):
    """language_model.go:100-112: a short context is left-wrapped with <S>, a long one keeps its last order-1 words,
    one of exactly `order` words keeps its FIRST order-1 (sic); ngram_model.go:64-98: unseen context -> nil scorer."""
    assert ora_lm.next_score(["i"], "am") == ora_lm.next_score(["<S>", "i"], "am", model_level=True)
    assert ora_lm.next_score(["sam", "i", "am", "sam"], "</S>") == ora_lm.next_score(["am", "sam"], "</S>", model_level=True)
    assert ora_lm.next_score(["i", "am", "sam"], "sam") == ora_lm.next_score(["i", "am"], "sam", model_level=True)
    assert ora_lm.next_score(["ham", "i"], "am")[0] == 1            # "ham i" never occurs: no scorer, no error
    assert ora_lm.next_score(["dont"], "know")[0] == 1              # unknown word in the context
    assert ora_lm.next_score(["i", "am"], "i") == (0, -100.0)       # known context, unseen continuation
    assert ora_lm.next_score([], "i", model_level=True)[0] == 2     # "nGrams length should be less than the nGramModel order"
    assert ora_lm.next_score(["i", "am", "sam"], "i", model_level=True)[0] == 2


def test_tokenizer(lm_golden):                                # sentence_retriever_test.go:17-27 (tokens of one sentence)
    lm = oracle.OracleLM(LM_DIR, 3, alphabet=("english", "russian", "numbers"))
    t = lm_golden["sentence_retriever_tokens"]
    assert [x.decode() for x in lm.tokenize(t["text"])] == t["tokens"]
    assert [x.decode() for x in lm.tokenize("  Hello, WORLD 42!  ")] == ["hello", "world", "42"]


def test_predict_small_vocabulary(ora_lm):
    """SpellChecker.Predict over the fixture vocabulary (index description of cmd/spellchecker/cmd/eval.go:16-23)."""
    words = ora_lm.words()
    ix = oracle.OracleIndex(words, ngram_size=3, wrap=("^", "$"), pad="$", alphabet=("english", "russian", "numbers", "$^'"))
    queries = [b"i am sa", b"green eg", b"i do", b"sam i am sam i am sa", b"gren egs", b"i an", b"<s> i am", b"ha", b"i am xyzxyz", b"", b"i a"]
    qb, qo = oracle.pack_strings(queries)
    ids, cnt = ora_lm.predict_batch(ix, qb, qo, 5, 0.3)
    got = {q: [words[i].decode() for i in ids[n, :cnt[n]]] for n, q in enumerate(queries)}
    assert got[b"i am sa"] == ["sam"] and got[b"green eg"] == ["eggs"] and got[b"i do"] == ["do"]
    assert got[b"gren egs"] == ["eggs"]               # no prefix match: the fuzzy search (Cosine) finds it
    assert got[b"i an"] == ["and"]
    assert got[b"<s> i am"] == ["am", "sam", "ham"]   # "am" completes itself; "sam", "ham" come from the fuzzy search
    assert got[b"i am xyzxyz"] == [] and got[b""] == []
    assert got[b"i a"] == []                          # "^a" is shorter than q: no n-gram, the reference finds nothing
