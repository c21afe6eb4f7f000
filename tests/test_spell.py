"""The spellchecker caller through the product: language model (host side of libsuggest_hip.so) against the
reference's goldens on CPU; SpellChecker.Predict (GPU: LM-ranked autocomplete + Cosine fuzzy search) against the oracle."""
import os

import numpy as np
import pytest

import oracle

LM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm")
SPELL_INDEX = dict(ngram_size=3, wrap=("^", "$"), pad="$", alphabet=("english", "russian", "numbers", "$^'"))   # eval.go:16-23


@pytest.fixture(scope="module")
def g(reference_tests):
    return reference_tests["lm"]


@pytest.fixture(scope="module")
def lm(g):
    from suggest_amd import LanguageModel
    return LanguageModel(LM_DIR, g["order"], g["startSymbol"], g["endSymbol"])


def test_lm_goldens(lm, g):
    tol = g["tolerance"]
    for words, expected in g["model_score"]:                  # ngram_model_test.go:126-149
        assert abs(lm.Score(words) - expected) < tol, words
    for context, word, expected in g["model_next"]:           # ngram_model_test.go:28-87
        status, score = lm.next_score(context, word, model_level=True)
        assert status == 0 and abs(score - expected) < tol, (context, word)
    for words, expected in g["score_sentence"]:               # language_model_test.go:52-70
        assert abs(lm.ScoreSentence(words) - expected) < tol, words


def test_lm_matches_oracle_everywhere(lm, g):
    """every context of up to 4 known/unknown words x every word: same status and bit-identical score as the oracle"""
    import itertools
    ora = oracle.OracleLM(LM_DIR, g["order"], g["startSymbol"], g["endSymbol"])
    vocab = [w.decode() for w in ora.words()] + ["dont"]
    assert [w.decode() for w in lm.words()] == vocab[:-1]
    rnd = np.random.RandomState(3)
    contexts = [[]] + [[w] for w in vocab] + [list(c) for c in itertools.product(vocab[:6] + ["dont"], repeat=2)]
    contexts += [[vocab[i] for i in rnd.randint(0, len(vocab), size=n)] for n in (3, 3, 3, 4, 4, 5) for _ in range(20)]
    for ctx in contexts:
        for model_level in (False, True):
            for w in vocab:
                assert lm.next_score(ctx, w, model_level) == ora.next_score(ctx, w, model_level), (ctx, w, model_level)
        assert lm.Score(ctx) == ora.score(ctx) and lm.ScoreSentence(ctx) == ora.score_sentence(ctx), ctx


def test_lm_tokenizer_matches_oracle(g):
    from suggest_amd import LanguageModel
    alpha = ("english", "russian", "numbers")
    lm = LanguageModel(LM_DIR, 3, alphabet=alpha)
    ora = oracle.OracleLM(LM_DIR, 3, alphabet=alpha)
    t = g["sentence_retriever_tokens"]
    assert [x.decode() for x in lm.Tokenize(t["text"])] == t["tokens"]
    for text in ["  Hello, WORLD 42!  ", "", "   ", "ЁЖИК в тумане", b"bad \xff\xfe utf8 \xc3", "İstanbul'da", "a-b.c", "x" * 300]:
        assert lm.Tokenize(text) == ora.tokenize(text), text


def _write_lm(tmp, vocab, order, sentences):
    """Google-format count files of a toy corpus (the counting itself is plain Python: test data, not the product)"""
    from collections import Counter
    counts = [Counter() for _ in range(order)]
    for s in sentences:
        seq = ["<S>"] + s + ["</S>"]
        for k in range(1, order + 1):
            for i in range(len(seq) - k + 1):
                counts[k - 1][tuple(seq[i:i + k])] += 1
    for w in vocab:
        counts[0].setdefault((w,), 1)
    with open(os.path.join(tmp, "1-gm"), "w") as f:
        for w in vocab:
            f.write("%s\t%d\n" % (w, counts[0][(w,)]))
    for k in range(2, order + 1):
        with open(os.path.join(tmp, "%d-gm" % k), "w") as f:
            for gram, c in counts[k - 1].items():
                f.write("%s\t%d\n" % (" ".join(gram), c))


def _assert_same_predictions(sc, ora_lm, ora_ix, queries, top_k, similarity):
    qb, qo = oracle.pack_strings(queries)
    ids, cnt = sc.predict_batch(blob=qb, offs=qo, top_k=top_k, similarity=similarity)
    oi, oc = ora_lm.predict_batch(ora_ix, qb, qo, top_k, similarity)
    bad = np.nonzero(cnt != oc)[0]
    assert bad.size == 0, ("counts differ", [(queries[i], int(cnt[i]), int(oc[i])) for i in bad[:5]])
    valid = (np.arange(top_k + 1)[None, :] < np.minimum(cnt, top_k + 1)[:, None]) & (cnt < 0xFFFFFFF0)[:, None]
    rows = np.nonzero((valid & (ids != oi)).any(axis=1))[0]
    assert rows.size == 0, ("rows differ", [(queries[i], ids[i].tolist(), oi[i].tolist()) for i in rows[:5]])


@pytest.mark.gpu
def test_predict_reference_fixture(g):
    from suggest_amd import LanguageModel, SpellChecker
    lm = LanguageModel(LM_DIR, g["order"], g["startSymbol"], g["endSymbol"])
    sc = SpellChecker(lm)
    ora_lm = oracle.OracleLM(LM_DIR, g["order"], g["startSymbol"], g["endSymbol"])
    ora_ix = oracle.OracleIndex(ora_lm.words(), **SPELL_INDEX)
    queries = [b"i am sa", b"green eg", b"i do", b"sam i am sam i am sa", b"gren egs", b"i an", b"<s> i am", b"ha", b"i am xyzxyz", b"", b"i a",
               b"I AM SAM", b"dont kno", b"eggs and ha", b"  ", b"sam"]
    for top_k, sim in ((1, 0.5), (2, 0.3), (5, 0.3), (5, 0.9)):
        _assert_same_predictions(sc, ora_lm, ora_ix, queries, top_k, sim)
    assert sc.Predict("i am sa", 5, 0.3) == ["sam"]
    assert sc.Predict("<s> i am", 5, 0.3) == ["am", "sam", "ham"]
    assert sc.Predict("", 5, 0.3) == []


@pytest.mark.gpu
def test_predict_synthetic_language_model(tmp_path):
    """A 20 000-word vocabulary with families of similar words and a random bigram/trigram corpus: contexts with many
    continuations (64-ary search over several rounds), prefixes with hundreds of completions, ties, unknown words."""
    import torch  # noqa: F401  (before the library: both must resolve the same libamdhip64, suggest_amd/_lib.py)
    from suggest_amd import LanguageModel, SpellChecker, synth
    blob, offs = synth.make_dict(20000, seed=51, families=3)
    vocab = sorted(set(w.decode() for w in synth.unpack(blob, offs)))
    rnd = np.random.RandomState(7)
    hot = [vocab[i] for i in rnd.randint(0, len(vocab), size=40)]
    sentences = []
    for _ in range(30000):
        n = int(rnd.randint(2, 7))
        s = [vocab[int(i)] for i in rnd.zipf(1.3, size=n) % len(vocab)]
        if rnd.rand() < 0.5:
            s[0] = hot[int(rnd.randint(0, len(hot)))]
        sentences.append(s)
    _write_lm(str(tmp_path), vocab, 3, sentences)
    alpha = ("english", "numbers")
    lm = LanguageModel(str(tmp_path), 3, alphabet=alpha)
    sc = SpellChecker(lm)
    ora_lm = oracle.OracleLM(str(tmp_path), 3, alphabet=alpha)
    ora_ix = oracle.OracleIndex(ora_lm.words(), **SPELL_INDEX)
    queries = []
    for i in range(600):
        s = sentences[int(rnd.randint(0, len(sentences)))]
        cut = int(rnd.randint(1, len(s) + 1))
        ctx, word = s[max(0, cut - 1 - int(rnd.randint(0, 4))):cut - 1], s[cut - 1]
        kind = i % 4
        if kind == 0:
            word = word[:int(rnd.randint(2, max(3, len(word))))]                # a prefix
        elif kind == 1:
            word = word[:2]                                                      # a short prefix: many completions
        elif kind == 2 and len(word) > 3:
            p = int(rnd.randint(0, len(word)))
            word = word[:p] + "x" + word[p + 1:]                                 # a typo: the fuzzy search has to find it
        if i % 7 == 0:
            ctx = ctx + ["zzzunknown"]
        queries.append((" ".join(ctx + [word])).encode())
    for top_k, sim in ((5, 0.5), (20, 0.4), (1, 0.7)):
        _assert_same_predictions(sc, ora_lm, ora_ix, queries, top_k, sim)
    # the word tokeniser runs on the device (spell_tokenize_kernel): texts a keyboard produces — upper case, separators of
    # every kind between and around the words, several in a row, non-ASCII runes outside the model's alphabet, invalid
    # UTF-8, a text that ends in separators (the last word is then the one before them), nothing but separators
    messy = []
    for i, q in enumerate(queries[:200]):
        w = q.decode().split(" ")
        sep = [" ", "  ", ", ", " - ", "\t", ". ", "! ", " \u00e9 ", " \u4e2d "][i % 9]
        t = sep.join(x.upper() if (i + j) % 3 == 0 else x for j, x in enumerate(w))
        t = ["", " ", "...", "\u00ab"][i % 4] + t + ["", " ", "!!", " ?", "\u00bb "][i % 5]
        messy.append(t.encode("utf-8"))
    messy += [b"", b"   ", b"?!", b"\xff\xfe", b"abc\xffdef ghi", vocab[5].encode() + b"\xc3", (vocab[7] + " " + vocab[8]).encode() + b"\xe2\x82"]
    _assert_same_predictions(sc, ora_lm, ora_ix, messy, 5, 0.5)
    # the device-resident entry point gives the rows of the host-buffer one
    import torch
    from suggest_amd.index import pack_strings
    dev = torch.device("cuda", 0)
    qb, qo = pack_strings(queries + messy)
    n, k = len(qo) - 1, 5
    h_ids, h_cnt = sc.predict_batch(blob=qb, offs=qo, top_k=k, similarity=0.5)
    d_q = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.view(np.int64)).to(dev)
    d_ids = torch.full((n, k + 1), 7, dtype=torch.int32, device=dev); d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    sc.predict_batch_device(d_q.data_ptr(), d_o.data_ptr(), n, int(qo[-1]), k, 0.5, d_ids.data_ptr(), d_cnt.data_ptr(),
                            stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    assert np.array_equal(d_cnt.cpu().numpy().view(np.uint32), h_cnt)
    assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), h_ids)


def _gram_files(directory, order=3):
    out = []
    for k in range(1, order + 1):
        with open(os.path.join(directory, "%d-gm" % k), "rb") as f:
            out.append(sorted(f.read().splitlines()))
    return out


def test_lm_builder_reproduces_the_reference_fixture(g, tmp_path):
    """`lm build-lm` (NGramBuilder + SentenceRetriever + googleNGramFormatWriter): the reference's fixture files are what
    its builder wrote for testdata/test.txt — product and oracle builders must write the same lines (as multisets)."""
    from suggest_amd import LanguageModel
    b = g["build"]
    want = _gram_files(LM_DIR)
    prod, ora = tmp_path / "prod", tmp_path / "ora"
    prod.mkdir(); ora.mkdir()
    LanguageModel.build_files(b["text"], str(prod), g["order"], g["startSymbol"], g["endSymbol"], b["alphabet"], b["separators"])
    oracle.lm_build_files(b["text"], str(ora), g["order"], g["startSymbol"], g["endSymbol"], b["alphabet"], b["separators"])
    assert _gram_files(str(prod)) == want
    assert _gram_files(str(ora)) == want
    lm = LanguageModel(str(prod), g["order"], g["startSymbol"], g["endSymbol"])
    for words, expected in g["score_sentence"]:
        assert abs(lm.ScoreSentence(words) - expected) < g["tolerance"]


def test_lm_builder_matches_oracle_on_messy_text(tmp_path):
    from suggest_amd import LanguageModel
    rnd = np.random.RandomState(11)
    vocab = ["alpha", "beta", "Gamma", "дельта", "ЭПСИЛОН", "x-ray", "e.g", "42", "naïve", "don't"]
    text = ""
    for _ in range(400):
        text += " ".join(vocab[int(i)] for i in rnd.randint(0, len(vocab), size=int(rnd.randint(0, 9))))
        text += ["\n", ".", "!", " ?", "\n\n", " ", ";"][int(rnd.randint(0, 7))]
    text = text.encode() + b"\xff tail \xc3\n"
    for seps, alpha in (((".", "?", "!", "\n"), ("english", "russian", "numbers", "-'")), (("\n",), ("english", "numbers"))):
        prod, ora = tmp_path / ("p%d" % len(seps)), tmp_path / ("o%d" % len(seps))
        prod.mkdir(); ora.mkdir()
        LanguageModel.build_files(text, str(prod), 4, "<S>", "</S>", alpha, seps)
        oracle.lm_build_files(text, str(ora), 4, "<S>", "</S>", alpha, seps)
        assert _gram_files(str(prod), 4) == _gram_files(str(ora), 4)
