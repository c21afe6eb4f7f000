"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference path).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

METRICS = {"jaccard": 0, "cosine": 1, "dice": 2, "exact": 3, "overlap": 4}
ALGOS = {"cp_merge": 0, "scan_count": 1, "merge_skip": 2, "divide_skip": 3, "intersector": 4}
STATUS_REFERENCE_PANICS = -1      # suggester.go:62 make(chan, negative)
STATUS_REFERENCE_DEADLOCKS = -2   # suggester.go:62/70/115 capacity-0 channel, no workers


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        srcs = [os.path.join(ROOT, "oracle", n) for n in ("suggest_oracle.cpp", "spell_oracle.inc")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(x) for x in srcs):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.or_index_build.restype = C.c_void_p
        L.or_index_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.or_index_free.argtypes = [C.c_void_p]
        L.or_index_segments.argtypes = [C.c_void_p]
        L.or_index_num_lists.restype = C.c_uint64
        L.or_index_num_lists.argtypes = [C.c_void_p]
        L.or_index_list_at.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                       C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_int)]
        L.or_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.or_ngram_tokenize.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.or_alphabet_has.argtypes = [C.c_char_p, C.c_uint32]
        L.or_to_lower.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        for f in (L.or_metric_min_y, L.or_metric_max_y):
            f.argtypes = [C.c_int, C.c_double, C.c_int]
        L.or_metric_threshold.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int]
        for f in (L.or_metric_distance, L.or_metric_score):
            f.restype = C.c_double
            f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.or_merge.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.or_candidate_increment.argtypes = [C.c_uint32]
        L.or_topk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                              C.POINTER(C.c_int), C.c_double]
        L.or_suggest.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.or_autocomplete.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.or_suggest_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_double, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        L.or_autocomplete_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.or_query_algorithmic_bytes.restype = C.c_uint64
        L.or_query_algorithmic_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int]
        L.or_lm_load.restype = C.c_void_p
        L.or_lm_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.or_lm_load_ex.restype = C.c_void_p
        L.or_lm_load_ex.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.or_lm_load_binary.restype = C.c_void_p
        L.or_lm_load_binary.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.or_lm_level.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint64)),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.or_lm_order.argtypes = [C.c_void_p]
        L.or_go_sort.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.or_lm_free.argtypes = [C.c_void_p]
        L.or_lm_build_google.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.or_lm_words.restype = C.c_uint32
        L.or_lm_words.argtypes = [C.c_void_p]
        L.or_lm_word.restype = C.c_char_p
        L.or_lm_word.argtypes = [C.c_void_p, C.c_uint32]
        L.or_lm_word_id.restype = C.c_uint32
        L.or_lm_word_id.argtypes = [C.c_void_p, C.c_char_p]
        for f in (L.or_lm_score, L.or_lm_score_word_ids):
            f.restype = C.c_double
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for f in (L.or_lm_model_next_score, L.or_lm_next_score):
            f.restype = C.c_double
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_int)]
        L.or_lm_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.or_spell_predict_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_double,
                                             C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def pack_strings(strings):
    """list of bytes/str -> (uint8 blob, uint64 offsets[n+1])"""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strings]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return blob, offs


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


class OracleIndex:
    """Index built with the reference's indexing semantics (pkg/suggest/indexer.go:14-45)."""

    def __init__(self, docs=None, ngram_size=3, wrap=("$", "$"), pad="$", alphabet=("english", "numbers", "$"),
                 blob=None, offs=None):
        L = lib()
        if blob is None:
            blob, offs = pack_strings(docs)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        self.n_docs = len(offs) - 1
        self._h = L.or_index_build(blob.ctypes.data, offs.ctypes.data, self.n_docs, ngram_size, _b(wrap[0]), _b(wrap[1]),
                                   _b(pad), "\n".join(alphabet).encode("utf-8"))
        self.k_cap = 0

    def close(self):
        if self._h:
            lib().or_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_segments(self):
        return lib().or_index_segments(self._h)

    def lists(self):
        """-> {(segment, term_bytes): (raw_len, [stored postings])}"""
        L = lib()
        out = {}
        seg, term, tl, post, ns = C.c_int(), C.c_char_p(), C.c_int(), C.POINTER(C.c_uint32)(), C.c_int()
        for i in range(L.or_index_num_lists(self._h)):
            raw = L.or_index_list_at(self._h, i, C.byref(seg), C.byref(term), C.byref(tl), C.byref(post), C.byref(ns))
            t = C.string_at(term, tl.value)
            out[(seg.value, t)] = (raw, np.ctypeslib.as_array(post, shape=(ns.value,)).tolist() if ns.value else [])
        return out

    def tokenize(self, text, autocomplete=False):
        t = _b(text)
        buf = C.create_string_buffer(64 * (len(t) + 16) + 64)
        n = lib().or_tokenize(self._h, t, len(t), 1 if autocomplete else 0, buf, len(buf))
        assert n >= 0
        return _split(buf.raw, n)

    def suggest(self, query, metric, similarity, k, tighten=False, algo="cp_merge"):
        """-> list[(doc_id, score)] best first, or a negative STATUS_* int"""
        q = _b(query)
        ids = np.zeros(k, dtype=np.uint32)
        sc = np.zeros(k, dtype=np.float64)
        n = lib().or_suggest(self._h, q, len(q), METRICS[metric], similarity, k, 1 if tighten else 0, ALGOS[algo],
                             ids.ctypes.data, sc.ctypes.data)
        if n < 0:
            return n
        return [(int(ids[i]), float(sc[i])) for i in range(n)]

    def autocomplete(self, query, limit):
        q = _b(query)
        ids = np.zeros(max(limit, 1), dtype=np.uint32)
        n = lib().or_autocomplete(self._h, q, len(q), limit, ids.ctypes.data)
        return [int(x) for x in ids[:n]]

    def suggest_batch(self, blob, offs, metric, similarity, k, threads=0):
        """-> (ids[n_q,k] u32, scores[n_q,k] f64, counts[n_q] u32, threads_used).
        counts 0xFFFFFFFF / 0xFFFFFFFE flag queries on which the reference panics / dead-locks."""
        n_q = len(offs) - 1
        ids = np.zeros((n_q, k), dtype=np.uint32)
        sc = np.zeros((n_q, k), dtype=np.float64)
        cnt = np.zeros(n_q, dtype=np.uint32)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        used = lib().or_suggest_batch(self._h, blob.ctypes.data, offs.ctypes.data, n_q, METRICS[metric], similarity, k, threads,
                                      ids.ctypes.data, sc.ctypes.data, cnt.ctypes.data)
        return ids, sc, cnt, used

    def autocomplete_batch(self, blob, offs, limit, threads=0):
        n_q = len(offs) - 1
        ids = np.zeros((n_q, limit), dtype=np.uint32)
        cnt = np.zeros(n_q, dtype=np.uint32)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        used = lib().or_autocomplete_batch(self._h, blob.ctypes.data, offs.ctypes.data, n_q, limit, threads, ids.ctypes.data,
                                           cnt.ctypes.data)
        return ids, cnt, used

    def algorithmic_bytes(self, query, metric, similarity, k):
        q = _b(query)
        return int(lib().or_query_algorithmic_bytes(self._h, q, len(q), METRICS[metric], similarity, k))


def _split(raw, n):
    out, i = [], 0
    for _ in range(n):
        j = raw.index(b"\0", i)
        out.append(raw[i:j])
        i = j + 1
    return out


def ngram_tokenize(text, n):
    t = _b(text)
    buf = C.create_string_buffer(16 * (len(t) + 8) + 64)
    c = lib().or_ngram_tokenize(t, len(t), n, buf, len(buf))
    assert c >= 0
    return _split(buf.raw, c)


def alphabet_has(spec, ch):
    return bool(lib().or_alphabet_has("\n".join(spec).encode("utf-8"), ord(ch)))


def to_lower(text):
    t = _b(text)
    buf = C.create_string_buffer(4 * len(t) + 8)
    n = lib().or_to_lower(t, len(t), buf, len(buf))
    return buf.raw[:n]


def merge(algo, rid, threshold):
    """-> list[(position, overlap)] in collection order, or -2 on the reference's 'overlap overflow' panic"""
    flat = np.array([x for l in rid for x in l], dtype=np.uint32)
    lens = np.array([len(l) for l in rid], dtype=np.uint32)
    cap = max(16, len(flat) + 16)
    pos = np.zeros(cap, dtype=np.uint32)
    ov = np.zeros(cap, dtype=np.uint32)
    n = lib().or_merge(ALGOS[algo], flat.ctypes.data, lens.ctypes.data, len(rid), threshold, pos.ctypes.data, ov.ctypes.data, cap)
    if n < 0:
        return n
    return [(int(pos[i]), int(ov[i])) for i in range(n)]


def topk(k, inserts, probe=0.0):
    keys = np.array([x[0] for x in inserts], dtype=np.uint32)
    sc = np.array([x[1] for x in inserts], dtype=np.float64)
    ok = np.zeros(max(k, 1), dtype=np.uint32)
    os_ = np.zeros(max(k, 1), dtype=np.float64)
    lowest, can = C.c_double(), C.c_int()
    n = lib().or_topk(k, keys.ctypes.data, sc.ctypes.data, len(inserts), ok.ctypes.data, os_.ctypes.data, C.byref(lowest),
                      C.byref(can), probe)
    return [(int(ok[i]), float(os_[i])) for i in range(n)], lowest.value, bool(can.value)


def metric_min_y(m, a, s):
    return lib().or_metric_min_y(METRICS[m], a, s)


def metric_max_y(m, a, s):
    return lib().or_metric_max_y(METRICS[m], a, s)


def metric_threshold(m, a, sa, sb):
    return lib().or_metric_threshold(METRICS[m], a, sa, sb)


def metric_score(m, inter, sa, sb):
    return lib().or_metric_score(METRICS[m], inter, sa, sb)


def lm_build_files(text, directory, order=3, start_symbol="<S>", end_symbol="</S>", alphabet=("english", "russian", "numbers", "-."),
                   separators=("\n",)):
    """or_lm_build_google: corpus -> <directory>/{1..order}-gm (NGramBuilder + googleNGramFormatWriter)"""
    raw = _b(text)
    rc = lib().or_lm_build_google(raw, len(raw), int(order), _b(start_symbol), _b(end_symbol), b"\n".join(_b(a) for a in alphabet),
                                  b"\x1f".join(_b(a) for a in separators), _b(directory))
    if rc:
        raise IOError("cannot write n-gram files to %s" % directory)


class OracleLM:
    """or_lm_*: the language model of the spellchecker caller (pkg/lm), loaded from Google-format n-gram count files
    <dir>/{1..order}-gm; word ids = line numbers of 1-gm."""

    def __init__(self, directory=None, order=3, start_symbol="<S>", end_symbol="</S>", alphabet=("english", "russian", "numbers", "-."),
                 id_order="lines", binary=None, dictionary=None):
        """id_order "lines": word ids = 1-gm line numbers (the reference's test indexer); "count": (count desc, word asc) like
        buildDictionary (binary.go:101-199).  binary + dictionary: RetrieveLMFromBinary from <name>.lm / <name>.cdb."""
        err = C.create_string_buffer(256)
        alpha = b"\n".join(_b(a) for a in alphabet)
        if binary is not None:
            self._h = lib().or_lm_load_binary(_b(binary), _b(dictionary), _b(start_symbol), _b(end_symbol), alpha, err, 256)
        else:
            self._h = lib().or_lm_load_ex(_b(directory), int(order), _b(start_symbol), _b(end_symbol), alpha, {"lines": 0, "count": 1}[id_order], err, 256)
        if not self._h:
            raise IOError(err.value.decode())

    def level(self, i):
        """-> (containers u64[], values u64[], total) of level i, as packedArray.Store writes them"""
        c, v = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        nc, nv, tot = C.c_uint32(), C.c_uint32(), C.c_uint32()
        assert lib().or_lm_level(self._h, i, C.byref(c), C.byref(nc), C.byref(v), C.byref(nv), C.byref(tot)) == 0
        return (np.ctypeslib.as_array(c, shape=(nc.value,)).copy() if nc.value else np.zeros(0, np.uint64),
                np.ctypeslib.as_array(v, shape=(nv.value,)).copy() if nv.value else np.zeros(0, np.uint64), int(tot.value))

    @property
    def order(self):
        return int(lib().or_lm_order(self._h))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().or_lm_free(self._h)
                self._h = None
        except Exception:
            pass

    def words(self):
        L = lib()
        return [L.or_lm_word(self._h, i) for i in range(L.or_lm_words(self._h))]

    def word_id(self, w):
        return int(lib().or_lm_word_id(self._h, _b(w)))

    def _ids(self, words):
        return np.array([self.word_id(w) for w in words], dtype=np.uint32)

    def score(self, words):
        """NGramModel.Score over the ids of `words` (ngram_model.go:44-62)"""
        ids = self._ids(words)
        return float(lib().or_lm_score(self._h, ids.ctypes.data, len(ids)))

    def score_sentence(self, words):
        """LanguageModel.ScoreSentence (language_model.go:66-86)"""
        ids = self._ids(words)
        return float(lib().or_lm_score_word_ids(self._h, ids.ctypes.data, len(ids)))

    def next_score(self, context, word, model_level=False):
        """Next(context).ScoreNext(word): (status, score); status 0 scorer / 1 nil scorer / 2 error"""
        ids = self._ids(context)
        st = C.c_int(0)
        f = lib().or_lm_model_next_score if model_level else lib().or_lm_next_score
        v = f(self._h, ids.ctypes.data, len(ids), self.word_id(word), C.byref(st))
        return st.value, float(v)

    def tokenize(self, text):
        raw = _b(text)
        buf = C.create_string_buffer(len(raw) * 2 + 64)
        n = lib().or_lm_tokenize(self._h, raw, len(raw), buf, len(buf))
        return buf.value.split(b"\n") if n else []

    def predict_batch(self, index, blob, offs, top_k, similarity, threads=0):
        """SpellChecker.Predict per query -> (ids [n, top_k+1], counts [n]); `index` is an OracleIndex over self.words()"""
        n = len(offs) - 1
        ids = np.zeros((n, top_k + 1), dtype=np.uint32)
        cnt = np.zeros(n, dtype=np.uint32)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lib().or_spell_predict_batch(index._h, self._h, blob.ctypes.data if blob.size else None, offs.ctypes.data, n, int(top_k),
                                     float(similarity), ids.ctypes.data, cnt.ctypes.data, threads or (os.cpu_count() or 1))
        return ids, cnt


def go_sort(keys):
    """the oracle's Go 1.14 sort.Sort restatement: permutation of the indices of `keys`"""
    k = np.ascontiguousarray(keys, dtype=np.uint32)
    out = np.zeros(len(k), dtype=np.uint32)
    lib().or_go_sort(k.ctypes.data, len(k), out.ctypes.data)
    return out.tolist()
