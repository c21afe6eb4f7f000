"""The spellchecker caller of the fuzzy-search path: pkg/lm's n-gram language model and
pkg/spellchecker.SpellChecker.Predict over the MI355X engine.

  lm = LanguageModel(directory, order=3, start_symbol="<S>", end_symbol="</S>", alphabet=(...))   # Google n-gram files
  sc = SpellChecker(lm)                      # builds + uploads the fuzzy index over the model's vocabulary
  sc.Predict("i am sa", 5, 0.5)              # -> ["sam", ...]            (spellchecker.go:40-92)
"""
import ctypes as C

import numpy as np

from . import _lib
from .index import IndexDescription, NGramIndex, _c_desc, _enc, pack_strings

# cmd/spellchecker/cmd/eval.go:16-23
SPELLCHECKER_INDEX = dict(name="words", ngram_size=3, wrap=("^", "$"), pad="$", alphabet=("english", "russian", "numbers", "$^'"))


class LanguageModel:
    """lm.LanguageModel (pkg/lm/language_model.go) loaded from <dir>/{1..order}-gm (pkg/lm/ngram_reader.go)."""

    def __init__(self, directory=None, order=3, start_symbol="<S>", end_symbol="</S>", alphabet=("english", "russian", "numbers", "-."),
                 id_order="lines", binary=None, dictionary=None):
        """directory + id_order: Google count files; "lines" numbers the words by 1-gm line (the reference's test indexer,
        indexer.go:88-114), "count" by (count desc, word asc) like the production build (binary.go:101-199).
        binary + dictionary: RetrieveLMFromBinary (binary.go:59-98) from <name>.lm and <name>.cdb."""
        alpha = (C.c_char_p * len(alphabet))(*[_enc(a) for a in alphabet])
        h = C.c_void_p()
        L = _lib.lib()
        if binary is not None:
            _lib.check(L.sg_lm_load_binary(_enc(binary), _enc(dictionary), _enc(start_symbol), _enc(end_symbol), alpha, len(alphabet), C.byref(h)))
        else:
            _lib.check(L.sg_lm_load_google_ex(_enc(directory), int(order), _enc(start_symbol), _enc(end_symbol), alpha, len(alphabet),
                                              {"lines": 0, "count": 1}[id_order], C.byref(h)))
        self._h = h
        self.order = int(L.sg_lm_order(h))

    def level(self, i):
        """-> (containers u64[], values u64[], total): level i in the reference's packed form (packed_array.go:12-16)"""
        L = _lib.lib()
        nc, nv, tot = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(L.sg_lm_level(self._h, i, None, 0, C.byref(nc), None, 0, C.byref(nv), C.byref(tot)))
        c = np.zeros(nc.value, dtype=np.uint64)
        v = np.zeros(nv.value, dtype=np.uint64)
        _lib.check(L.sg_lm_level(self._h, i, c.ctypes.data, nc.value, C.byref(nc), v.ctypes.data, nv.value, C.byref(nv), C.byref(tot)))
        return c, v, int(tot.value)

    @staticmethod
    def build_files(text, directory, order=3, start_symbol="<S>", end_symbol="</S>", alphabet=("english", "russian", "numbers", "-."),
                    separators=("\n",)):
        """`lm build-lm`: count the k-grams of a corpus into <directory>/{1..order}-gm (pkg/lm/ngram_builder.go, ngram_writer.go)"""
        raw = _enc(text)
        alpha = (C.c_char_p * len(alphabet))(*[_enc(a) for a in alphabet])
        seps = (C.c_char_p * len(separators))(*[_enc(a) for a in separators])
        _lib.check(_lib.lib().sg_lm_build_google(raw, len(raw), int(order), _enc(start_symbol), _enc(end_symbol), alpha, len(alphabet),
                                                 seps, len(separators), _enc(directory)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().sg_lm_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(_lib.lib().sg_lm_num_words(self._h))

    def word(self, i):
        buf = C.create_string_buffer(256)
        n = _lib.lib().sg_lm_word(self._h, int(i), buf, 256)
        if n > 256:
            buf = C.create_string_buffer(n)
            _lib.lib().sg_lm_word(self._h, int(i), buf, n)
        return buf.raw[:n]

    def words(self):
        return [self.word(i) for i in range(len(self))]

    def GetWordID(self, token):                                # indexer.go:50-63 (0xFFFFFFFF = unknown)
        t = _enc(token)
        return int(_lib.lib().sg_lm_word_id(self._h, t, len(t)))

    def _ids(self, words):
        return np.array([self.GetWordID(w) for w in words], dtype=np.uint32)

    def Score(self, words):                                    # NGramModel.Score
        ids = self._ids(words)
        return float(_lib.lib().sg_lm_score(self._h, ids.ctypes.data, len(ids)))

    def ScoreSentence(self, words):                            # LanguageModel.ScoreSentence
        ids = self._ids(words)
        return float(_lib.lib().sg_lm_score_word_ids(self._h, ids.ctypes.data, len(ids)))

    def next_score(self, context, word, model_level=False):
        """Next(context).ScoreNext(word) -> (status, score); status 0 scorer, 1 nil scorer, 2 error"""
        ids = self._ids(context)
        v = C.c_double(0)
        st = _lib.lib().sg_lm_next_score(self._h, ids.ctypes.data, len(ids), self.GetWordID(word), 1 if model_level else 0, C.byref(v))
        return int(st), float(v.value)

    def Tokenize(self, text):                                  # lm.NewTokenizer(alphabet).Tokenize
        raw = _enc(text)
        buf = C.create_string_buffer(len(raw) * 2 + 64)
        n = _lib.lib().sg_lm_tokenize(self._h, raw, len(raw), buf, len(buf))
        return buf.value.split(b"\n") if n else []


class SpellChecker:
    """spellchecker.SpellChecker (pkg/spellchecker/spellchecker.go:14-37) wired like
    internal/spellchecker/dep/spellchecker.go:14-53: fuzzy index over the model's vocabulary, docID = word id."""

    def __init__(self, model, description=None, device=0):
        self.model = model
        self.description = description or IndexDescription(**SPELLCHECKER_INDEX)
        desc = _c_desc(self.description)
        h = C.c_void_p()
        _lib.check(_lib.lib().sg_spell_index_build(model._h, C.byref(desc), int(device), C.byref(h)))
        self.index = NGramIndex(description=self.description, device=device, upload=False, _handle=h)
        self.index.device = int(device)

    def predict_batch(self, queries=None, top_k=5, similarity=0.5, blob=None, offs=None):
        """-> (ids uint32 [n, top_k + 1], counts uint32 [n]) — word ids of the predictions, best first"""
        if blob is None:
            blob, offs = pack_strings(queries)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        ids = np.zeros((n, int(top_k) + 1), dtype=np.uint32)
        cnt = np.zeros(n, dtype=np.uint32)
        _lib.check(_lib.lib().sg_spell_predict_batch(self.index._h, self.model._h, blob.ctypes.data if blob.size else None, offs.ctypes.data,
                                                     n, int(top_k), float(similarity), ids.ctypes.data, cnt.ctypes.data))
        return ids, cnt

    def predict_batch_device(self, d_blob, d_offs, n_q, q_bytes, top_k, similarity, d_ids, d_counts, stream=0):
        """sg_spell_predict_batch_device: raw device pointers (torch tensors' data_ptr()), asynchronous on `stream`"""
        _lib.check(_lib.lib().sg_spell_predict_batch_device(self.index._h, self.model._h, d_blob, d_offs, int(n_q), int(q_bytes), int(top_k),
                                                            float(similarity), d_ids, d_counts, stream))

    def Predict(self, query, topK, similarity):
        ids, cnt = self.predict_batch([query], topK, similarity)
        c = int(cnt[0])
        if c == _lib.SG_COUNT_REF_PANIC:
            raise RuntimeError("reference behaviour: panic: makechan: size out of range (suggester.go:62)")
        if c == _lib.SG_COUNT_REF_DEADLOCK:
            raise RuntimeError("reference behaviour: deadlock (suggester.go:62)")
        if c == _lib.SG_COUNT_LM_ERROR:
            raise ValueError("nGrams length should be less than the nGramModel order")
        if c == _lib.SG_COUNT_TOO_LONG:
            raise ValueError("query word has more than SG_MAX_QUERY_TERMS n-grams")
        return [self.model.word(int(i)).decode("utf-8", "replace") for i in ids[0, :c]]
