"""NGramIndex on the GPU: host CSR build + HBM replica + batched Suggest / Autocomplete.

Mirrors suggest.Builder / suggest.NGramIndex (pkg/suggest/ngram_index_builder.go:14-83, ngram_index.go:7-35)
for the hot path; every call goes through the C ABI of include/suggest_hip.h.
"""
import contextlib
import ctypes as C
import threading

import numpy as np

from . import _lib
from .metric import resolve


def pack_strings(strings):
    """list[str|bytes] -> (uint8 blob, uint64 offsets[n+1])"""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strings]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return blob, offs


class IndexDescription:
    """pkg/suggest/config.go:25-35"""

    def __init__(self, name="index", ngram_size=3, wrap=("$", "$"), pad="$", alphabet=("english", "numbers", "$"),
                 driver="RAM", source=None, output=None):
        self.name, self.ngram_size, self.wrap, self.pad = name, int(ngram_size), tuple(wrap), pad
        self.alphabet, self.driver, self.source, self.output = tuple(alphabet), driver, source, output

    @classmethod
    def from_json(cls, d):
        return cls(name=d.get("name", "index"), ngram_size=d["nGramSize"], wrap=tuple(d["wrap"]), pad=d["pad"],
                   alphabet=tuple(d["alphabet"]), driver=d.get("driver", "RAM"), source=d.get("source"), output=d.get("output"))


def _enc(s):
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def _c_desc(d):
    alpha = (C.c_char_p * len(d.alphabet))(*[_enc(a) for a in d.alphabet])
    desc = _lib.SgDesc(d.ngram_size, _enc(d.wrap[0]), _enc(d.wrap[1]), _enc(d.pad), alpha, len(d.alphabet))
    desc._keep = alpha
    return desc


class MetricTables:
    """sg_metric_tables: an opaque metric.Metric at one similarity, tabulated and resident in HBM."""

    def __init__(self, handle, a_max):
        self._h, self.a_max = handle, a_max

    def close(self):
        h, self._h = self._h, None
        if h:
            _lib.lib().sg_metric_tables_release(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _max_terms(offs, description):
    """upper bound of the n-grams of the longest query of a batch (bytes + wrap runes)"""
    longest = int((offs[1:] - offs[:-1]).max()) if len(offs) > 1 else 0
    # (capped: the tables are dense — (a_max + 1)^2 x segments doubles — and filled by a Python loop; a longer query comes back
    #  SG_COUNT_TOO_LONG from a tabulated launch)
    return min(256, longest + len(description.wrap[0]) + len(description.wrap[1]) + 1)


class Ticket:
    """A batch in flight (sg_ticket).  wait() must be called exactly once; the buffers of the submit are kept alive here."""

    def __init__(self, handle, keep):
        self._t, self._keep = handle, keep

    def wait(self):
        t, self._t = self._t, None
        if t is None:
            raise ValueError("ticket already waited for")
        try:
            _lib.check(_lib.lib().sg_ticket_wait(t))
        finally:
            self._keep = None

    def __del__(self):
        if getattr(self, "_t", None) is not None:     # never waited for: the engine still owns the slot and a reference
            try:
                _lib.lib().sg_ticket_wait(self._t)
            except Exception:
                pass


def pinned_array(shape, dtype):
    """numpy array over pinned host memory (sg_host_alloc): the buffers of sg_*_submit that need no staging copy.  The
    memory is handed back (sg_host_free) when the last view of the array goes."""
    import weakref
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    n = max(count * dtype.itemsize, 16)
    p = C.c_void_p()
    _lib.check(_lib.lib().sg_host_alloc(n, C.byref(p)))
    buf = (C.c_uint8 * n).from_address(p.value)
    weakref.finalize(buf, _lib.lib().sg_host_free, p.value)
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


class NGramIndex:
    def __init__(self, docs=None, description=None, blob=None, offs=None, device=0, upload=True, _handle=None, build="host", min_segments=0):
        """build="host": sg_index_build (CPU tokenise + CSR); build="device": sg_index_build_device (same arrays, built on the
        GPU `device`; documents with more than 128 n-grams are not supported there)."""
        L = _lib.lib()
        d = description or IndexDescription()
        self.description = d
        self.device = None
        self._hlock = threading.Lock()
        if _handle is not None:
            self._h = _handle
            self.n_docs = self.stats()["n_docs"]
        else:
            if blob is None:
                blob, offs = pack_strings(docs)
            blob = np.ascontiguousarray(blob, dtype=np.uint8)
            offs = np.ascontiguousarray(offs, dtype=np.uint64)
            self.n_docs = len(offs) - 1
            desc = _c_desc(d)
            h = C.c_void_p()
            if build not in ("host", "device"):
                raise ValueError("build must be 'host' or 'device'")
            _lib.check(L.sg_index_build_ex(blob.ctypes.data if blob.size else None, offs.ctypes.data, self.n_docs, C.byref(desc),
                                           int(min_segments), int(device) if build == "device" else -1, C.byref(h)))
            self._h = h
        if upload:
            self.upload(device)

    @classmethod
    def from_reference_files(cls, hd_path, dl_path, description, device=0, upload=True):
        """NewFSBuilder (pkg/suggest/ngram_index_builder.go:44-52): open an index the reference built
        (<name>.hd gob header + <name>.dl posting lists)."""
        desc = _c_desc(description)
        h = C.c_void_p()
        _lib.check(_lib.lib().sg_index_load_reference(_enc(hd_path), _enc(dl_path), C.byref(desc), C.byref(h)))
        return cls(description=description, device=device, upload=upload, _handle=h)

    def digest(self):
        """64-bit digests of the host CSR arrays (postings, seg_off, list lengths, term keys + repeated-term table)"""
        out = (C.c_uint64 * 4)()
        with self._use() as h:
            _lib.check(_lib.lib().sg_index_digest(h, out))
        return tuple(int(x) for x in out)

    def upload(self, device=0):
        with self._use() as h:
            _lib.check(_lib.lib().sg_index_upload(h, int(device)))
        if self.device is None:
            self.device = int(device)
        return self

    def replicate(self, devices):
        """sg_index_replicate: one host build, a replica in the HBM of every listed GPU."""
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        with self._use() as h:
            _lib.check(_lib.lib().sg_index_replicate(h, arr, len(devices)))
        if self.device is None and devices:
            self.device = int(devices[0])
        return self

    def forward(self, first, n, cap=160):
        """-> (card[n], n_terms[n], keys[n, cap]) of the device's forward index (doc -> distinct term keys)"""
        card = np.zeros(n, dtype=np.uint32); nt = np.zeros(n, dtype=np.uint32); keys = np.zeros((n, cap), dtype=np.uint64)
        with self._use() as h:
            _lib.check(_lib.lib().sg_index_forward(h, int(first), int(n), int(cap), card.ctypes.data, nt.ctypes.data, keys.ctypes.data))
        return card, nt, keys

    def tune(self, **knobs):
        """sg_index_tune: e.g. tune(SG_T_FLOOR=6, SG_FILTER_LEVEL=5) — results never depend on the knobs"""
        with self._use() as h:
            for k, v in knobs.items():
                _lib.check(_lib.lib().sg_index_tune(h, k.encode(), int(v)))
        return self

    def replicas(self):
        out = (C.c_int * 64)()
        with self._use() as h:
            n = _lib.lib().sg_index_replicas(h, out, 64)
        return [int(out[i]) for i in range(min(n, 64))]

    @contextlib.contextmanager
    def _use(self):
        """The handle, retained for the duration of a C call: a concurrent close() (Service re-indexing while queries are in
        flight — ctypes releases the GIL) only drops its own reference; the index is freed when the last call returns."""
        L = _lib.lib()
        with self._hlock:
            h = self._h
            if not h:
                raise ValueError("index is closed")
            L.sg_index_retain(h)
        try:
            yield h
        finally:
            L.sg_index_release(h)

    def close(self):
        lock = getattr(self, "_hlock", None)
        if lock is None:
            return
        with lock:
            h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().sg_index_release(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- search: host buffers ------------------------------------------------------------
    def suggest_batch(self, queries=None, metric="jaccard", similarity=0.5, k=10, blob=None, offs=None, multi=False, tables=None):
        """-> (ids[n_q,k] u32, scores[n_q,k] f64, counts[n_q] u32); row i best first.  multi=True: sg_suggest_batch_multi
        (the batch sliced over every replica)."""
        if blob is None:
            blob, offs = pack_strings(queries)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n_q = len(offs) - 1
        ids = np.zeros((n_q, k), dtype=np.uint32)
        sc = np.zeros((n_q, k), dtype=np.float64)
        cnt = np.zeros(n_q, dtype=np.uint32)
        L = _lib.lib()
        m = resolve(metric)
        if m.code is None or tables is not None:      # a Metric implementation the device has no twin of: tabulated (sg_suggest_batch_tables)
            tb = tables or self.metric_tables(m, similarity, _max_terms(offs, self.description))
            with self._use() as h:
                _lib.check(L.sg_suggest_batch_tables(h, blob.ctypes.data if blob.size else None, offs.ctypes.data, n_q, tb._h, int(k),
                                                     ids.ctypes.data, sc.ctypes.data, cnt.ctypes.data))
            return ids, sc, cnt
        with self._use() as h:
            _lib.check((L.sg_suggest_batch_multi if multi else L.sg_suggest_batch)(
                h, blob.ctypes.data if blob.size else None, offs.ctypes.data, n_q, m.code, float(similarity), int(k),
                ids.ctypes.data, sc.ctypes.data, cnt.ctypes.data))
        return ids, sc, cnt

    def metric_tables(self, metric, similarity, a_max):
        """MinY / MaxY / Threshold / 1 - Distance of `metric` (any object with the four methods of metric.Metric,
        pkg/metric/metric.go:7-16) at `similarity`, tabulated for queries of up to a_max n-grams and uploaded
        (sg_metric_tables_create).  Reusable across calls; released with the object."""
        m = resolve(metric)
        a_max = int(max(1, a_max))
        S = int(self.stats()["n_segments"])
        alpha = float(similarity)
        n_a = a_max + 1
        i32 = lambda v: int(max(-2**31, min(2**31 - 1, v)))                                  # noqa: E731
        min_y = np.array([i32(m.MinY(alpha, a)) for a in range(n_a)], dtype=np.int32)
        max_y = np.array([i32(m.MaxY(alpha, a)) for a in range(n_a)], dtype=np.int32)
        thr = np.zeros((n_a, S), dtype=np.int32)
        score = np.zeros((n_a, S, n_a), dtype=np.float64)
        for a in range(1, n_a):
            for b in range(max(0, int(min_y[a])), min(S - 1, int(max_y[a])) + 1):
                thr[a, b] = i32(m.Threshold(alpha, a, b))
                for o in range(max(0, int(thr[a, b])), a + 1):          # (a query that repeats a term: overlaps up to a)
                    try:
                        score[a, b, o] = 1 - m.Distance(o, a, b)          # metricScorer.Score, pkg/suggest/scorer.go:29-31
                    except ZeroDivisionError:
                        score[a, b, o] = float("nan")
        t = C.c_void_p()
        with self._use() as h:
            _lib.check(_lib.lib().sg_metric_tables_create(h, a_max, min_y.ctypes.data, max_y.ctypes.data, thr.ctypes.data, score.ctypes.data, C.byref(t)))
        return MetricTables(t, a_max)

    def suggest_batch_from(self, queries=None, metric="jaccard", similarity=0.5, first_doc=0, limit=1024, blob=None, offs=None, tables=None):
        """nGramSuggester.Suggest for ANY collector (suggester.go:78-99): per query the `limit` smallest docIDs >= first_doc
        among ALL documents whose overlap reaches their segment's threshold -> (ids, scores, aux, counts); aux = segment << 16
        | overlap.  sg_suggest_batch_from."""
        if blob is None:
            blob, offs = pack_strings(queries)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n_q = len(offs) - 1
        ids = np.zeros((n_q, limit), dtype=np.uint32)
        sc = np.zeros((n_q, limit), dtype=np.float64)
        aux = np.zeros((n_q, limit), dtype=np.uint32)
        cnt = np.zeros(n_q, dtype=np.uint32)
        m = resolve(metric)
        if m.code is None and tables is None:
            tables = self.metric_tables(m, similarity, _max_terms(offs, self.description))
        with self._use() as h:
            _lib.check(_lib.lib().sg_suggest_batch_from(h, blob.ctypes.data if blob.size else None, offs.ctypes.data, n_q, m.code or 0, float(similarity),
                                                        tables._h if tables is not None else None, int(first_doc), int(limit), ids.ctypes.data,
                                                        sc.ctypes.data, aux.ctypes.data, cnt.ctypes.data))
        return ids, sc, aux, cnt

    def suggest_all(self, query, metric="jaccard", similarity=0.5, page=1024, tables=None):
        """EVERY candidate of one query, ascending docID: [(docID, score, overlap, segment)] — what the reference's Suggest
        streams to the caller's collector — paged through sg_suggest_batch_from.  A document that repeats a term can appear
        several times; a page that ends inside such a run is resumed AT that docID and the entries already
        delivered are skipped."""
        out, first, skip = [], 0, 0
        m = resolve(metric)
        if m.code is None and tables is None:
            tables = self.metric_tables(m, similarity, len(_enc(query)) + 16)
        while True:
            ids, sc, aux, cnt = self.suggest_batch_from([query], m, similarity, first, page, tables=tables)
            c = int(cnt[0])
            if c >= _lib.SG_COUNT_LM_ERROR:
                raise ValueError("query cannot be answered (status %#x)" % c)
            n = min(c, page)
            rows = [(int(ids[0, j]), float(sc[0, j]), int(aux[0, j]) & 0xFFFF, int(aux[0, j]) >> 16) for j in range(n)]
            out.extend(rows[skip:])
            if n < page:
                return out
            last = rows[-1][0]
            run = sum(1 for r in rows if r[0] == last)     # (rows ascend: the entries of the last document are the page's tail)
            if run == n:                                    # a whole page of one document: ask again with a larger page
                del out[len(out) - (n - skip):]
                page *= 2
                continue
            first, skip = last, run                         # resume AT the last docID, past its entries already delivered

    def autocomplete_batch(self, queries=None, limit=10, blob=None, offs=None, multi=False):
        if blob is None:
            blob, offs = pack_strings(queries)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n_q = len(offs) - 1
        ids = np.zeros((n_q, limit), dtype=np.uint32)
        cnt = np.zeros(n_q, dtype=np.uint32)
        L = _lib.lib()
        with self._use() as h:
            _lib.check((L.sg_autocomplete_batch_multi if multi else L.sg_autocomplete_batch)(
                h, blob.ctypes.data if blob.size else None, offs.ctypes.data, n_q, int(limit), ids.ctypes.data, cnt.ctypes.data))
        return ids, cnt

    def autocomplete_all(self, query, page=1024):
        """EVERY document the prefix completes, ascending docID — what the reference's Autocomplete hands to the caller's
        collector (pkg/suggest/autocomplete.go:40-77) — paged through sg_autocomplete_one_from."""
        q = _enc(query)
        out, first, skip = [], 0, 0
        cnt = C.c_uint32()
        while True:
            ids = np.zeros(page, dtype=np.uint32)
            with self._use() as h:
                _lib.check(_lib.lib().sg_autocomplete_one_from(h, q, len(q), int(first), int(page), ids.ctypes.data, C.addressof(cnt)))
            c = int(cnt.value)
            if c >= _lib.SG_COUNT_LM_ERROR:
                raise ValueError("query cannot be answered (status %#x)" % c)
            n = min(c, page)
            rows = [int(x) for x in ids[:n]]
            out.extend(rows[skip:])
            if n < page:
                return out
            # a document that repeats a term is emitted once per secondary match: a page that ends inside such a run is resumed
            # AT that docID, past the entries of it already delivered (resuming at last + 1 dropped the rest of the run)
            last = rows[-1]
            run = sum(1 for r in rows if r == last)
            if run == n:
                del out[len(out) - (n - skip):]
                page *= 2
                continue
            first, skip = last, run

    # ---- search: host buffers, asynchronous (sg_suggest_submit / sg_ticket_wait) -----------------
    def suggest_submit(self, blob, offs, metric, similarity, k, ids, scores, counts, replica=0):
        """Enqueues copy in -> search -> copy out for one batch and returns a ticket; `ticket.wait()` blocks until the rows
        are in ids / scores / counts.  All six arrays must stay alive and untouched until then; arrays from `pinned_array`
        are read / written by the DMA engine directly.  Two tickets in flight hide PCIe behind the kernel.  `replica`: which
        replica of the index runs it (0 = the primary) — a slice per GPU from one host thread."""
        n_q = len(offs) - 1
        assert offs.dtype == np.uint64 and blob.dtype == np.uint8 and ids.dtype == np.uint32 and scores.dtype == np.float64 and counts.dtype == np.uint32
        assert ids.size >= n_q * k and scores.size >= n_q * k and counts.size >= n_q
        t = C.c_void_p()
        with self._use() as h:
            code = resolve(metric).code
            if code is None:
                raise ValueError("suggest_submit takes the metrics with a device twin (jaccard, cosine, dice, exact, overlap); tabulate others with metric_tables + suggest_batch")
            _lib.check(_lib.lib().sg_suggest_submit_on(h, int(replica), blob.ctypes.data if blob.size else None, offs.ctypes.data, n_q, code,
                                                       float(similarity), int(k), ids.ctypes.data, scores.ctypes.data, counts.ctypes.data, C.byref(t)))
        return Ticket(t, (blob, offs, ids, scores, counts))

    def autocomplete_submit(self, blob, offs, limit, ids, counts, first_doc=0):
        n_q = len(offs) - 1
        assert offs.dtype == np.uint64 and blob.dtype == np.uint8 and ids.dtype == np.uint32 and counts.dtype == np.uint32
        t = C.c_void_p()
        with self._use() as h:
            _lib.check(_lib.lib().sg_autocomplete_submit(h, blob.ctypes.data if blob.size else None, offs.ctypes.data, n_q, int(first_doc), int(limit),
                                                         ids.ctypes.data, counts.ctypes.data, C.byref(t)))
        return Ticket(t, (blob, offs, ids, counts))

    # ---- search: device-resident buffers (raw pointers; torch tensors' data_ptr()) ---------
    def suggest_batch_device(self, d_blob, d_offs, n_q, metric, similarity, k, d_ids, d_scores, d_counts, stream=0):
        with self._use() as h:
            _lib.check(_lib.lib().sg_suggest_batch_device(h, d_blob, d_offs, int(n_q), resolve(metric).code, float(similarity),
                                                          int(k), d_ids, d_scores, d_counts, stream))

    def autocomplete_batch_device(self, d_blob, d_offs, n_q, limit, d_ids, d_counts, stream=0):
        with self._use() as h:
            _lib.check(_lib.lib().sg_autocomplete_batch_device(h, d_blob, d_offs, int(n_q), int(limit), d_ids, d_counts, stream))

    # ---- NGramIndex interface (single query) -----------------------------------------------
    def suggest(self, query, similarity, metric, k):
        """Suggester.Suggest (pkg/suggest/suggester.go:17-20) -> list[(docID, score)] best first."""
        q = _enc(query)
        ids = np.zeros((1, k), dtype=np.uint32)
        sc = np.zeros((1, k), dtype=np.float64)
        cnt = C.c_uint32()
        m = resolve(metric)
        if m.code is None:          # a Metric implementation of the caller's: tabulated (sg_suggest_batch_tables)
            ids, sc, cn = self.suggest_batch([q], m, similarity, k)
            cnt = C.c_uint32(int(cn[0]))
        else:
            with self._use() as h:      # one query per call: coalesced with the other callers' (sg_suggest_one)
                _lib.check(_lib.lib().sg_suggest_one(h, q, len(q), m.code, float(similarity), int(k), ids.ctypes.data,
                                                     sc.ctypes.data, C.addressof(cnt)))
        c = int(cnt.value)
        if c == _lib.SG_COUNT_REF_PANIC:
            raise RuntimeError("query window is empty: the reference panics here (suggester.go:62, negative channel capacity)")
        if c == _lib.SG_COUNT_REF_DEADLOCK:
            raise RuntimeError("query window is empty: the reference dead-locks here (suggester.go:62, zero channel capacity)")
        if c == _lib.SG_COUNT_TOO_LONG:
            raise ValueError("query has more than %d n-grams" % _lib.SG_MAX_QUERY_TERMS)
        return [(int(ids[0, i]), float(sc[0, i])) for i in range(c)]

    def autocomplete(self, query, limit):
        q = _enc(query)
        ids = np.zeros((1, limit), dtype=np.uint32)
        cnt = C.c_uint32()
        with self._use() as h:
            _lib.check(_lib.lib().sg_autocomplete_one(h, q, len(q), int(limit), ids.ctypes.data, C.addressof(cnt)))
        c = int(cnt.value)
        if c == _lib.SG_COUNT_TOO_LONG:
            raise ValueError("query has more than %d n-grams" % _lib.SG_MAX_QUERY_TERMS)
        return [int(ids[0, i]) for i in range(c)]

    # ---- introspection ---------------------------------------------------------------------
    def launch_stats(self):
        """sampled launch counters (cumulative; full, sampled, results mod 2^32, chunks 64 bits wide): {full, sampled, results, chunks} — sg_index_launch_stats"""
        out = (C.c_uint64 * 4)()
        with self._use() as h:
            _lib.check(_lib.lib().sg_index_launch_stats(h, out))
        return {"full": int(out[0]), "sampled": int(out[1]), "results": int(out[2]), "chunks": int(out[3])}

    def pipe_stats(self):
        """queries the plan -> stream -> verify pipeline left to the fused kernel (cumulative): {unplanned, overflow, repeats}, and `queries`: those of all launches that took the pipeline — sg_index_pipe_stats"""
        out = (C.c_uint64 * 4)()
        with self._use() as h:
            _lib.check(_lib.lib().sg_index_pipe_stats(h, out))
        return {"unplanned": int(out[0]), "overflow": int(out[1]), "repeats": int(out[2]), "queries": int(out[3])}

    def pipe_volumes(self):
        """the pipeline's sampled volumes (cumulative): {sampled, groups, lists, rows, candidates} + the store's chunks, descriptor width and the stream workgroup of the latest launch — sg_index_pipe_volumes"""
        out = (C.c_uint64 * 8)()
        with self._use() as h:
            _lib.check(_lib.lib().sg_index_pipe_volumes(h, out))
        return {"sampled": int(out[0]), "groups": int(out[1]), "lists": int(out[2]), "rows": int(out[3]), "candidates": int(out[4]),
                "packed_chunks": int(out[5]), "wide": bool(out[6]),
                "stream_shape": ("2 wavefronts x 2^11 counters", "4 x 2^12", "8 x 2^13", "fixed by knobs")[min(int(out[7]), 3)]}

    def stats(self):
        st = _lib.SgStats()
        with self._use() as h:
            _lib.check(_lib.lib().sg_index_stats(h, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def tokenize_keys(self, text, autocomplete=False):
        t = _enc(text)
        cap = 4 * len(t) + 64
        buf = np.zeros(cap, dtype=np.uint64)
        n = _lib.check(_lib.lib().sg_tokenize(self._h, t, len(t), 1 if autocomplete else 0, buf.ctypes.data, cap))
        return [int(x) for x in buf[:n]]

    def term_string(self, key):
        buf = C.create_string_buffer(64)
        n = _lib.check(_lib.lib().sg_term_string(self._h, int(key), buf, 64))
        return buf.raw[:n]

    def tokenize(self, text, autocomplete=False):
        return [self.term_string(k) for k in self.tokenize_keys(text, autocomplete)]

    def lists(self):
        """-> {(segment, term_bytes): (raw_len, [stored docIDs])} from the host CSR"""
        L = _lib.lib()
        n = L.sg_index_lists(self._h, None, None, 0)
        segs = np.zeros(n, dtype=np.uint32)
        keys = np.zeros(n, dtype=np.uint64)
        L.sg_index_lists(self._h, segs.ctypes.data, keys.ctypes.data, n)
        out = {}
        raw = C.c_uint64()
        buf = np.zeros(1 << 16, dtype=np.uint32)
        for s, k in zip(segs.tolist(), keys.tolist()):
            ln = L.sg_index_list(self._h, s, k, buf.ctypes.data, buf.size, C.byref(raw))
            if ln > buf.size:
                buf = np.zeros(int(ln), dtype=np.uint32)
                ln = L.sg_index_list(self._h, s, k, buf.ctypes.data, buf.size, C.byref(raw))
            out[(s, self.term_string(k))] = (int(raw.value), buf[:ln].tolist())
        return out

    def algorithmic_bytes(self, blob, offs, metric, similarity, k):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        out = C.c_uint64()
        with self._use() as h:
            _lib.check(_lib.lib().sg_suggest_algorithmic_bytes(h, blob.ctypes.data if blob.size else None, offs.ctypes.data,
                                                               len(offs) - 1, resolve(metric).code, float(similarity), int(k), C.byref(out)))
        return int(out.value)
