"""suggest.Service — the reference's public API (pkg/suggest/service.go:20-173) over the MI355X engine.

Same method names, argument meaning and error behaviour as the Go type, plus additive *Batch methods
(the GPU earns its keep on batches).  Dictionaries stay host side (docID -> string, line order), as in
pkg/dictionary/memory_dictionary.go.
"""
import json
import os
import threading

from .index import IndexDescription, NGramIndex
from . import _lib
from .metric import resolve


class ResultItem:
    """pkg/suggest/service.go:12-17"""
    __slots__ = ("score", "value")

    def __init__(self, score, value):
        self.score, self.value = score, value

    def __repr__(self):
        return "ResultItem(score=%r, value=%r)" % (self.score, self.value)

    def __eq__(self, o):
        return isinstance(o, ResultItem) and (self.score, self.value) == (o.score, o.value)


class SearchConfig:
    """NewSearchConfig — pkg/suggest/search.go:18-35 (same validation, same messages)"""

    def __init__(self, query, top_k, metric, similarity):
        if top_k <= 0:
            raise ValueError("topK should be greater or equal to 1")
        if similarity <= 0 or similarity > 1:
            raise ValueError("similarity shouble be in (0.0, 1.0]")
        self.query, self.top_k, self.metric, self.similarity = query, int(top_k), resolve(metric), float(similarity)


def read_dictionary(path):
    """OpenRAMDictionary — pkg/dictionary/helpers.go:25-48 (bufio.Scanner lines; docID = line number)"""
    with open(path, "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    return [l[:-1] if l.endswith(b"\r") else l for l in lines]


def read_cdb_dictionary(path):
    """OpenCDBDictionary (pkg/dictionary/helpers.go:14-22, cdb_dictionary.go): D. J. Bernstein's constant database
    written by BuildCDBDictionary (helpers.go:52-100) with key = docID as 4-byte little endian, value = the string.
    Records start at byte 2048 and run to the first hash table; returned in docID order."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 2048:
        raise IOError("failed to open cdb dictionary file: %s" % path)
    tables = [struct.unpack_from("<II", data, 8 * i) for i in range(256)]
    end = min([pos for pos, n in tables if n] or [len(data)])            # (an empty table carries no position)
    out = {}
    pos = 2048
    while pos < end:
        klen, dlen = struct.unpack_from("<II", data, pos)
        key = data[pos + 8:pos + 8 + klen]
        out[struct.unpack("<I", key)[0] if klen == 4 else None] = data[pos + 8 + klen:pos + 8 + klen + dlen]
        pos += 8 + klen + dlen
    n = len(out)
    return [out.get(i, b"") for i in range(n)]


def read_configs(path):
    """ReadConfigs — pkg/suggest/config.go:84-112 (paths relative to the config file)"""
    with open(path, encoding="utf-8") as f:
        raw = json.load(f)
    base = os.path.dirname(os.path.abspath(path))
    out = []
    for d in raw:
        desc = IndexDescription.from_json(d)
        if desc.source and not os.path.isabs(desc.source):
            desc.source = os.path.join(base, desc.source)
        if desc.output and not os.path.isabs(desc.output):
            desc.output = os.path.join(base, desc.output)
        out.append(desc)
    return out


class Service:
    def __init__(self, device=0):
        self._lock = threading.RLock()
        self._indexes = {}
        self._dictionaries = {}
        self.device = device

    # -- AddIndexByDescription / AddRunTimeIndex / AddIndex (service.go:35-91) --
    def add_index_by_description(self, description):
        if description.driver == "RAM":
            return self.add_run_time_index(description)
        return self.add_on_disc_index(description)

    def add_on_disc_index(self, description):
        """AddOnDiscIndex (service.go:61-75): <output>/<name>.cdb dictionary + <name>.hd/.dl built by the reference's
        indexer; the posting lists are decoded once and kept as CSR in HBM."""
        base = os.path.join(description.output, description.name)
        dictionary = read_cdb_dictionary(base + ".cdb")
        index = NGramIndex.from_reference_files(base + ".hd", base + ".dl", description, device=self.device)
        return self._install(description.name, index, dictionary)

    def add_run_time_index(self, description):
        if not description.source or not os.path.exists(description.source):
            raise IOError("failed to create RAMDriver builder: open %s: no such file" % description.source)
        return self.add_index(description.name, read_dictionary(description.source), description)

    def add_index(self, name, dictionary, description):
        """dictionary: sequence of str/bytes, docID = position (dictionary.NewInMemoryDictionary)"""
        index = NGramIndex(dictionary, description, device=self.device)
        return self._install(name, index, dictionary)

    def _install(self, name, index, dictionary):
        with self._lock:                       # service.go:85-88
            old = self._indexes.get(name)
            self._indexes[name] = index
            self._dictionaries[name] = list(dictionary)
        if old is not None:
            old.close()                        # handle is reference counted on the C side
        return None

    def get_dictionaries(self):
        with self._lock:
            return list(self._dictionaries)

    def _lookup(self, name):
        with self._lock:
            index, dictionary = self._indexes.get(name), self._dictionaries.get(name)
        if index is None or dictionary is None:
            raise KeyError("given dictionary %s is not exists" % name)   # service.go:111-113
        return index, dictionary

    @staticmethod
    def _value(dictionary, doc_id):
        v = dictionary[doc_id] if 0 <= doc_id < len(dictionary) else ""
        return v.decode("utf-8", "replace") if isinstance(v, (bytes, bytearray)) else v

    # -- Suggest / Autocomplete (service.go:105-173) --
    def suggest(self, dict_name, config):
        index, dictionary = self._lookup(dict_name)
        cands = index.suggest(config.query, config.similarity, config.metric, config.top_k)
        return [ResultItem(score, self._value(dictionary, doc)) for doc, score in cands]

    def autocomplete(self, dict_name, query, limit):
        index, dictionary = self._lookup(dict_name)
        return [ResultItem(0, self._value(dictionary, doc)) for doc in index.autocomplete(query, limit)]

    # -- additive batch API --
    def suggest_batch(self, dict_name, queries, top_k, metric, similarity):
        SearchConfig("", top_k, metric, similarity)
        index, dictionary = self._lookup(dict_name)
        ids, sc, cnt = index.suggest_batch(queries, metric, similarity, top_k)
        out = []
        for i in range(len(queries)):
            c = int(cnt[i])
            if c >= _lib.SG_COUNT_TOO_LONG:
                out.append(None)
                continue
            out.append([ResultItem(float(sc[i, j]), self._value(dictionary, int(ids[i, j]))) for j in range(c)])
        return out

    # Go-style aliases so reference call sites read the same
    AddIndexByDescription = add_index_by_description
    AddRunTimeIndex = add_run_time_index
    AddOnDiscIndex = add_on_disc_index
    AddIndex = add_index
    GetDictionaries = get_dictionaries
    Suggest = suggest
    Autocomplete = autocomplete
