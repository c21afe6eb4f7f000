"""Deterministic synthetic workloads of BASELINE.json / SURVEY.md §8(d).

Counter-based splitmix64: draw(seed, i, j) = mix(seed * 0x9E3779B97F4A7C15 + i * 64 + j), so the
generator vectorises and any slice of the dictionary can be produced independently.
  dict(N, seed=1):   doc i has length 8 + draw(seed,i,0) % 25 over [a-z0-9]; char c of doc i is
                     alphabet[byte (c%8) of draw(seed,i,1+c//8) % 36]
  queries(M, seed=2): query q = doc draw(seed,q,0) % N with 1 + draw(seed,q,1) % 2 random edits
                      (substitute / delete / insert at a random position, random symbol)
Normalisation is the identity on this alphabet, so no document repeats a term.

Variants of the dictionary, reported separately from the headline:
  skewed:    symbols drawn Zipf(s=1) over the 36-symbol alphabet instead of uniformly -> long posting lists at q=3
  families:  documents come in families of 1+F: a base string followed by F copies with 1-3 random edits each
             (substitute / delete / insert), so that a query has several near matches and top-k / ties are exercised
"""
import numpy as np

ALPHABET = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
DESCRIPTION = dict(ngram_size=3, wrap=("$", "$"), pad="$", alphabet=("english", "numbers", "$"))
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix(x):
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def draw(seed, i, j):
    with np.errstate(over="ignore"):
        return _mix(np.uint64(seed) * _G + np.asarray(i, dtype=np.uint64) * np.uint64(64) + np.asarray(j, dtype=np.uint64))


_ZIPF_CDF = np.cumsum(1.0 / np.arange(1, 37))
_ZIPF_CDF = (_ZIPF_CDF / _ZIPF_CDF[-1] * 65536.0).astype(np.uint32)   # thresholds on a 16-bit draw
_ZIPF_CDF[-1] = 65536
_W = 36                                                                  # padded row width (32 + up to 3 inserts)


def _base_rows(idx, seed, skewed):
    """-> (chars [m,_W] uint8 (garbage past the length), lengths [m] int64) of the base strings `idx`"""
    m = len(idx)
    L = 8 + (draw(seed, idx, 0) % np.uint64(25)).astype(np.int64)
    if not skewed:
        j = np.arange(4, dtype=np.uint64)
        words = draw(seed, idx[:, None], j[None, :] + np.uint64(1))          # [m,4] u64 -> 32 bytes per doc
        ch = ALPHABET[np.ascontiguousarray(words).view(np.uint8).reshape(m, 32) % np.uint8(36)]
    else:
        j = np.arange(8, dtype=np.uint64)
        words = draw(seed, idx[:, None], j[None, :] + np.uint64(1))          # [m,8] u64 -> 32 u16 per doc
        u = np.ascontiguousarray(words).view(np.uint16).reshape(m, 32).astype(np.uint32)
        ch = ALPHABET[np.searchsorted(_ZIPF_CDF, u, side="right")]
    rows = np.zeros((m, _W), dtype=np.uint8)
    rows[:, :32] = ch
    return rows, L


def _edit_rows(rows, L, idx, seed, max_edits=3):
    """1 + draw % max_edits random edits per row, vectorised (draw columns 16..)"""
    m = len(idx)
    n_edits = 1 + (draw(seed, idx, 16) % np.uint64(max_edits)).astype(np.int64)
    col = np.arange(_W, dtype=np.int64)[None, :]
    for e in range(max_edits):
        act = n_edits > e
        kind = (draw(seed, idx, 17 + 3 * e) % np.uint64(3)).astype(np.int64)          # 0 substitute, 1 delete, 2 insert
        sym = ALPHABET[(draw(seed, idx, 18 + 3 * e) % np.uint64(36)).astype(np.int64)]
        r = draw(seed, idx, 19 + 3 * e)
        kind = np.where((kind == 1) & (L <= 1), 2, kind)
        pos = np.where(kind == 2, r % (L + 1).astype(np.uint64), r % np.maximum(L, 1).astype(np.uint64)).astype(np.int64)
        dele = act & (kind == 1)
        ins = act & (kind == 2)
        sub = act & (kind == 0)
        src = col + (dele[:, None] & (col >= pos[:, None])) - (ins[:, None] & (col > pos[:, None]))
        rows = np.take_along_axis(rows, np.clip(src, 0, _W - 1), axis=1)
        w = ins | sub
        rows[np.nonzero(w)[0], pos[w]] = sym[w]
        L = L + ins.astype(np.int64) - dele.astype(np.int64)
    return rows, L


def make_dict(n, seed=1, chunk=1 << 20, skewed=False, families=0):
    """-> (blob uint8, offs uint64[n+1]);  skewed / families: the variants described in the module docstring"""
    lens = np.empty(n, dtype=np.int64)
    parts = []
    F1 = families + 1
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = np.arange(s, e, dtype=np.uint64)
        base = idx - idx % np.uint64(F1) if families else idx
        rows, L = _base_rows(base, seed, skewed)
        if families:
            member = (idx % np.uint64(F1)) != 0
            er, eL = _edit_rows(rows[member], L[member], idx[member], seed)
            rows[member] = er
            L[member] = eL
        lens[s:e] = L
        mask = np.arange(_W)[None, :] < L[:, None]
        parts.append(rows[mask])
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens).astype(np.uint64)
    return (np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)), offs


def make_queries(m, blob, offs, seed=2, start=0):
    """-> (blob uint8, offs uint64[m+1]); query ids start..start+m-1 (lets ranks draw disjoint batches)"""
    n = len(offs) - 1
    qi = np.arange(start, start + m, dtype=np.uint64)
    docs = (draw(seed, qi, 0) % np.uint64(n)).astype(np.int64)
    n_edits = 1 + (draw(seed, qi, 1) % np.uint64(2)).astype(np.int64)
    r = draw(seed, qi[:, None], np.arange(2, 10, dtype=np.uint64)[None, :])
    out = []
    offs_i = offs.astype(np.int64)
    for q in range(m):
        s = bytearray(blob[offs_i[docs[q]]:offs_i[docs[q] + 1]].tobytes())
        for e in range(int(n_edits[q])):
            kind = int(r[q, 3 * e] % np.uint64(3))
            sym = int(ALPHABET[int(r[q, 3 * e + 2] % np.uint64(36))])
            if kind == 0 and s:
                s[int(r[q, 3 * e + 1] % np.uint64(len(s)))] = sym
            elif kind == 1 and len(s) > 1:
                del s[int(r[q, 3 * e + 1] % np.uint64(len(s)))]
            else:
                s.insert(int(r[q, 3 * e + 1] % np.uint64(len(s) + 1)), sym)
        out.append(bytes(s))
    qoffs = np.zeros(m + 1, dtype=np.uint64)
    qoffs[1:] = np.cumsum([len(x) for x in out]).astype(np.uint64)
    return np.frombuffer(b"".join(out), dtype=np.uint8).copy(), qoffs


def unpack(blob, offs):
    o = offs.astype(np.int64)
    b = blob.tobytes()
    return [b[o[i]:o[i + 1]] for i in range(len(o) - 1)]
