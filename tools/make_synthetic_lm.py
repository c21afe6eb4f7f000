#!/usr/bin/env python3
"""Synthetic language model for BASELINE config 5 ("50M-token LM n-gram index"), written in the reference's own
production formats: <dir>/synth.lm (nGramModel.Store, pkg/lm/ngram_model.go:100-121 + packed_array.go:96-116) and
<dir>/synth.cdb (BuildCDBDictionary, pkg/dictionary/helpers.go:52-100) — what `lm build-lm` leaves for
RetrieveLMFromBinary (pkg/lm/binary.go:59-98).  (The MPH table the reference appends to the .lm is not written: neither
loader here reads it.)

Corpus: sentences of 6..21 words, words drawn Zipf(s) from a vocabulary of random lower-case strings; every sentence is
wrapped in <S> .. </S> and its 1/2/3-grams are counted (ngram_builder.go:39-64).  Word ids: count descending, word
ascending (binary.go:141-199).  Deterministic (numpy PCG64 seeds).  ~40 s and ~6 GB of host memory at 50M tokens."""
import argparse
import os
import struct
import sys

import numpy as np

NO_CONTEXT = 0xFFFFFFFD


def make(directory, tokens=50_000_000, vocab=1_000_000, zipf=1.07, seed=7, verbose=True):
    os.makedirs(directory, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(seed))
    log = (lambda *a: print("[lm]", *a, file=sys.stderr, flush=True)) if verbose else (lambda *a: None)
    # ---- vocabulary: distinct random strings of 3..12 letters
    need = int(vocab * 1.3) + 64                               # (short strings collide: oversample, then cut)
    ln = rng.integers(3, 13, size=need)
    chars = rng.integers(0, 26, size=(need, 12), dtype=np.uint8) + ord("a")
    chars[np.arange(12)[None, :] >= ln[:, None]] = 0
    raw = np.unique(np.ascontiguousarray(chars).view("S12").ravel())
    raw = raw[rng.permutation(len(raw))[:vocab]]
    vocab = len(raw)
    raw = np.concatenate([raw, np.array([b"<S>", b"</S>"], dtype="S12")])
    V = vocab
    S_ID, E_ID = V, V + 1
    # ---- corpus
    n_sent = max(1, tokens // 14)
    lens = rng.integers(6, 22, size=n_sent)
    n_words = int(lens.sum())
    w = 1.0 / np.arange(1, V + 1) ** zipf
    cdf = np.cumsum(w); cdf /= cdf[-1]
    draws = np.searchsorted(cdf, rng.random(n_words)).astype(np.uint32)
    perm = rng.permutation(V).astype(np.uint32)                 # rank -> raw word (so that ids are not alphabetical)
    draws = perm[draws]
    total = n_words + 2 * n_sent
    T = np.empty(total, dtype=np.uint32)
    starts = np.concatenate([[0], np.cumsum(lens + 2)[:-1]])
    T[starts] = S_ID
    T[starts + lens + 1] = E_ID
    mask = np.ones(total, dtype=bool); mask[starts] = False; mask[starts + lens + 1] = False
    T[mask] = draws
    del draws, mask
    log("corpus: %d sentences, %d tokens incl. markers" % (n_sent, total))
    # ---- word ids by (count desc, word asc)
    cnt1 = np.bincount(T, minlength=V + 2).astype(np.int64)
    order = np.lexsort((raw, -cnt1))
    order = order[cnt1[order] > 0]
    new_id = np.full(V + 2, 0xFFFFFFFF, dtype=np.uint32)
    new_id[order] = np.arange(len(order), dtype=np.uint32)
    words = raw[order]
    T = new_id[T]
    e_new = int(new_id[E_ID])
    n1 = len(order)
    c1 = cnt1[order].astype(np.uint64)
    lv1_vals = (np.arange(n1, dtype=np.uint64) << np.uint64(32)) | c1
    lv1_cont = np.array([(NO_CONTEXT << 32) | 0], dtype=np.uint64)
    # ---- bigrams / trigrams inside sentences (a window must not start at or run over a </S>)
    a, b = T[:-1].astype(np.uint64), T[1:].astype(np.uint64)
    ok = T[:-1] != e_new
    k2, c2 = np.unique((a[ok] << np.uint64(32)) | b[ok], return_counts=True)
    log("bigrams: %d distinct" % len(k2))
    ctx2 = (k2 >> np.uint64(32)).astype(np.uint64)              # context offset of a unigram = its id
    lv2_vals = ((k2 & np.uint64(0xFFFFFFFF)) << np.uint64(32)) | c2.astype(np.uint64)
    first2 = np.nonzero(np.concatenate([[True], ctx2[1:] != ctx2[:-1]]))[0]
    lv2_cont = (ctx2[first2] << np.uint64(32)) | first2.astype(np.uint64)
    ok3 = (T[:-2] != e_new) & (T[1:-1] != e_new)
    ab = (T[:-2].astype(np.uint64)[ok3] << np.uint64(32)) | T[1:-1].astype(np.uint64)[ok3]
    ctx3 = np.searchsorted(k2, ab).astype(np.uint64)             # offset of the bigram (a, b) in level 2
    k3, c3 = np.unique((ctx3 << np.uint64(32)) | T[2:].astype(np.uint64)[ok3], return_counts=True)
    del ab, ctx3
    log("trigrams: %d distinct" % len(k3))
    c3ctx = k3 >> np.uint64(32)
    lv3_vals = ((k3 & np.uint64(0xFFFFFFFF)) << np.uint64(32)) | c3.astype(np.uint64)
    first3 = np.nonzero(np.concatenate([[True], c3ctx[1:] != c3ctx[:-1]]))[0]
    lv3_cont = (c3ctx[first3] << np.uint64(32)) | first3.astype(np.uint64)
    with open(os.path.join(directory, "synth.lm"), "wb") as f:
        f.write(b"0.0.2" + bytes([3]))
        for cont, vals, tot in ((lv1_cont, lv1_vals, int(c1.sum())), (lv2_cont, lv2_vals, int(c2.sum())), (lv3_cont, lv3_vals, int(c3.sum()))):
            f.write(("%d %d %d\n" % (len(cont) * 8, len(vals) * 8, tot & 0xFFFFFFFF)).encode())
            f.write(cont.astype("<u8").tobytes()); f.write(vals.astype("<u8").tobytes())
    # ---- <name>.cdb: key = id (4 bytes LE), value = word
    wl = [x for x in words.tolist()]
    keys = np.arange(n1, dtype=np.uint32)
    kb = keys.view(np.uint8).reshape(n1, 4).astype(np.uint32)
    h = np.full(n1, 5381, dtype=np.uint32)
    for j in range(4):
        h = (((h << np.uint32(5)) + h) ^ kb[:, j]).astype(np.uint32)
    recs = bytearray()
    pos = np.empty(n1, dtype=np.uint32)
    p = 2048
    for i, wd in enumerate(wl):
        pos[i] = p
        recs += struct.pack("<II", 4, len(wd)) + struct.pack("<I", i) + wd
        p += 12 + len(wd)
    tables = bytearray()
    header = bytearray()
    tpos = p
    slot_of = h & np.uint32(255)
    order_t = np.argsort(slot_of, kind="stable")
    bounds = np.searchsorted(slot_of[order_t], np.arange(257))
    for t in range(256):
        idx = order_t[bounds[t]:bounds[t + 1]]
        n = len(idx) * 2
        header += struct.pack("<II", tpos if n else 0, n)
        if not n:
            continue
        slots = [(0, 0)] * n
        for i in idx.tolist():
            s = (int(h[i]) >> 8) % n
            while slots[s][1]:
                s = (s + 1) % n
            slots[s] = (int(h[i]), int(pos[i]))
        for hv, pv in slots:
            tables += struct.pack("<II", hv, pv)
        tpos += 8 * n
    with open(os.path.join(directory, "synth.cdb"), "wb") as f:
        f.write(bytes(header)); f.write(bytes(recs)); f.write(bytes(tables))
    log("wrote %s (%d words; .lm %d MB)" % (directory, n1, os.path.getsize(os.path.join(directory, "synth.lm")) >> 20))
    return {"words": n1, "tokens": int(total), "bigrams": int(len(k2)), "trigrams": int(len(k3)), "corpus_sample": T[: 4_000_000].copy(), "end_id": e_new,
            "start_id": int(new_id[S_ID]), "word_list": wl}


def make_queries(info, n_q, seed):
    """cfg 5's query stream: two context words of a corpus position + the next word cut to a prefix (2 of 3) or with a typo
    (1 of 3) -> list[bytes]"""
    T, words = info["corpus_sample"], info["word_list"]
    rng = np.random.Generator(np.random.PCG64(seed))
    markers = (info["start_id"], info["end_id"])
    out = []
    while len(out) < n_q:
        p = int(rng.integers(2, len(T)))
        a, b, c = int(T[p - 2]), int(T[p - 1]), int(T[p])
        if a in markers or b in markers or c in markers:
            continue
        w = words[c]
        if len(out) % 3 == 2 and len(w) > 3:
            j = int(rng.integers(1, len(w)))
            w = w[:j] + bytes([ord("a") + int(rng.integers(0, 26))]) + w[j + 1:]
        else:
            w = w[:max(2, (len(w) * 2 + 2) // 3)]
        out.append(words[a] + b" " + words[b] + b" " + w)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--tokens", type=int, default=50_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    a = ap.parse_args()
    r = make(a.directory, a.tokens, a.vocab)
    print({k: v for k, v in r.items() if k not in ("corpus_sample", "word_list")})
