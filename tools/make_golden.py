#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference's own test data (run in the authoring container,
where /root/reference exists; the GPU box only ever sees the committed outputs).

Only DATA is carried over (inputs and expected outputs the reference's tests hold):
  cars.dict, config.json, db/cars.{hd,dl,cdb} copied verbatim from pkg/suggest/testdata[/db]
  db/words_subset.{hd,dl}                    a SUBSET of pkg/suggest/testdata/db/words.{hd,dl}: the lists of a few edge
                                             segments plus the longest (roaring-coded) lists, posting-list bytes
                                             verbatim, header re-encoded as gob with the original type messages
  words.dict.xz                              pkg/suggest/testdata/words.dict, xz-compressed
  words_index_digest.json                    digest of pkg/suggest/testdata/db/words.{hd,dl}
                                             decoded with tests/refindex.py (the 4 MB .dl is not committed)
reference_tests.json (hand-transcribed expectations of the Go unit tests) is not generated here.
"""
import hashlib, json, lzma, os, random, shutil, struct, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refindex  # noqa: E402

REF = "/root/reference/pkg/suggest/testdata"
OUT = os.path.join(ROOT, "tests", "golden")


def segment_digests(lists):
    """lists: {(seg, term_bytes): (raw_len, postings)} -> {seg: (n_lists, n_postings, sha256hex)}"""
    per = {}
    for (seg, term) in sorted(lists):
        raw_len, post = lists[(seg, term)]
        h = per.setdefault(seg, [0, 0, hashlib.sha256()])
        h[0] += 1
        h[1] += len(post)
        h[2].update(struct.pack("<II", seg, len(term)) + term + struct.pack("<II", raw_len, len(post)))
        h[2].update(struct.pack("<%dI" % len(post), *post))
    return {str(s): [v[0], v[1], v[2].hexdigest()] for s, v in per.items()}


def _gob_uint(v):
    if v < 128:
        return bytes([v])
    b = v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def _gob_int(v):
    return _gob_uint((~v << 1) | 1 if v < 0 else v << 1)


def write_words_subset():
    """db/words_subset.{hd,dl}: edge segments + the 40 longest lists of words.{hd,dl} (all three codecs)."""
    hd = open(os.path.join(REF, "db", "words.hd"), "rb").read()
    dl = open(os.path.join(REF, "db", "words.dl"), "rb").read()
    version, indices, terms = refindex.read_header(os.path.join(REF, "db", "words.hd"))
    # type-definition messages (negative type ids) are copied verbatim; find where the value message starts
    g = refindex._Gob(memoryview(hd))
    type_id = None
    while g.i < len(hd):
        start = g.i
        n = g.uint()
        end = g.i + n
        tid = g.int_()
        if tid >= 0:
            type_id, value_start = tid, start
            break
        g.i = end
    longest = sorted(terms, key=lambda t: -t[4])[:40]
    keep = [t for t in terms if t[1] in (1, 2, 3, 4, 21, 22, 23, 24)] + longest
    keep = sorted(set(keep), key=lambda t: (t[1], t[0]))
    out_dl = bytearray()
    body = bytearray()
    body += _gob_int(type_id)
    body += _gob_uint(1) + _gob_uint(len(version)) + version.encode()        # field 0 Version
    body += _gob_uint(1) + _gob_uint(indices)                                 # field 1 Indices
    body += _gob_uint(1) + _gob_uint(len(keep))                               # field 2 Terms
    for term, indice, size, pos, length in keep:
        new_pos = len(out_dl)
        out_dl += dl[pos:pos + size]
        rec = bytearray()
        f = -1
        for idx, val in enumerate((term, indice, size, new_pos, length)):
            if idx == 0:
                if len(val) == 0:
                    continue
                rec += _gob_uint(idx - f) + _gob_uint(len(val)) + val
            else:
                if val == 0:
                    continue
                rec += _gob_uint(idx - f) + _gob_uint(val)
            f = idx
        body += rec + b"\x00"
    body += b"\x00"
    with open(os.path.join(OUT, "db", "words_subset.hd"), "wb") as f:
        f.write(hd[:value_start] + _gob_uint(len(body)) + bytes(body))
    with open(os.path.join(OUT, "db", "words_subset.dl"), "wb") as f:
        f.write(bytes(out_dl))
    # self-check with the reader
    n_idx, lists = refindex.read_index(os.path.join(OUT, "db", "words_subset.hd"), os.path.join(OUT, "db", "words_subset.dl"))
    assert n_idx == indices and len(lists) == len(keep)
    classes = [sum(1 for v in lists.values() if lo <= v[0] <= hi) for lo, hi in ((0, 65), (66, 256), (257, 1 << 30))]
    print("words subset:", len(keep), "lists, codec classes", classes, "dl bytes", len(out_dl))


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in ("cars.dict", "config.json"):
        shutil.copyfile(os.path.join(REF, name), os.path.join(OUT, name))
    os.makedirs(os.path.join(OUT, "db"), exist_ok=True)
    for name in ("cars.hd", "cars.dl", "cars.cdb"):
        shutil.copyfile(os.path.join(REF, "db", name), os.path.join(OUT, "db", name))
    for name in ("cars.dict", "config.json", "db/cars.hd", "db/cars.dl", "db/cars.cdb"):
        os.chmod(os.path.join(OUT, name), 0o644)
    # pkg/lm/testdata/fixtures: the Google-format n-gram count files the reference's LM tests read (data, not source)
    os.makedirs(os.path.join(OUT, "lm"), exist_ok=True)
    LM_REF = os.path.join(os.path.dirname(REF), "..", "lm", "testdata", "fixtures")
    for name in ("1-gm", "2-gm", "3-gm"):
        shutil.copyfile(os.path.join(LM_REF, name), os.path.join(OUT, "lm", name))
        os.chmod(os.path.join(OUT, "lm", name), 0o644)
    write_words_subset()
    with open(os.path.join(REF, "words.dict"), "rb") as f:
        raw = f.read()
    with open(os.path.join(OUT, "words.dict.xz"), "wb") as f:
        f.write(lzma.compress(raw, preset=9))
    n_idx, lists = refindex.read_index(os.path.join(REF, "db", "words.hd"), os.path.join(REF, "db", "words.dl"))
    rng = random.Random(20260928)
    keys = sorted(lists)
    sample = rng.sample(keys, 300)
    # make sure the three storage classes are all represented
    for lo, hi in ((0, 65), (66, 256), (257, 1 << 30)):
        cls = [k for k in keys if lo <= lists[k][0] <= hi]
        sample += rng.sample(cls, min(20, len(cls)))
    digest = {
        "source": "pkg/suggest/testdata/db/words.{hd,dl}",
        "n_indices": n_idx,
        "n_lists": len(lists),
        "n_postings_raw": sum(v[0] for v in lists.values()),
        "n_postings_stored": sum(len(v[1]) for v in lists.values()),
        "segments": segment_digests(lists),
        "samples": [[k[0], k[1].hex(), lists[k][0], lists[k][1]] for k in sorted(set(sample))],
    }
    with open(os.path.join(OUT, "words_index_digest.json"), "w") as f:
        json.dump(digest, f, separators=(",", ":"))
    print("wrote", OUT, {k: digest[k] for k in ("n_indices", "n_lists", "n_postings_raw")})


if __name__ == "__main__":
    main()
