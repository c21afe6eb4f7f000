#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference's own test data (run in the authoring container,
where /root/reference exists; the GPU box only ever sees the committed outputs).

Only DATA is carried over (inputs and expected outputs the reference's tests hold):
  cars.dict, cars.hd, cars.dl, config.json   copied verbatim from pkg/suggest/testdata[/db]
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


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in ("cars.dict", "config.json"):
        shutil.copyfile(os.path.join(REF, name), os.path.join(OUT, name))
    for name in ("cars.hd", "cars.dl"):
        shutil.copyfile(os.path.join(REF, "db", name), os.path.join(OUT, name))
    for name in ("cars.dict", "config.json", "cars.hd", "cars.dl"):
        os.chmod(os.path.join(OUT, name), 0o644)
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
