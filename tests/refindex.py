"""Pure-Python reader of the reference's on-disk index files (<name>.hd / <name>.dl).

Test infrastructure: it lets the parity tests compare an index built from a dictionary by
the oracle (and by the HIP engine's host builder) with the index files the reference itself
committed under pkg/suggest/testdata/db — the strongest pin available without a Go toolchain.

Formats (reference file:line, paths relative to /root/reference):
  * header  = encoding/gob stream of `header{Version string; Indices uint32; Terms []termDescription}`
              with `termDescription{Term string; Indice, PostingListBytesSize, PostingListPosition,
              PostingListLen uint32}`            pkg/index/indexer_writer.go:50-63,148-167
  * lists   = by raw length (pkg/index/codec.go:39-51): <=65 VB deltas (compression/varint.go:36-55),
              <=256 skip blocks of 64 (compression/skipping.go:67-113: u16 LE block byte length incl.
              itself, bit 15 = last block; first value of a block is a delta to the previous block's
              first value), else a roaring bitmap in the portable serialisation (compression/bitmap.go:18-29).
"""
import struct


class _Gob:
    def __init__(self, buf):
        self.b = buf
        self.i = 0

    def uint(self):
        c = self.b[self.i]
        self.i += 1
        if c < 128:
            return c
        n = 256 - c
        v = int.from_bytes(self.b[self.i:self.i + n], "big")
        self.i += n
        return v

    def int_(self):
        u = self.uint()
        return ~(u >> 1) if u & 1 else u >> 1

    def string(self):
        n = self.uint()
        s = self.b[self.i:self.i + n]
        self.i += n
        return bytes(s)


def read_header(path):
    """-> (version:str, n_indices:int, terms:list[(term:bytes, indice, bytes_size, position, length)])"""
    data = memoryview(open(path, "rb").read())
    g = _Gob(data)
    while g.i < len(data):
        n = g.uint()
        end = g.i + n
        type_id = g.int_()
        if type_id < 0:          # type definition message: skip
            g.i = end
            continue
        # value message of the header struct: delta-encoded fields, zero values omitted
        version, indices, terms = "", 0, []
        field = -1
        while True:
            d = g.uint()
            if d == 0:
                break
            field += d
            if field == 0:
                version = g.string().decode()
            elif field == 1:
                indices = g.uint()
            elif field == 2:
                cnt = g.uint()
                for _ in range(cnt):
                    rec = [b"", 0, 0, 0, 0]
                    f = -1
                    while True:
                        dd = g.uint()
                        if dd == 0:
                            break
                        f += dd
                        rec[f] = g.string() if f == 0 else g.uint()
                    terms.append(tuple(rec))
            else:
                raise ValueError("unexpected header field %d" % field)
        assert g.i == end, (g.i, end)
        return version, indices, terms
    raise ValueError("no value message in gob stream")


def _varints(buf, i, end):
    out = []
    while i < end:
        v = 0
        s = 0
        while True:
            b = buf[i]
            i += 1
            v |= (b & 0x7F) << s
            s += 7
            if b < 0x80:
                break
        out.append(v)
    return out


def decode_vb(buf, length):
    deltas = _varints(buf, 0, len(buf))
    assert len(deltas) == length, (len(deltas), length)
    out, prev = [], 0
    for d in deltas:
        prev = (prev + d) & 0xFFFFFFFF
        out.append(prev)
    return out


def decode_skipping(buf, length, gap=64):
    out = []
    i = 0
    block_first = 0
    while True:
        (packed,) = struct.unpack_from("<H", buf, i)
        size, last = packed & 0x7FFF, bool(packed & 0x8000)
        deltas = _varints(buf, i + 2, i + size)
        prev = block_first
        first = True
        for d in deltas:
            prev = (prev + d) & 0xFFFFFFFF
            if first:
                block_first = prev
                first = False
            out.append(prev)
        assert len(deltas) <= gap
        i += size
        if last:
            break
    assert i == len(buf) and len(out) == length, (i, len(buf), len(out), length)
    return out


def decode_roaring(buf):
    """RoaringBitmap portable serialisation -> ascending list of uint32."""
    (cookie,) = struct.unpack_from("<I", buf, 0)
    i = 4
    run_flags = None
    if cookie & 0xFFFF == 12347:
        n = (cookie >> 16) + 1
        nb = (n + 7) // 8
        run_flags = bytes(buf[i:i + nb])
        i += nb
    elif cookie == 12346:
        (n,) = struct.unpack_from("<I", buf, i)
        i += 4
    else:
        raise ValueError("bad roaring cookie %x" % cookie)
    keys = []
    for _ in range(n):
        k, c = struct.unpack_from("<HH", buf, i)
        keys.append((k, c + 1))
        i += 4
    if run_flags is None or n >= 4:
        i += 4 * n                       # offset header
    out = []
    for idx, (k, card) in enumerate(keys):
        base = k << 16
        is_run = run_flags is not None and (run_flags[idx // 8] >> (idx % 8)) & 1
        if is_run:
            (nr,) = struct.unpack_from("<H", buf, i)
            i += 2
            for _ in range(nr):
                s, l = struct.unpack_from("<HH", buf, i)
                i += 4
                out.extend(range(base + s, base + s + l + 1))
        elif card > 4096:
            words = struct.unpack_from("<1024Q", buf, i)
            i += 8192
            for w, word in enumerate(words):
                while word:
                    t = word & -word
                    out.append(base + w * 64 + t.bit_length() - 1)
                    word ^= t
        else:
            vals = struct.unpack_from("<%dH" % card, buf, i)
            i += 2 * card
            out.extend(base + v for v in vals)
    return out


def decode_list(buf, length):
    """Decode one stored posting list given its raw (header) length."""
    if length <= 65:
        return decode_vb(buf, length)
    if length <= 256:
        return decode_skipping(buf, length)
    return decode_roaring(buf)


def read_index(hd_path, dl_path):
    """-> (n_indices, {(indice, term_bytes): (raw_len, [postings as stored])})"""
    version, indices, terms = read_header(hd_path)
    assert version == "v5.1", version
    dl = memoryview(open(dl_path, "rb").read())
    lists = {}
    for term, indice, size, pos, length in terms:
        lists[(indice, term)] = (length, decode_list(dl[pos:pos + size], length))
    return indices, lists


# ---- writer (tests only): lets a test lay down <name>.hd/.dl in the reference's format for ANY set of lists, e.g. a
#      roaring-coded list that dropped a document's repeated postings (codec.go:39-51) — the reference's fixtures hold none ----
def _gob_uint(v):
    if v < 128:
        return bytes([v])
    b = v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def _gob_int(v):
    return _gob_uint((~v << 1) | 1 if v < 0 else v << 1)


def _enc_varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def encode_vb(postings):
    out, prev = bytearray(), 0
    for p in postings:
        out += _enc_varint(p - prev)
        prev = p
    return bytes(out)


def encode_skipping(postings, gap=64):
    out = bytearray()
    block_first = 0
    blocks = [postings[i:i + gap] for i in range(0, len(postings), gap)]
    for bi, blk in enumerate(blocks):
        body = bytearray()
        prev = block_first
        for j, p in enumerate(blk):
            body += _enc_varint(p - prev)
            prev = p
            if j == 0:
                block_first = p
        size = len(body) + 2
        assert size < 0x8000
        out += struct.pack("<H", size | (0x8000 if bi == len(blocks) - 1 else 0)) + body
    return bytes(out)


def encode_roaring(values):
    """portable format without run containers (cookie 12346): array containers up to 4096 values, bitmaps above"""
    cont = {}
    for v in values:
        cont.setdefault(v >> 16, []).append(v & 0xFFFF)
    keys = sorted(cont)
    head = struct.pack("<II", 12346, len(keys))
    for k in keys:
        head += struct.pack("<HH", k, len(cont[k]) - 1)
    bodies = []
    for k in keys:
        vals = cont[k]
        if len(vals) > 4096:
            words = [0] * 1024
            for v in vals:
                words[v >> 6] |= 1 << (v & 63)
            bodies.append(struct.pack("<1024Q", *words))
        else:
            bodies.append(struct.pack("<%dH" % len(vals), *vals))
    off = len(head) + 4 * len(keys)
    offs = b""
    for b in bodies:
        offs += struct.pack("<I", off)
        off += len(b)
    return head + offs + b"".join(bodies)


def write_index(hd_path, dl_path, n_indices, lists, type_prefix_from):
    """lists: {(indice, term_bytes): (raw_len, stored postings)} as read_index returns them.  The gob type-definition
    messages are taken verbatim from an existing header file (`type_prefix_from`)."""
    hd = open(type_prefix_from, "rb").read()
    g = _Gob(memoryview(hd))
    while g.i < len(hd):
        start = g.i
        n = g.uint()
        end = g.i + n
        tid = g.int_()
        if tid >= 0:
            type_id, value_start = tid, start
            break
        g.i = end
    dl = bytearray()
    body = bytearray()
    body += _gob_int(type_id)
    body += _gob_uint(1) + _gob_uint(4) + b"v5.1"
    body += _gob_uint(1) + _gob_uint(n_indices)
    body += _gob_uint(1) + _gob_uint(len(lists))
    for (indice, term) in sorted(lists):
        raw_len, post = lists[(indice, term)]
        enc = encode_vb(post) if raw_len <= 65 else encode_skipping(post) if raw_len <= 256 else encode_roaring(post)
        pos = len(dl)
        dl += enc
        f = -1
        rec = bytearray()
        for idx, val in enumerate((term, indice, len(enc), pos, raw_len)):
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
    with open(hd_path, "wb") as f:
        f.write(hd[:value_start] + _gob_uint(len(body)) + bytes(body))
    with open(dl_path, "wb") as f:
        f.write(bytes(dl))
