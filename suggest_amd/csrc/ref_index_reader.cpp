// ref_index_reader.cpp — loads an index the reference itself built (its DISC driver files) into the
// engine's CSR layout:
//   <name>.hd  encoding/gob stream of header{Version string; Indices uint32; Terms []termDescription{Term string;
//              Indice, PostingListBytesSize, PostingListPosition, PostingListLen uint32}}
//              — pkg/index/indexer_writer.go:50-63,148-167, read by pkg/index/index_reader.go:57-120
//   <name>.dl  posting lists, codec chosen by raw length (pkg/index/codec.go:39-51):
//              <= 65  VB deltas            pkg/compression/varint.go:36-78
//              <= 256 skip blocks of 64    pkg/compression/skipping.go:67-151 (u16 LE block length incl. itself,
//                                          bit 15 = last block; a block's first value is a delta to the previous block's first)
//              else   roaring bitmap, portable serialisation (pkg/compression/bitmap.go:18-29; RoaringBitmap/roaring v0.5.5)
// Only the byte formats are restated here; no reference code is used.

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>

#include "sg_internal.h"

namespace sg {

namespace {

struct Reader {
  const uint8_t* p; size_t n, i = 0; bool ok = true;
  uint64_t gob_uint() {
    if (i >= n) { ok = false; return 0; }
    uint8_t c = p[i++];
    if (c < 128) return c;
    int cnt = 256 - c;
    if (cnt > 8 || i + cnt > n) { ok = false; return 0; }
    uint64_t v = 0;
    for (int k = 0; k < cnt; k++) v = (v << 8) | p[i++];
    return v;
  }
  int64_t gob_int() { uint64_t u = gob_uint(); return (u & 1) ? ~(int64_t)(u >> 1) : (int64_t)(u >> 1); }
  std::string gob_string() {
    uint64_t len = gob_uint();
    if (!ok || len > n - i) { ok = false; return {}; }       // (no i + len: an untrusted length may wrap)
    std::string s((const char*)p + i, (size_t)len);
    i += len;
    return s;
  }
};

struct TermDesc { std::string term; uint32_t indice = 0, size = 0, pos = 0, len = 0; };

bool read_file(const char* path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz < 0) { fclose(f); return false; }
  out.resize((size_t)sz);
  bool ok = sz == 0 || fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
  fclose(f);
  return ok;
}

bool parse_header(const std::vector<uint8_t>& buf, std::string& version, uint32_t& indices, std::vector<TermDesc>& terms) {
  Reader r{buf.data(), buf.size()};
  while (r.ok && r.i < r.n) {
    uint64_t len = r.gob_uint();
    if (!r.ok || len > r.n - r.i) return false;
    size_t end = r.i + (size_t)len;
    int64_t type_id = r.gob_int();
    if (type_id < 0) { r.i = end; continue; }            // type definition message
    int field = -1;
    for (;;) {                                             // header struct: delta-encoded fields, zero values omitted
      uint64_t d = r.gob_uint();
      if (!r.ok) return false;
      if (d == 0) break;
      field += (int)d;
      if (field == 0) version = r.gob_string();
      else if (field == 1) indices = (uint32_t)r.gob_uint();
      else if (field == 2) {
        uint64_t cnt = r.gob_uint();
        if (!r.ok || cnt > r.n - r.i) return false;            // every term description takes at least a byte
        terms.reserve((size_t)cnt);
        for (uint64_t k = 0; k < cnt && r.ok; k++) {
          TermDesc td;
          int f = -1;
          for (;;) {
            uint64_t dd = r.gob_uint();
            if (!r.ok || dd == 0) break;
            f += (int)dd;
            if (f == 0) td.term = r.gob_string();
            else { uint32_t v = (uint32_t)r.gob_uint(); if (f == 1) td.indice = v; else if (f == 2) td.size = v; else if (f == 3) td.pos = v; else if (f == 4) td.len = v; else return false; }
          }
          terms.push_back(std::move(td));
        }
      } else return false;
    }
    return r.ok && r.i == end;
  }
  return false;
}

bool varints(const uint8_t* b, size_t beg, size_t end, std::vector<uint32_t>& out) {
  size_t i = beg;
  while (i < end) {
    uint32_t v = 0; int s = 0;
    for (;;) {
      if (i >= end || s > 28) return false;
      uint8_t c = b[i++];
      v |= (uint32_t)(c & 0x7F) << s; s += 7;
      if (c < 0x80) break;
    }
    out.push_back(v);
  }
  return true;
}

bool decode_vb(const uint8_t* b, size_t n, uint32_t len, std::vector<uint32_t>& out) {
  std::vector<uint32_t> d;
  if (!varints(b, 0, n, d) || d.size() != len) return false;
  uint32_t prev = 0;
  for (uint32_t x : d) { prev += x; out.push_back(prev); }
  return true;
}

bool decode_skipping(const uint8_t* b, size_t n, uint32_t len, std::vector<uint32_t>& out) {
  size_t i = 0;
  uint32_t block_first = 0;
  for (;;) {
    if (i + 2 > n) return false;
    uint16_t packed = (uint16_t)(b[i] | (b[i + 1] << 8));
    size_t size = packed & 0x7FFF;
    bool last = packed & 0x8000;
    if (size < 2 || i + size > n) return false;
    std::vector<uint32_t> d;
    if (!varints(b, i + 2, i + size, d)) return false;
    uint32_t prev = block_first;
    bool first = true;
    for (uint32_t x : d) { prev += x; if (first) { block_first = prev; first = false; } out.push_back(prev); }
    i += size;
    if (last) break;
  }
  return i == n && out.size() == len;
}

bool decode_roaring(const uint8_t* b, size_t n, std::vector<uint32_t>& out) {
  auto u16 = [&](size_t o) { return (uint32_t)(b[o] | (b[o + 1] << 8)); };
  auto u32 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8) | ((uint32_t)b[o + 2] << 16) | ((uint32_t)b[o + 3] << 24); };
  if (n < 8) return false;
  uint32_t cookie = u32(0);
  size_t i = 4;
  uint32_t cnt;
  std::vector<uint8_t> run_flags;
  bool has_runs = false;
  if ((cookie & 0xFFFF) == 12347) {
    cnt = (cookie >> 16) + 1; has_runs = true;
    size_t nb = (cnt + 7) / 8;
    if (i + nb > n) return false;
    run_flags.assign(b + i, b + i + nb); i += nb;
  } else if (cookie == 12346) { cnt = u32(i); i += 4; }
  else return false;
  if (i + 4ull * cnt > n) return false;
  std::vector<std::pair<uint32_t, uint32_t>> keys;
  for (uint32_t k = 0; k < cnt; k++) { keys.push_back({u16(i), u16(i + 2) + 1}); i += 4; }
  if (!has_runs || cnt >= 4) i += 4ull * cnt;              // offset header
  for (uint32_t k = 0; k < cnt; k++) {
    uint32_t base = keys[k].first << 16, card = keys[k].second;
    bool is_run = has_runs && ((run_flags[k / 8] >> (k % 8)) & 1);
    if (is_run) {
      if (i + 2 > n) return false;
      uint32_t nr = u16(i); i += 2;
      if (i + 4ull * nr > n) return false;
      for (uint32_t r = 0; r < nr; r++) { uint32_t s = u16(i), l = u16(i + 2); i += 4; for (uint32_t v = s; v <= std::min(s + l, 0xFFFFu); v++) out.push_back(base + v); }   // (a run stays inside its container)
    } else if (card > 4096) {
      if (i + 8192 > n) return false;
      for (uint32_t w = 0; w < 1024; w++) {
        uint64_t word = 0;
        for (int q = 7; q >= 0; q--) word = (word << 8) | b[i + w * 8 + q];
        while (word) { int t = __builtin_ctzll(word); out.push_back(base + w * 64 + t); word &= word - 1; }
      }
      i += 8192;
    } else {
      if (i + 2ull * card > n) return false;
      for (uint32_t c = 0; c < card; c++) { out.push_back(base + u16(i)); i += 2; }
    }
  }
  return true;
}

}  // namespace

// Builds the host CSR (same layout as build_host_index) from reference-built <name>.hd / <name>.dl.
int load_reference_index(const char* hd_path, const char* dl_path, const sg_desc* desc, HostIndex& ix, std::string& err) {
  int rc = init_description(desc, ix, err);
  if (rc) return rc;
  std::vector<uint8_t> hd, dl;
  if (!read_file(hd_path, hd)) { err = std::string("failed to open header: ") + hd_path; return SG_E_INVALID; }
  if (!read_file(dl_path, dl)) { err = std::string("failed to open document list: ") + dl_path; return SG_E_INVALID; }
  std::string version;
  uint32_t indices = 0;
  std::vector<TermDesc> terms;
  if (!parse_header(hd, version, indices, terms)) { err = "failed to retrieve header: malformed gob stream"; return SG_E_INVALID; }
  if (version != "v5.1") { err = "index version mismatch, expected v5.1 version"; return SG_E_INVALID; }   // index_reader.go:71-73
  const uint32_t S = indices;
  ix.n_segments = S;
  // intern terms: every rune of a stored term must be a symbol of the description
  struct L { uint32_t t, b, raw; std::vector<uint32_t> v; };
  std::vector<L> lists;
  lists.reserve(terms.size());
  uint32_t max_doc = 0;
  bool any = false;
  for (const auto& td : terms) {
    if (td.size == 0 || td.indice >= S) continue;
    uint64_t key;
    if (!term_string_key(ix, td.term, &key)) { err = "stored term '" + td.term + "' does not fit the description's alphabet/pad"; return SG_E_UNSUPPORTED; }
    uint32_t t;
    auto it = ix.term_of.find(key);
    if (it == ix.term_of.end()) { t = (uint32_t)ix.term_key.size(); ix.term_key.push_back(key); ix.term_of.emplace(key, t); }
    else t = it->second;
    if ((uint64_t)td.pos + td.size > dl.size()) { err = "posting list outside the document list file"; return SG_E_INVALID; }
    L l{t, td.indice, td.len, {}};
    const uint8_t* b = dl.data() + td.pos;
    bool ok = td.len <= 65 ? decode_vb(b, td.size, td.len, l.v) : td.len <= 256 ? decode_skipping(b, td.size, td.len, l.v) : decode_roaring(b, td.size, l.v);
    if (!ok || l.v.empty()) { err = "malformed posting list of term '" + td.term + "'"; return SG_E_INVALID; }
    max_doc = std::max(max_doc, l.v.back());
    any = true;
    ix.n_postings_raw += td.len;
    lists.push_back(std::move(l));
  }
  ix.n_docs = any ? (uint64_t)max_doc + 1 : 0;
  const size_t nT = ix.term_key.size();
  ix.list_len.assign(nT * (size_t)S, 0);
  for (auto& l : lists) {                                   // runs of equal docIDs = a doc repeating the term
    std::vector<uint32_t> ded;
    for (size_t i = 0; i < l.v.size();) {
      size_t j = i;
      while (j < l.v.size() && l.v[j] == l.v[i]) j++;
      ded.push_back(l.v[i]);
      if (j - i > 1) ix.dups.push_back(DupEntry{l.t, l.b, l.v[i], (uint32_t)(j - i)});
      i = j;
    }
    if (l.raw > 256 && l.raw > ded.size())                  // roaring dropped the repeats: keep the raw length (codec dispatch)
      ix.dups.push_back(DupEntry{l.t, l.b, 0xFFFFFFFFu, (uint32_t)(l.raw - ded.size()) + 1});
    l.v.swap(ded);
    ix.list_len[(size_t)l.t * S + l.b] = (uint32_t)l.v.size();
    ix.n_postings += l.v.size();
    ix.n_lists++;
  }
  ix.seg_off.assign(nT * (size_t)(S + 1) + 1, 0);
  uint64_t chunk = 0;
  for (size_t t = 0; t < nT; t++) {
    for (uint32_t b = 0; b < S; b++) { ix.seg_off[t * (S + 1) + b] = (uint32_t)chunk; chunk += (ix.list_len[t * S + b] + 3) / 4; }
    ix.seg_off[t * (S + 1) + S] = (uint32_t)chunk;
    if (chunk >= 0xFFFFFFF0ull) { err = "posting store exceeds 2^32 16-byte chunks"; return SG_E_UNSUPPORTED; }
  }
  ix.postings.assign((size_t)chunk * 4, 0);
  for (const auto& l : lists) {
    uint32_t* p = ix.postings.data() + (size_t)ix.seg_off[(size_t)l.t * (S + 1) + l.b] * 4;
    std::copy(l.v.begin(), l.v.end(), p);
    for (size_t i = l.v.size(); i < ((l.v.size() + 3) & ~(size_t)3); i++) p[i] = l.v.back();
  }
  std::sort(ix.dups.begin(), ix.dups.end(), [](const DupEntry& x, const DupEntry& y) {
    if (x.term != y.term) return x.term < y.term;
    if (x.segment != y.segment) return x.segment < y.segment;
    return x.doc < y.doc;
  });
  build_term_table(ix);
  return SG_OK;
}

}  // namespace sg
