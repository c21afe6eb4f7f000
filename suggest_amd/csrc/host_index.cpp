// host_index.cpp — host side of libsuggest_hip: tokenizer, term-key packing and the CSR index
// builder.  Replaces (for the GPU engine) the reference's
//   NewSuggestTokenizer        pkg/suggest/tokenizer.go:9-34 (+ pkg/analysis, pkg/alphabet)
//   suggest.Index              pkg/suggest/indexer.go:14-45
//   Writer.AddDocument/Commit  pkg/index/indexer_writer.go:66-145
// Layout decisions are in DESIGN.md §Data layout.  No code here is shared with the test oracle.

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <numeric>
#include <exception>
#include <thread>

#include "sg_internal.h"

namespace sg {

namespace {

struct LowerPair { uint32_t from, to; };
const LowerPair kLower[] = {
#include "unicode_lower.inc"
};

// Go `range` decoding: invalid byte -> U+FFFD, width 1
inline uint32_t next_rune(const uint8_t* s, size_t n, size_t* adv) {
  uint32_t c0 = s[0];
  *adv = 1;
  if (c0 < 0x80) return c0;
  if (c0 < 0xC2 || c0 > 0xF4) return kRuneError;
  if (c0 < 0xE0) {
    if (n < 2 || (s[1] & 0xC0) != 0x80) return kRuneError;
    *adv = 2;
    return ((c0 & 0x1F) << 6) | (s[1] & 0x3F);
  }
  if (c0 < 0xF0) {
    if (n < 3) return kRuneError;
    uint32_t lo = c0 == 0xE0 ? 0xA0 : 0x80, hi = c0 == 0xED ? 0x9F : 0xBF;
    if (s[1] < lo || s[1] > hi || (s[2] & 0xC0) != 0x80) return kRuneError;
    *adv = 3;
    return ((c0 & 0x0F) << 12) | ((s[1] & 0x3F) << 6) | (s[2] & 0x3F);
  }
  if (n < 4) return kRuneError;
  uint32_t lo = c0 == 0xF0 ? 0x90 : 0x80, hi = c0 == 0xF4 ? 0x8F : 0xBF;
  if (s[1] < lo || s[1] > hi || (s[2] & 0xC0) != 0x80 || (s[3] & 0xC0) != 0x80) return kRuneError;
  *adv = 4;
  return ((c0 & 0x07) << 18) | ((s[1] & 0x3F) << 12) | ((s[2] & 0x3F) << 6) | (s[3] & 0x3F);
}

inline uint32_t utf8_width(uint32_t r) { return r < 0x80 ? 1 : r < 0x800 ? 2 : r < 0x10000 ? 3 : 4; }

inline uint32_t lower_rune(uint32_t r) {
  if (r < 0x80) return (r - 'A' < 26u) ? r + 32 : r;
  size_t lo = 0, hi = SG_UNICODE_LOWER_COUNT;
  while (lo < hi) {
    size_t mid = (lo + hi) >> 1;
    if (kLower[mid].from < r) lo = mid + 1; else hi = mid;
  }
  return (lo < SG_UNICODE_LOWER_COUNT && kLower[lo].from == r) ? kLower[lo].to : r;
}

void decode_all(const std::string& s, std::vector<uint32_t>& out) {
  size_t i = 0;
  while (i < s.size()) {
    size_t adv;
    out.push_back(next_rune((const uint8_t*)s.data() + i, s.size() - i, &adv));
    i += adv;
  }
}

inline void sym_lookup(const Symbols& sy, uint32_t r, uint8_t* id, bool* alpha) {
  if (r < 128) { *id = sy.ascii_sym[r]; *alpha = sy.ascii_alpha[r] != 0; return; }
  auto it = std::lower_bound(sy.na_rune.begin(), sy.na_rune.end(), r);
  if (it != sy.na_rune.end() && *it == r) {
    size_t k = it - sy.na_rune.begin();
    *id = sy.na_sym[k]; *alpha = sy.na_alpha[k] != 0;
  } else { *id = 0; *alpha = false; }
}

// normaliseFilter (pkg/analysis/normalizer.go:21-37) fused with key packing
inline bool pack_key(const Symbols& sy, const uint32_t* runes, uint32_t n, uint64_t* key) {
  uint64_t k = 0;
  uint32_t len = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint8_t id; bool alpha;
    sym_lookup(sy, runes[i], &id, &alpha);
    if (alpha) {
      if (len >= 8) return false;
      k |= (uint64_t)id << (8 * len++);
    } else {
      for (uint32_t p = 0; p < sy.n_pad; p++) {
        if (len >= 8) return false;
        k |= (uint64_t)sy.pad_sym[p] << (8 * len++);
      }
    }
  }
  *key = k;
  return true;
}

bool alphabet_part_has(const std::string& spec, uint32_t r) {
  // pkg/alphabet/alphabet.go:23-36 + sequential/simple/russian alphabets
  if (spec == "english") return r >= 'a' && r <= 'z';
  if (spec == "numbers") return r >= '0' && r <= '9';
  if (spec == "russian") { uint32_t c = r == 0x451 ? 0x435 : r; return c >= 0x430 && c <= 0x44F; }
  std::vector<uint32_t> rs; decode_all(spec, rs);
  return std::find(rs.begin(), rs.end(), r) != rs.end();
}

int build_symbols(HostIndex& ix, std::string& err) {
  std::vector<uint32_t> alpha_runes;
  for (const auto& spec : ix.alphabet_spec) {
    if (spec == "english") for (uint32_t r = 'a'; r <= 'z'; r++) alpha_runes.push_back(r);
    else if (spec == "numbers") for (uint32_t r = '0'; r <= '9'; r++) alpha_runes.push_back(r);
    else if (spec == "russian") { for (uint32_t r = 0x430; r <= 0x44F; r++) alpha_runes.push_back(r); alpha_runes.push_back(0x451); }
    else decode_all(spec, alpha_runes);
  }
  std::vector<uint32_t> pad_runes; decode_all(ix.pad_s, pad_runes);
  std::vector<uint32_t> all = alpha_runes;
  all.insert(all.end(), pad_runes.begin(), pad_runes.end());
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  if (all.size() > 255) { err = "alphabet + pad have more than 255 distinct runes"; return SG_E_UNSUPPORTED; }
  if (pad_runes.size() > 8 || ix.q * std::max<size_t>(1, pad_runes.size()) > 8) {
    err = "ngram_size * len(pad runes) exceeds the 8-symbol term key"; return SG_E_UNSUPPORTED;
  }
  Symbols& sy = ix.sym;
  memset(sy.ascii_sym, 0, sizeof sy.ascii_sym);
  memset(sy.ascii_alpha, 0, sizeof sy.ascii_alpha);
  sy.sym_rune.assign(1, 0);
  std::sort(alpha_runes.begin(), alpha_runes.end());
  for (size_t i = 0; i < all.size(); i++) {
    uint32_t r = all[i];
    uint8_t id = (uint8_t)(i + 1);
    bool in_alpha = std::binary_search(alpha_runes.begin(), alpha_runes.end(), r);
    sy.sym_rune.push_back(r);
    if (r < 128) { sy.ascii_sym[r] = id; sy.ascii_alpha[r] = in_alpha; }
    else { sy.na_rune.push_back(r); sy.na_sym.push_back(id); sy.na_alpha.push_back(in_alpha); }
  }
  sy.n_pad = (uint32_t)pad_runes.size();
  for (size_t i = 0; i < pad_runes.size(); i++) { uint8_t id; bool a; sym_lookup(sy, pad_runes[i], &id, &a); sy.pad_sym[i] = id; }
  return SG_OK;
}

}  // namespace

// text helpers shared with lm.cpp (same decoding / lower-casing rules as the index tokeniser)
uint32_t host_next_rune(const uint8_t* s, size_t n, size_t* adv) { return next_rune(s, n, adv); }
uint32_t host_lower_rune(uint32_t r) { return lower_rune(r); }
uint32_t host_utf8_width(uint32_t r) { return utf8_width(r); }
bool host_alphabet_has(const std::vector<std::string>& spec, uint32_t r) {
  for (const auto& part : spec) if (alphabet_part_has(part, r)) return true;
  return false;
}

uint64_t mix64(uint64_t k) {  // splitmix64 finaliser
  k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull;
  k ^= k >> 27; k *= 0x94D049BB133111EBull;
  k ^= k >> 31;
  return k;
}

int metric_min_y(int m, double alpha, int size) {
  switch (m) {
    case SG_JACCARD: return (int)std::ceil(alpha * (double)size);                 // jaccard.go:12-14
    case SG_COSINE: return (int)std::ceil(alpha * alpha * (double)size);          // cosine.go:12-14
    case SG_DICE: return (int)std::ceil(alpha / (2 - alpha) * (double)size);      // dice.go:12-14
    case SG_EXACT: return size;
    default: return 1;
  }
}
int metric_max_y(int m, double alpha, int size) {
  switch (m) {
    case SG_JACCARD: return (int)std::floor((double)size / alpha);
    case SG_COSINE: return (int)std::floor((double)size / (alpha * alpha));
    case SG_DICE: return (int)std::floor((2 - alpha) / alpha * (double)size);
    case SG_EXACT: return size;
    default: return 32767;
  }
}
int metric_threshold(int m, double alpha, int a, int b) {
  switch (m) {
    case SG_JACCARD: return (int)std::ceil(alpha * (double)(a + b) / (1 + alpha));
    case SG_COSINE: return (int)std::ceil(alpha * std::sqrt((double)(a * b)));
    case SG_DICE: return (int)std::ceil(0.5 * alpha * (double)(a + b));
    case SG_EXACT: return a;
    default: return (int)std::ceil(alpha * std::fmin((double)a, (double)b));
  }
}

// wrap -> lower -> trim -> q-gram (first-occurrence dedup) -> normalise, as packed keys
bool tokenize_keys(const HostIndex& ix, const uint8_t* s, size_t n, bool autocomplete, std::vector<uint64_t>& out) {
  out.clear();
  const uint32_t q = ix.q;
  thread_local std::vector<uint32_t> runes;
  runes.clear();
  bool ascii = true;
  for (size_t i = 0; i < n; i++) if (s[i] >= 0x80) { ascii = false; break; }
  // wrap symbols are stored as runes already; they take part in lower-casing like the text
  for (uint32_t r : ix.wrap0) runes.push_back(r);
  if (ascii) {
    for (size_t i = 0; i < n; i++) runes.push_back(s[i]);
  } else {
    size_t i = 0;
    while (i < n) { size_t adv; runes.push_back(next_rune(s + i, n - i, &adv)); i += adv; }
  }
  if (!autocomplete) for (uint32_t r : ix.wrap1) runes.push_back(r);
  size_t byte_len = 0;
  for (auto& r : runes) { r = lower_rune(r); byte_len += utf8_width(r); }
  // strings.Trim(text, " ")
  size_t a = 0, b = runes.size();
  while (a < b && runes[a] == ' ') { a++; byte_len--; }
  while (b > a && runes[b - 1] == ' ') { b--; byte_len--; }
  if (byte_len < q) return true;                       // ngram_tokenizer.go:18
  const uint32_t* r = runes.data() + a;
  size_t R = b - a;
  if (R <= q) {                                        // fewer runes than q: the whole text, once
    uint64_t key;
    if (!pack_key(ix.sym, r, (uint32_t)R, &key)) return false;
    out.push_back(key);
    return true;
  }
  size_t G = R - q + 1;
  for (size_t g = 0; g < G; g++) {
    bool dup = false;                                  // appendUnique, ngram_tokenizer.go:46-54
    for (size_t h = 0; h < g && !dup; h++) {
      bool eq = true;
      for (uint32_t t = 0; t < q; t++) if (r[h + t] != r[g + t]) { eq = false; break; }
      dup = eq;
    }
    if (dup) continue;
    uint64_t key;
    if (!pack_key(ix.sym, r + g, q, &key)) return false;
    out.push_back(key);
  }
  return true;
}

int init_description(const sg_desc* desc, HostIndex& ix, std::string& err) {
  if (!desc || desc->ngram_size < 1 || desc->ngram_size > 8) { err = "ngram_size must be in 1..8"; return SG_E_INVALID; }
  ix.q = desc->ngram_size;
  ix.wrap0_s = desc->wrap_start ? desc->wrap_start : "";
  ix.wrap1_s = desc->wrap_end ? desc->wrap_end : "";
  ix.pad_s = desc->pad ? desc->pad : "";
  for (uint32_t i = 0; i < desc->n_alphabet; i++) ix.alphabet_spec.emplace_back(desc->alphabet[i]);
  decode_all(ix.wrap0_s, ix.wrap0);
  decode_all(ix.wrap1_s, ix.wrap1);
  return build_symbols(ix, err);
}

// key of an already normalised term string (every rune must be a symbol)
bool term_string_key(const HostIndex& ix, const std::string& term, uint64_t* key) {
  std::vector<uint32_t> rs;
  decode_all(term, rs);
  if (rs.size() > 8) return false;
  uint64_t k = 0;
  for (size_t i = 0; i < rs.size(); i++) {
    uint8_t id; bool alpha;
    sym_lookup(ix.sym, rs[i], &id, &alpha);
    if (!id) return false;
    k |= (uint64_t)id << (8 * i);
  }
  *key = k;
  return true;
}

// term-key hash table (open addressing, linear probing, load <= 0.5)
void build_term_table(HostIndex& ix) {
  const size_t nT = ix.term_key.size();
  size_t cap = 16;
  while (cap < nT * 2) cap <<= 1;
  ix.slots.assign(cap, TermSlot{0, kNoTerm, 0});
  for (size_t t = 0; t < nT; t++) {
    size_t h = mix64(ix.term_key[t]) & (cap - 1);
    while (ix.slots[h].term != kNoTerm) h = (h + 1) & (cap - 1);
    ix.slots[h] = TermSlot{ix.term_key[t], (uint32_t)t, 0};
  }
}

// Runs fn(block) for block = 0 .. n-1 on n threads (block 0 on the caller's).  An exception in any block (bad_alloc at
// 10M docs x 32 threads is the plausible one) is carried to the caller once every thread has been joined: nothing
// escapes a worker (std::terminate) and no joinable thread is destroyed, so the C ABI still answers SG_E_NOMEM.
template <class F>
static void run_blocks(uint32_t n, F fn) {
  std::vector<std::exception_ptr> failed(n);
  auto guarded = [&fn, &failed](uint32_t b) {
    try { fn(b); } catch (...) { failed[b] = std::current_exception(); }
  };
  std::vector<std::thread> th;
  try {
    for (uint32_t b = 1; b < n; b++) th.emplace_back(guarded, b);
  } catch (...) {                                    // thread creation failed: the blocks without a thread run here
    const uint32_t started = (uint32_t)th.size() + 1;
    for (uint32_t b = started; b < n; b++) guarded(b);
  }
  guarded(0);
  for (auto& t : th) t.join();
  for (auto& e : failed) if (e) std::rethrow_exception(e);
}

// What one thread learns about its contiguous block of docs in pass 1.
struct BuildBlock {
  uint32_t d0 = 0, d1 = 0;
  std::vector<uint64_t> keys;                        // local term id -> key, first-occurrence order
  std::unordered_map<uint64_t, uint32_t> local_of;
  std::vector<uint32_t> terms;                       // unique term ids per doc, concatenated (local, then global ids)
  std::vector<uint64_t> off;                         // [d1 - d0 + 1] into terms
  struct Dup { uint32_t doc, term, mult; };
  std::vector<Dup> dups;
  std::vector<uint32_t> to_global;
  std::vector<uint32_t> cursor;                      // [nT * S]: counts of this block, then its write cursors
  uint32_t max_card = 0;
  uint64_t raw = 0;
  bool bad_key = false;
};

int build_host_index(const uint8_t* utf8, const uint64_t* offs, uint32_t n_docs, const sg_desc* desc, HostIndex& ix,
                     std::string& err) {
  int rc = init_description(desc, ix, err);
  if (rc) return rc;
  ix.n_docs = n_docs;

  // Contiguous blocks of docs, one thread each (SG_BUILD_THREADS, default: the cores this process may run on, at most
  // 32).  Every step below is arranged so that the result is the one the sequential build gives: term ids in
  // first-occurrence order over ascending docIDs, lists ascending.
  uint32_t n_thr = std::min<uint32_t>(32, std::max<uint32_t>(1, std::thread::hardware_concurrency()));
  if (const char* e = getenv("SG_BUILD_THREADS")) n_thr = (uint32_t)std::max(1, atoi(e));
  n_thr = std::max<uint32_t>(1, std::min<uint32_t>(n_thr, n_docs / 4096 + 1));
  std::vector<BuildBlock> blk(n_thr);
  for (uint32_t b = 0; b < n_thr; b++) {
    blk[b].d0 = (uint32_t)((uint64_t)n_docs * b / n_thr);
    blk[b].d1 = (uint32_t)((uint64_t)n_docs * (b + 1) / n_thr);
  }
  std::vector<uint32_t> doc_card(n_docs);

  // pass 1: tokenise, intern terms per block, remember each doc's (de-duplicated) term ids and cardinality
  run_blocks(n_thr, [&](uint32_t b) {
    BuildBlock& B = blk[b];
    B.off.assign((size_t)(B.d1 - B.d0) + 1, 0);
    B.terms.reserve((size_t)(B.d1 - B.d0) * 16);
    B.local_of.reserve(1 << 16);
    std::vector<uint64_t> keys;
    std::vector<uint32_t> tids, sorted;
    for (uint32_t d = B.d0; d < B.d1; d++) {
      if (!tokenize_keys(ix, utf8 + offs[d], (size_t)(offs[d + 1] - offs[d]), false, keys)) { B.bad_key = true; return; }
      const uint32_t card = (uint32_t)keys.size();
      doc_card[d] = card;
      B.max_card = std::max(B.max_card, card);
      B.raw += card;
      tids.clear();
      for (uint64_t k : keys) {
        auto it = B.local_of.find(k);
        uint32_t t;
        if (it == B.local_of.end()) { t = (uint32_t)B.keys.size(); B.keys.push_back(k); B.local_of.emplace(k, t); }
        else t = it->second;
        tids.push_back(t);
      }
      // a doc may repeat a term after normalisation; the CSR keeps (term,doc) once
      bool has_dup = false;
      for (size_t i = 1; i < tids.size() && !has_dup; i++)
        for (size_t j = 0; j < i; j++) if (tids[i] == tids[j]) { has_dup = true; break; }
      if (has_dup) {
        sorted = tids;
        std::sort(sorted.begin(), sorted.end());
        tids.clear();
        for (size_t i = 0; i < sorted.size();) {
          size_t j = i;
          while (j < sorted.size() && sorted[j] == sorted[i]) j++;
          tids.push_back(sorted[i]);
          if (j - i > 1) B.dups.push_back(BuildBlock::Dup{d, sorted[i], (uint32_t)(j - i)});
          i = j;
        }
      }
      B.terms.insert(B.terms.end(), tids.begin(), tids.end());
      B.off[d - B.d0 + 1] = B.terms.size();
    }
  });
  uint32_t max_card = 0;
  for (auto& B : blk) {
    if (B.bad_key) { err = "a term does not fit the 8-symbol key"; return SG_E_UNSUPPORTED; }
    max_card = std::max(max_card, B.max_card);
    ix.n_postings_raw += B.raw;
    ix.n_postings += B.terms.size();
  }
  // global term ids: blocks in docID order, each block's terms in its own first-occurrence order
  ix.term_of.reserve(1 << 16);
  for (auto& B : blk) {
    B.to_global.resize(B.keys.size());
    for (size_t l = 0; l < B.keys.size(); l++) {
      auto it = ix.term_of.find(B.keys[l]);
      if (it == ix.term_of.end()) {
        B.to_global[l] = (uint32_t)ix.term_key.size();
        ix.term_of.emplace(B.keys[l], (uint32_t)ix.term_key.size());
        ix.term_key.push_back(B.keys[l]);
      } else B.to_global[l] = it->second;
    }
    std::unordered_map<uint64_t, uint32_t>().swap(B.local_of);
  }
  // indexer_writer.go:69-73: len(indices) = max cardinality + 1
  const uint32_t S = n_docs ? std::max(max_card + 1, ix.min_segments) : 0;
  ix.n_segments = S;
  const size_t nT = ix.term_key.size();
  const size_t nTS = nT * (size_t)S;
  // the per-block counters below take nTS words per block: past 2 GiB in total, one shared set and a sequential scatter
  const bool shared = n_thr > 1 && (nTS * n_thr > (1ull << 29) || getenv("SG_BUILD_SHARED_COUNTERS"));

  // pass 2: count per (term, segment) and block, lay the lists out term-major, each padded to 4 postings
  run_blocks(n_thr, [&](uint32_t b) {
    BuildBlock& B = blk[b];
    for (auto& t : B.terms) t = B.to_global[t];
    for (auto& dp : B.dups) dp.term = B.to_global[dp.term];
    if (shared) return;
    B.cursor.assign(nTS, 0);
    for (uint32_t d = B.d0; d < B.d1; d++)
      for (uint64_t p = B.off[d - B.d0]; p < B.off[d - B.d0 + 1]; p++) B.cursor[(size_t)B.terms[p] * S + doc_card[d]]++;
  });
  ix.list_len.assign(nTS, 0);
  if (shared) {
    for (auto& B : blk)
      for (uint32_t d = B.d0; d < B.d1; d++)
        for (uint64_t p = B.off[d - B.d0]; p < B.off[d - B.d0 + 1]; p++) ix.list_len[(size_t)B.terms[p] * S + doc_card[d]]++;
    blk[0].cursor.assign(nTS, 0);
  } else run_blocks(n_thr, [&](uint32_t b) {     // per (term, segment): total length; each block's count becomes its first slot
    const size_t i0 = nTS * b / n_thr, i1 = nTS * (b + 1) / n_thr;
    for (size_t i = i0; i < i1; i++) {
      uint32_t run = 0;
      for (auto& B : blk) { const uint32_t c = B.cursor[i]; B.cursor[i] = run; run += c; }
      ix.list_len[i] = run;
    }
  });
  ix.seg_off.assign(nT * (size_t)(S + 1) + 1, 0);
  uint64_t chunk = 0;
  for (size_t t = 0; t < nT; t++) {
    for (uint32_t b = 0; b < S; b++) {
      ix.seg_off[t * (S + 1) + b] = (uint32_t)chunk;
      uint32_t len = ix.list_len[t * S + b];
      if (len) ix.n_lists++;
      chunk += (len + 3) / 4;
    }
    ix.seg_off[t * (S + 1) + S] = (uint32_t)chunk;
    if (chunk >= 0xFFFFFFF0ull) { err = "posting store exceeds 2^32 16-byte chunks"; return SG_E_UNSUPPORTED; }
  }
  ix.postings.resize((size_t)chunk * 4);
  auto scatter = [&](BuildBlock& B, std::vector<uint32_t>& cursor) {
    for (uint32_t d = B.d0; d < B.d1; d++) {
      const uint32_t s = doc_card[d];
      for (uint64_t p = B.off[d - B.d0]; p < B.off[d - B.d0 + 1]; p++) {
        const size_t t = B.terms[p];
        ix.postings[(size_t)ix.seg_off[t * (S + 1) + s] * 4 + cursor[t * S + s]++] = d;
      }
    }
  };
  // ascending docID within a block, blocks in docID order => every list ascending
  if (shared) for (auto& B : blk) scatter(B, blk[0].cursor);
  else run_blocks(n_thr, [&](uint32_t b) { scatter(blk[b], blk[b].cursor); });
  // pad every list to a whole 16-byte chunk by repeating its last docID: the kernel's lossy counters
  // stay upper bounds and its binary searches stay valid without a per-posting sentinel test
  run_blocks(n_thr, [&](uint32_t b) {
    for (size_t t = nT * b / n_thr; t < nT * (b + 1) / n_thr; t++)
      for (uint32_t s = 0; s < S; s++) {
        const uint32_t len = ix.list_len[t * S + s];
        if (!len || !(len & 3)) continue;
        uint32_t* p = ix.postings.data() + (size_t)ix.seg_off[t * (S + 1) + s] * 4;
        for (uint32_t i = len; i < ((len + 3) & ~3u); i++) p[i] = p[len - 1];
      }
  });
  for (const auto& B : blk)
    for (const auto& rd : B.dups) ix.dups.push_back(DupEntry{rd.term, doc_card[rd.doc], rd.doc, rd.mult});
  std::sort(ix.dups.begin(), ix.dups.end(), [](const DupEntry& x, const DupEntry& y) {
    if (x.term != y.term) return x.term < y.term;
    if (x.segment != y.segment) return x.segment < y.segment;
    return x.doc < y.doc;
  });

  build_term_table(ix);
  return SG_OK;
}

}  // namespace sg
