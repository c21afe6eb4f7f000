// suggest_oracle.cpp — CPU restatement of suggest-go's n-gram fuzzy-search path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under suggest_amd/ may include, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as
// the checker / the timed CPU baseline (kind "port"), never as the product.
//
// It is a behavioural restatement (not a copy) of the Go reference, function by
// function; every block cites the reference file:line it follows (paths relative to
// /root/reference).  Parity pins: tests/test_oracle_golden.py checks it against every
// golden vector of the reference's own tests for this path and against the committed
// on-disk index fixtures (tests/golden/cars.{hd,dl}); see DESIGN.md §Oracle.
// One piece is "parity unpinned": go_sort() restates Go 1.14's sort.Sort (the
// reference's Dockerfile pins golang:1.14.4) from its published algorithm; it only
// decides the order of equal-length posting lists, which only matters for documents
// with repeated terms.
//
// Build: make -C oracle   (g++ -O2 -fopenmp -shared)

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <cstdio>
#include <iterator>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------
// UTF-8 helpers with Go semantics: `for _, r := range s` yields U+FFFD, width 1, for
// every byte that does not start a valid encoding (Go spec, "For statements").
// ---------------------------------------------------------------------------------
constexpr uint32_t kRuneError = 0xFFFD;

static inline uint32_t decode_rune(const unsigned char* s, size_t n, int* width) {
  if (n == 0) { *width = 0; return kRuneError; }
  unsigned c0 = s[0];
  if (c0 < 0x80) { *width = 1; return c0; }
  auto bad = [&]() { *width = 1; return kRuneError; };
  if (c0 < 0xC2) return bad();
  if (c0 < 0xE0) {
    if (n < 2 || (s[1] & 0xC0) != 0x80) return bad();
    *width = 2; return ((c0 & 0x1F) << 6) | (s[1] & 0x3F);
  }
  if (c0 < 0xF0) {
    if (n < 3) return bad();
    unsigned lo = 0x80, hi = 0xBF;
    if (c0 == 0xE0) lo = 0xA0;
    if (c0 == 0xED) hi = 0x9F;
    if (s[1] < lo || s[1] > hi || (s[2] & 0xC0) != 0x80) return bad();
    *width = 3; return ((c0 & 0x0F) << 12) | ((s[1] & 0x3F) << 6) | (s[2] & 0x3F);
  }
  if (c0 < 0xF5) {
    if (n < 4) return bad();
    unsigned lo = 0x80, hi = 0xBF;
    if (c0 == 0xF0) lo = 0x90;
    if (c0 == 0xF4) hi = 0x8F;
    if (s[1] < lo || s[1] > hi || (s[2] & 0xC0) != 0x80 || (s[3] & 0xC0) != 0x80) return bad();
    *width = 4;
    return ((c0 & 0x07) << 18) | ((s[1] & 0x3F) << 12) | ((s[2] & 0x3F) << 6) | (s[3] & 0x3F);
  }
  return bad();
}

static inline void append_rune(std::string& out, uint32_t r) {
  if (r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF)) r = kRuneError;
  if (r < 0x80) out.push_back((char)r);
  else if (r < 0x800) { out.push_back((char)(0xC0 | (r >> 6))); out.push_back((char)(0x80 | (r & 0x3F))); }
  else if (r < 0x10000) {
    out.push_back((char)(0xE0 | (r >> 12))); out.push_back((char)(0x80 | ((r >> 6) & 0x3F)));
    out.push_back((char)(0x80 | (r & 0x3F)));
  } else {
    out.push_back((char)(0xF0 | (r >> 18))); out.push_back((char)(0x80 | ((r >> 12) & 0x3F)));
    out.push_back((char)(0x80 | ((r >> 6) & 0x3F))); out.push_back((char)(0x80 | (r & 0x3F)));
  }
}

static std::vector<uint32_t> runes_of(const std::string& s) {
  std::vector<uint32_t> r;
  size_t i = 0;
  while (i < s.size()) {
    int w; uint32_t c = decode_rune((const unsigned char*)s.data() + i, s.size() - i, &w);
    r.push_back(c); i += w;
  }
  return r;
}

struct LowerPair { uint32_t from, to; };
static const LowerPair kLower[] = {
#include "unicode_lower.inc"
};

// unicode.ToLower (simple case mapping), used by strings.ToLower
static inline uint32_t rune_lower(uint32_t r) {
  if (r < 0x80) return (r >= 'A' && r <= 'Z') ? r + 32 : r;
  size_t lo = 0, hi = SG_UNICODE_LOWER_COUNT;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (kLower[mid].from < r) lo = mid + 1; else hi = mid;
  }
  if (lo < SG_UNICODE_LOWER_COUNT && kLower[lo].from == r) return kLower[lo].to;
  return r;
}

// strings.ToLower — pkg/analysis/filter_tokenizer.go:21.  ASCII strings are mapped byte
// wise; otherwise strings.Map re-encodes every rune, replacing invalid bytes by U+FFFD.
static std::string go_to_lower(const std::string& s) {
  bool ascii = true;
  for (unsigned char c : s) if (c >= 0x80) { ascii = false; break; }
  std::string out;
  out.reserve(s.size());
  if (ascii) {
    for (unsigned char c : s) out.push_back((c >= 'A' && c <= 'Z') ? (char)(c + 32) : (char)c);
    return out;
  }
  size_t i = 0;
  while (i < s.size()) {
    int w; uint32_t c = decode_rune((const unsigned char*)s.data() + i, s.size() - i, &w);
    append_rune(out, rune_lower(c)); i += w;
  }
  return out;
}

// strings.Trim(text, " ") — filter_tokenizer.go:22
static std::string go_trim_spaces(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && s[a] == ' ') a++;
  while (b > a && s[b - 1] == ' ') b--;
  return s.substr(a, b - a);
}

// ---------------------------------------------------------------------------------
// pkg/alphabet — alphabet.go:23-36 (CreateAlphabet), sequential_alphabet.go:23,
// simple_alphabet.go:24, composite_alphabet.go:35, russian_alphabet.go:16-22.
// ---------------------------------------------------------------------------------
struct Alphabet {
  struct Part { bool seq; uint32_t lo, hi; bool russian; std::vector<uint32_t> set; };
  std::vector<Part> parts;

  static Alphabet create(const std::vector<std::string>& spec) {
    Alphabet a;
    for (const auto& s : spec) {
      Part p{};
      if (s == "english") { p.seq = true; p.lo = 'a'; p.hi = 'z'; }
      else if (s == "numbers") { p.seq = true; p.lo = '0'; p.hi = '9'; }
      else if (s == "russian") { p.seq = true; p.lo = 0x430; p.hi = 0x44F; p.russian = true; }
      else { p.seq = false; p.set = runes_of(s); }
      a.parts.push_back(std::move(p));
    }
    return a;
  }
  bool has(uint32_t r) const {
    for (const auto& p : parts) {
      if (p.seq) {
        uint32_t c = r;
        if (p.russian && r == 0x451) c = 0x435;  // 'ё' is looked up as 'е' (russian_alphabet.go:17-19)
        if (c >= p.lo && c <= p.hi) return true;
      } else {
        for (uint32_t x : p.set) if (x == r) return true;
      }
    }
    return false;
  }
};

// ---------------------------------------------------------------------------------
// pkg/analysis
// ---------------------------------------------------------------------------------
// nGramTokenizer.Tokenize + appendUnique — ngram_tokenizer.go:17-55
static std::vector<std::string> ngram_tokenize(const std::string& text, int n) {
  std::vector<std::string> result;
  if ((int)text.size() < n) return result;            // :18 byte-length guard
  int prev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int i = 0;
  auto append_unique = [&](std::string g) {           // :46-54
    for (const auto& y : result) if (y == g) return;
    result.push_back(std::move(g));
  };
  size_t pos = 0;
  while (pos < text.size()) {                          // for index := range text
    int w; decode_rune((const unsigned char*)text.data() + pos, text.size() - pos, &w);
    int index = (int)pos;
    i++;
    if (i > n) {
      int top = prev[(i - n) % n];
      append_unique(text.substr(top, index - top));
    }
    prev[i % n] = index;
    pos += w;
  }
  int top = prev[(i + 1) % n];
  append_unique(text.substr(top));
  return result;
}

struct Description {
  int q = 3;
  std::string wrap0, wrap1, pad;
  Alphabet alphabet;
};

// normalizeFilter.Filter — normalizer.go:21-37 (no re-dedup afterwards)
static void normalize(std::vector<std::string>& toks, const Description& d) {
  for (auto& t : toks) {
    std::string res;
    size_t i = 0;
    while (i < t.size()) {
      int w; uint32_t r = decode_rune((const unsigned char*)t.data() + i, t.size() - i, &w);
      if (d.alphabet.has(r)) append_rune(res, r); else res += d.pad;
      i += w;
    }
    t.swap(res);
  }
}

// NewSuggestTokenizer / NewAutocompleteTokenizer — pkg/suggest/tokenizer.go:9-34;
// wrapTokenizer.Tokenize wrap_tokenizer.go:18; filterTokenizer.Tokenize filter_tokenizer.go:20-27
static std::vector<std::string> tokenize(const Description& d, const std::string& text, bool autocomplete) {
  std::string s = d.wrap0 + text + (autocomplete ? std::string() : d.wrap1);
  s = go_trim_spaces(go_to_lower(s));
  auto toks = ngram_tokenize(s, d.q);
  normalize(toks, d);
  return toks;
}

// ---------------------------------------------------------------------------------
// pkg/metric — jaccard.go:12-27, cosine.go:12-26, dice.go:12-26, exact.go, overlap.go.
// All IEEE binary64, evaluation order as written in Go; built with -ffp-contract=off.
// ---------------------------------------------------------------------------------
enum Metric { JACCARD = 0, COSINE = 1, DICE = 2, EXACT = 3, OVERLAP = 4 };

static int metric_min_y(int m, double alpha, int size) {
  switch (m) {
    case JACCARD: return (int)std::ceil(alpha * (double)size);
    case COSINE: return (int)std::ceil(alpha * alpha * (double)size);
    case DICE: return (int)std::ceil(alpha / (2 - alpha) * (double)size);
    case EXACT: return size;
    default: return 1;
  }
}
static int metric_max_y(int m, double alpha, int size) {
  switch (m) {
    case JACCARD: return (int)std::floor((double)size / alpha);
    case COSINE: return (int)std::floor((double)size / (alpha * alpha));
    case DICE: return (int)std::floor((2 - alpha) / alpha * (double)size);
    case EXACT: return size;
    default: return 32767;  // math.MaxInt16
  }
}
static int metric_threshold(int m, double alpha, int a, int b) {
  switch (m) {
    case JACCARD: return (int)std::ceil(alpha * (double)(a + b) / (1 + alpha));
    case COSINE: return (int)std::ceil(alpha * std::sqrt((double)(a * b)));
    case DICE: return (int)std::ceil(0.5 * alpha * (double)(a + b));
    case EXACT: return a;
    default: return (int)std::ceil(alpha * std::fmin((double)a, (double)b));
  }
}
static double metric_distance(int m, int inter, int a, int b) {
  switch (m) {
    case JACCARD: return 1 - (double)inter / (double)(a + b - inter);
    case COSINE: return 1 - (double)inter / std::sqrt((double)(a * b));
    case DICE: return 1 - (double)(2 * inter) / (double)(a + b);
    case EXACT: return 0;
    default: return 1 - (double)inter / std::fmin((double)a, (double)b);
  }
}

// ---------------------------------------------------------------------------------
// pkg/merger — ListIterator semantics (list_iterator.go:14-27) as implemented by the
// stored posting lists (index/posting_list.go:24-108): LowerBound never moves backwards
// and stays on an element that is already >= `to`.
// ---------------------------------------------------------------------------------
struct Iter {
  const uint32_t* p = nullptr;
  int n = 0;      // elements
  int len = 0;    // Len() as the header reports it (see StoredList)
  int idx = 0;
  bool valid() const { return idx < n; }
  bool get(uint32_t* v) const { if (!valid()) return false; *v = p[idx]; return true; }
  bool has_next() const { return idx + 1 < n; }
  bool next(uint32_t* v) { if (!has_next()) return false; idx++; *v = p[idx]; return true; }
  bool lower_bound(uint32_t to, uint32_t* v) {
    if (!valid()) return false;
    if (p[idx] >= to) { *v = p[idx]; return true; }
    const uint32_t* it = std::lower_bound(p + idx, p + n, to);
    if (it == p + n) { idx = n; return false; }
    idx = (int)(it - p); *v = *it; return true;
  }
};

struct Cand { uint32_t pos; uint32_t overlap; };  // MergeCandidate list_merger.go:33-49
struct OverlapOverflow {};
static inline void increment(Cand& c) {            // list_merger.go:51-57
  if (c.overlap == 0xFFFF) throw OverlapOverflow();
  c.overlap++;
}

using Collect = std::function<bool(Cand)>;  // returns false == ErrCollectionTerminated

// Go 1.14 sort.Sort (quickSort + ShellSort pass + insertionSort + heapSort), restated
// from the published algorithm of the Go standard library the reference builds with.
template <class Less, class Swap>
struct GoSort {
  Less less;     // (functors, inlined: the sort runs once per segment and query on the CPU baseline's hot path)
  Swap swap;
  void insertion(int a, int b) {
    for (int i = a + 1; i < b; i++)
      for (int j = i; j > a && less(j, j - 1); j--) swap(j, j - 1);
  }
  void sift_down(int lo, int hi, int first) {
    int root = lo;
    for (;;) {
      int child = 2 * root + 1;
      if (child >= hi) return;
      if (child + 1 < hi && less(first + child, first + child + 1)) child++;
      if (!less(first + root, first + child)) return;
      swap(first + root, first + child);
      root = child;
    }
  }
  void heap_sort(int a, int b) {
    int first = a, lo = 0, hi = b - a;
    for (int i = (hi - 1) / 2; i >= 0; i--) sift_down(i, hi, first);
    for (int i = hi - 1; i >= 0; i--) { swap(first, first + i); sift_down(lo, i, first); }
  }
  void median3(int m1, int m0, int m2) {
    if (less(m1, m0)) swap(m1, m0);
    if (less(m2, m1)) { swap(m2, m1); if (less(m1, m0)) swap(m1, m0); }
  }
  void do_pivot(int lo, int hi, int* midlo, int* midhi) {
    int m = (int)((unsigned)(lo + hi) >> 1);
    if (hi - lo > 40) {
      int s = (hi - lo) / 8;
      median3(lo, lo + s, lo + 2 * s);
      median3(m, m - s, m + s);
      median3(hi - 1, hi - 1 - s, hi - 1 - 2 * s);
    }
    median3(lo, m, hi - 1);
    int pivot = lo;
    int a = lo + 1, c = hi - 1;
    for (; a < c && less(a, pivot); a++) {}
    int b = a;
    for (;;) {
      for (; b < c && !less(pivot, b); b++) {}
      for (; b < c && less(pivot, c - 1); c--) {}
      if (b >= c) break;
      swap(b, c - 1); b++; c--;
    }
    bool protect = hi - c < 5;
    if (!protect && hi - c < (hi - lo) / 4) {
      int dups = 0;
      if (!less(pivot, hi - 1)) { swap(c, hi - 1); c++; dups++; }
      if (!less(b - 1, pivot)) { b--; dups++; }
      if (!less(m, pivot)) { swap(m, b - 1); b--; dups++; }
      protect = dups > 1;
    }
    if (protect) {
      for (;;) {
        for (; a < b && !less(b - 1, pivot); b--) {}
        for (; a < b && less(a, pivot); a++) {}
        if (a >= b) break;
        swap(a, b - 1); a++; b--;
      }
    }
    swap(pivot, b - 1);
    *midlo = b - 1; *midhi = c;
  }
  void quick(int a, int b, int depth) {
    while (b - a > 12) {
      if (depth == 0) { heap_sort(a, b); return; }
      depth--;
      int mlo, mhi; do_pivot(a, b, &mlo, &mhi);
      if (mlo - a < b - mhi) { quick(a, mlo, depth); a = mhi; }
      else { quick(mhi, b, depth); b = mlo; }
    }
    if (b - a > 1) {
      for (int i = a + 6; i < b; i++) if (less(i, i - 6)) swap(i, i - 6);
      insertion(a, b);
    }
  }
  void sort(int n) {
    int depth = 0;
    for (int i = n; i > 0; i >>= 1) depth++;
    quick(0, n, depth * 2);
  }
};

template <class Less, class Swap>
static void go_sort(int n, Less less, Swap swap) { GoSort<Less, Swap> g{less, swap}; g.sort(n); }

// test hook: the permutation go_sort gives n keys (out[i] = original index of the element that ends at position i)
extern "C" void or_go_sort(const uint32_t* keys, int n, uint32_t* out) {
  std::vector<uint32_t> k(keys, keys + n);
  for (int i = 0; i < n; i++) out[i] = (uint32_t)i;
  go_sort(n, [&](int i, int j) { return k[i] < k[j]; }, [&](int i, int j) { std::swap(k[i], k[j]); std::swap(out[i], out[j]); });
}

// sort.Sort(rid) with Rid.Less = Len() < Len() — list_merger.go:23-31
static void sort_rid(std::vector<Iter>& rid, bool reverse = false) {
  auto swap = [&](int i, int j) { std::swap(rid[i], rid[j]); };
  if (!reverse) go_sort((int)rid.size(), [&](int i, int j) { return rid[i].len < rid[j].len; }, swap);
  else go_sort((int)rid.size(), [&](int i, int j) { return rid[j].len < rid[i].len; }, swap);   // sort.Reverse
}

// One list folded into the running candidate array: the loop shared by
// cp_merge.go:47-78 and scan_count.go:33-64.
static void fold_list(Iter& list, std::vector<Cand>& cands, std::vector<Cand>& tmp) {
  uint32_t current = 0;
  bool is_valid = list.get(&current);
  tmp.clear();
  size_t j = 0, end = cands.size();
  while (j < end || is_valid) {
    if (j >= end || (is_valid && cands[j].pos > current)) {
      tmp.push_back(Cand{current, 1});
      if (list.has_next()) list.next(&current); else is_valid = false;
    } else if (!is_valid || (j < end && cands[j].pos < current)) {
      tmp.push_back(cands[j]); j++;
    } else {
      increment(cands[j]);
      tmp.push_back(cands[j]); j++;
      if (list.has_next()) list.next(&current); else is_valid = false;
    }
  }
  cands.swap(tmp);
}

// cpMerge.Merge — cp_merge.go:19-120
static void cp_merge(std::vector<Iter>& rid, int threshold, const Collect& collect) {
  int len_rid = (int)rid.size();
  int min_queries = len_rid - threshold + 1;
  sort_rid(rid);                                                       // :24
  std::vector<Cand> cands, tmp;
  for (int i = 0; i < min_queries; i++) fold_list(rid[i], cands, tmp);  // :32-81
  for (int i = min_queries; i < len_rid && !cands.empty(); i++) {       // :83-103
    tmp.clear();
    for (Cand c : cands) {
      uint32_t cur;
      if (rid[i].lower_bound(c.pos, &cur) && cur == c.pos) increment(c);
      if ((int)c.overlap + (len_rid - i - 1) >= threshold) tmp.push_back(c);
    }
    cands.swap(tmp);
  }
  for (Cand c : cands)                                                  // :105-117
    if ((int)c.overlap >= threshold) if (!collect(c)) return;
}

// scanCount.Merge — scan_count.go:14-88
static void scan_count(std::vector<Iter>& rid, int threshold, const Collect& collect) {
  std::vector<Cand> cands, tmp;
  for (auto& l : rid) fold_list(l, cands, tmp);
  for (Cand c : cands) if ((int)c.overlap >= threshold) if (!collect(c)) return;
}

// intersector.Intersect — list_intersector.go:23-81
static void intersect(std::vector<Iter>& rid, const Collect& collect) {
  uint32_t n = (uint32_t)rid.size();
  if (n == 0) return;
  sort_rid(rid);
  Iter& first = rid[0];
  uint32_t item;
  if (!first.get(&item)) return;   // Get() error is returned to the caller, nothing collected
  for (;;) {
    bool good = true;
    for (size_t k = 1; k < rid.size(); k++) {
      uint32_t lower;
      bool ok = rid[k].lower_bound(item, &lower);
      if (!ok || lower != item) { good = false; break; }
    }
    if (good) if (!collect(Cand{item, n})) return;
    if (!first.has_next()) break;
    first.next(&item);
  }
}

// container/heap on merge_skip.go's recordHeap (min-heap by position)
struct Rec { uint32_t rid, pos; };
struct RecHeap {
  std::vector<Rec> s; int size = 0;
  bool less(int i, int j) const { return s[i].pos < s[j].pos; }
  void up(int j) { for (;;) { int i = (j - 1) / 2; if (i == j || !less(j, i)) break; std::swap(s[i], s[j]); j = i; } }
  bool down(int i0, int n) {
    int i = i0;
    for (;;) {
      int j1 = 2 * i + 1;
      if (j1 >= n || j1 < 0) break;
      int j = j1, j2 = j1 + 1;
      if (j2 < n && less(j2, j1)) j = j2;
      if (!less(j, i)) break;
      std::swap(s[i], s[j]); i = j;
    }
    return i > i0;
  }
  void init() { int n = size; for (int i = n / 2 - 1; i >= 0; i--) down(i, n); }
  void push() { size++; up(size - 1); }                       // heap.Push: Push() then up(Len()-1)
  void pop() { int n = size - 1; std::swap(s[0], s[n]); down(0, n); size--; }  // heap.Pop
};

static void merge_dispatch(int algo, std::vector<Iter>& rid, int threshold, const Collect& collect);

// mergeSkip.Merge — merge_skip.go:52-151
static void merge_skip(std::vector<Iter>& rid, int threshold, const Collect& collect) {
  int len_rid = (int)rid.size();
  RecHeap h; h.s.resize(len_rid); h.size = len_rid;
  for (int i = 0; i < len_rid; i++) { uint32_t r = 0; rid[i].get(&r); h.s[i] = Rec{(uint32_t)i, r}; }
  h.init();
  while (h.size > 0) {
    int popped = 0;
    Rec t = h.s[0];
    while (h.size > 0 && t.pos >= h.s[0].pos) { h.pop(); popped++; }
    if (popped >= threshold) {
      if (!collect(Cand{t.pos, (uint32_t)popped})) return;
      int start = h.size;
      for (int i = 0; i < popped; i++) {
        Rec item = h.s[start + i];
        Iter& cur = rid[item.rid];
        if (cur.has_next()) { uint32_t r; cur.next(&r); h.s[h.size] = Rec{item.rid, r}; h.push(); }
      }
    } else {
      for (int j = threshold - 1 - popped; j > 0 && h.size > 0; j--) { h.pop(); popped++; }
      if (h.size == 0) break;
      uint32_t top_pos = h.s[0].pos;
      int start = h.size;
      for (int i = 0; i < popped; i++) {
        Rec item = h.s[start + i];
        Iter& cur = rid[item.rid];
        if (cur.len == 0) continue;
        uint32_t r;
        if (cur.lower_bound(top_pos, &r)) { h.s[h.size] = Rec{item.rid, r}; h.push(); }
      }
    }
  }
}

// divideSkip.Merge — divide_skip.go:25-74 (inner merger = MergeSkip behind mergerOptimizer)
static void divide_skip(std::vector<Iter>& rid, int threshold, double mu, const Collect& collect) {
  sort_rid(rid, true);
  double M = (double)rid[0].len;
  int l = (int)((double)threshold / (mu * std::log(M) + 1));
  std::vector<Iter> l_long(rid.begin(), rid.begin() + l), l_short(rid.begin() + l, rid.end());
  if (l_short.empty()) { merge_dispatch(2, rid, threshold, collect); return; }
  std::vector<Cand> res;
  merge_dispatch(2, l_short, threshold - l, [&](Cand c) { res.push_back(c); return true; });
  for (Cand c : res) {
    for (auto& ll : l_long) { uint32_t r; if (ll.lower_bound(c.pos, &r) && r == c.pos) increment(c); }
    if ((int)c.overlap >= threshold) if (!collect(c)) return;
  }
}

// mergerOptimizer.Merge — list_merger.go:73-85.  algo: 0 cp_merge, 1 scan_count, 2 merge_skip, 3 divide_skip(0.01)
static void merge_dispatch(int algo, std::vector<Iter>& rid, int threshold, const Collect& collect) {
  int n = (int)rid.size();
  if (n < threshold || n == 0 || threshold < 0) return;
  if (n == threshold) { intersect(rid, collect); return; }
  switch (algo) {
    case 0: cp_merge(rid, threshold, collect); break;
    case 1: scan_count(rid, threshold, collect); break;
    case 2: merge_skip(rid, threshold, collect); break;
    default: divide_skip(rid, threshold, 0.01, collect); break;
  }
}

// ---------------------------------------------------------------------------------
// pkg/suggest top-k — collector.go:20-26 (Candidate.Less), topk.go:66-175
// (container/heap min-heap whose root is the *worst* kept candidate).
// ---------------------------------------------------------------------------------
struct Candidate { uint32_t key; double score; };
static inline bool cand_less(const Candidate& c, const Candidate& o) {
  if (c.score == o.score) return c.key > o.key;
  return c.score < o.score;
}
struct TopK {
  int k; std::vector<Candidate> h;
  explicit TopK(int k_) : k(k_) {}
  void up(int j) { for (;;) { int i = (j - 1) / 2; if (i == j || !cand_less(h[j], h[i])) break; std::swap(h[i], h[j]); j = i; } }
  bool down(int i0, int n) {
    int i = i0;
    for (;;) {
      int j1 = 2 * i + 1;
      if (j1 >= n || j1 < 0) break;
      int j = j1, j2 = j1 + 1;
      if (j2 < n && cand_less(h[j2], h[j1])) j = j2;
      if (!cand_less(h[j], h[i])) break;
      std::swap(h[i], h[j]); i = j;
    }
    return i > i0;
  }
  bool full() const { return (int)h.size() == k; }
  bool can_take(double score) const { return !full() || h[0].score <= score; }   // topk.go:114-120
  double lowest() const { return h.empty() ? -INFINITY : h[0].score; }
  void add(uint32_t key, double score) {                                            // topk.go:82-102
    if (!can_take(score)) return;
    Candidate c{key, score};
    if ((int)h.size() < k) { h.push_back(c); up((int)h.size() - 1); return; }
    if (cand_less(h[0], c)) { h[0] = c; if (!down(0, (int)h.size())) up(0); }      // heap.Fix
  }
  void merge(const TopK& o) { for (const auto& c : o.h) add(c.key, c.score); }     // topk.go:150-165
  std::vector<Candidate> candidates() const {                                       // topk.go:127-147
    TopK t = *this;
    std::vector<Candidate> sorted(t.h.size());
    while (!t.h.empty()) {
      int n = (int)t.h.size() - 1;
      std::swap(t.h[0], t.h[n]); t.down(0, n);
      sorted[n] = t.h.back(); t.h.pop_back();
    }
    return sorted;
  }
};

// ---------------------------------------------------------------------------------
// pkg/index — Writer.AddDocument indexer_writer.go:66-86; storage rule codec.go:39-51
// (lists longer than 256 become roaring bitmaps: duplicates vanish and Len() is the
// cardinality, bitmap_posting_list.go:99; the header keeps the raw length, which is
// what resolvePostingList codec.go:76-89 dispatches on).
// ---------------------------------------------------------------------------------
struct StoredList { std::vector<uint32_t> v; int raw_len = 0; };
struct Segment {
  std::unordered_map<std::string, StoredList> terms;
  std::vector<const StoredList*> by_id;         // commit(): the same lists by dense term id (nullptr: not in this segment)
};

struct Index {
  Description d;
  std::vector<std::unique_ptr<Segment>> segs;   // nullptr == empty segment (indices.go:27-33)
  std::unordered_map<std::string, int> term_id; // commit(): every term of the index -> dense id; a query resolves its
                                                // terms once instead of hashing the strings again in every segment
};
static std::vector<int> term_ids(const Index& ix, const std::vector<std::string>& terms) {
  std::vector<int> ids(terms.size(), -1);
  for (size_t i = 0; i < terms.size(); i++) { auto it = ix.term_id.find(terms[i]); if (it != ix.term_id.end()) ids[i] = it->second; }
  return ids;
}

static void add_document(Index& ix, uint32_t id, const std::vector<std::string>& terms) {
  size_t card = terms.size();
  if (ix.segs.size() <= card) ix.segs.resize(card + 1);
  if (!ix.segs[card]) ix.segs[card] = std::make_unique<Segment>();
  for (const auto& t : terms) ix.segs[card]->terms[t].v.push_back(id);
}

static void commit(Index& ix) {
  for (auto& s : ix.segs) {
    if (!s) continue;
    if (s->terms.empty()) { s.reset(); continue; }   // segment 0 of an empty-token doc has no terms
    for (auto& kv : s->terms) {
      StoredList& l = kv.second;
      l.raw_len = (int)l.v.size();
      if (l.raw_len > 256) l.v.erase(std::unique(l.v.begin(), l.v.end()), l.v.end());
      l.v.shrink_to_fit();
    }
  }
  ix.term_id.clear();
  for (auto& s : ix.segs) if (s) for (auto& kv : s->terms) ix.term_id.emplace(kv.first, (int)ix.term_id.size());
  for (auto& s : ix.segs) {
    if (!s) continue;
    s->by_id.assign(ix.term_id.size(), nullptr);
    for (auto& kv : s->terms) s->by_id[ix.term_id.at(kv.first)] = &kv.second;    // (node addresses are stable)
  }
}

static inline int stored_len(const StoredList& l) { return l.raw_len > 256 ? (int)l.v.size() : l.raw_len; }

// searcher.Search + filterTermsByExistence — searcher.go:28-78
static void search(const Segment& seg, const std::vector<int>& terms, int threshold, int algo,
                   const Collect& collect) {
  int n = (int)terms.size();
  std::vector<const StoredList*> kept;
  for (int i = 0; i < n && ((int)kept.size() + n - i) >= threshold; i++) {
    const StoredList* l = terms[i] >= 0 ? seg.by_id[terms[i]] : nullptr;
    if (l) kept.push_back(l);
  }
  if ((int)kept.size() < threshold) return;
  std::vector<Iter> rid;
  rid.reserve(kept.size());
  for (auto* l : kept) { Iter it; it.p = l->v.data(); it.n = (int)l->v.size(); it.len = stored_len(*l); rid.push_back(it); }
  merge_dispatch(algo, rid, threshold, collect);
}

// nGramSuggester.Suggest — suggester.go:46-131.  Segments are visited in the feeder's
// inside-out order (:113-121) by one worker.  tighten=false is the parity definition
// (SURVEY.md §A.6); tighten=true emulates one worker applying :101-103 after every segment.
//
// *status: 0 ok; 1 = the reference panics here (suggester.go:62 make(chan, negative) when the
// clipped window is empty by more than one); 2 = the reference dead-locks here (capacity 0:
// no workers are started, :70, and the first send at :115 blocks forever).  Both happen only
// when the query is so much longer than every dictionary entry that bMin > nSegments-1.
static std::vector<Candidate> suggest(const Index& ix, const std::string& query, int metric, double similarity,
                                      int k, bool tighten, int algo, int* status = nullptr) {
  if (status) *status = 0;
  auto tokens = tokenize(ix.d, query, false);
  if (tokens.empty()) return {};
  int size_a = (int)tokens.size();
  int b_min = metric_min_y(metric, similarity, size_a), b_max = metric_max_y(metric, similarity, size_a);
  int len_indices = (int)ix.segs.size();
  if (b_max >= len_indices) b_max = len_indices - 1;
  if (b_max - b_min + 1 < 0) { if (status) *status = 1; return {}; }
  if (b_max - b_min + 1 == 0) { if (status) *status = 2; return {}; }
  TopK global(k);
  double sim = similarity;
  const std::vector<int> token_ids = term_ids(ix, tokens);
  auto visit = [&](int size_b) {
    int threshold = metric_threshold(metric, sim, size_a, size_b);
    if (threshold == 0 || threshold > size_b || threshold > size_a) return;
    if (size_b < 0 || size_b >= len_indices || !ix.segs[size_b]) return;
    TopK local(k);
    search(*ix.segs[size_b], token_ids, threshold, algo, [&](Cand c) {
      local.add(c.pos, 1 - metric_distance(metric, (int)c.overlap, size_a, size_b));  // scorer.go:29-31
      return true;
    });
    global.merge(local);
    if (tighten && global.full() && global.lowest() > sim) sim = global.lowest();
  };
  for (int i = size_a, j = size_a + 1; i >= b_min || j <= b_max; i--, j++) {
    if (i >= b_min) visit(i);
    if (j <= b_max) visit(j);
  }
  return global.candidates();
}

// nGramAutocomplete.Autocomplete — autocomplete.go:40-77 with firstKCollector
// collector.go:48-66 and FirstKCollectorManager.Collect collector.go:97-111 (score = -docID)
static std::vector<Candidate> autocomplete(const Index& ix, const std::string& query, int limit, int algo) {
  auto terms = tokenize(ix.d, query, true);
  int terms_len = (int)terms.size();
  TopK queue(limit);
  const std::vector<int> ids = term_ids(ix, terms);
  for (int size = terms_len; size < (int)ix.segs.size(); size++) {
    if (!ix.segs[size]) continue;
    std::vector<uint32_t> items;
    search(*ix.segs[size], ids, terms_len, algo, [&](Cand c) {
      if ((int)items.size() == limit) return false;
      items.push_back(c.pos);
      return true;
    });
    for (uint32_t p : items) queue.add(p, -(double)p);
  }
  return queue.candidates();
}

static std::vector<std::string> split_alphabet(const char* spec) {
  // '\n'-separated list of alphabet descriptors
  std::vector<std::string> out;
  std::string cur;
  for (const char* p = spec; *p; p++) { if (*p == '\n') { out.push_back(cur); cur.clear(); } else cur.push_back(*p); }
  out.push_back(cur);
  return out;
}

}  // namespace

// ---------------------------------------------------------------------------------
// C ABI for ctypes (tests/, bench.py cpu_baseline)
// ---------------------------------------------------------------------------------
extern "C" {

struct or_index { Index ix; std::vector<std::pair<int, const std::string*>> flat; };

or_index* or_index_build(const uint8_t* blob, const uint64_t* offs, uint32_t n_docs, int q, const char* wrap0,
                         const char* wrap1, const char* pad, const char* alphabet_nl) {
  auto* h = new or_index();
  h->ix.d.q = q; h->ix.d.wrap0 = wrap0; h->ix.d.wrap1 = wrap1; h->ix.d.pad = pad;
  h->ix.d.alphabet = Alphabet::create(split_alphabet(alphabet_nl));
  // Documents are added in docID order (suggest.Index, indexer.go:32-34).  For large dictionaries the
  // work is split into contiguous docID ranges built in parallel and concatenated in range order,
  // which yields exactly the lists a sequential AddDocument loop produces.
  int n_chunks = 1;
#ifdef _OPENMP
  if (n_docs >= 200000) n_chunks = std::min(32, omp_get_max_threads());
#endif
  if (n_chunks <= 1) {
    for (uint32_t i = 0; i < n_docs; i++) {
      std::string s((const char*)blob + offs[i], (size_t)(offs[i + 1] - offs[i]));
      add_document(h->ix, i, tokenize(h->ix.d, s, false));
    }
  } else {
    std::vector<Index> parts(n_chunks);
#pragma omp parallel for schedule(static, 1) num_threads(n_chunks)
    for (int c = 0; c < n_chunks; c++) {
      parts[c].d = h->ix.d;
      uint64_t lo = (uint64_t)n_docs * c / n_chunks, hi = (uint64_t)n_docs * (c + 1) / n_chunks;
      for (uint64_t i = lo; i < hi; i++) {
        std::string s((const char*)blob + offs[i], (size_t)(offs[i + 1] - offs[i]));
        add_document(parts[c], (uint32_t)i, tokenize(h->ix.d, s, false));
      }
    }
    size_t n_seg = 0;
    for (auto& p : parts) n_seg = std::max(n_seg, p.segs.size());
    h->ix.segs.resize(n_seg);
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_chunks)
    for (int64_t sg = 0; sg < (int64_t)n_seg; sg++) {
      for (auto& p : parts) {
        if ((size_t)sg >= p.segs.size() || !p.segs[sg]) continue;
        if (!h->ix.segs[sg]) h->ix.segs[sg] = std::make_unique<Segment>();
        for (auto& kv : p.segs[sg]->terms) {
          auto& dst = h->ix.segs[sg]->terms[kv.first].v;
          dst.insert(dst.end(), kv.second.v.begin(), kv.second.v.end());
        }
        p.segs[sg].reset();
      }
    }
  }
  commit(h->ix);
  return h;
}
void or_index_free(or_index* h) { delete h; }
int or_index_segments(const or_index* h) { return (int)h->ix.segs.size(); }

// flat enumeration of (segment, term) lists for fixture comparison
uint64_t or_index_num_lists(or_index* h) {
  if (h->flat.empty())
    for (size_t s = 0; s < h->ix.segs.size(); s++)
      if (h->ix.segs[s]) for (auto& kv : h->ix.segs[s]->terms) h->flat.push_back({(int)s, &kv.first});
  return h->flat.size();
}
// returns raw (header) length; *n_stored = elements actually iterated by a search
int or_index_list_at(or_index* h, uint64_t i, int* seg, const char** term, int* term_len, const uint32_t** postings,
                     int* n_stored) {
  auto& e = h->flat[i];
  const StoredList& l = h->ix.segs[e.first]->terms.at(*e.second);
  *seg = e.first; *term = e.second->data(); *term_len = (int)e.second->size();
  *postings = l.v.data(); *n_stored = (int)l.v.size();
  return l.raw_len;
}

// tokens are written '\0'-separated; returns the token count or -1 if cap is too small
int or_tokenize(const or_index* h, const uint8_t* text, int len, int autocomplete_mode, char* out, int cap) {
  auto toks = tokenize(h->ix.d, std::string((const char*)text, len), autocomplete_mode != 0);
  int used = 0;
  for (auto& t : toks) {
    if (used + (int)t.size() + 1 > cap) return -1;
    memcpy(out + used, t.data(), t.size()); used += (int)t.size(); out[used++] = 0;
  }
  return (int)toks.size();
}
int or_ngram_tokenize(const uint8_t* text, int len, int n, char* out, int cap) {
  auto toks = ngram_tokenize(std::string((const char*)text, len), n);
  int used = 0;
  for (auto& t : toks) {
    if (used + (int)t.size() + 1 > cap) return -1;
    memcpy(out + used, t.data(), t.size()); used += (int)t.size(); out[used++] = 0;
  }
  return (int)toks.size();
}
int or_alphabet_has(const char* alphabet_nl, uint32_t rune) {
  return Alphabet::create(split_alphabet(alphabet_nl)).has(rune) ? 1 : 0;
}
int or_to_lower(const uint8_t* text, int len, char* out, int cap) {
  std::string s = go_to_lower(std::string((const char*)text, len));
  if ((int)s.size() > cap) return -1;
  memcpy(out, s.data(), s.size());
  return (int)s.size();
}

int or_metric_min_y(int m, double a, int size) { return metric_min_y(m, a, size); }
int or_metric_max_y(int m, double a, int size) { return metric_max_y(m, a, size); }
int or_metric_threshold(int m, double a, int sa, int sb) { return metric_threshold(m, a, sa, sb); }
double or_metric_distance(int m, int inter, int sa, int sb) { return metric_distance(m, inter, sa, sb); }
double or_metric_score(int m, int inter, int sa, int sb) { return 1 - metric_distance(m, inter, sa, sb); }

// algo: 0 cp_merge, 1 scan_count, 2 merge_skip, 3 divide_skip, 4 intersector.  lists are given flat.
// returns the number of collected candidates (written up to cap), -2 on "overlap overflow" panic
int or_merge(int algo, const uint32_t* flat, const uint32_t* lens, int n_lists, int threshold, uint32_t* out_pos,
             uint32_t* out_overlap, int cap) {
  std::vector<Iter> rid;
  size_t off = 0;
  for (int i = 0; i < n_lists; i++) { Iter it; it.p = flat + off; it.n = it.len = (int)lens[i]; rid.push_back(it); off += lens[i]; }
  int n = 0;
  auto coll = [&](Cand c) { if (n < cap) { out_pos[n] = c.pos; out_overlap[n] = c.overlap; } n++; return true; };
  try {
    if (algo == 4) intersect(rid, coll); else merge_dispatch(algo, rid, threshold, coll);
  } catch (OverlapOverflow&) { return -2; }
  return n;
}
int or_candidate_increment(uint32_t overlap) {  // list_merger.go:51-57; -2 == panic
  Cand c{1, overlap};
  try { increment(c); } catch (OverlapOverflow&) { return -2; }
  return (int)c.overlap;
}

int or_topk(int k, const uint32_t* keys, const double* scores, int n, uint32_t* out_keys, double* out_scores,
            double* lowest, int* can_take_probe, double probe) {
  TopK t(k);
  for (int i = 0; i < n; i++) t.add(keys[i], scores[i]);
  auto c = t.candidates();
  for (size_t i = 0; i < c.size(); i++) { out_keys[i] = c[i].key; out_scores[i] = c[i].score; }
  if (lowest) *lowest = t.lowest();
  if (can_take_probe) *can_take_probe = t.can_take(probe) ? 1 : 0;
  return (int)c.size();
}

// returns the result count, or -1 / -2 when the reference panics / dead-locks on this query
int or_suggest(const or_index* h, const uint8_t* q, int qlen, int metric, double similarity, int k, int tighten,
               int algo, uint32_t* ids, double* scores) {
  int status = 0;
  auto r = suggest(h->ix, std::string((const char*)q, qlen), metric, similarity, k, tighten != 0, algo, &status);
  if (status) return -status;
  for (size_t i = 0; i < r.size(); i++) { ids[i] = r[i].key; scores[i] = r[i].score; }
  return (int)r.size();
}
int or_autocomplete(const or_index* h, const uint8_t* q, int qlen, int limit, uint32_t* ids) {
  auto r = autocomplete(h->ix, std::string((const char*)q, qlen), limit, 0);
  for (size_t i = 0; i < r.size(); i++) ids[i] = r[i].key;
  return (int)r.size();
}

// batch drivers: queries are independent, parallelised across queries with OpenMP.
// ids/scores are n_q*k, counts n_q.  Returns the number of threads used.
int or_suggest_batch(const or_index* h, const uint8_t* blob, const uint64_t* offs, uint32_t n_q, int metric,
                     double similarity, int k, int n_threads, uint32_t* ids, double* scores, uint32_t* counts) {
  int used = 1;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
  used = n_threads > 0 ? n_threads : omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < (int64_t)n_q; i++) {
    int status = 0;
    auto r = suggest(h->ix, std::string((const char*)blob + offs[i], (size_t)(offs[i + 1] - offs[i])), metric,
                     similarity, k, false, 0, &status);
    counts[i] = status ? (0xFFFFFFFFu - (uint32_t)status + 1) : (uint32_t)r.size();  // 0xFFFFFFFF / 0xFFFFFFFE flag status 1 / 2
    for (size_t j = 0; j < r.size(); j++) { ids[i * k + j] = r[j].key; scores[i * k + j] = r[j].score; }
  }
  return used;
}
int or_autocomplete_batch(const or_index* h, const uint8_t* blob, const uint64_t* offs, uint32_t n_q, int limit,
                          int n_threads, uint32_t* ids, uint32_t* counts) {
  int used = 1;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
  used = n_threads > 0 ? n_threads : omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < (int64_t)n_q; i++) {
    auto r = autocomplete(h->ix, std::string((const char*)blob + offs[i], (size_t)(offs[i + 1] - offs[i])), limit, 0);
    counts[i] = (uint32_t)r.size();
    for (size_t j = 0; j < r.size(); j++) ids[i * limit + j] = r[j].key;
  }
  return used;
}

// algorithmic bytes of one query: 4 * sum of |postings| of every query
// term present in every admissible segment + len(query) + 12*k
uint64_t or_query_algorithmic_bytes(const or_index* h, const uint8_t* q, int qlen, int metric, double similarity, int k) {
  const Index& ix = h->ix;
  auto tokens = tokenize(ix.d, std::string((const char*)q, qlen), false);
  uint64_t bytes = (uint64_t)qlen + 12ull * (uint64_t)k;
  if (tokens.empty()) return bytes;
  int a = (int)tokens.size();
  int b_min = metric_min_y(metric, similarity, a), b_max = metric_max_y(metric, similarity, a);
  if (b_max >= (int)ix.segs.size()) b_max = (int)ix.segs.size() - 1;
  for (int b = std::max(b_min, 0); b <= b_max; b++) {
    int t = metric_threshold(metric, similarity, a, b);
    if (t == 0 || t > b || t > a || !ix.segs[b]) continue;
    for (auto& tok : tokens) {
      auto it = ix.segs[b]->terms.find(tok);
      if (it != ix.segs[b]->terms.end()) bytes += 4ull * it->second.v.size();
    }
  }
  return bytes;
}

}  // extern "C"

#include "spell_oracle.inc"
