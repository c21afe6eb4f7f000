// engine.hip — MI355X (gfx950 / CDNA4) engine of libsuggest_hip: device index replica, the fused
// per-query search kernel and the C ABI of include/suggest_hip.h.
//
// One 64-lane wavefront (= one workgroup) owns one query end to end:
//   tokenise (pkg/suggest/tokenizer.go:9-34)  ->  window [MinY,MaxY] (pkg/metric/*.go)
//   -> for every admissible cardinality segment B: stream the query terms' posting lists once
//      with 16-byte coalesced loads, count candidates in LDS (lossy u16 counters + exact
//      verification), keep docs with overlap >= T(B)
//      (the result set of searcher.Search + cpMerge.Merge, pkg/index/searcher.go:28-78,
//      pkg/merger/cp_merge.go:19-120)
//   -> score 1-Distance in IEEE double (pkg/suggest/scorer.go:29-31)
//   -> wave-level top-k by (score desc, docID asc) (pkg/suggest/collector.go:20-26, topk.go:82-147).
// Integer/index work: no MFMA; the bound is HBM bandwidth on the posting stream.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <functional>
#include <string>
#include <sys/syscall.h>
#include <unistd.h>

#include "sg_internal.h"

namespace sg {

#define SG_MAX_A 128           // == SG_MAX_QUERY_TERMS
#define SG_MAX_RUNES 144       // SG_MAX_A + 2*8 wrap runes
#define SG_K_LDS 64
#define SG_WRAP_MAX 8
#define SG_DUP_SCRATCH (2 * SG_MAX_A + 64 + 64)   // LDS words of the repeated-term (secondary entry) path (borrowed from the row table)
#define SG_TILE_MAX 64     // segments per tile (one lane each)
#define SG_UNROLL 4        // 16-byte loads in flight per lane
#define SG_PPC 7           // postings per 16-byte chunk of the packed store: {u32 first x, 6 x u16 gaps} (packed_store.inc)
#define SG_X_MASK 0x1FFFFFFFu   // first word of a chunk: x of its first posting | (postings in the chunk - 1) << 29
#define SG_PAD_GAP 41u          // gap of the padding slots behind a chunk's last posting (packed_store.inc)
#define SG_PPC8 13         // ... of a dense term's chunk: {u32 first x, 12 x u8 gaps} (packed_store.inc; the term's format = bit 31 of its seg_off entries)
#define SG_X_MASK8 0x0FFFFFFFu  // first word of such a chunk: x | (postings in the chunk - 1) << 28
#define SG_G8_FLAG 0x80000000u
#define SG_SUB 2           // rows counted per LDS round trip: SG_SUB * SG_PPC = 14 atomics issued, one wait
#define SG_EPOCHS 4           // group passes whose candidates may wait in the queue together (ring of their streamed-list masks + docID ranges)
#define SG_EPOCH_WORDS 6
// LDS layout of a search wavefront.  Two sets of table sizes: the full one, and a slim one that — with 2^11 counter words,
// k <= 20 and the small candidate queue — fits 12 800 B = 10 allocation granules, i.e. a 12th wavefront per CU (headline
// +4.5 %, cfg 2 +6 %, cfg 3 +5 %).  Where the slim tables buy no wavefront (2^12 counters: long-list indexes; the large
// queue of launches with many results) they only cost (skewed 10M -11 %, families -3.5 %), so the host picks per launch.
template <bool kSlim> struct Lds {
  static constexpr uint32_t rows_cap = kSlim ? 448u : 512u;     // u32 entries of seg_off rows kept in LDS per tile
  static constexpr uint32_t rowtab_cap = kSlim ? 64u : 84u;     // row descriptors (8 B + 1 B) per streaming window
  static constexpr uint32_t qh = kSlim ? 144u : 192u;           // slots of the query-term hash: A <= 128 occurrences (typical A ~ 20)
  static constexpr uint32_t dedup = kSlim ? 96u : 128u;         // slots of the in-batch (doc, list) table of flush_queue
  static constexpr uint32_t rowtab_words = 2u * (rowtab_cap + 2u * SG_UNROLL) + (rowtab_cap + 2u * SG_UNROLL + 7u) / 8u * 2u;   // descriptors + list ids
  // everything besides the counters, the candidate queue and the top-k rows
  static constexpr uint32_t fixed_words = SG_MAX_A + rows_cap + rowtab_words + 64u + qh + qh / 4u + SG_EPOCH_WORDS * SG_EPOCHS + 4u;
  static_assert(qh > SG_MAX_A && qh % 4u == 0u && qh <= 256u, "the query-term hash needs a free slot and whole words of positions");
  static_assert(SG_DUP_SCRATCH <= rowtab_words + 64u + qh + qh / 4u, "dup scratch must fit row table + dummies + hash");
  static_assert(2u * dedup <= rowtab_words + 64u, "dedup table must fit row table + dummies");
};
// The queue gets what is left of the wavefront's share of the CU's LDS: 160 KB in allocation granules of 1 280 B (320
// words), so a wavefront that needs g granules with the smallest useful queue (40 entries; 112 for the launches that
// expect many candidates) runs 128 / g to a CU and may as well use all of 128 / (128 / g) granules.
inline uint32_t sg_topk_words(uint32_t k) { const uint32_t kk = k < SG_K_LDS ? k : SG_K_LDS; return ((kk + 1u) & ~1u) + 2u * kk; }
inline uint32_t sg_lds_waves(uint32_t log2_cnt, uint32_t k, bool roomy, bool slim) {
  const uint32_t used = (1u << log2_cnt) + (slim ? Lds<true>::fixed_words : Lds<false>::fixed_words) + sg_topk_words(k);
  return 128u / ((used + 2u * (roomy ? 112u : 40u) + 319u) / 320u);
}
inline uint32_t sg_queue_cap(uint32_t log2_cnt, uint32_t k, bool roomy, bool slim) {
  const uint32_t used = (1u << log2_cnt) + (slim ? Lds<true>::fixed_words : Lds<false>::fixed_words) + sg_topk_words(k);
  const uint32_t waves = sg_lds_waves(log2_cnt, k, roomy, slim);
  if (waves == 0u) return 64u;
  const uint32_t c = (((128u / waves) * 320u - used) / 2u) & ~7u;
  return c < 112u ? c : 112u;
}
#define SG_MAX_PARTS 32       // parts a heavy query is cut into

struct DeviceIndex {
  // The packed posting store (packed_store.inc): documents are numbered x = seg_base[cardinality] + rank inside the segment
  // (ascending docID), a 16-byte chunk holds SG_PPC postings as {u32 first x, 6 x u16 gaps}.  Everything below the top-k
  // works in x; orig_of[] maps back where a document is offered to the top-k (whose order is by ORIGINAL docID).
  const uint32_t* postings;
  const uint32_t* cut_sample;  // [ceil(n_chunks/16)+1] x of the first posting of every 16th chunk of the posting store
  const uint32_t* seg_off;     // [n_terms * (S+1)] first chunk of list (term, segment), term-major
  const uint32_t* orig_of;     // [n_docs] x -> docID
  const uint32_t* seg_base;    // [S+1] first x of every cardinality segment
  const TermSlot* slots;
  const uint8_t* ascii_sym;    // [128]
  const uint8_t* ascii_alpha;  // [128]
  const uint32_t* na_rune;
  const uint8_t* na_sym;
  const uint8_t* na_alpha;
  const uint32_t* lower_from;
  const uint32_t* lower_to;
  // documents that repeat a term: null / 0 when the dictionary has none
  const uint32_t* dup_ts;      // [n_dups] term*S+segment, ascending (then by doc)
  const uint32_t* dup_doc;     // [n_dups]
  const uint32_t* dup_mult;    // [n_dups] occurrences of the term in the doc (>= 2)
  const uint32_t* dup_docs;    // [n_dup_docs] ascending docIDs with any repeated term
  const uint32_t* dup_bits;    // [ceil(n_docs/32)] bit d set: document d repeats a term (one load instead of a search per emitted doc)
  const uint32_t* extra_ts;    // [n_extra] term*S+segment of lists holding repeats, ascending
  const uint32_t* extra_cnt;   // [n_extra] raw length - stored length of that list
  const uint32_t* list_len;    // [n_terms*S] stored (de-duplicated) list lengths
  // forward index (doc -> its distinct terms), derived from the CSR on the device (forward_index.inc): verifying a
  // candidate is ONE coalesced read of its term list instead of a binary search in every query term's posting list
  const uint2* fwd_rec;        // [n_docs] indexed by x: {first 16-byte chunk of the doc's terms in fwd_terms, cardinality B | distinct terms << 16}
  const uint32_t* fwd_terms;   // term ids, a doc's list padded to a whole chunk with 0xFFFFFFFF
  const uint32_t* fx_base;     // [S+1] [r6] (dictionaries of <= 63 segments; else null) fwd_terms is in x order at a fixed stride per segment: document
                               // x of segment B at chunk fx_base[B] + (x - seg_base[B]) * ceil(B / 4) — fwd_rec[x].x says the same
  uint32_t n_dups, n_dup_docs, n_extra;
  uint32_t has_g8;             // some term's lists have 8-bit gaps (bit 31 of its seg_off entries): the launches take the kG8 instantiations
  uint32_t slot_mask, n_na, n_lower;
  uint32_t S, n_terms, q, n_docs;
  uint32_t wrap0[SG_WRAP_MAX], wrap1[SG_WRAP_MAX];
  uint32_t n_wrap0, n_wrap1, n_pad;
  uint8_t pad_sym[8];
};

#define SG_TABLE 5   // BatchArgs::metric: MinY / MaxY / Threshold / 1 - Distance come from host-built tables
struct MetricTab {   // [a] = query cardinality 0..a_max, [b] = segment 0..S-1, [o] = overlap 0..a_max
  const int32_t* min_y;      // [a_max + 1]
  const int32_t* max_y;      // [a_max + 1]
  const int32_t* thr;        // [(a_max + 1) * S]            Threshold(alpha, a, b)
  const double* score;       // [(a_max + 1) * S * (a_max + 1)]   1 - Distance(o, a, b), as the scorer computes it (scorer.go:29-31)
  uint32_t a_max, S;
};

struct BatchArgs {
  DeviceIndex ix;
  const uint8_t* q_blob;
  const uint64_t* q_offs;
  const uint32_t* q_len;   // null: query i is q_blob[q_offs[i] .. q_offs[i + 1]); else q_blob[q_offs[i] .. q_offs[i] + q_len[i]) (Predict's last words)
  uint32_t* out_ids;
  double* out_scores;   // null in autocomplete mode
  uint32_t* out_counts;
  uint64_t* scratch_s;  // [n_q*k] top-k working rows when k > SG_K_LDS
  uint32_t* scratch_id;
  double alpha;
  uint32_t n_q, k;
  int metric, autocomplete;   // autocomplete: 0 fuzzy top-k by score, 1 prefix search (first-k docIDs), 2 fuzzy, first-k docIDs >= ac_first (any collector)
  uint32_t log2_cnt;    // LDS counter words per wave = 1 << log2_cnt
  int t_floor;          // lowest flag threshold list skipping may leave
  uint32_t filter_level;  // row of kBucketsPer16Postings: how rarely a bucket may reach T by chance
  uint32_t cq_cap;        // entries of the candidate queue (sg_queue_cap: what the LDS budget leaves)
  // ---- heavy queries are cut into parts (ranges of segments) that other wavefronts take over ----
  uint32_t* split_ctl;    // [2] items taken (second launch), items queued (first launch); null: splitting off.  Zeroed per batch.
  uint32_t* items;        // [item_cap][4] {query, seg_lo | seg_hi << 16, slot | part << 24, -}
  uint32_t* slot_ctl;     // [slot_cap][2] parts finished, parts in total (slot = query number, queries >= slot_cap are not split)
  uint64_t* part_s;       // [slot_cap][SG_MAX_PARTS][k] top-k of each part (unordered), score bits
  uint32_t* part_id;      //   ... docIDs
  uint32_t* part_n;       // [slot_cap][SG_MAX_PARTS] entries
  uint32_t item_cap, slot_cap;
  uint32_t split_chunks;  // 16-byte chunks of postings per part of a split query ...
  uint32_t split_min;     // ... which is a query (tile) whose admissible lists hold at least this many
  // ---- spellchecker mode of autocomplete: candidates ranked by the language model ----
  const uint64_t* lm_values;   // word << 32 | count of every n-gram level, flattened; null: plain autocomplete
  const uint32_t* lm_from;     // [n_q] the continuations of query i's context are lm_values[lm_from[i] .. lm_to[i])
  const uint32_t* lm_to;       //       (sorted by word; from == to: no scorer, every candidate scores alike)
  // a launch over a subset of the batch (the spellchecker's fuzzy top-up): workgroup b runs query q_sel[b], b < *q_sel_n
  const uint32_t* q_sel;
  const uint32_t* q_sel_n;
  // ---- queries beyond the wavefront kernel's tables (more than SG_MAX_A n-grams): sg_long_kernel, HBM working memory ----
  uint8_t* long_scratch;  // [SG_LONG_SLOTS] slots of long_slot_bytes; null: such queries are flagged SG_COUNT_TOO_LONG
  uint32_t* long_lock;    // [SG_LONG_SLOTS] 0 free / 1 taken (launches on several streams share the replica's slots)
  uint64_t long_slot_bytes;
  uint32_t* long_list;    // [1 + n_q] the wavefront kernel's list of such queries: [0] = how many (zeroed per launch), then their indices
  uint32_t long_max_seg;  // documents of the largest cardinality segment (the slot's counter array)
  uint32_t ac_first;      // autocomplete / by_doc: only documents with docID >= this (a caller that wants every match pages through them)
  // ---- an opaque metric.Metric implementation (pkg/metric/metric.go:7-16), tabulated by the host: metric == SG_TABLE ----
  MetricTab mt;
  // ---- fuzzy search for ANY collector (suggester.go:78-99 hands every candidate with overlap >= T to the caller's collector):
  //      the `k` smallest docIDs >= ac_first among them instead of the k best scores; top-k key = ~docID << 32 | segment << 16 |
  //      overlap (the score is computed from it on the way out), out_aux receives the key's low word ----
  //      = autocomplete == 2 (one mode word: a second flag beside it made the compiler keep all of BatchArgs on the stack)
  uint32_t* out_aux;      // [n_q][k] segment << 16 | overlap of every row (autocomplete == 2 only; may be null)
  // ---- the tokeniser as a launch of its own (sg_terms_kernel, big batches): the search kernel then starts from the term ids ----
  int32_t* pre_A;         // [n_q] d_tokenize's result per query (null: the search kernel tokenises itself)
  uint32_t* pre_terms;    // [n_q][SG_MAX_A] its term ids
  uint32_t* fill_stat;    // {sampled fuzzy queries whose top-k ended full, sampled fuzzy queries, their results, -, u64: 16-byte chunks of postings they streamed}: cumulative
  uint32_t fill_mask;     // ... sampled: queries with (index & fill_mask) == 0 — one in 32 of a large batch, every one of a small
  unsigned long long* prof;  // phase cycle counters (only read by SG_PHASE_TIMING builds)
  uint32_t dbg_skip;         // ablation bits (SG_PHASE_TIMING builds only; results are wrong when set)
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t popc64(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }
__device__ __forceinline__ uint32_t readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// pkg/metric — IEEE binary64, Go's evaluation order; built with -ffp-contract=off
__device__ double d_floor_div_clamp(double x, double hi) { double f = floor(x); return f > hi ? hi : f; }
__device__ __forceinline__ int d_min_y(int m, double alpha, int size, const MetricTab& mt) {
  switch (m) {
    case SG_TABLE: return mt.min_y[size];
    case SG_JACCARD: return (int)ceil(alpha * (double)size);
    case SG_COSINE: return (int)ceil(alpha * alpha * (double)size);
    case SG_DICE: return (int)ceil(alpha / (2 - alpha) * (double)size);
    case SG_EXACT: return size;
    default: return 1;
  }
}
__device__ __forceinline__ int d_max_y(int m, double alpha, int size, int cap, const MetricTab& mt) {  // result clamped to cap (>= any usable bMax)
  double v;
  switch (m) {
    case SG_TABLE: { const int t = mt.max_y[size]; return t > cap ? cap : t; }
    case SG_JACCARD: v = floor((double)size / alpha); break;
    case SG_COSINE: v = floor((double)size / (alpha * alpha)); break;
    case SG_DICE: v = floor((2 - alpha) / alpha * (double)size); break;
    case SG_EXACT: v = (double)size; break;
    default: v = 32767.0; break;
  }
  return v > (double)cap ? cap : (int)v;
}
__device__ __forceinline__ int d_threshold(int m, double alpha, int a, int b, const MetricTab& mt) {
  switch (m) {
    case SG_TABLE: return (uint32_t)b < mt.S ? mt.thr[(uint32_t)a * mt.S + (uint32_t)b] : 0;
    case SG_JACCARD: return (int)ceil(alpha * (double)(a + b) / (1 + alpha));
    case SG_COSINE: return (int)ceil(alpha * sqrt((double)((long long)a * (long long)b)));   // (Go's int is 64 bits wide)
    case SG_DICE: return (int)ceil(0.5 * alpha * (double)(a + b));
    case SG_EXACT: return a;
    default: return (int)ceil(alpha * fmin((double)a, (double)b));
  }
}
__device__ __forceinline__ double d_score(int m, int inter, int a, int b, const MetricTab& mt) {  // 1 - Distance(...), two roundings
  double dist;
  switch (m) {
    case SG_TABLE: return mt.score[((uint64_t)((uint32_t)a * mt.S + (uint32_t)b)) * (mt.a_max + 1u) + (uint32_t)min(inter, (int)mt.a_max)];
    case SG_JACCARD: dist = 1 - (double)inter / (double)(a + b - inter); break;
    case SG_COSINE: dist = 1 - (double)inter / sqrt((double)((long long)a * (long long)b)); break;
    case SG_DICE: dist = 1 - (double)(2 * inter) / (double)(a + b); break;
    case SG_EXACT: dist = 0; break;
    default: dist = 1 - (double)inter / fmin((double)a, (double)b); break;
  }
  return 1 - dist;
}
// order-preserving map double -> u64 (bigger = better score)
__device__ __forceinline__ uint64_t score_bits(double x);
// Threshold tightening (suggester.go:67-68,101-103, restated exactly): once the top-k is full, a document of segment b can
// only enter it with a score >= the k-th best, i.e. with an overlap >= the smallest o whose score reaches it.  Scores are
// compared as the bit patterns the results carry, upwards from the metric's own threshold — no inverse formula, no
// rounding argument; ties stay in (the docID decides them at the insertion).  Returns omax + 1 when no overlap will do.
__device__ __noinline__ int d_tighten(int m, int t, int omax, int a, int b, uint64_t worst_s, const MetricTab mt);

__device__ __forceinline__ uint64_t score_bits(double x) {
  uint64_t b = (uint64_t)__double_as_longlong(x);
  return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
__device__ __forceinline__ double bits_score(uint64_t k) {
  uint64_t b = (k >> 63) ? (k ^ 0x8000000000000000ull) : ~k;
  return __longlong_as_double((long long)b);
}
__device__ __noinline__ int d_tighten(int m, int t, int omax, int a, int b, uint64_t worst_s, const MetricTab mt) {   // (by value: a reference into the kernel argument would put all of BatchArgs on the stack)
  while (t <= omax && score_bits(d_score(m, t, a, b, mt)) < worst_s) t++;
  return t;
}
// by_doc mode: the top-k keeps the SMALLEST docIDs; segment and overlap ride in the low word (the score is computed from
// them when the rows are written out: the same d_score call as the score-ordered path, bit for bit)
__device__ __forceinline__ uint64_t by_doc_key(uint32_t d, int overlap, int b) {
  return ((uint64_t)(~d) << 32) | (uint64_t)(((uint32_t)b & 0xFFFFu) << 16) | (uint64_t)((uint32_t)overlap & 0xFFFFu);
}
__device__ __forceinline__ bool better(uint64_t s1, uint32_t i1, uint64_t s2, uint32_t i2) {
  return s1 > s2 || (s1 == s2 && i1 < i2);  // Candidate.Less inverted, collector.go:20-26
}

__device__ uint32_t d_lower(const DeviceIndex& ix, uint32_t r) {
  if (r < 0x80) return (r - 'A' < 26u) ? r + 32 : r;
  uint32_t lo = 0, hi = ix.n_lower;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (ix.lower_from[mid] < r) lo = mid + 1; else hi = mid;
  }
  return (lo < ix.n_lower && ix.lower_from[lo] == r) ? ix.lower_to[lo] : r;
}
__device__ __forceinline__ uint32_t d_width(uint32_t r) { return r < 0x80 ? 1 : r < 0x800 ? 2 : r < 0x10000 ? 3 : 4; }

// Go `range` decoding of one rune (invalid byte -> U+FFFD, width 1)
__device__ uint32_t d_next_rune(const uint8_t* s, uint32_t n, uint32_t* adv) {
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

// normalizeFilter (normalizer.go:21-37) fused with key packing; n <= 8 runes
__device__ uint64_t d_pack_key(const DeviceIndex& ix, const uint32_t* runes, uint32_t n) {
  uint64_t k = 0;
  uint32_t len = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t r = runes[i];
    uint32_t id = 0, alpha = 0;
    if (r < 128) { id = ix.ascii_sym[r]; alpha = ix.ascii_alpha[r]; }
    else {
      uint32_t lo = 0, hi = ix.n_na;
      while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (ix.na_rune[mid] < r) lo = mid + 1; else hi = mid; }
      if (lo < ix.n_na && ix.na_rune[lo] == r) { id = ix.na_sym[lo]; alpha = ix.na_alpha[lo]; }
    }
    if (alpha) { k |= (uint64_t)id << (8 * len); len++; }
    else for (uint32_t p = 0; p < ix.n_pad; p++) { k |= (uint64_t)ix.pad_sym[p] << (8 * len); len++; }
  }
  return k;
}

// ASCII symbol table held in registers: lane l keeps {sym, alpha} of runes l and 64 + l; a lookup is two ds_bpermute
// (LDS crossbar, no memory access).  Every lane of the wave must take part in a lookup.
struct AsciiTab { uint32_t lo, hi; };
__device__ __forceinline__ AsciiTab d_ascii_tab(const DeviceIndex& ix, int lane) {
  AsciiTab t;
  t.lo = (uint32_t)ix.ascii_sym[lane] | ((uint32_t)ix.ascii_alpha[lane] << 8);
  t.hi = (uint32_t)ix.ascii_sym[64 + lane] | ((uint32_t)ix.ascii_alpha[64 + lane] << 8);
  return t;
}
// key of one n-gram of ASCII runes (all < 128), normalised like d_pack_key
__device__ __forceinline__ uint64_t d_pack_key_ascii(const DeviceIndex& ix, const AsciiTab& tab, const uint32_t* runes, uint32_t n) {
  if (ix.n_pad == 1u && n <= 4u) {
    // the usual description — one pad symbol: a rune outside the alphabet is one byte of the key like any other, the key is n
    // bytes in place (no per-lane branch over the pad string; 31 of the tokeniser's 72 us on 65 536 queries went there)
    const uint32_t pad0 = ix.pad_sym[0];
    uint32_t k32 = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4u; i++) {
      if (i < n) {
        const uint32_t r = runes[i] & 127u;
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r & 63u) << 2), (int)tab.lo);
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r & 63u) << 2), (int)tab.hi);
        const uint32_t v = r < 64u ? a : b;
        k32 |= ((v >> 8) ? (v & 0xFFu) : pad0) << (8u * i);
      }
    }
    return (uint64_t)k32;
  }
  uint64_t k = 0;
  uint32_t len = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t r = runes[i] & 127u;
    const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r & 63u) << 2), (int)tab.lo);
    const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((r & 63u) << 2), (int)tab.hi);
    const uint32_t v = r < 64u ? a : b;
    if (v >> 8) { k |= (uint64_t)(v & 0xFFu) << (8 * len); len++; }
    else for (uint32_t p = 0; p < ix.n_pad; p++) { k |= (uint64_t)ix.pad_sym[p] << (8 * len); len++; }
  }
  return k;
}

__device__ uint64_t d_mix64(uint64_t k) {
  k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull;
  k ^= k >> 27; k *= 0x94D049BB133111EBull;
  k ^= k >> 31;
  return k;
}
template <uint32_t kQH> __device__ __forceinline__ uint32_t d_qhash(uint32_t term) { return (((term * 0x9E3779B1u) >> 24) * kQH) >> 8; }   // 8 hash bits onto kQH slots
template <uint32_t kQH> __device__ __forceinline__ uint32_t d_qnext(uint32_t h) { return h + 1u == kQH ? 0u : h + 1u; }
__device__ uint32_t d_term_lookup(const DeviceIndex& ix, uint64_t key) {
  uint32_t h = (uint32_t)d_mix64(key) & ix.slot_mask;
  for (;;) {
    TermSlot s = ix.slots[h];
    if (s.term == kNoTerm) return kNoTerm;
    if (s.key == key) return s.term;
    h = (h + 1) & ix.slot_mask;
  }
}

// count of `word` among the continuations values[from .. to) of a context (sorted by word): a 64-ary search, one probe
// per lane and round (scorerNext.ScoreNext, pkg/lm/scorer_next.go:15-23: the score is monotone in this count).
__device__ uint32_t d_lm_count(const uint64_t* values, uint32_t from, uint32_t to, uint32_t word, int lane) {
  while (to - from > 64u) {
    const uint32_t step = (to - from + 63u) >> 6, pos = from + (uint32_t)lane * step;
    const uint32_t w = pos < to ? (uint32_t)(values[pos] >> 32) : 0xFFFFFFFFu;
    const uint64_t m = ballot(pos < to && w <= word);          // a prefix of the lanes
    if (!m) return 0u;
    const uint32_t at = popc64(m) - 1u;
    from += at * step;
    to = min(to, from + step);
  }
  const uint32_t pos = from + (uint32_t)lane;
  const uint64_t v = pos < to ? values[pos] : ~0ull;
  const uint64_t m = ballot(pos < to && (uint32_t)(v >> 32) == word);
  if (!m) return 0u;
  return readlane((uint32_t)v, __builtin_ctzll(m));
}

// Tokeniser: wrap -> lower -> trim -> q-grams (first-occurrence dedup) -> normalise -> term ids.
// Returns the token count A (tokens absent from the dictionary keep their slot as kNoTerm), or -1
// when the query exceeds SG_MAX_A tokens / SG_MAX_RUNES runes.  Wave-uniform control flow.
__device__ int d_tokenize(const BatchArgs& a, const uint8_t* q, uint32_t qlen, uint32_t* runes, uint64_t* keys,
                          uint32_t* term, int lane) {
  const DeviceIndex& ix = a.ix;
  const uint32_t n_w0 = ix.n_wrap0, n_w1 = a.autocomplete == 1 ? 0u : ix.n_wrap1;
  const AsciiTab tab = d_ascii_tab(ix, lane);           // (asked for up front: its loads travel with the query bytes' instead of behind them)
  // ASCII?
  bool na = false;
  for (uint32_t i = lane; i < qlen; i += 64) {            // one pass over the bytes: ASCII test and, optimistically, the runes
    const uint32_t b = q[i];
    na |= b >= 0x80;
    if (n_w0 + i < SG_MAX_RUNES) runes[n_w0 + i] = (b - 'A' < 26u) ? b + 32u : b;
  }
  const bool ascii = ballot(na) == 0;
  uint32_t R = 0, byte_len = 0;
  if (ascii) {
    R = n_w0 + qlen + n_w1;
    if (R > SG_MAX_RUNES) return -1;
    // (the wrap strings by uniform index: scalar reads of the kernel arguments — indexed by lane they were two vector loads from
    //  the argument block, a memory round trip each, at the head of every query)
    byte_len = qlen;
    for (uint32_t i = 0; i < n_w0; i++) { const uint32_t r = d_lower(ix, ix.wrap0[i]); if (lane == 0) runes[i] = r; byte_len += d_width(r); }
    for (uint32_t i = 0; i < n_w1; i++) { const uint32_t r = d_lower(ix, ix.wrap1[i]); if (lane == 0) runes[n_w0 + qlen + i] = r; byte_len += d_width(r); }
  } else {
    // rare path: every lane runs the same sequential decode (uniform), lane 0 stores
    bool too_long = false;
    for (uint32_t i = 0; i < n_w0; i++) { uint32_t r = d_lower(ix, ix.wrap0[i]); if (lane == 0) runes[R] = r; R++; byte_len += d_width(r); }
    uint32_t i = 0;
    while (i < qlen) {
      uint32_t adv;
      uint32_t r = d_lower(ix, d_next_rune(q + i, qlen - i, &adv));
      i += adv;
      if (R >= SG_MAX_RUNES - SG_WRAP_MAX) { too_long = true; break; }
      if (lane == 0) runes[R] = r;
      R++; byte_len += d_width(r);
    }
    if (too_long) return -1;
    for (uint32_t j = 0; j < n_w1; j++) { uint32_t r = d_lower(ix, ix.wrap1[j]); if (lane == 0) runes[R] = r; R++; byte_len += d_width(r); }
  }
  __syncthreads();
  // strings.Trim(text, " ")
  uint32_t t0 = 0, t1 = R;
  while (t0 < t1 && runes[t0] == ' ') { t0++; byte_len--; }
  while (t1 > t0 && runes[t1 - 1] == ' ') { t1--; byte_len--; }
  const uint32_t q_n = ix.q;
  if (byte_len < q_n) return 0;                         // ngram_tokenizer.go:18
  const uint32_t Rt = t1 - t0;
  uint32_t n_tok = 0;
  if (Rt <= q_n) {                                       // one short gram: the whole text
    if (lane == 0) {
      uint32_t w[8];
      for (uint32_t t = 0; t < Rt; t++) w[t] = runes[t0 + t];
      keys[0] = d_pack_key(ix, w, Rt);
    }
    n_tok = 1;
  } else {
    const uint32_t G = Rt - q_n + 1;
    bool big = false;                                          // (the wrap strings may hold non-ASCII runes)
    for (uint32_t i = t0 + lane; i < t1; i += 64) big |= runes[i] > 127u;
    if (ascii && q_n <= 3u && G <= 64u && ballot(big) == 0) {
      // the common case in registers: a gram of <= 3 ASCII runes is a 21-bit fingerprint, appendUnique
      // (ngram_tokenizer.go:46-54) is G wave-uniform readlanes, normalisation two ds_bpermute per rune
      const uint32_t g = (uint32_t)lane;
      const bool valid = g < G;
      uint32_t w[3] = {0, 0, 0};
      for (uint32_t t = 0; t < q_n; t++) w[t] = valid ? runes[t0 + g + t] : 0u;
      const uint32_t fp = w[0] | (w[1] << 7) | (w[2] << 14) | (valid ? 1u << 21 : 0u);
      bool dup = false;
      for (uint32_t h = 0; h + 1 < G; h++) dup |= readlane(fp, (int)h) == fp && h < g;
      const bool keep = valid && !dup;
      const uint64_t m = ballot(keep);
      const uint32_t rank = popc64(m & ((1ull << lane) - 1ull));
      const uint64_t key = d_pack_key_ascii(ix, tab, w, q_n);    // every lane takes part in the lookups
      if (keep && rank < SG_MAX_A) keys[rank] = key;
      n_tok = popc64(m);
    } else
    for (uint32_t base = 0; base < G; base += 64) {
      const uint32_t g = base + lane;
      const bool valid = g < G;
      uint32_t w[8];
      for (uint32_t t = 0; t < q_n; t++) w[t] = valid ? runes[t0 + g + t] : 0u;
      bool dup = false;                                  // appendUnique, ngram_tokenizer.go:46-54
      const uint32_t hmax = min(base + 63u, G - 1);
      for (uint32_t h = 0; h < hmax; h++) {
        bool eq = true;
        for (uint32_t t = 0; t < q_n; t++) eq &= runes[t0 + h + t] == w[t];
        dup |= eq && h < g;
      }
      const bool keep = valid && !dup;
      const uint64_t m = ballot(keep);
      const uint32_t rank = n_tok + popc64(m & ((1ull << lane) - 1ull));
      if (keep && rank < SG_MAX_A) keys[rank] = d_pack_key(ix, w, q_n);
      n_tok += popc64(m);
    }
    if (n_tok > SG_MAX_A) return -1;
  }
  __syncthreads();
  if (ix.slots) for (uint32_t i = lane; i < n_tok; i += 64) term[i] = d_term_lookup(ix, keys[i]);
  __syncthreads();
  return (int)n_tok;
}

// inclusive wave scan (64 lanes) on the DPP network: Hillis-Steele inside rows of 16 lanes
// (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 carry the row totals — no LDS traffic.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane) {
  (void)lane;
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// Go 1.14 sort.Sort (quickSort with ninther pivot, heapSort fallback, ShellSort pass + insertionSort below
// 13 elements) on (key,val) pairs in LDS ordered by key, restated from the published algorithm of the Go
// standard library the reference builds with (golang:1.14.4).  It decides the order of EQUAL-length posting
// lists in cpMerge.Merge (cp_merge.go:24) / Intersect (list_intersector.go:30), which only matters for
// documents that repeat a term.  Executed uniformly by the whole wave (rare path).
struct PairSort {
  uint32_t* k; uint32_t* v; int* stk;
  __device__ bool less(int i, int j) const { return k[i] < k[j]; }
  __device__ void swap(int i, int j) { uint32_t t = k[i]; k[i] = k[j]; k[j] = t; t = v[i]; v[i] = v[j]; v[j] = t; }
  __device__ void insertion(int a, int b) {
    for (int i = a + 1; i < b; i++) for (int j = i; j > a && less(j, j - 1); j--) swap(j, j - 1);
  }
  __device__ void sift_down(int lo, int hi, int first) {
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
  __device__ void heap_sort(int a, int b) {
    const int first = a, hi = b - a;
    for (int i = (hi - 1) / 2; i >= 0; i--) sift_down(i, hi, first);
    for (int i = hi - 1; i >= 0; i--) { swap(first, first + i); sift_down(0, i, first); }
  }
  __device__ void median3(int m1, int m0, int m2) {
    if (less(m1, m0)) swap(m1, m0);
    if (less(m2, m1)) { swap(m2, m1); if (less(m1, m0)) swap(m1, m0); }
  }
  __device__ void do_pivot(int lo, int hi, int* midlo, int* midhi) {
    const int m = (int)((unsigned)(lo + hi) >> 1);
    if (hi - lo > 40) {
      const int s = (hi - lo) / 8;
      median3(lo, lo + s, lo + 2 * s);
      median3(m, m - s, m + s);
      median3(hi - 1, hi - 1 - s, hi - 1 - 2 * s);
    }
    median3(lo, m, hi - 1);
    const int pivot = lo;
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
  __device__ void sort(int n) {
    int depth = 0;
    for (int i = n; i > 0; i >>= 1) depth++;
    depth *= 2;
    int sp = 0;
    stk[0] = 0; stk[1] = n; stk[2] = depth; sp = 1;
    while (sp > 0) {
      sp--;
      int a = stk[sp * 3], b = stk[sp * 3 + 1], d = stk[sp * 3 + 2];
      while (b - a > 12) {
        if (d == 0) { heap_sort(a, b); a = b; break; }
        d--;
        int mlo, mhi;
        do_pivot(a, b, &mlo, &mhi);
        if (mlo - a < b - mhi) { stk[sp * 3] = a; stk[sp * 3 + 1] = mlo; stk[sp * 3 + 2] = d; sp++; a = mhi; }
        else { stk[sp * 3] = mhi; stk[sp * 3 + 1] = b; stk[sp * 3 + 2] = d; sp++; b = mlo; }
      }
      if (b - a > 1) {
        for (int i = a + 6; i < b; i++) if (less(i, i - 6)) swap(i, i - 6);
        insertion(a, b);
      }
    }
  }
};

__device__ uint32_t d_lower_bound_u32(const uint32_t* p, uint32_t n, uint32_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (p[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

// Secondary entries of a doc that repeats a term (reference quirk): cpMerge keeps one
// candidate per COPY of the doc in the n-T+1 shortest lists (cp_merge.go:47-78); copy j >= 2 ends with
// overlap #{merged lists holding >= j copies} + #{probed lists holding the doc} and is collected if that
// reaches T; the intersector (n == T) collects the doc once per copy in the shortest list
// (list_intersector.go:37-70).  fm0/fm1 = query terms (lanes, round 0/1) whose list in the doc's segment
// holds it.  Every extra overlap goes to `emit_extra` as it is found — a document that repeats a term hundreds of times
// (long documents) has as many secondary entries, the reference keeps them all; `scratch` is the row-table region, idle
// whenever candidates are emitted (NOT the counters: the overflow pass emits while it still reads them).
// (A real call would cost the kernel ~100 VGPRs: kept inline.)
template <class EmitExtra>
__device__ __forceinline__ void dup_secondary_overlaps(const DeviceIndex& ix, const uint32_t* term, const uint32_t* rows,
                                                    uint32_t* scratch, int A, uint32_t stride, int w, uint32_t B, int T,
                                                    uint32_t d, uint64_t fm0, uint64_t fm1, int lane, EmitExtra&& emit_extra) {
  const uint32_t S32 = ix.S;
  uint32_t* sk = scratch;
  uint32_t* sv = scratch + SG_MAX_A;
  int* stk = (int*)(scratch + 2 * SG_MAX_A);
  const int a_rounds = (A + 63) >> 6;
  int n = 0;
  for (int r = 0; r < a_rounds; r++) {
    const int i = r * 64 + lane;
    bool present = false;
    uint32_t len = 0, mult = 0;
    if (i < A) {
      const uint32_t t = term[i];
      present = t != kNoTerm && rows[i * stride + w + 1] != rows[i * stride + w];
      if (present) {
        const uint32_t ts = t * S32 + B;
        len = ix.list_len[ts];                          // stored (de-duplicated) length
        const uint32_t e = d_lower_bound_u32(ix.extra_ts, ix.n_extra, ts);
        const uint32_t raw = len + ((e < ix.n_extra && ix.extra_ts[e] == ts) ? ix.extra_cnt[e] : 0u);
        const bool has = ((r ? fm1 : fm0) >> lane) & 1ull;
        mult = has ? 1u : 0u;
        if (raw <= 256u) {                              // VB / skip lists keep repeats; roaring (> 256) drops them
          len = raw;
          if (has) {
            uint32_t lo = d_lower_bound_u32(ix.dup_ts, ix.n_dups, ts);
            while (lo < ix.n_dups && ix.dup_ts[lo] == ts && ix.dup_doc[lo] < d) lo++;
            if (lo < ix.n_dups && ix.dup_ts[lo] == ts && ix.dup_doc[lo] == d) mult = ix.dup_mult[lo];
          }
        }
      }
    }
    const uint64_t pm = ballot(present);
    const int pos = n + (int)popc64(pm & ((1ull << lane) - 1ull));
    if (present) { sk[pos] = len; sv[pos] = mult; }
    n += (int)popc64(pm);
  }
  __syncthreads();
  PairSort ps{sk, sv, stk};
  ps.sort(n);                                           // sort.Sort(rid) by Len
  __syncthreads();
  if (n == T) {                                         // intersector: once per copy in the shortest list
    const int copies = (int)sv[0];
    for (int c = 1; c < copies; c++) emit_extra(n);
  } else {
    const int min_q = n - T + 1;
    uint32_t maxm = 0;
    int tail = 0;
    for (int p = 0; p < n; p++) { if (p < min_q) maxm = max(maxm, sv[p]); else tail += sv[p] ? 1 : 0; }
    for (uint32_t j = 2; j <= maxm; j++) {
      int c = tail;
      for (int p = 0; p < min_q; p++) c += sv[p] >= j ? 1 : 0;
      if (c >= T) emit_extra(c);
    }
  }
  __syncthreads();
}

struct TopK {  // wave-uniform state; arrays live in LDS (k <= SG_K_LDS) or in the query's output row
  uint64_t* s;
  uint32_t* id;
  uint32_t n, k;
  uint64_t worst_s;
  uint32_t worst_id, worst_pos;
};

__device__ __forceinline__ void topk_recompute_worst(TopK& tk, int lane) {
  uint64_t ws = ~0ull; uint32_t wi = 0, wp = 0xFFFFFFFFu;
  for (uint32_t i = lane; i < tk.n; i += 64) {
    uint64_t s = tk.s[i]; uint32_t d = tk.id[i];
    if (wp == 0xFFFFFFFFu || better(ws, wi, s, d)) { ws = s; wi = d; wp = i; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint64_t os = __shfl_xor(ws, off, 64); uint32_t oi = __shfl_xor(wi, off, 64), op = __shfl_xor(wp, off, 64);
    // pick the worse of the two (ties on (s,id): lower position, for determinism)
    bool take = op != 0xFFFFFFFFu && (wp == 0xFFFFFFFFu || better(ws, wi, os, oi) || (ws == os && wi == oi && op < wp));
    if (take) { ws = os; wi = oi; wp = op; }
  }
  tk.worst_s = ws; tk.worst_id = wi; tk.worst_pos = wp;
}

// topKQueue.Add (topk.go:82-102): keep the k best under (score desc, docID asc)
__device__ __forceinline__ void topk_insert(TopK& tk, uint64_t s, uint32_t d, int lane) {
  if (tk.n < tk.k) {
    if (lane == 0) { tk.s[tk.n] = s; tk.id[tk.n] = d; }
    tk.n++;
    if (tk.n == tk.k) { __syncthreads(); topk_recompute_worst(tk, lane); }
    return;
  }
  if (!better(s, d, tk.worst_s, tk.worst_id)) return;
  if (lane == 0) { tk.s[tk.worst_pos] = s; tk.id[tk.worst_pos] = d; }
  __syncthreads();
  topk_recompute_worst(tk, lane);
}

// Buckets a group of `postings` postings needs so that flagged postings that are no matches stay rare: postings / lambda(T).
// m16[level][T] = ceil(16 / lambda), lambda = postings per bucket (Poisson; tables generated with scipy.stats.poisson):
//   levels 0-3: a BUCKET reaches T by chance with probability 3e-5 / 1e-5 / 3e-6 / 1e-6 — what a false candidate was worth
//               when verifying one cost a binary search in every query term's list;
//   levels 4-7: a POSTING finds T-1 others in its bucket with probability 1e-3 / 3e-3 / 1e-2 / 3e-2 — verification through
//               the forward index is one 128-byte read, so far more false candidates are affordable: 2-3x fewer buckets, i.e.
//               larger groups, fewer docID-range passes, deeper list skipping.  T > 32 uses T = 32.
__device__ const uint8_t kBucketsPer16Postings[8][33] = {
    {255, 255, 255, 255, 95, 47, 28, 19, 14, 11, 9, 7, 6, 6, 5, 4, 4, 4, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2},
    {255, 255, 255, 255, 126, 59, 35, 23, 17, 13, 10, 8, 7, 6, 5, 5, 4, 4, 4, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2},
    {255, 255, 255, 255, 171, 76, 43, 28, 19, 15, 12, 9, 8, 7, 6, 5, 5, 4, 4, 4, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2},
    {255, 255, 255, 255, 226, 95, 52, 33, 23, 17, 13, 11, 9, 7, 6, 6, 5, 5, 4, 4, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2},
    {255, 255, 255, 255, 84, 38, 22, 15, 11, 9, 7, 6, 5, 4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1},
    {255, 255, 255, 202, 57, 28, 17, 12, 9, 7, 6, 5, 4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1},
    {255, 255, 255, 108, 37, 20, 13, 9, 7, 6, 5, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1},
    {255, 255, 255, 60, 25, 14, 10, 7, 6, 5, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1}};
// The launch's row of the table lives in a register — lane t holds m16[level][min(t, 32)] — because the lookups sit on
// the critical path of every group's setup (group sizes, list skipping, counter geometry: two dependent lookups per group)
// and a load from the table in memory is a VMEM round trip under a saturated memory system: ~1 us each, a fifth of a
// headline query's time.  Uniform T: v_readlane; T per lane: ds_bpermute.
__device__ __forceinline__ uint32_t buckets_needed(uint32_t postings, int T, uint32_t m16_lane) {       // T uniform
  const uint32_t m16 = readlane(m16_lane, T < 0 ? 0 : (T > 32 ? 32 : T));
  return (postings * m16) >> 4;
}
__device__ __forceinline__ uint32_t buckets_needed_lane(uint32_t postings, int T, uint32_t m16_lane) {  // T differs by lane
  const uint32_t m16 = (uint32_t)__builtin_amdgcn_ds_bpermute((T < 0 ? 0 : (T > 32 ? 32 : T)) << 2, (int)m16_lane);
  return (postings * m16) >> 4;
}

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8v __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
static_assert(SG_UNROLL == 4 && SG_SUB == 2 && SG_SUB * SG_PPC <= 16, "u32x16 below holds SG_SUB rows x SG_PPC postings");

// The SG_PPC slots of one packed chunk: first x, then six 16-bit gaps.  Returns how many of them are postings (the top
// three bits of the first word); the slots behind them are padding with gaps of SG_PAD_GAP — phantom numbers that no consumer but
// the lossy counters of the stream may take for postings.
template <class V>
__device__ __forceinline__ uint32_t decode_chunk(const uint4& v, V& p, int at) {
  p[at] = v.x & SG_X_MASK;
  p[at + 1] = p[at] + (v.y & 0xFFFFu); p[at + 2] = p[at + 1] + (v.y >> 16);
  p[at + 3] = p[at + 2] + (v.z & 0xFFFFu); p[at + 4] = p[at + 3] + (v.z >> 16);
  p[at + 5] = p[at + 4] + (v.w & 0xFFFFu); p[at + 6] = p[at + 5] + (v.w >> 16);
  return (v.x >> 29) + 1u;
}

// ... and the SG_PPC8 slots of a dense term's chunk: first x, then twelve 8-bit gaps
template <class V>
__device__ __forceinline__ uint32_t decode_chunk8(const uint4& v, V& p, int at) {
  p[at] = v.x & SG_X_MASK8;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    p[at + 1 + e] = p[at + e] + ((v.y >> (8 * e)) & 0xFFu);
  }
#pragma unroll
  for (int e = 0; e < 4; e++) p[at + 5 + e] = p[at + 4 + e] + ((v.z >> (8 * e)) & 0xFFu);
#pragma unroll
  for (int e = 0; e < 4; e++) p[at + 9 + e] = p[at + 8 + e] + ((v.w >> (8 * e)) & 0xFFu);
  return (v.x >> 28) + 1u;
}

// Counts SG_SUB rows (a row = up to 64 consecutive 16-byte chunks of ONE posting list, one chunk = SG_PPC postings per
// lane): one LDS atomic per posting (U8: four u8 counters per word, else one u32 counter per word), issued back to back
// and waited for once.  live[u] is 1 for lanes inside the list, 0 for lanes past its end (they re-read the list's last
// chunk and add 0).  Returns the ballot of lanes holding a posting whose bucket reached T; `pp` receives the decoded
// postings, `was` the counts seen.
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// LDS byte address of the counter word of doc d.  amask selects the bucket bits of the docID and is
// already a byte mask, cbase is the LDS address of the counter array, so the address is ONE v_and_or_b32:
//   u32 counters: bucket = docID bits [2, lg+2)   -> amask = ((1<<lg)-1) << 2
//   u8  counters: bucket = docID bits [0, lg), four per word -> amask = ((1<<lg)-1) & ~3, byte lane = d & 3
__device__ __forceinline__ lds_u32* counter_word(uint32_t d, uint32_t amask, uint32_t cbase) {
  return (lds_u32*)(uintptr_t)((d & amask) | cbase);
}

template <bool U8>
__device__ __forceinline__ uint64_t count_rows(const uint4 (&v)[SG_SUB], const uint32_t (&live)[SG_SUB], uint32_t amask,
                                               uint32_t cbase, uint32_t dummy, uint32_t Tm1, u32x16& pp, u32x16& was, uint32_t& mx) {
  uint32_t old[SG_SUB * SG_PPC];
#pragma unroll
  for (int u = 0; u < SG_SUB; u++) {
    // the six gaps.  EVERY slot adds 1: the padding of a chunk (the end of a list, or a chunk cut short by a gap above
    // 65 535) has gaps of SG_PAD_GAP, i.e. its slots are phantom postings of numbers behind the chunk's last one — each in a
    // bucket of its own, a per cent or two of noise in counters that are upper bounds anyway (they overcount, never
    // undercount; what enters the candidate queue is checked against the chunk's posting count, flagged()).  Padding as
    // repeats of the last posting (gaps of 0, rounds 1-3) needed an increment of its own per slot — min(gap, live), one
    // VALU instruction in seven — because counted, the copies pre-loaded one bucket with up to six postings.
    const uint32_t g[SG_PPC] = {0u, v[u].y & 0xFFFFu, v[u].y >> 16, v[u].z & 0xFFFFu, v[u].z >> 16, v[u].w & 0xFFFFu, v[u].w >> 16};
    // Lanes past the end of the list hold copies of its last chunk; aimed at the real counters they would
    // count it again.  They add to a lane-private dummy word instead (never read): per ROW two selects, per
    // posting still one v_and_or_b32.
    const uint32_t am = live[u] ? amask : 0u;
    const uint32_t cb = live[u] ? cbase : dummy;
    uint32_t d = v[u].x & SG_X_MASK;
#pragma unroll
    for (int e = 0; e < SG_PPC; e++) {
      if (e) d += g[e];
      pp[u * SG_PPC + e] = d;
      const uint32_t inc = 1u;
      lds_u32* w = counter_word(d, am, cb);
      if (U8) {
        // byte lane = docID & 3: the shift amount is (d << 3) mod 32 — the hardware shifters and v_bfe use
        // only the low 5 bits, so no masking instruction is needed
        old[u * SG_PPC + e] = __hip_atomic_fetch_add(w, inc << ((d << 3) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        old[u * SG_PPC + e] = __hip_atomic_fetch_add(w, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  // one wait for all the returns (left alone the compiler may stage it: lgkmcnt(11), (10), (8) ... — a dozen more
  // instructions in a loop that is issue-bound)
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), vmcnt / expcnt untouched
  mx = 0;
#pragma unroll
  for (int u = 0; u < SG_SUB; u++) {
    uint32_t mu = 0;
#pragma unroll
    for (int e = 0; e < SG_PPC; e++) {
      uint32_t o = old[u * SG_PPC + e];
      if (U8) o = __builtin_amdgcn_ubfe(o, pp[u * SG_PPC + e] << 3, 8u);   // v_bfe_u32 reads offset[4:0] only
      was[u * SG_PPC + e] = o;
      mu = max(mu, o);
    }
    mx = max(mx, live[u] ? mu : 0u);
  }
  return ballot(mx >= Tm1);
}

// ONE row of a dense term (8-bit gaps: SG_PPC8 = 13 postings per lane) — as count_rows, 13 atomics and one wait.
template <bool U8>
__device__ __forceinline__ uint64_t count_row8(const uint4& v, uint32_t live, uint32_t amask, uint32_t cbase, uint32_t dummy, uint32_t Tm1,
                                               u32x16& pp, u32x16& was, uint32_t& mx) {
  uint32_t old[SG_PPC8];
  const uint32_t am = live ? amask : 0u;
  const uint32_t cb = live ? cbase : dummy;
  uint32_t d = v.x & SG_X_MASK8;
#pragma unroll
  for (int e = 0; e < SG_PPC8; e++) {
    if (e) { const uint32_t w = e <= 4 ? v.y : e <= 8 ? v.z : v.w; d += (w >> (8 * ((e - 1) & 3))) & 0xFFu; }   // (v_add_u32_sdwa BYTE_n)
    pp[e] = d;
    lds_u32* w = counter_word(d, am, cb);
    if (U8) old[e] = __hip_atomic_fetch_add(w, 1u << ((d << 3) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else old[e] = __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  uint32_t mu = 0;
#pragma unroll
  for (int e = 0; e < SG_PPC8; e++) {
    uint32_t o = old[e];
    if (U8) o = __builtin_amdgcn_ubfe(o, pp[e] << 3, 8u);
    was[e] = o;
    mu = max(mu, o);
  }
  mx = live ? mu : 0u;
  return ballot(mx >= Tm1);
}

#ifdef SG_PHASE_TIMING   // tools/phase_timing.py: where do a wavefront's cycles go (s_memtime brackets)
#define PH_DECL long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ph_last = clock64();
#define PH(n) { const long long ph_now = clock64(); ph_acc[n] += ph_now - ph_last; ph_last = ph_now; }
#define PH_FLUSH if (lane < 8 && a.prof && !((a.dbg_skip & 65536u) && !kLM) && !((a.dbg_skip & 131072u) && kLM)) { atomicAdd(a.prof + (qi & 4095u) * 8 + lane, (unsigned long long)ph_acc[lane]); \
                                             atomicAdd(a.prof + 4096 * 8 + (qi & 4095u) * 8 + lane, (unsigned long long)dbg_n[lane]); }
#define DBG_SKIP(bit) (a.dbg_skip & (bit))
#define DBG_COUNT(slot, v) dbg_n[slot] += (v);
#define DBG_DECL long long dbg_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
#define DBG_SKIP(bit) false
#define DBG_COUNT(slot, v)
#define DBG_DECL
#define PH_DECL
#define PH(n)
#define PH_FLUSH
#endif

// minimum over the wave (DPP network, as wave_scan_incl)
__device__ __forceinline__ int wave_min_i32(int v) {
  const int I = 0x7FFFFFFF;
  v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x111, 0xf, 0xf, false));   // row_shr:1 (lanes without a source keep I)
  v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x112, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x114, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x118, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return (int)readlane((uint32_t)v, 63);
}

// ------------------------------------------------------------------------------------------
// The fused search kernel.  grid = n_q workgroups of one wavefront; dynamic LDS per wave:
//   cnt[1<<log2_cnt] u32 (tokeniser scratch aliases it) | term | rows | cand, candw | topk
//
// Per query: the admissible window of cardinality segments is cut into tiles; for a tile the
// chunk offsets seg_off[term][b] of every query term are fetched once into LDS (`rows`), lane w
// then owns segment w of the tile (its threshold T, posting volume, number of present terms).
// Consecutive valid segments are merged into groups (term-major CSR keeps a term's postings for
// consecutive segments contiguous, so a group is still one contiguous range per term) and each
// group is streamed once, list by list: a row = 64 consecutive 16-byte chunks of one list (one
// coalesced 1 KiB read), SG_UNROLL rows in flight and the next batch's loads issued before the
// current batch is counted  ->  one LDS atomic per posting  ->  postings whose bucket reaches T
// are verified exactly (binary searches in the term lists of the doc's own segment), scored and
// offered to the wave's top-k.
// ------------------------------------------------------------------------------------------
// kParts = false: one workgroup per query (the batch launch).  kParts = true: the second launch, a few thousand
// persistent wavefronts that take the queued parts of split queries off the item queue until it is empty.
// kLM = true: the spellchecker's autocomplete (candidates ranked by the language model) — its own instantiation, so the
// search kernel proper carries none of its registers (with the LM code inlined the shared kernel spilled 750 bytes per
// lane and the headline batch went from 2.6 to 4.4 ms).
// kTight = true: threshold tightening (see d_tighten) — pays where queries have many more matches than k (real
// dictionaries: cars +19 %, words +38 %) and costs the others ~4 % in registers, so it is its own instantiation too; the
// host picks per launch from the share of recent queries whose top-k filled (fill_stat).
// kG8 = true: the index has dense terms with 8-bit gaps (13 postings per chunk; packed_store.inc) — their rows are decoded
// and counted one at a time (count_row8).  Such an index is a long-list index: 2^12 counter words, 7 wavefronts per CU by
// the LDS, so these instantiations may take the registers of two wavefronts per SIMD.
// kLoop = true [r5]: the launch behind the three-launch pipeline (pipeline.inc) — a few thousand persistent wavefronts that walk
// the list of queries the pipeline left to this kernel (q_sel / q_sel_n), a stride of the grid apart; an empty list: they leave
// at once.  Its own instantiation: the batch launches carry none of it.
template <bool kParts, bool kLM, bool kTight, bool kSlim, bool kG8, bool kLoop = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kG8 ? 2 : 3, kG8 ? 2 : 3))) void sg_search_kernel_t(const BatchArgs a) {
  using L = Lds<kSlim>;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x;
  const DeviceIndex& ix = a.ix;
  const uint32_t cnt_words = 1u << a.log2_cnt;
  uint32_t* cnt = smem;
  const uint32_t cbase = (uint32_t)(uintptr_t)(lds_u32*)cnt;   // LDS byte address of the counters (aligned to their size)
  uint32_t* term = cnt + cnt_words;
  uint32_t* rows = term + SG_MAX_A;                 // (moved up against the query's last term once A is known: see wt_max)
  const uint32_t cq_cap = a.cq_cap;                                  // (what the LDS budget leaves: sg_queue_cap, set by the host —
                                                                     //  computed here, the LDS pointers behind the queue cost registers)
  uint32_t* cq_doc = rows + L::rows_cap;            // candidate queue: docs whose bucket reached the flag threshold ...
  uint32_t* cq_jj = cq_doc + cq_cap;                // ... and the query position of the list each was met in (later: the verdict)
  uint32_t* rowtab = cq_jj + cq_cap;                // the row table: {first chunk, live lanes} per row (8-byte aligned) ...
  uint8_t* rowlist = (uint8_t*)(rowtab + 2 * (L::rowtab_cap + 2 * SG_UNROLL));   // ... and the list (query position) of each row
  uint32_t* dummy_w = rowtab + 2 * (L::rowtab_cap + 2 * SG_UNROLL) + (L::rowtab_cap + 2 * SG_UNROLL + 7) / 8 * 2;   // then one
  const uint32_t dummy_lane = (uint32_t)(uintptr_t)(lds_u32*)(dummy_w + lane);   // private dummy counter word per lane
  // the query's terms as an LDS hash (linear probing; an occurrence = an entry, so a repeated term sits in consecutive
  // probe slots): what a candidate's own term list (forward index) is matched against
  uint32_t* qh_key = dummy_w + 64;
  uint8_t* qh_pos = (uint8_t*)(qh_key + L::qh);     // query position of the entry
  // the repeated-term path (documents that repeat a term: rare) borrows row table + dummies + hash — all idle while
  // candidates are emitted — and rebuilds the hash afterwards
  uint32_t* dup_scratch = rowtab;
  uint32_t* ep_ring = qh_key + L::qh + L::qh / 4;   // [SG_EPOCHS][6] streamed-list mask (128 bits over query positions) + docID range of a group pass
  // {k-th best score the tile's thresholds were tightened against (2 words), first segment still to come, -}: in LDS, not
  // in scalar registers — the stream loop has none to spare
  uint32_t* tile_state = ep_ring + SG_EPOCH_WORDS * SG_EPOCHS;
  uint32_t* tk_id_lds = tile_state + 4;                          // top-k rows: min(k, SG_K_LDS) ids, then as many 64-bit scores
  uint64_t* tk_s_lds = (uint64_t*)(tk_id_lds + ((min(a.k, (uint32_t)SG_K_LDS) + 1u) & ~1u));
  // tokeniser scratch inside the counter region: runes[SG_MAX_RUNES] then keys[SG_MAX_A]
  uint32_t* runes = cnt;
  uint64_t* keys = (uint64_t*)(cnt + SG_MAX_RUNES);

  const uint32_t k = a.k;
  const uint32_t m16_lane = kBucketsPer16Postings[a.filter_level][min(lane, 32)];   // (see buckets_needed)
  PH_DECL
  DBG_DECL

  const bool splitting = a.split_ctl != nullptr;
  const bool primary = !kParts;
  uint32_t qi = blockIdx.x;
  if (kLoop) {
    if (qi >= __builtin_amdgcn_readfirstlane(*a.q_sel_n)) return;
    if (lane == 0) tile_state[3] = qi;                    // (the position in the list lives in LDS across a query: no register for it)
    qi = __builtin_amdgcn_readfirstlane(a.q_sel[qi]);
  } else
  if (!kParts && a.q_sel) {
    // (both loads issued together — the list has an entry for every workgroup of the grid —: one memory round trip at the
    //  head of every query instead of two)
    const uint32_t sel_n = *a.q_sel_n, sel_q = a.q_sel[qi];
    if (qi >= __builtin_amdgcn_readfirstlane(sel_n)) return;
    qi = __builtin_amdgcn_readfirstlane(sel_q);
  }
  int r_lo = 0, r_hi = 0x7FFFFFFF;                      // segments this wavefront handles (a part of a split query)
  uint32_t my_slot = 0xFFFFFFFFu, my_part = 0;
  for (;;) {
  if (kParts) {
    const uint32_t n_items = min(__builtin_amdgcn_readfirstlane(a.split_ctl[1]), a.item_cap);   // queued by the first launch
    if (n_items == 0u) break;
    uint32_t idx = 0;
    if (lane == 0) idx = __hip_atomic_fetch_add(a.split_ctl + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    idx = __builtin_amdgcn_readfirstlane(idx);
    if (idx >= n_items) break;
    const uint32_t* it = a.items + (uint64_t)idx * 4;
    qi = __builtin_amdgcn_readfirstlane(it[0]);
    if (qi == 0xFFFFFFFFu) continue;                    // a hole: reserved by a query that then found no room
    r_lo = (int)(__builtin_amdgcn_readfirstlane(it[1]) & 0xFFFFu); r_hi = (int)(__builtin_amdgcn_readfirstlane(it[1]) >> 16);
    my_slot = __builtin_amdgcn_readfirstlane(it[2]) & 0xFFFFFFu; my_part = __builtin_amdgcn_readfirstlane(it[2]) >> 24;
    __syncthreads();
  }
  do {   // one query (or one part of one): `break` leaves it
  const uint32_t lm_from = kLM ? a.lm_from[qi] : 0u, lm_to = kLM ? a.lm_to[qi] : 0u;
  // The continuations of the query's context, one (word << 32 | count) per lane when there are at most 64 of them (the
  // usual case: a bigram context of a 50 M-token model has a handful): ScoreNext of a candidate is then a compare across
  // the wave instead of a search in HBM per prefix match — that search, one dependent round trip per match inside the
  // serial emit loop, was what the LM-ranked autocomplete spent its time on (0.05 of the HBM peak).
  const bool lm_small = kLM && lm_to - lm_from <= 64u;
  uint32_t lm_w = 0xFFFFFFFFu, lm_c = 0u;
  if (kLM && lm_small && lm_from + (uint32_t)lane < lm_to) { const uint64_t v = a.lm_values[lm_from + (uint32_t)lane]; lm_w = (uint32_t)(v >> 32); lm_c = (uint32_t)v; }
  uint32_t* out_ids = a.out_ids + (uint64_t)qi * k;
  double* out_scores = a.out_scores ? a.out_scores + (uint64_t)qi * k : nullptr;

  if (DBG_SKIP(32u)) { if (lane == 0) a.out_counts[qi] = 0; break; }
  int A;
  if (a.pre_A) {
    // sg_terms_kernel has tokenised the batch: the term ids are one coalesced read (issued together with the count)
    // instead of the tokeniser's chain of dependent round trips — offsets, bytes, term table — at the head of every query
    const uint32_t* pt = a.pre_terms + (uint64_t)qi * SG_MAX_A;
    const uint32_t t0 = pt[lane];
    A = __builtin_amdgcn_readfirstlane(a.pre_A[qi]);
    if (lane < A) term[lane] = t0;
    if (A > 64 && 64 + lane < A) term[64 + lane] = pt[64 + lane];
    __syncthreads();
  } else {
    const uint64_t qb = a.q_offs[qi], qe = a.q_len ? qb + a.q_len[qi] : a.q_offs[qi + 1];
    A = d_tokenize(a, a.q_blob + qb, (uint32_t)(qe - qb), runes, keys, term, lane);
  }
  PH(0)
  if (DBG_SKIP(64u)) { if (lane == 0) a.out_counts[qi] = (uint32_t)A; break; }
  if (A < 0) {                                                 // beyond this kernel's tables: flagged, and listed for sg_long_kernel
    if (lane == 0) {
      a.out_counts[qi] = SG_COUNT_TOO_LONG;
      if (a.long_list) a.long_list[1u + atomicAdd(a.long_list, 1u)] = qi;
    }
    break;
  }
  if (A == 0) { if (lane == 0) a.out_counts[qi] = 0; break; }
  if (a.metric == SG_TABLE && (uint32_t)A > a.mt.a_max) { if (lane == 0) a.out_counts[qi] = SG_COUNT_TOO_LONG; break; }   // the tables end before this cardinality
  auto build_qhash = [&]() {
    for (uint32_t i = lane; i < L::qh; i += 64) qh_key[i] = kNoTerm;
    __syncthreads();
    for (int i = lane; i < A; i += 64) {
      const uint32_t x = term[i];
      if (x == kNoTerm) continue;
      for (uint32_t h = d_qhash<L::qh>(x);; h = d_qnext<L::qh>(h))
        if (atomicCAS(qh_key + h, kNoTerm, x) == kNoTerm) { qh_pos[h] = (uint8_t)i; break; }
    }
    __syncthreads();
  };
  build_qhash();

  const int S = (int)ix.S;
  int b_min, b_max;
  if (a.autocomplete == 1) { b_min = A; b_max = S - 1; }     // autocomplete.go:47
  else {
    b_min = d_min_y(a.metric, a.alpha, A, a.mt);
    b_max = d_max_y(a.metric, a.alpha, A, S, a.mt);       // suggester.go:54-59
    if (b_max >= S) b_max = S - 1;
    const int span = b_max - b_min + 1;                    // suggester.go:62 make(chan int, span)
    if (span < 0) { if (lane == 0) a.out_counts[qi] = SG_COUNT_REF_PANIC; break; }
    if (span == 0) { if (lane == 0) a.out_counts[qi] = SG_COUNT_REF_DEADLOCK; break; }
  }
  b_min = max(max(b_min, 0), r_lo);                        // a part of a split query: its range of segments
  b_max = min(b_max, r_hi);

  TopK tk;
  const bool tk_in_lds = k <= SG_K_LDS;
  tk.s = tk_in_lds ? tk_s_lds : a.scratch_s + (uint64_t)qi * k;
  tk.id = tk_in_lds ? tk_id_lds : a.scratch_id + (uint64_t)qi * k;
  tk.n = 0; tk.k = k; tk.worst_s = 0; tk.worst_id = 0; tk.worst_pos = 0;

  const uint4* __restrict__ post4 = (const uint4*)ix.postings;
  const int a_rounds = (A + 63) >> 6;                          // 1 or 2 (A <= SG_MAX_A = 128)
  // The term ids take A of their SG_MAX_A words; the rest joins the segment table behind them: a query of 20 terms gets 556
  // words instead of 448 (slim layout) and its whole window — 23 segments at Jaccard 0.5 — becomes ONE tile instead of two
  // (one fetch of the seg_off rows, one round of statistics and thresholds, groups that merge across the former seam).
  const uint32_t term_words = ((uint32_t)A + 3u) & ~3u;
  rows = term + term_words;
  const int wt_max = min(SG_TILE_MAX, (int)((L::rows_cap + SG_MAX_A - term_words) / (uint32_t)A) - 1);   // A <= 128 -> >= 3
  const uint32_t max_buckets = cnt_words * 4u;                 // u8 mode
  // A dictionary with no more documents than the wavefront has u8 counters (the reference's own test dictionaries; 8 192
  // at the default size): every document gets a counter of its own.  Nothing is skipped and nothing is flagged on the way
  // — the counters are read once after the group's stream, and what reaches the lowest threshold of the group is verified
  // as usual (the padding's phantom postings make the counts upper bounds even here).  The lossy scheme on such a
  // dictionary queued every match once per streamed list behind its T'-th and verified ~15 candidates per result (cars:
  // verification 43 % of a wavefront's time, the flagged path another 10 %).
  const bool tiny = ix.n_docs <= max_buckets;

  uint32_t pushed = 0;                                         // parts of this query queued for the second launch
  uint32_t q_chunks = 0;                                       // 16-byte chunks of postings streamed for this query (sampled into fill_stat[3])
  for (int tb = b_min; tb <= b_max; tb += wt_max) {
    const int Wt = min(wt_max, b_max - tb + 1);
    const uint32_t stride = (uint32_t)Wt + 1;
    // ---- chunk offsets of every query term for segments tb .. tb+Wt (searcher.go:38-58) ----
    __syncthreads();
    // (e / stride by a 20-bit reciprocal: exact for e * stride < 2^20 — e < 128 * 65 — and two full-rate instructions
    //  where the compiler's unsigned division is twenty)
    const uint32_t rcp = (1u << 20) / stride + 1u;
    // (eight loads in flight per lane: as a plain loop — load, wait, store to LDS — the table of a 20-term query was nine
    //  dependent memory round trips, a fifth of a small-dictionary query's time)
    const uint32_t n_e = (uint32_t)A * stride;
    for (uint32_t e0 = lane; e0 < n_e; e0 += 64u * 8u) {
      uint32_t got[8];
#pragma unroll
      for (uint32_t u = 0; u < 8u; u++) {
        const uint32_t e = e0 + 64u * u;
        got[u] = 0u;
        if (e < n_e) {
          const uint32_t i = __umul24(e, rcp) >> 20, w = e - __umul24(i, stride);
          const uint32_t t = term[i];
          if (t != kNoTerm) got[u] = ix.seg_off[(uint64_t)t * (uint32_t)(S + 1) + (uint32_t)tb + w];
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < 8u; u++) {
        const uint32_t e = e0 + 64u * u;
        if (e < n_e) rows[e] = got[u];
      }
    }
    cnt[lane] = 0u; cnt[64 + lane] = 0u;                          // (the counters are idle between groups: scratch of the statistics)
    if (kG8) cnt[128 + lane] = 0u;
    __syncthreads();
    // ---- posting volume and present terms per segment: every lane takes (term, segment) pairs — all 64 lanes busy — and
    //      adds into the segment's two words; lane w then owns segment tb+w ----
    for (uint32_t e = lane; e < (uint32_t)A * stride; e += 64) {
      const uint32_t i = __umul24(e, rcp) >> 20, w = e - __umul24(i, stride);
      if (w < (uint32_t)Wt) {
        const uint32_t len = rows[e + 1] - rows[e];             // (kG8: bit 31 of a term's entries is its format — the same in both)
        if (len) {
          atomicAdd(cnt + w, len); atomicAdd(cnt + 64 + w, 1u);
          if (kG8) atomicAdd(cnt + 128 + w, len * ((rows[e] & SG_G8_FLAG) ? (uint32_t)SG_PPC8 : (uint32_t)SG_PPC));   // posting slots
        }
      }
    }
    __syncthreads();
    const uint32_t seg_vol = cnt[lane], seg_ne = cnt[64 + lane];
    const uint32_t seg_slots = kG8 ? cnt[128 + lane] : 0u;
    __syncthreads();
    // ---- lane w owns segment tb+w: posting volume, present terms, threshold.  kTight: computed again (for the segments
    //      still to come, w >= w_start) whenever the k-th best score has moved: the thresholds tighten with it. ----
    // kTight: the segments from the query's own cardinality upwards go first, those below it second — the reference walks
    // the window inside-out for the same reason (suggester.go:64-75): the best matches sit next to |A|, the top-k fills
    // with them and the thresholds of everything further out tighten before it is streamed.  (Ascending from b_min, a low
    // similarity over a dictionary of near-duplicates filled the top-k with the worst admissible documents first: a
    // million verifications per query.)  The result does not depend on the order: the top-k's order is total.
    const bool may_split = splitting && primary && k <= SG_K_LDS && qi < a.slot_cap;
    int h_n = 1, h_lo[2] = {0, 0}, h_hi[2] = {Wt, Wt};
    if (kTight && !may_split && !a.autocomplete && A > tb && A - tb < Wt) { h_n = 2; h_lo[0] = A - tb; h_hi[1] = A - tb; }
    for (int half = 0; half < h_n; half++) {
    if (kTight && lane == 0) tile_state[2] = (uint32_t)h_lo[half];
    for (;;) {
    int w_start = 0;
    if (kTight) { __syncthreads(); w_start = (int)tile_state[2]; }
    uint32_t seg_tot = 0;
    int seg_T = 0;
    bool seg_valid = false;
    if (lane < Wt) {
      const uint32_t ne = seg_ne;
      seg_tot = seg_vol;
      const int B = tb + lane;
      if (a.autocomplete == 1) { seg_T = A; seg_valid = true; }
      else {
        seg_T = d_threshold(a.metric, a.alpha, A, B, a.mt);
        seg_valid = !(seg_T == 0 || seg_T > B || seg_T > A);       // suggester.go:76
        if (kTight && seg_valid && tk.n == k) {                      // the top-k is full (the docID-ordered modes never get here: seg_T is set above / the host takes the plain instantiation)
          // (a query that repeats a term counts it per occurrence: overlaps reach A even where B is smaller)
          seg_T = d_tighten(a.metric, seg_T, A, A, B, tk.worst_s, a.mt);
          seg_valid = seg_T <= A;
        }
      }
      seg_valid = seg_valid && (int)ne >= seg_T && (!kTight || (lane >= w_start && lane < h_hi[half]));   // searcher.go:32 (fewer present terms than T)
    }
    const uint64_t vmask = ballot(seg_valid);
    if (kTight && lane == 0) {                                       // the k-th best score these thresholds were tightened against
      const uint64_t ts = tk.n == k ? tk.worst_s : 0ull;
      tile_state[0] = (uint32_t)ts; tile_state[1] = (uint32_t)(ts >> 32);
    }
    bool again = false;
    // ---- split decision (first launch, top-k in LDS): a tile whose admissible lists hold >= 2 x split_chunks chunks
    //      is cut into parts of consecutive segments of about equal volume, which are queued for the second launch;
    //      this wavefront goes on with the next tile.  A lone heavy query would otherwise be one wavefront's serial
    //      work (skewed dictionaries: 40x the mean) and set the batch's latency. ----
    if (may_split && (!kTight || w_start == 0)) {
      const uint32_t vol = seg_valid ? seg_tot : 0u;
      const uint32_t incl = wave_scan_incl(vol, lane);
      const uint32_t total = readlane(incl, 63);
      const uint32_t P = min(min((uint32_t)SG_MAX_PARTS - 1u - pushed, total / max(a.split_chunks, 1u)), popc64(vmask));
      if (P >= 2u && total >= a.split_min) {
        // part of a segment: where the middle of its volume falls; parts are runs of consecutive valid segments
        const uint32_t part_of = min(P - 1u, (uint32_t)(((uint64_t)(incl - vol + (vol >> 1)) * P) / max(total, 1u)));
        uint32_t n_parts = 0;
        for (uint32_t j = 0; j < P; j++) n_parts += ballot(seg_valid && part_of == j) ? 1u : 0u;
        // room in the item queue: ONE fetch-add per splitting query (a CAS loop collapses when thousands of
        // wavefronts reach this point together).  A reservation that runs over the end marks its slots as holes;
        // once the queue is full nobody tries any more.
        uint32_t base = 0xFFFFFFFFu;
        if (lane == 0 && __hip_atomic_load(a.split_ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.item_cap) {
          base = __hip_atomic_fetch_add(a.split_ctl + 1, n_parts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (base >= a.item_cap) base = 0xFFFFFFFFu;
          else if (n_parts > a.item_cap - base) {
            for (uint32_t x = base; x < a.item_cap; x++) a.items[(uint64_t)x * 4] = 0xFFFFFFFFu;
            base = 0xFFFFFFFFu;
          }
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (base != 0xFFFFFFFFu) {
          uint32_t dense = 0;
          for (uint32_t j = 0; j < P; j++) {
            const uint64_t mj = ballot(seg_valid && part_of == j);
            if (!mj) continue;
            if (lane == 0) {
              uint32_t* it = a.items + (uint64_t)(base + dense) * 4;
              it[0] = qi;
              it[1] = (uint32_t)(tb + __builtin_ctzll(mj)) | ((uint32_t)(tb + 63 - __builtin_clzll(mj)) << 16);
              it[2] = qi | ((pushed + dense) << 24);       // the parts' top-k rows are indexed by the query
            }
            dense++;
          }
          pushed += n_parts;
          break;                                             // this tile's parts run in the second launch (on to the next tile)
        }
      }
    }
    PH(1)

    // ---- candidate queue: docs whose bucket reached the flag threshold wait here (with the query position of the list
    //      they were met in) and are verified together at the end of their group — or earlier when the queue fills ----
    uint32_t qn = 0;
    uint32_t epoch = 0, flushed_at = 0;                         // running group number; its value at the last flush
    uint64_t str_m[2] = {0, 0};                                 // query positions whose list the current group streams
    uint32_t lo_doc = 0, hi_doc = 0xFFFFFFFFu;                  // docID range of the current pass (whole range outside pass groups)
    // (d = the ORIGINAL docID: the top-k's order, the LM's word ids and the results are in the caller's numbering)
    auto offer = [&](uint32_t d, int overlap, int w) {
      if (kLM) {                                                            // lmCollector: score = ScoreNext(doc), monotone in the count
        uint32_t c;
        if (lm_small) { const uint64_t m = ballot(lm_w == d); c = m ? readlane(lm_c, __builtin_ctzll(m)) : 0u; }
        else c = d_lm_count(a.lm_values, lm_from, lm_to, d, lane);
        topk_insert(tk, (uint64_t)c, d, lane);
      }
      else if (a.autocomplete) { if (d >= a.ac_first) topk_insert(tk, a.autocomplete == 2 ? by_doc_key(d, overlap, tb + w) : ~(uint64_t)d, d, lane); }   // score = -docID, collector.go:104-106
      else topk_insert(tk, score_bits(d_score(a.metric, overlap, A, tb + w, a.mt)), d, lane);
    };
    // which query-term occurrences hold doc d in segment w, by binary search in their lists: only documents that repeat a
    // term come here (the secondary entries of SURVEY.md §A.3 need the per-list view)
    auto exact_masks = [&](uint32_t x, int w, uint64_t (&fm)[2]) -> int {
      fm[0] = fm[1] = 0;
      int c = 0;
      for (int r = 0; r < a_rounds; r++) {
        const int i = r * 64 + lane;
        bool found = false;
        if (i < A) {
          const uint32_t s0f = rows[i * stride + w], nch = rows[i * stride + w + 1] - s0f;
          const bool g8 = kG8 && (s0f & SG_G8_FLAG);
          const uint32_t s0 = kG8 ? s0f & ~SG_G8_FLAG : s0f, xm = g8 ? SG_X_MASK8 : SG_X_MASK;
          if (nch) {                                             // the last chunk that begins at or before x, then its postings
            const uint32_t* p = ix.postings + (uint64_t)s0 * 4;
            uint32_t lo = 0, hi = nch;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((p[(uint64_t)mid * 4] & xm) <= x) lo = mid + 1; else hi = mid; }
            if (lo && g8) {
              u32x16 pc;
              const uint32_t np = decode_chunk8(post4[s0 + lo - 1u], pc, 0);
#pragma unroll
              for (int e = 0; e < SG_PPC8; e++) found |= (uint32_t)e < np && pc[e] == x;
            } else if (lo) {
              u32x8v pc;
              const uint32_t np = decode_chunk(post4[s0 + lo - 1u], pc, 0);
#pragma unroll
              for (int e = 0; e < SG_PPC; e++) found |= (uint32_t)e < np && pc[e] == x;
            }
          }
        }
        const uint64_t m = ballot(found);
        c += (int)popc64(m);
        if (r == 0) fm[0] = m; else fm[1] = m;
      }
      return c;
    };
    auto emit = [&](uint32_t x, uint32_t d, int overlap, int w) {   // x: the store's number of the document, d: its docID
      const int T = (int)readlane((uint32_t)seg_T, w);
      if (DBG_SKIP(2048u)) { topk_insert(tk, score_bits((double)overlap + (double)T / 1000.0 + (double)w / 1e6), d, lane); return; }
      if (overlap < T) return;
      DBG_COUNT(5, 1)
      offer(d, overlap, w);
      if (ix.n_dup_docs) {                                  // dictionaries whose docs never repeat a term skip this
        if ((ix.dup_bits[d >> 5] >> (d & 31u)) & 1u) {
          uint64_t fm[2];
          __syncthreads();
          exact_masks(x, w, fm);
          // (which secondary entries CPMerge produces depends on the threshold it ran with: the metric's own, not the tightened one)
          const int T0 = a.autocomplete == 1 ? A : d_threshold(a.metric, a.alpha, A, tb + w, a.mt);
          dup_secondary_overlaps(ix, term, rows, dup_scratch, A, stride, w, (uint32_t)(tb + w), T0, d, fm[0], fm[1], lane,
                                 [&](int extra_overlap) { offer(d, extra_overlap, w); });
          __syncthreads();
          build_qhash();                                    // (the scratch lay over it)
        }
      }
    };
    // Verification through the forward index: a candidate's own term list (one contiguous read) is matched against the
    // query-term hash.  Lanes are (candidate, term slot) pairs, 64/gsz candidates side by side, 4 interleaved per lane.
    //   overlap = sum over the doc's distinct terms of their occurrences in the query   (query-side repeats
    //             count per occurrence, doc-side once)
    // A matching doc is flagged in every streamed list from its T'-th on; it is emitted ONCE — at its occurrence in the LAST
    // streamed list that holds it (bucket counts only grow, so that occurrence is always flagged): `later` = some streamed
    // query position behind the one it was met in holds one of its terms.  No dedup set, no bound on candidates per group.
    auto flush_queue = [&]() {
      flushed_at = epoch;
      if (qn == 0) return;
      PH(2)
      __syncthreads();
      for (uint32_t base = 0; base < qn; base += 64) {          // 64 candidates (one per lane for the record loads) at a time
      uint32_t n = min(64u, qn - base);
      uint32_t* qd = cq_doc + base;
      uint32_t* qj = cq_jj + base;
      uint32_t my_doc = (uint32_t)lane < n ? qd[lane] : 0u;
      uint32_t my_jj = (uint32_t)lane < n ? qj[lane] : 0u;
      if (n >= 8u) {
        // A matching document is queued once per list that flagged it (dictionaries of near-duplicates: dozens of times).
        // Of the entries of one document in this batch only the one met in the LAST list can be its last occurrence — the
        // others are "late" by the mere presence of that one, and go without a single load.  128-slot table in the row
        // table's LDS (idle outside the stream): max over (doc, list) per slot; a slot two documents share favours the
        // larger docID, the other is simply verified as usual.
        unsigned long long* dh = (unsigned long long*)rowtab;
        dh[lane] = 0ull; if (64 + lane < L::dedup) dh[64 + lane] = 0ull;
        __syncthreads();
        const uint32_t slot = (((my_doc * 0x9E3779B1u) >> 25) * (uint32_t)L::dedup) >> 7;
        const unsigned long long mine = ((unsigned long long)my_doc << 32) | (unsigned long long)(my_jj & 0xFFu);
        if ((uint32_t)lane < n) __hip_atomic_fetch_max(dh + slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        const unsigned long long top = dh[slot];
        const bool alive = (uint32_t)lane < n && !((uint32_t)(top >> 32) == my_doc && (uint32_t)top > (my_jj & 0xFFu));
        const uint64_t am = ballot(alive);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
        __syncthreads();
        if (alive) { qd[rank] = my_doc; qj[rank] = my_jj; }
        __syncthreads();
        n = popc64(am);
        my_doc = (uint32_t)lane < n ? qd[lane] : 0u;
        my_jj = (uint32_t)lane < n ? qj[lane] : 0u;
      }
      uint2 rec = make_uint2(0u, 0u);
      uint32_t my_orig = 0u;                                    // (one load per candidate, side by side: the serial emit loop has it at hand)
      if ((uint32_t)lane < n) { rec = ix.fwd_rec[my_doc]; my_orig = ix.orig_of[my_doc]; }
      uint32_t nd_max = rec.y >> 16;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) nd_max = max(nd_max, (uint32_t)__shfl_xor((int)nd_max, off, 64));
      const int gsz = nd_max <= 8u ? 8 : nd_max <= 16u ? 16 : nd_max <= 32u ? 32 : 64, ngrp = 64 / gsz;
      const int gi = lane / gsz, ti = lane - gi * gsz;
      const uint64_t gmask = gsz == 64 ? ~0ull : ((1ull << gsz) - 1ull) << (gi * gsz);
      // one candidate per lane group and iteration; the next iteration's term loads are issued before this one's are used
      auto fetch_terms = [&](uint32_t c0, uint32_t s0, uint32_t& x_off, uint32_t& x_nd, uint32_t& x_jj) -> uint32_t {
        const uint32_t cc = c0 + (uint32_t)gi;
        const int src = (int)(min(cc, 63u) << 2);
        // (ds_bpermute returns 0 from a source lane that is switched off: the permutes must not end up under a branch —
        //  a candidate's record lives in lane cc, which may belong to a lane group with nothing to do this round)
        x_off = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rec.x);
        const uint32_t ry = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rec.y);
        x_jj = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)my_jj);
        asm volatile("" : "+v"(x_off), "+v"(x_jj) : "v"(ry));      // pin the three permutes here, unpredicated
        x_nd = cc < n ? ry >> 16 : 0u;
        const uint32_t slot = s0 + (uint32_t)ti;
        return slot < x_nd ? ix.fwd_terms[(uint64_t)x_off * 4u + slot] : kNoTerm;
      };
      const uint32_t rounds = (nd_max + (uint32_t)gsz - 1u) / (uint32_t)gsz;   // 1 unless a document has more than 64 distinct terms
      uint32_t n_off, n_nd, n_jj;
      uint32_t x_next = fetch_terms(0u, 0u, n_off, n_nd, n_jj);
      for (uint32_t c0 = 0; c0 < n; c0 += (uint32_t)ngrp) {
        int ov = 0;
        bool late = false;
        const uint32_t cc = c0 + (uint32_t)gi;
        for (uint32_t rd = 0; rd < rounds; rd++) {
          const uint32_t x = x_next, c_jj = n_jj & 0xFFu, c_ep = (n_jj >> 8) & (SG_EPOCHS - 1u);
          const bool last_rd = rd + 1u == rounds;
          if (!last_rd || c0 + (uint32_t)ngrp < n) x_next = fetch_terms(last_rd ? c0 + (uint32_t)ngrp : c0, last_rd ? 0u : (rd + 1u) * (uint32_t)gsz, n_off, n_nd, n_jj);
          uint32_t mult = 0;
          bool lt = false;
          if (x != kNoTerm) {
            for (uint32_t h = d_qhash<L::qh>(x);; h = d_qnext<L::qh>(h)) {
              const uint32_t kx = qh_key[h];
              if (kx == kNoTerm) break;
              if (kx == x) {
                const uint32_t pp = qh_pos[h];
                mult++;
                lt |= pp > c_jj && ((ep_ring[c_ep * SG_EPOCH_WORDS + (pp >> 5)] >> (pp & 31u)) & 1u);
              }
            }
          }
          const uint64_t mm = ballot(mult != 0u);
          ov += (int)popc64(mm & gmask);
          uint64_t m2 = ballot(mult > 1u);                       // a query that repeats a term: the extra occurrences
          while (m2) {
            const int l = __builtin_ctzll(m2);
            m2 &= m2 - 1;
            const uint32_t extra = readlane(mult, l) - 1u;
            if (l / gsz == gi) ov += (int)extra;
          }
          late |= (ballot(lt) & gmask) != 0ull;
        }
        // verdict into the queue slot (consumed): bit 31 = emit, bits 0..7 overlap
        if (ti == 0 && cc < n) {
          const uint32_t dd = qd[cc], v_ep = (qj[cc] >> 8) & (SG_EPOCHS - 1u);      // (its pass's docID range: a document on a
          const bool keep = !late && dd >= ep_ring[v_ep * SG_EPOCH_WORDS + 4u] && dd < ep_ring[v_ep * SG_EPOCH_WORDS + 5u];   // pass boundary is met twice)
          qj[cc] = keep ? (0x80000000u | (uint32_t)ov) : 0u;
        }
      }
      __syncthreads();
      if (DBG_SKIP(2048u | 4096u)) {
#pragma nounroll
        for (uint32_t c = 0; c < n; c++) {
          const uint32_t v = qj[c];
          DBG_COUNT(7, 1)
          if (DBG_SKIP(4096u)) { topk_insert(tk, score_bits((double)(v & 0xFFu) + (double)c / 1000.0 + (double)n / 1e6 + (v >> 31 ? 0.5 : 0.0)), qd[c], lane); continue; }
          if (!(v >> 31)) continue;
          DBG_COUNT(6, 1)
          const uint32_t card = readlane(rec.y, (int)c) & 0xFFFFu;
          emit(qd[c], readlane(my_orig, (int)c), (int)(v & 0xFFu), (int)card - tb);
        }
      } else {
        // Lane c <-> candidate c: verdict, the threshold of the document's own segment and the top-k key of all of them
        // side by side.  Once the top-k is full, ONE compare per lane against the k-th best leaves the few that can still
        // enter it (re-filtered whenever one of them does): a prefix or a near-duplicate family with dozens of matches
        // per query costs a ballot, not a serial insertion attempt per match (autocomplete: 45 % of a wavefront's time).
        // Documents that repeat a term take the serial path: their secondary entries need the per-list view (emit).
        const uint32_t v = (uint32_t)lane < n ? qj[lane] : 0u;
        const int ov = (int)(v & 0xFFu);
        const int wseg = (int)(rec.y & 0xFFFFu) - tb;
        const int Tc = __shfl(seg_T, wseg & 63, 64);
        bool pass = (v >> 31) != 0u && ov >= Tc;
        DBG_COUNT(7, n)
        DBG_COUNT(6, popc64(ballot((v >> 31) != 0u)))
        DBG_COUNT(5, popc64(ballot(pass)))
        bool dupd = false;
        if (ix.n_dup_docs && pass) dupd = (ix.dup_bits[my_orig >> 5] >> (my_orig & 31u)) & 1u;
        uint64_t key;
        if (kLM) {                                              // lmCollector: ScoreNext is monotone in the continuation count
          uint32_t myc = 0;
          if (lm_small) {
            uint64_t pm = ballot(pass);
            while (pm) {
              const int l = __builtin_ctzll(pm);
              pm &= pm - 1;
              const uint32_t d = readlane(my_orig, l);
              const uint64_t mm = ballot(lm_w == d);
              const uint32_t c = mm ? readlane(lm_c, __builtin_ctzll(mm)) : 0u;
              if (lane == l) myc = c;
            }
          } else if (pass) {
            // A context with more than 64 continuations (the frequent words: thousands): every lane searches ITS candidate's
            // word in the sorted list — log2(n) dependent loads for the 64 candidates of the flush together.  One wave-wide
            // 64-ary search per candidate (d_lm_count), a memory round trip or two each inside a serial loop, made a
            // two-letter prefix behind a frequent context (1 400 matches) a 3 ms wavefront: the launch's tail, with the
            // machine half empty behind it (SQ_WAVE_CYCLES: 46 % of the wave slots occupied, profiles/r04a_pmc_cfg5.txt).
            uint32_t lo = lm_from, hi = lm_to;
            while (lo < hi) {
              const uint32_t mid = lo + ((hi - lo) >> 1);
              if ((uint32_t)(a.lm_values[mid] >> 32) < my_orig) lo = mid + 1u; else hi = mid;
            }
            if (lo < lm_to) { const uint64_t v = a.lm_values[lo]; if ((uint32_t)(v >> 32) == my_orig) myc = (uint32_t)v; }
          }
          key = (uint64_t)myc;
        } else if (a.autocomplete) {                            // score = -docID, collector.go:104-106
          key = a.autocomplete == 2 ? by_doc_key(my_orig, ov, tb + wseg) : ~(uint64_t)my_orig;
          pass = pass && (dupd || my_orig >= a.ac_first);
        } else key = score_bits(d_score(a.metric, ov, A, tb + wseg, a.mt));
        uint64_t m = ballot(pass && !dupd);
        while (m) {
          if (tk.n == k) { m &= ballot(better(key, my_orig, tk.worst_s, tk.worst_id)); if (!m) break; }
          const int l = __builtin_ctzll(m);
          m &= m - 1;
          const uint64_t ks = (uint64_t)readlane((uint32_t)key, l) | ((uint64_t)readlane((uint32_t)(key >> 32), l) << 32);
          topk_insert(tk, ks, readlane(my_orig, l), lane);
        }
        // (a document that repeats a term: its secondary entries have its own docID and at most the primary entry's overlap —
        //  none of them enters a full top-k the primary does not; the per-list view and the pair sort, ~60 us, only for the rest)
        uint64_t dm = ballot(pass && dupd);
        while (dm) {
          if (tk.n == k) { dm &= ballot(better(key, my_orig, tk.worst_s, tk.worst_id)); if (!dm) break; }
          const int l = __builtin_ctzll(dm);
          dm &= dm - 1;
          emit(qd[l], readlane(my_orig, l), (int)(readlane(v, l) & 0xFFu), (int)readlane((uint32_t)wseg, l));
        }
      }
      __syncthreads();
      }
      PH(4)
      qn = 0;
    };

    // buckets a segment needs once its longest lists are skipped down to a.t_floor (estimate: equal
    // lengths), computed by the segment's own lane; groups are then cut by accumulating these.
    uint32_t seg_need = 0, need_p = 0;
    int need_T = 0;
    if (seg_valid) {
      const int k = (seg_T > a.t_floor && !DBG_SKIP(8u)) ? min(seg_T - a.t_floor, A - 1) : 0;
      const uint32_t tot_v = kG8 ? seg_slots : seg_tot;                               // (kG8: posting slots — chunks hold 7 or 13)
      const uint32_t rem = tot_v - (uint32_t)((float)tot_v * (float)k * (1.0f / (float)A));
      need_T = seg_T - k; need_p = kG8 ? rem : rem * SG_PPC;
    }
    {   // (every lane takes part in the permute: the table row sits in lanes 0..32)
      const uint32_t bn = buckets_needed_lane(need_p, need_T, m16_lane);
      if (seg_valid) seg_need = tiny ? 0u : max(1u, bn);       // (tiny: every run of valid segments is one group)
    }

    const uint32_t P_need = wave_scan_incl(seg_need, lane), P_tot = wave_scan_incl(seg_valid ? seg_tot : 0u, lane);
    const uint32_t P_slots = kG8 ? wave_scan_incl(seg_valid ? seg_slots : 0u, lane) : 0u;      // (kG8: chunks hold 7 or 13 postings)
    // Modes without a score (autocomplete, LM ranking) have nothing to tighten against.
    const bool tightening = kTight && !kLM && !a.autocomplete;
    auto tightened_against = [&]() -> uint64_t { return (uint64_t)tile_state[0] | ((uint64_t)tile_state[1] << 32); };
    int wnext = DBG_SKIP(16u) ? Wt : 0;
    while (wnext < Wt) {
      if (tightening && tk.n == k && tk.worst_s != tightened_against()) {   // the k-th best score moved: the segments still
        if (lane == 0) tile_state[2] = (uint32_t)wnext;                      // to come get their thresholds again (the queue
        again = true;                                                        // is empty here: see the end of the pass loop)
        break;
      }
      const uint64_t rest = (vmask >> wnext) << wnext;
      if (!rest) break;
      const int g0 = __builtin_ctzll(rest);
      // merge following valid segments while the counters still resolve the group: the run of valid segments from g0 on,
      // cut where the running sum of their bucket needs passes the counters (always at least g0 itself) — from the
      // tile's prefix sums, a ballot and two bit scans (as a loop it was three v_readlane per segment, each feeding a
      // scalar compare: 25 dependent rounds per query)
      const uint32_t base_need = g0 ? readlane(P_need, g0 - 1) : 0u, base_tot = g0 ? readlane(P_tot, g0 - 1) : 0u;
      const uint64_t inv = ~vmask >> g0;
      const int run_len = inv ? __builtin_ctzll(inv) : 64 - g0;
      const uint64_t unfit = ~(ballot(P_need - base_need <= max_buckets) >> g0);
      const int fit_len = unfit ? __builtin_ctzll(unfit) : 64;
      const int g1 = g0 + max(1, min(run_len, fit_len)) - 1;
      const uint32_t L = readlane(P_tot, g1) - base_tot;      // 16-byte chunks of the group
      const uint32_t Lv = kG8 ? readlane(P_slots, g1) - (g0 ? readlane(P_slots, g0 - 1) : 0u) : L;   // its volume: posting slots (kG8), else chunks
      const int Tmin = wave_min_i32((lane >= g0 && lane <= g1) ? seg_T : 0x7FFFFFFF);
      wnext = g1 + 1;

      // ---- the group's lists: list i = postings of term i over segments tb+g0 .. tb+g1;
      //      lane i keeps its start chunk and chunk count (round r covers terms 64r .. 64r+63) ----
      uint32_t ls_r[2] = {0, 0}, ln_r[2] = {0, 0};
      bool g8_r[2] = {false, false};                           // (kG8) the list's term has 8-bit gaps
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int i = r * 64 + lane;
        if (r < a_rounds && i < A) {
          ls_r[r] = rows[i * stride + g0]; ln_r[r] = rows[i * stride + g1 + 1] - ls_r[r];
          if (kG8) { g8_r[r] = (ls_r[r] & SG_G8_FLAG) != 0u; ls_r[r] &= ~SG_G8_FLAG; }
        }
      }
      const uint64_t g8_m[2] = {kG8 ? ballot(g8_r[0]) : 0ull, kG8 ? ballot(g8_r[1]) : 0ull};
      // a list's volume in the unit of Lv
      const uint32_t lv_r[2] = {kG8 ? ln_r[0] * (g8_r[0] ? (uint32_t)SG_PPC8 : (uint32_t)SG_PPC) : ln_r[0],
                                kG8 ? ln_r[1] * (g8_r[1] ? (uint32_t)SG_PPC8 : (uint32_t)SG_PPC) : ln_r[1]};
      // ---- skip the longest lists (the pigeonhole behind CPMerge, cp_merge.go:22-31): a doc that is
      //      in >= T of the n lists is in >= T-k of ANY n-k of them, so k lists need not be streamed
      //      if postings are flagged at T-k; the exact overlap always comes from the verification over
      //      all lists.  Any k lists are valid; up to T - a.t_floor lists are taken in tiers of
      //      relative length (> 2x, 1.25x, 1x, 0.5x the mean), longest tiers first, lowest lanes first
      //      inside a tier (rank among the tier's lanes by v_mbcnt: no loops). ----
      uint64_t skip_m[2] = {0, 0};
      int Teff = Tmin;
      uint32_t Leff = L, Lveff = Lv;
      if (Tmin > a.t_floor && !DBG_SKIP(8u) && !tiny) {
        const uint32_t n_ne = popc64(ballot(ln_r[0] != 0)) + (a_rounds > 1 ? popc64(ballot(ln_r[1] != 0)) : 0u);
        const uint32_t th[4] = {Lv * 2u, Lv + (Lv >> 2), Lv, Lv >> 1};
        uint64_t pick0 = 0, pick1 = 0;
        int budget = Tmin - a.t_floor;
        auto take = [&](uint32_t x, uint32_t thr, uint64_t& pick) {
          const uint64_t m = ballot(x > thr) & ~pick;
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          const uint64_t keep = ballot(((m >> lane) & 1ull) && (int)rank < budget);
          pick |= keep;
          budget -= (int)popc64(keep);
        };
        const uint32_t x0 = lv_r[0] * n_ne;
#pragma unroll
        for (int f = 0; f < 4; f++) {
          if (budget > 0) take(x0, th[f], pick0);
          if (a_rounds > 1 && budget > 0) take(lv_r[1] * n_ne, th[f], pick1);
        }
        if (pick0 | pick1) {
          const uint32_t sk = (((pick0 >> lane) & 1ull) ? lv_r[0] : 0u) + (((pick1 >> lane) & 1ull) ? lv_r[1] : 0u);
          const uint32_t skipped = readlane(wave_scan_incl(sk, lane), 63);
          const int k_skip = (int)(popc64(pick0) + popc64(pick1));
          if (buckets_needed((Lv - skipped) * (kG8 ? 1u : (uint32_t)SG_PPC), Tmin - k_skip, m16_lane) <= max_buckets) {
            skip_m[0] = pick0; skip_m[1] = pick1; Teff = Tmin - k_skip; Lveff = Lv - skipped;
            if (kG8) {                                          // (the chunks that go with them: what the query streams is counted in chunks)
              const uint32_t skc = (((pick0 >> lane) & 1ull) ? ln_r[0] : 0u) + (((pick1 >> lane) & 1ull) ? ln_r[1] : 0u);
              Leff = L - readlane(wave_scan_incl(skc, lane), 63);
            } else Leff = Lveff;
          }
        }
      }
      // ---- counter geometry: lossy per-bucket counts are upper bounds of per-doc counts ----
      // u32 counters (cheapest per posting) when they resolve the group, else four u8 counters per
      // word; a u8 counter that nears saturation re-runs the group with u32 counters.
      const uint32_t need = tiny ? max_buckets : buckets_needed(Lveff * (kG8 ? 1u : (uint32_t)SG_PPC), Teff, m16_lane);
      bool u8 = need > cnt_words && Teff <= 200;
      bool exact = tiny && u8;                                   // (the u32 re-run of a saturated group is lossy again)
      // the group's documents are the numbers [x_lo, x_hi): what the read-out below may take for candidates
      const uint32_t x_lo = tiny ? ix.seg_base[tb + g0] : 0u, x_hi = tiny ? ix.seg_base[tb + g1 + 1] : 0u;
      uint32_t lg = 8;
      {
        const uint32_t lg_max = u8 ? a.log2_cnt + 2 : a.log2_cnt;
        while (lg < lg_max && (1u << lg) < need) lg++;
      }
      DBG_COUNT(0, 1)
      q_chunks += Leff;                                          // (what this query streams: the accounting of bench.py's roofline.model_bytes)
      bool saturated = false, overflow = false;
      // the group's streamed lists as a mask over query positions (what `later` in flush_queue is asked against); candidates
      // wait in the queue across groups — up to SG_EPOCHS of them — so the masks of the recent groups are kept in a ring
      str_m[0] = ballot(ln_r[0] != 0u) & ~skip_m[0];
      str_m[1] = a_rounds > 1 ? ballot(ln_r[1] != 0u) & ~skip_m[1] : 0ull;
      uint32_t ep_tag = 0, q0 = qn;                             // (set at the top of every pass)
      // slow path of one counted batch whose row descriptors are rows4[0..3]: the flagged postings go to the queue
      // (per posting slot one ballot + a prefix count: the lanes store their own postings)
      auto flagged = [&](const u32x16& vv, const uint32_t (&live)[SG_SUB], const u32x16& was, uint32_t mx, uint32_t row0, uint32_t Tm1,
                         uint32_t last0, uint32_t last1) {   // last0/1: slot of the last posting in the lane's chunk of either row
        // a u8 counter about to wrap would carry into its neighbour and — worse — undercount its own bucket.  Every
        // increment returns the value it found, and a counter passes through every value on its way up, so "some
        // posting found >= 250" is seen before any wrap (reading the counter afterwards is not: identical lists of a
        // query that repeats a term can push one bucket past 255 within a single batch).
        if (u8 && ballot(mx >= 250u)) saturated = true;
        if (overflow) return;                                    // (the saturation watch above goes on)
        // the posting slots flagged in any lane: one compare per slot, its lane mask IS the ballot (no per-lane bit
        // fields, no reduction over the wave); which lanes, and whether they hold a real posting, is settled per slot below
        uint32_t any_fl = 0;
#pragma unroll
        for (int ue = 0; ue < SG_SUB * SG_PPC; ue++) any_fl |= (ballot(live[ue / SG_PPC] && was[ue] >= Tm1) ? 1u : 0u) << ue;
        while (any_fl) {                                         // (vv[ue]: a uniform dynamic index into the register vector)
          const int ue = __builtin_ctz(any_fl);
          any_fl &= any_fl - 1u;
          const int e = ue >= SG_PPC ? ue - SG_PPC : ue;
          // (the slots behind a chunk's last posting are padding: phantom numbers, never candidates)
          const bool mine = (ue >= SG_PPC ? live[1] : live[0]) && was[ue] >= Tm1 && (uint32_t)e <= (ue >= SG_PPC ? last1 : last0);
          const uint64_t m = ballot(mine);
          if (!m) continue;
          const uint32_t cnt_f = popc64(m);
          DBG_COUNT(2, cnt_f)
          // a full queue is not emptied here (verification in the middle of the stream loop costs the loop its registers):
          // the group's flagged postings are collected again after the stream, from the final counters
          if (qn + cnt_f > cq_cap || DBG_SKIP(1024u)) { overflow = true; break; }
          DBG_COUNT(3, cnt_f)
          const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (mine) { cq_doc[pos] = vv[ue]; cq_jj[pos] = ((uint32_t)rowlist[row0 + (uint32_t)(ue >= SG_PPC ? 1 : 0)] & (kG8 ? 0x7Fu : 0xFFu)) | ep_tag; }
          qn += cnt_f;
        }
      };
      // the same for ONE counted row of a dense term (count_row8: SG_PPC8 slots, all of row `row`)
      auto flagged8 = [&](const u32x16& vv, uint32_t live, const u32x16& was, uint32_t mx, uint32_t row, uint32_t Tm1, uint32_t last) {
        if (u8 && ballot(mx >= 250u)) saturated = true;
        if (overflow) return;
        uint32_t any_fl = 0;
#pragma unroll
        for (int ue = 0; ue < SG_PPC8; ue++) any_fl |= (ballot(live && was[ue] >= Tm1) ? 1u : 0u) << ue;
        while (any_fl) {
          const int ue = __builtin_ctz(any_fl);
          any_fl &= any_fl - 1u;
          const bool mine = live && was[ue] >= Tm1 && (uint32_t)ue <= last;
          const uint64_t m = ballot(mine);
          if (!m) continue;
          const uint32_t cnt_f = popc64(m);
          DBG_COUNT(2, cnt_f)
          if (qn + cnt_f > cq_cap) { overflow = true; break; }
          DBG_COUNT(3, cnt_f)
          const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (mine) { cq_doc[pos] = vv[ue]; cq_jj[pos] = ((uint32_t)rowlist[row] & 0x7Fu) | ep_tag; }
          qn += cnt_f;
        }
      };

      PH(2)
      // ---- docID-range passes: a single segment whose postings outnumber what the counters resolve
      //      (long lists: q = 2, skewed terms) is streamed in R passes over disjoint docID ranges.  Lists
      //      are ascending, so a range is one contiguous run of chunks per list (rounded outwards to whole
      //      chunks: every doc of the range is seen completely in its own pass). ----
      uint32_t n_pass = 1;
      if (g0 == g1 && need > max_buckets) n_pass = min(1024u, (need + max_buckets - 1u) / max_buckets);   // (as many as needed: a power of two wasted up to half)
      const uint32_t full_ls[2] = {ls_r[0], ls_r[1]}, full_ln[2] = {ln_r[0], ln_r[1]};
      uint32_t prev_lo[2] = {0, 0};                             // per list: a chunk that begins below the range (or the list's first)
      uint32_t g_cur[2] = {(full_ls[0] + 15u) >> 4, (full_ls[1] + 15u) >> 4};   // per list: cursor into cut_sample
      // (a single segment: its documents are the numbers seg_base[B] .. seg_base[B + 1], and the passes divide THAT range)
      const uint32_t pass_x0 = n_pass > 1 ? ix.seg_base[tb + g0] : 0u, pass_span = n_pass > 1 ? ix.seg_base[tb + g0 + 1] - pass_x0 : 0u;
      for (uint32_t pass = 0; pass < n_pass; pass++) {
      DBG_COUNT(4, 1)
      if (n_pass > 1) {
        lo_doc = pass_x0 + (uint32_t)(((uint64_t)pass_span * pass) / n_pass);
        hi_doc = pass + 1 == n_pass ? 0xFFFFFFFFu : pass_x0 + (uint32_t)(((uint64_t)pass_span * (pass + 1)) / n_pass);
        // Cut every list at hi_doc, in whole chunks: [prev_lo, hi) is streamed, where chunk hi begins at or above hi_doc
        // (everything in it belongs to later passes) and the next pass starts at lo, a chunk that begins below hi_doc — all
        // chunks before it lie wholly below (a chunk ends where the next begins).  The cut need not be tight: a few
        // postings counted in two passes only loosen the filter, and a candidate is emitted in the pass that owns its number.
        // Binary-searching the lists themselves costs ~17 random 128 B lines per list and pass — as much HBM traffic as
        // the postings the pass streams.  Instead a cursor per list advances over cut_sample (contiguous, one u32 per 16
        // chunks, under 1 % of the store) to the 16-chunk block that straddles hi_doc, and one round of probes inside it.
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const uint32_t n = full_ln[r];                        // chunks
          uint32_t lo = prev_lo[r], hi = n;
          if (pass + 1 != n_pass && r < a_rounds) {
            const uint32_t s0 = full_ls[r], g_first = (s0 + 15u) >> 4, g_end = (s0 + full_ln[r] + 15u) >> 4;
            // bracket [glo, ghi]: every sample before glo is < hi_doc, sample ghi is >= hi_doc (or ghi == g_end).
            // 8 independent loads per round: around the position the remaining samples predict when the cursor
            // has far to go (the numbers of a list are close to uniform over the segment's range), then 9-ary, then 8
            // consecutive samples.
            uint32_t glo = g_cur[r], ghi = g_end;
            const uint32_t step = (g_end - glo) / (n_pass - pass);
            bool predict = true;
            while (ballot(glo < ghi)) {
              uint32_t pos[8], sv[8];
              const uint32_t span = ghi - glo;
              // (where the boundary falls in a list of n postings has a standard deviation of sqrt(p(1-p)n) postings — about a
              //  sample for n < 10^4 — so eight CONSECUTIVE samples around the prediction bracket it in one round trip
              //  almost always; a miss falls through to the 9-ary rounds)
              const uint32_t base = glo + step > 3u ? glo + step - 3u : 0u;
#pragma unroll
              for (int i = 0; i < 8; i++) {
                uint32_t x = span <= 8u || (predict && step <= 6u) ? glo + (uint32_t)i
                             : predict ? base + (uint32_t)i : glo + (span * (uint32_t)(i + 1)) / 9u;
                pos[i] = min(max(x, glo), g_end);
                sv[i] = ix.cut_sample[pos[i]];
              }
#pragma unroll
              for (int i = 0; i < 8; i++) {
                if (pos[i] < g_end && sv[i] < hi_doc) glo = max(glo, pos[i] + 1u); else ghi = min(ghi, pos[i]);
              }
              glo = min(glo, ghi);
              predict = false;
            }
            const uint32_t g = ghi;
            g_cur[r] = g;
            lo = max(lo, g > g_first ? ((g - 1u) << 4) - s0 : 0u);      // sample g - 1 (chunk 16 (g - 1)) begins below hi_doc
            hi = min(n, g < g_end ? (g << 4) - s0 : n);                 // sample g begins at or above it
            lo = min(lo, hi);
            // one round of probes on the first postings of the chunks in between (two 128 B lines this pass and the next
            // stream anyway): cuts to within two chunks — with many passes a 16-chunk slack would re-read short lists whole
            const uint32_t* p = ix.postings + (uint64_t)s0 * 4;
            while (ballot(hi - lo > 2u)) {
              uint32_t pos[8], val[8];
#pragma unroll
              for (int i = 0; i < 8; i++) {
                pos[i] = min(lo + ((hi - lo) * (uint32_t)(i + 1)) / 9u, n ? n - 1u : 0u);
                val[i] = n ? p[(uint64_t)pos[i] * 4] & (g8_r[r] ? SG_X_MASK8 : SG_X_MASK) : 0xFFFFFFFFu;
              }
              uint32_t nlo = lo, nhi = hi;
#pragma unroll
              for (int i = 0; i < 8; i++) {
                if (val[i] < hi_doc) nlo = max(nlo, pos[i]); else nhi = min(nhi, pos[i]);
              }
              hi = nhi; lo = min(nlo, nhi);
            }
          }
          const uint32_t c_start = prev_lo[r], c_end = hi;
          ls_r[r] = full_ls[r] + c_start;
          ln_r[r] = c_end > c_start ? c_end - c_start : 0u;
          prev_lo[r] = lo;
        }
        lg = u8 ? a.log2_cnt + 2 : a.log2_cnt;                 // all buckets for every pass
      }
      // ---- streaming passes: normally one; a saturated u8 pass is repeated with u32 counters ----
      saturated = false; overflow = false;
      // this pass's slot in the epoch ring (the flush at the end of the previous pass made sure it is free)
      epoch++;
      ep_tag = (epoch & (SG_EPOCHS - 1u)) << 8;
      q0 = qn;                                                  // the queue below q0 belongs to earlier passes / groups
      if (lane == 0) {
        uint32_t* er = ep_ring + (epoch & (SG_EPOCHS - 1u)) * SG_EPOCH_WORDS;
        er[0] = (uint32_t)str_m[0]; er[1] = (uint32_t)(str_m[0] >> 32); er[2] = (uint32_t)str_m[1]; er[3] = (uint32_t)(str_m[1] >> 32);
        er[4] = lo_doc; er[5] = hi_doc;
      }
      for (int attempt = 0; attempt < 2; attempt++) {
        const uint32_t words = u8 ? (1u << lg) >> 2 : (1u << lg);     // (cleared below, behind the first row loads)
        // (exact: no posting is flagged on the way — only the watch for a u8 counter about to wrap stays on)
        const uint32_t amask = u8 ? (((1u << lg) - 1u) & ~3u) : (((1u << lg) - 1u) << 2), Tm1 = exact ? 249u : (uint32_t)Teff - 1u;
        __syncthreads();
        PH(3)
        // Rows of 64 chunks, list after list.  Lane i knows how many rows its list has (nr) and the running
        // row number where they start (pr, wave scan) and writes their descriptors {first chunk, live lanes,
        // list} to the LDS row table; the stream loop then needs one uniform LDS read per row — no ballots,
        // no VALU->SALU->readlane dependency chains.  Tables larger than L::rowtab_cap are done in windows.
        uint32_t nr[2], pr[2];
        uint32_t n_rows = 0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const bool skipped = (skip_m[r] >> lane) & 1ull;
#ifdef SG_EXP_DROP_TAILS   // (timing experiment only — wrong results: a list's last row is dropped when it is less than half full: what a layout
                          //  without lanes behind the ends of the lists could gain at best, profiles/r04y_drop_tails.txt)
          nr[r] = (r < a_rounds && !skipped) ? (ln_r[r] + 31u) >> 6 : 0u;
#else
          nr[r] = (r < a_rounds && !skipped) ? (ln_r[r] + 63u) >> 6 : 0u;
#endif
          const uint32_t incl = wave_scan_incl(nr[r], lane);
          pr[r] = n_rows + incl - nr[r];
          n_rows += readlane(incl, 63);
        }
        uint2* rowtab2 = (uint2*)rowtab;
        for (uint32_t w0 = 0; w0 < n_rows && !DBG_SKIP(512u); w0 += L::rowtab_cap) {
          const uint32_t wn = min(L::rowtab_cap, n_rows - w0);
          __syncthreads();
#pragma unroll
          for (int r = 0; r < 2; r++) {
            if (r < a_rounds) {
              for (uint32_t x = 0; x < nr[r]; x++) {
                const uint32_t R = pr[r] + x;
                if (R >= w0 && R < w0 + wn) {
                  rowtab2[R - w0] = make_uint2(ls_r[r] + x * 64u, min(64u, ln_r[r] - x * 64u));
                  rowlist[R - w0] = (uint8_t)((r * 64 + lane) | (g8_r[r] ? 0x80 : 0));   // (kG8: bit 7 = rows of 13 postings per chunk)
                }
              }
            }
          }
          if (lane < 2 * SG_UNROLL) { rowtab2[wn + lane] = make_uint2(0u, 0u); if (kG8) rowlist[wn + lane] = 0; }   // dead rows behind the last batch
          __syncthreads();
          // The row loads are issued and awaited by hand (inline asm): the compiler's own wait-count insertion kept the
          // "wait for this batch only, the next one stays in flight" schedule for a while and then — after an unrelated
          // change elsewhere in the kernel — fell back to s_waitcnt vmcnt(0) right behind the prefetch (−5 % on the
          // headline).  vmcnt counts in issue order, so vmcnt(4) with the 4 loads of the next batch behind them means
          // "these four are here" whatever else is in flight.
          typedef u32x4v u32x4;
          u32x4 v[SG_UNROLL], vn[SG_UNROLL];
          uint32_t live[SG_UNROLL], liven[SG_UNROLL];
          uint32_t next_row = 0;
          auto fetch = [&](u32x4 (&vv)[SG_UNROLL], uint32_t (&lv)[SG_UNROLL]) {
#pragma unroll
            for (int u = 0; u < SG_UNROLL; u++) {
              const uint2 t = rowtab2[next_row + (uint32_t)u];   // uniform address: one broadcast LDS read
              lv[u] = (uint32_t)lane < t.y ? 1u : 0u;
              const uint4* src = post4 + (t.x + min((uint32_t)lane, t.y ? t.y - 1 : 0u));
              asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(vv[u]) : "v"(src) : "memory");
            }
            next_row += SG_UNROLL;
          };
          // ping-pong between two register sets: the next batch's loads are in flight while one is counted
          auto process = [&](u32x4 (&qv)[SG_UNROLL], const uint32_t (&pl)[SG_UNROLL], uint32_t row0) {
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3]) :: "memory");
#pragma unroll
            for (int h = 0; h < SG_UNROLL / SG_SUB; h++) {     // SG_SUB rows = 14 postings per LDS round trip
              if (h && row0 + (uint32_t)(SG_SUB * h) >= wn) break;   // the window's last batch: both rows of this half are dead rows
              const uint4 pv[SG_SUB] = {make_uint4(qv[2 * h].x, qv[2 * h].y, qv[2 * h].z, qv[2 * h].w),
                                        make_uint4(qv[2 * h + 1].x, qv[2 * h + 1].y, qv[2 * h + 1].z, qv[2 * h + 1].w)};
              const uint32_t sl[SG_SUB] = {pl[2 * h], pl[2 * h + 1]};
              u32x16 pp, was;
              uint32_t mx = 0;
              uint64_t any = 0;
              if (kG8) {   // rows of dense terms (13 postings per chunk) are counted one at a time; so is a 7-per-chunk row next to one
                const uint32_t f2 = __builtin_amdgcn_readfirstlane((uint32_t)*(const uint16_t*)(rowlist + row0 + (uint32_t)(SG_SUB * h)));
                if (f2 & 0x8080u) {
#pragma unroll
                  for (int u = 0; u < SG_SUB; u++) {
                    const uint32_t row = row0 + (uint32_t)(SG_SUB * h + u);
                    if ((f2 >> (8 * u + 7)) & 1u) {
                      any = u8 ? count_row8<true>(pv[u], sl[u], amask, cbase, dummy_lane, Tm1, pp, was, mx) : count_row8<false>(pv[u], sl[u], amask, cbase, dummy_lane, Tm1, pp, was, mx);
                      if (any) flagged8(pp, sl[u], was, mx, row, Tm1, pv[u].x >> 28);
                    } else {
                      const uint4 px[SG_SUB] = {pv[u], pv[u]};
                      const uint32_t lx[SG_SUB] = {sl[u], 0u};
                      any = u8 ? count_rows<true>(px, lx, amask, cbase, dummy_lane, Tm1, pp, was, mx) : count_rows<false>(px, lx, amask, cbase, dummy_lane, Tm1, pp, was, mx);
                      if (any) flagged(pp, lx, was, mx, row, Tm1, pv[u].x >> 29, 0u);
                    }
                  }
                  DBG_COUNT(1, 1)
                  continue;
                }
              }
              if (DBG_SKIP(4u)) asm volatile("" :: "v"(pv[0].x), "v"(pv[1].x));
              else any = u8 ? count_rows<true>(pv, sl, amask, cbase, dummy_lane, Tm1, pp, was, mx) : count_rows<false>(pv, sl, amask, cbase, dummy_lane, Tm1, pp, was, mx);
              DBG_COUNT(1, 1)
              if (any) { PH(5) flagged(pp, sl, was, mx, row0 + (uint32_t)(SG_SUB * h), Tm1, pv[0].x >> 29, pv[1].x >> 29); PH(6) }
            }
          };
          // fetches are unconditional (rows past the last are dead rows: they load chunk 0): every process() has the four
          // loads of the following batch behind its own
          const uint32_t n_batches = (wn + SG_UNROLL - 1) / SG_UNROLL;
          fetch(v, live);
          if (w0 == 0u) {       // the counters are cleared while the group's first rows are on their way (one wavefront: its
            // LDS instructions execute in order, the stores are ahead of the first atomics; a barrier here would carry a
            // fence that waits for the row loads)
            for (uint32_t w = lane * 4; w < words; w += 256) *(uint4*)(cnt + w) = make_uint4(0, 0, 0, 0);
            asm volatile("" ::: "memory");
          }
          for (uint32_t bi = 0; bi < n_batches; bi += 2) {
            fetch(vn, liven);
            process(v, live, bi * SG_UNROLL);
            if (bi + 1 >= n_batches) break;
            fetch(v, live);
            process(vn, liven, (bi + 1) * SG_UNROLL);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dead-row prefetch: nothing may be in flight into v / vn
        }
        PH(5)
        if (!(u8 && saturated)) break;
        // A u8 counter came close to wrapping: the group is counted again with u32 counters (nothing of it has reached
        // the top-k yet: its candidates are only queued)
        qn = q0; overflow = false;
        u8 = false; saturated = false; exact = false;
        lg = min(lg, a.log2_cnt);
        __syncthreads();
      }
      if (overflow || exact) {
        // exact: the read-out — lane <-> counter word, its four bytes are four documents; queued as met "behind every list"
        // (position 255: never late).
        // More flagged postings than the queue holds (dictionaries of near-duplicates: dozens of matches per query, each
        // flagged in every list from its T'-th on).  The lists are walked again against the FINAL counters — a superset of
        // what the stream flagged, counts only grow — and the queue is emptied whenever it fills; the verdict rule (emit at
        // the last streamed list holding the doc) keeps every document single.
        qn = q0;                                                 // this pass's entries go (what earlier ones queued stays)
        int lists_before = 0;
        const int n_outer = exact ? 1 : A;
        for (int i = 0; i < n_outer; i++) {
          uint32_t s = x_lo >> 2, n = ((x_hi + 3u) >> 2) - (x_lo >> 2);      // exact: the counter words of the group's documents
          bool l8 = false;                                                   // (kG8) a list of 13 postings per chunk
          if (!exact) {
            const int r = (i >> 6) & 1, li = i & 63;
            if (!(((r ? str_m[1] : str_m[0]) >> li) & 1ull)) continue;
            // a document in >= Teff streamed lists has its LAST occurrence in the Teff-th streamed list or later
            if (lists_before++ < Teff - 1) continue;
            s = readlane(r ? ls_r[1] : ls_r[0], li); n = readlane(r ? ln_r[1] : ln_r[0], li);
            l8 = kG8 && (((r ? g8_m[1] : g8_m[0]) >> li) & 1ull);
          }
          for (uint32_t c0 = 0; c0 < n; c0 += 64) {
            const uint32_t c = c0 + lane;
            typename std::conditional<kG8, u32x16, u32x8v>::type pc;
            uint32_t np = 4u;
            if (exact) {
              const uint32_t word = c < n ? cnt[s + c] : 0u;
              pc[0] = word & 0xFFu; pc[1] = (word >> 8) & 0xFFu; pc[2] = (word >> 16) & 0xFFu; pc[3] = word >> 24;
              pc[4] = pc[5] = pc[6] = 0u;
            } else {
              uint4 v = make_uint4(0, 0, 0, 0);
              if (c < n) v = post4[s + c];
              if constexpr (kG8) np = l8 ? decode_chunk8(v, pc, 0) : decode_chunk(v, pc, 0);
              else np = decode_chunk(v, pc, 0);
            }
#pragma nounroll
            for (int e = 0; e < (l8 ? SG_PPC8 : SG_PPC); e++) {
              const uint32_t d = exact ? (s + c) * 4u + (uint32_t)e : pc[e];
              bool flag = false;
              if (c < n && (uint32_t)e < np) {
                if (exact) flag = pc[e] >= (uint32_t)Teff && d >= x_lo && d < x_hi;
                else {
                  const uint32_t bk = d & ((1u << lg) - 1u);
                  const uint32_t now = u8 ? ((cnt[bk >> 2] >> ((bk & 3u) << 3)) & 0xFFu) : cnt[(d >> 2) & ((1u << lg) - 1u)];
                  flag = now >= (uint32_t)Teff;
                }
              }
              const uint64_t m_all = ballot(flag);
              if (!m_all) continue;
              // (the queue may hold fewer than 64 entries — never fewer than 32: the two halves of the wave go in turn)
#pragma nounroll
              for (int half = 0; half < 2; half++) {
                const uint64_t m = half ? (m_all >> 32) << 32 : m_all & 0xFFFFFFFFull;
                if (!m) continue;
                const uint32_t cnt_f = popc64(m);
                DBG_COUNT(3, cnt_f)
                if (qn + cnt_f > cq_cap) flush_queue();
                const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (flag && ((m >> lane) & 1ull)) { cq_doc[pos] = d; cq_jj[pos] = (exact ? 0xFFu : (uint32_t)i) | ep_tag; }
                qn += cnt_f;
              }
            }
          }
        }
      }
      // candidates wait across passes and groups (one verification of many instead of many of few); they are verified
      // when the next pass would reuse a live ring slot, when the queue is half full, and before the tile's segment table goes
      // A group after which the thresholds are due to be tightened (the k-th best score moved since the last time) ends
      // with an empty queue: the tightening at the top of the next group may leave nothing more to stream.
      const bool last_group = wnext >= Wt || (vmask >> wnext) == 0ull || (tightening && tk.n == k && tk.worst_s != tightened_against());
      if (epoch + 1u - flushed_at >= SG_EPOCHS || qn > cq_cap / 2 || (last_group && pass + 1u == n_pass)) flush_queue();
      }  // docID-range passes
      lo_doc = 0; hi_doc = 0xFFFFFFFFu;
    }
    if (!kTight || !again) break;
    }  // segment statistics, again after a tightening
    }  // the two halves of the tile
  }

  // ---- a part of a split query hands its top-k over; the part that finishes last merges them all (top-k of a
  //      union = top-k of the union of the parts' top-k: the order (score desc, docID asc) is total) ----
  if (kParts || pushed) {
    // (first launch: this wavefront's own tiles are one more part, and nobody has counted yet)
    if (!kParts) { my_slot = qi; my_part = pushed; }
    __syncthreads();
    const uint64_t pbase = (uint64_t)my_slot * SG_MAX_PARTS;
    // (agent-scope stores here and loads in the merge below, on top of the release / acquire around the counter: the rows of a slot's
    //  parts share cache lines — sixteen part_n words in one, short rows of several parts in another — that wavefronts on different
    //  XCDs write side by side, and two fuzz reports this round look like one stale line of part_id read by the
    //  merging part; these accesses go past the XCD's L2 whatever lines it holds)
    for (uint32_t i = lane; i < tk.n; i += 64) {
      __hip_atomic_store(a.part_s + (pbase + my_part) * k + i, tk.s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.part_id + (pbase + my_part) * k + i, tk.id[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) __hip_atomic_store(a.part_n + pbase + my_part, tk.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (what this wavefront streamed of a sampled query: a split query's chunks are the sum over its parts, and only the part
    //  that merges reaches the accounting at the end)
    const bool sampled = !kLM && !a.autocomplete && a.fill_stat && (qi & a.fill_mask) == 0u;
    if (!kParts) {                                         // the queued parts all run later: count this one as finished
      if (lane == 0) { a.slot_ctl[2 * qi] = 1u; a.slot_ctl[2 * qi + 1] = pushed + 1u; }
      if (sampled && lane == 0) atomicAdd((unsigned long long*)(a.fill_stat + 4), (unsigned long long)q_chunks);
      break;
    }
    // release: this part's rows are visible device-wide (across the XCDs' L2s) before it is counted; acquire: the
    // part that counts last sees every other part's rows.  One fence pair per >= 1 MiB part.
    uint32_t fin = 0;
    if (lane == 0) fin = __hip_atomic_fetch_add(a.slot_ctl + 2 * my_slot, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    fin = __builtin_amdgcn_readfirstlane(fin);
    const uint32_t n_parts = __builtin_amdgcn_readfirstlane(a.slot_ctl[2 * my_slot + 1]);   // written by the first launch
    if (fin != n_parts) { if (sampled && lane == 0) atomicAdd((unsigned long long*)(a.fill_stat + 4), (unsigned long long)q_chunks); break; }   // somebody else finishes later and merges
    for (uint32_t pp = 0; pp < n_parts; pp++) {
      if (pp == my_part) continue;
      const uint32_t np = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.part_n + pbase + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      for (uint32_t i0 = 0; i0 < np; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint64_t es = i < np ? __hip_atomic_load(a.part_s + (pbase + pp) * k + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        const uint32_t ed = i < np ? __hip_atomic_load(a.part_id + (pbase + pp) * k + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t cntl = min(64u, np - i0);
        for (uint32_t l = 0; l < cntl; l++) {
          const uint64_t sl = (uint64_t)readlane((uint32_t)es, (int)l) | ((uint64_t)readlane((uint32_t)(es >> 32), (int)l) << 32);
          topk_insert(tk, sl, readlane(ed, (int)l), lane);
        }
      }
    }
  }

  // ---- GetCandidates (topk.go:127-147): best first (rank sort, out of place) ----
  __syncthreads();
  const uint32_t n = tk.n;
  for (uint32_t i = lane; i < n; i += 64) {
    const uint64_t s = tk.s[i];
    const uint32_t d = tk.id[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) {
      const uint64_t sj = tk.s[j];
      const uint32_t dj = tk.id[j];
      rank += (better(sj, dj, s, d) || (sj == s && dj == d && j < i)) ? 1u : 0u;
    }
    out_ids[rank] = d;
    if (a.autocomplete == 2) {
      if (out_scores) out_scores[rank] = d_score(a.metric, (int)(s & 0xFFFFu), A, (int)((s >> 16) & 0xFFFFu), a.mt);
      if (a.out_aux) a.out_aux[(uint64_t)qi * k + rank] = (uint32_t)s;
    } else if (out_scores) out_scores[rank] = bits_score(s);
  }
  if (lane == 0) a.out_counts[qi] = n;
  if (!kLM && !a.autocomplete && a.fill_stat && lane == 0 && (qi & a.fill_mask) == 0u) {
    atomicAdd(a.fill_stat + 1, 1u);
    if (n == k) atomicAdd(a.fill_stat, 1u);
    if (n) atomicAdd(a.fill_stat + 2, n);
    atomicAdd((unsigned long long*)(a.fill_stat + 4), (unsigned long long)q_chunks);
  }
  if (DBG_SKIP(8192u) && lane == 0 && k >= 6) { out_ids[k - 1] = (uint32_t)A; out_ids[k - 3] = qi; }
  PH(7)
  } while (0);
  if (kLoop) {
    __syncthreads();
    const uint32_t pos = __builtin_amdgcn_readfirstlane(tile_state[3]) + gridDim.x;
    if (pos >= __builtin_amdgcn_readfirstlane(*a.q_sel_n)) break;
    __syncthreads();
    if (lane == 0) tile_state[3] = pos;
    qi = __builtin_amdgcn_readfirstlane(a.q_sel[pos]);
    __syncthreads();
    continue;
  }
  if (!kParts) break;
  }
  { const uint32_t qi = blockIdx.x; (void)qi; PH_FLUSH }
}


// ------------------------------------------------------------------------------------------
// The tokeniser as a launch of its own (batches large enough to fill the machine): one wavefront per query runs
// d_tokenize — wrap, lower, trim, q-grams, appendUnique, normalise, term table (pkg/suggest/tokenizer.go:9-34,
// pkg/analysis/ngram_tokenizer.go:17-55) — and leaves A and the term ids in HBM.  Inside the search kernel the same
// steps are four dependent memory round trips at the head of every query with 12 wavefronts per CU to hide them (9 % of
// a wavefront's time on the headline, more on small dictionaries); here the wavefronts need 2 KB of LDS and few
// registers, so a CU holds 32 of them and the round trips overlap.  Same function, same results.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sg_terms_kernel(const BatchArgs a) {
  __shared__ uint32_t s_runes[SG_MAX_RUNES];
  __shared__ uint64_t s_keys[SG_MAX_A];
  __shared__ uint32_t s_term[SG_MAX_A];
  const int lane = threadIdx.x;
  uint32_t qi = blockIdx.x;
  if (a.q_sel) {                                             // a launch over a subset of the batch: the same subset
    const uint32_t sel_n = *a.q_sel_n, sel_q = a.q_sel[qi];
    if (qi >= __builtin_amdgcn_readfirstlane(sel_n)) return;
    qi = __builtin_amdgcn_readfirstlane(sel_q);
  }
  const uint64_t qb = a.q_offs[qi], qe = a.q_len ? qb + a.q_len[qi] : a.q_offs[qi + 1];
  const int A = d_tokenize(a, a.q_blob + qb, (uint32_t)(qe - qb), s_runes, s_keys, s_term, lane);
  if (lane == 0) a.pre_A[qi] = A;
  for (int i = lane; i < A; i += 64) a.pre_terms[(uint64_t)qi * SG_MAX_A + i] = s_term[i];
}

// ------------------------------------------------------------------------------------------
// Queries beyond the wavefront kernel's tables (more than SG_MAX_A n-grams / SG_MAX_RUNES runes).  The reference has no
// such limit (pkg/suggest/suggester.go:46-59); the wavefront kernel marks these queries SG_COUNT_TOO_LONG and this
// kernel, launched behind it, answers them — exactly, on the device, with its working memory in HBM instead of LDS:
//   tokenise (any length up to SG_LONG_MAX_TERMS n-grams; appendUnique through a hash of first positions)
//   -> per admissible segment: zero one counter per document of the segment, add 1 per posting of every query-term
//      occurrence (ScanCount, pkg/merger/scan_count.go:14-88 — the same result set as CPMerge), collect counts >= T
//   -> score, top-k, the secondary entries of documents that repeat a term — as the wavefront kernel does.
// A few workgroups of one wavefront; each takes a slot of the replica's long-query scratch when it meets a marked query
// (launches on several streams share the slots: a lock word per slot).  Speed is beside the point here — such queries are
// rare — but nothing is approximated and nothing goes to the host.
// ------------------------------------------------------------------------------------------
#define SG_LONG_SLOTS 4
#define SG_LONG_MAX_TERMS 65536u
#define SG_LONG_MAX_RUNES (SG_LONG_MAX_TERMS + 2u * SG_WRAP_MAX + 8u)
#define SG_LONG_HASH (2u * SG_LONG_MAX_TERMS)
struct LongSlot {   // offsets (bytes) into a slot
  static constexpr uint64_t o_runes = 0;
  static constexpr uint64_t o_keys = o_runes + 4ull * SG_LONG_MAX_RUNES + 32;
  static constexpr uint64_t o_term = o_keys + 8ull * SG_LONG_MAX_TERMS;
  static constexpr uint64_t o_hash = o_term + 4ull * SG_LONG_MAX_TERMS;
  static constexpr uint64_t o_sk = o_hash + 4ull * SG_LONG_HASH;
  static constexpr uint64_t o_sv = o_sk + 4ull * SG_LONG_MAX_TERMS;
  static constexpr uint64_t o_stk = o_sv + 4ull * SG_LONG_MAX_TERMS;
  static constexpr uint64_t o_extra = o_stk + 4ull * 512;
  static constexpr uint64_t o_tks = o_extra + 4ull * 64;
  static constexpr uint64_t o_tkid = o_tks + 8ull * SG_K_LDS;
  static constexpr uint64_t o_cnt = o_tkid + 4ull * SG_K_LDS + 64;     // [long_max_seg] one counter per document of a segment
  static uint64_t bytes(uint32_t max_seg) { return (o_cnt + 4ull * ((uint64_t)max_seg + 64) + 255) & ~255ull; }
};
template <class T> __device__ __forceinline__ T ld_l2(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // past the L1: the word may have been changed by an atomic

__device__ __forceinline__ bool long_same_gram(const uint32_t* runes, uint32_t g1, uint32_t g2, uint32_t q_n) {
  bool eq = true;
  for (uint32_t t = 0; t < q_n; t++) eq &= runes[g1 + t] == runes[g2 + t];
  return eq;
}
__device__ __forceinline__ uint32_t long_gram_hash(const uint32_t* runes, uint32_t g, uint32_t q_n) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (uint32_t t = 0; t < q_n; t++) h = d_mix64(h ^ (uint64_t)runes[g + t]);
  return (uint32_t)h;
}

// The tokeniser of NewSuggestTokenizer / NewAutocompleteTokenizer (pkg/suggest/tokenizer.go:9-34) for texts of any length
// up to SG_LONG_MAX_TERMS bytes, with its tables in an HBM slot: wrap -> lower -> trim -> q-grams (first-occurrence
// dedup) -> normalise.  Returns the number of n-grams (0xFFFFFFFF: more than SG_LONG_MAX_TERMS of them before the dedup —
// the tables would overflow); their packed keys are in the slot's `keys`.  Shared by the
// long-query kernel and the device index builder's long documents.
__device__ uint32_t long_tokenize(const DeviceIndex& ix, bool autocomplete, const uint8_t* q, uint32_t qlen, uint8_t* slot, int lane) {
  uint32_t* runes = (uint32_t*)(slot + LongSlot::o_runes);
  uint64_t* keys = (uint64_t*)(slot + LongSlot::o_keys);
  uint32_t* htab = (uint32_t*)(slot + LongSlot::o_hash);
  // ---- wrap -> lower -> runes (the same uniform sequential decode as the wavefront kernel's rare path; ASCII in parallel) ----
  const uint32_t n_w0 = ix.n_wrap0, n_w1 = autocomplete ? 0u : ix.n_wrap1;
  bool na = false;
  for (uint32_t i = lane; i < qlen; i += 64) na |= q[i] >= 0x80;
  uint32_t R = 0, byte_len = 0;
  if (!ballot(na)) {
    R = n_w0 + qlen + n_w1;
    for (uint32_t i = lane; i < qlen; i += 64) { const uint32_t b = q[i]; runes[n_w0 + i] = (b - 'A' < 26u) ? b + 32u : b; }
    if ((uint32_t)lane < n_w0) runes[lane] = d_lower(ix, ix.wrap0[lane]);
    if ((uint32_t)lane < n_w1) runes[n_w0 + qlen + lane] = d_lower(ix, ix.wrap1[lane]);
    byte_len = qlen;
    for (uint32_t i = 0; i < n_w0; i++) byte_len += d_width(d_lower(ix, ix.wrap0[i]));
    for (uint32_t i = 0; i < n_w1; i++) byte_len += d_width(d_lower(ix, ix.wrap1[i]));
  } else {
    for (uint32_t i = 0; i < n_w0; i++) { const uint32_t r = d_lower(ix, ix.wrap0[i]); if (lane == 0) runes[R] = r; R++; byte_len += d_width(r); }
    for (uint32_t i = 0; i < qlen;) {
      uint32_t adv;
      const uint32_t r = d_lower(ix, d_next_rune(q + i, qlen - i, &adv));
      i += adv;
      if (lane == 0) runes[R] = r;
      R++; byte_len += d_width(r);
    }
    for (uint32_t j = 0; j < n_w1; j++) { const uint32_t r = d_lower(ix, ix.wrap1[j]); if (lane == 0) runes[R] = r; R++; byte_len += d_width(r); }
  }
  __syncthreads();
  uint32_t t0 = 0, t1 = R;                                      // strings.Trim(text, " ")
  while (t0 < t1 && runes[t0] == ' ') { t0++; byte_len--; }
  while (t1 > t0 && runes[t1 - 1] == ' ') { t1--; byte_len--; }
  const uint32_t q_n = ix.q;
  uint32_t A = 0;
  if (byte_len < q_n) A = 0;                                    // ngram_tokenizer.go:18
  else if (t1 - t0 <= q_n) {                                    // one short gram: the whole text
    if (lane == 0) { uint32_t w[8]; for (uint32_t t = 0; t < t1 - t0; t++) w[t] = runes[t0 + t]; keys[0] = d_pack_key(ix, w, t1 - t0); }
    A = 1;
  } else {
    // appendUnique (ngram_tokenizer.go:46-54): a gram stays where it FIRST occurs.  htab: the smallest position of every
    // distinct gram (open addressing on the runes of the window; equal windows meet in one slot and keep the minimum).
    const uint32_t G = t1 - t0 - q_n + 1;
    // (the slot's tables are sized in n-grams, not bytes: wrap strings of q runes or more — q = 2 with '$', '$' — make more
    //  n-grams than the text has bytes; beyond the tables the text is rejected like one of too many bytes)
    if (G > SG_LONG_MAX_TERMS) return 0xFFFFFFFFu;
    uint32_t H = 64;
    while (H < 2u * G) H <<= 1;
    for (uint32_t i = lane; i < H; i += 64) htab[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t g = lane; g < G; g += 64) {
      for (uint32_t h = long_gram_hash(runes, t0 + g, q_n) & (H - 1u);; h = (h + 1u) & (H - 1u)) {
        uint32_t cur = ld_l2(htab + h);
        if (cur == 0xFFFFFFFFu) { cur = atomicCAS(htab + h, 0xFFFFFFFFu, g); if (cur == 0xFFFFFFFFu) break; }
        if (long_same_gram(runes, t0 + cur, t0 + g, q_n)) { atomicMin(htab + h, g); break; }
      }
    }
    __syncthreads();
    for (uint32_t base = 0; base < G; base += 64) {
      const uint32_t g = base + (uint32_t)lane;
      bool keep = false;
      if (g < G)
        for (uint32_t h = long_gram_hash(runes, t0 + g, q_n) & (H - 1u);; h = (h + 1u) & (H - 1u)) {
          const uint32_t cur = ld_l2(htab + h);
          if (long_same_gram(runes, t0 + cur, t0 + g, q_n)) { keep = cur == g; break; }
        }
      const uint64_t m = ballot(keep);
      if (keep) {
        uint32_t w[8];
        for (uint32_t t = 0; t < q_n; t++) w[t] = runes[t0 + g + t];
        keys[A + popc64(m & ((1ull << lane) - 1ull))] = d_pack_key(ix, w, q_n);
      }
      A += popc64(m);
    }
  }
  __syncthreads();
  return A;
}

__device__ void long_query(const BatchArgs& a, uint32_t qi, uint8_t* slot, int lane) {
  const DeviceIndex& ix = a.ix;
  uint64_t* keys = (uint64_t*)(slot + LongSlot::o_keys);
  uint32_t* term = (uint32_t*)(slot + LongSlot::o_term);
  uint32_t* sk = (uint32_t*)(slot + LongSlot::o_sk);
  uint32_t* sv = (uint32_t*)(slot + LongSlot::o_sv);
  int* stk = (int*)(slot + LongSlot::o_stk);
  uint32_t* cnt = (uint32_t*)(slot + LongSlot::o_cnt);
  const uint64_t qb = a.q_offs[qi], qe = a.q_len ? qb + a.q_len[qi] : a.q_offs[qi + 1];
  const uint8_t* q = a.q_blob + qb;
  const uint64_t qlen64 = qe - qb;
  if (qlen64 > SG_LONG_MAX_TERMS) return;                       // stays SG_COUNT_TOO_LONG
  const uint32_t qlen = (uint32_t)qlen64, k = a.k;
  const uint32_t A = long_tokenize(ix, a.autocomplete == 1, q, qlen, slot, lane);
  if (A == 0xFFFFFFFFu) return;                                 // more n-grams than the slot's tables: stays SG_COUNT_TOO_LONG
  if (A == 0) { if (lane == 0) a.out_counts[qi] = 0; return; }
  if (a.metric == SG_TABLE && A > a.mt.a_max) return;            // the tables end before this cardinality: stays SG_COUNT_TOO_LONG
  if (ix.slots) for (uint32_t i = lane; i < A; i += 64) term[i] = d_term_lookup(ix, keys[i]);
  __syncthreads();
  // ---- window (suggester.go:53-62) ----
  const int S = (int)ix.S;
  int b_min, b_max;
  if (a.autocomplete == 1) { b_min = (int)A; b_max = S - 1; }
  else {
    b_min = d_min_y(a.metric, a.alpha, (int)A, a.mt);
    b_max = d_max_y(a.metric, a.alpha, (int)A, S, a.mt);
    if (b_max >= S) b_max = S - 1;
    const int span = b_max - b_min + 1;
    if (span < 0) { if (lane == 0) a.out_counts[qi] = SG_COUNT_REF_PANIC; return; }
    if (span == 0) { if (lane == 0) a.out_counts[qi] = SG_COUNT_REF_DEADLOCK; return; }
  }
  b_min = max(b_min, 0);
  TopK tk;
  tk.s = k <= SG_K_LDS ? (uint64_t*)(slot + LongSlot::o_tks) : a.scratch_s + (uint64_t)qi * k;
  tk.id = k <= SG_K_LDS ? (uint32_t*)(slot + LongSlot::o_tkid) : a.scratch_id + (uint64_t)qi * k;
  tk.n = 0; tk.k = k; tk.worst_s = 0; tk.worst_id = 0; tk.worst_pos = 0;
  const uint32_t lm_from = a.lm_values ? a.lm_from[qi] : 0u, lm_to = a.lm_values ? a.lm_to[qi] : 0u;
  const uint32_t S1 = (uint32_t)S + 1u;
  for (int B = b_min; B <= b_max; B++) {
    int T = (int)A;
    if (a.autocomplete != 1) {
      T = d_threshold(a.metric, a.alpha, (int)A, B, a.mt);
      if (T == 0 || T > B || T > (int)A) continue;               // suggester.go:76
    }
    const uint32_t x0 = ix.seg_base[B], n_seg = ix.seg_base[B + 1] - x0;
    if (n_seg == 0u || n_seg > a.long_max_seg) continue;
    uint32_t ne = 0;                                            // searcher.go:32: fewer present terms than T
    for (uint32_t i = lane; i < A; i += 64) { const uint32_t t = term[i]; if (t != kNoTerm) ne += ix.seg_off[(uint64_t)t * S1 + (uint32_t)B + 1u] != ix.seg_off[(uint64_t)t * S1 + (uint32_t)B]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ne += (uint32_t)__shfl_xor((int)ne, off, 64);
    if ((int)ne < T) continue;
    for (uint32_t j = lane; j < n_seg; j += 64) cnt[j] = 0u;
    __threadfence();
    __syncthreads();
    for (uint32_t i = 0; i < A; i++) {                           // one posting list per query-term OCCURRENCE
      const uint32_t t = term[i];
      if (t == kNoTerm) continue;
      const uint32_t c0f = ix.seg_off[(uint64_t)t * S1 + (uint32_t)B], c1 = ix.seg_off[(uint64_t)t * S1 + (uint32_t)B + 1u] & ~SG_G8_FLAG;
      const uint32_t c0 = c0f & ~SG_G8_FLAG;                       // (bit 31: the term's lists have 8-bit gaps, packed_store.inc)
      if (c0f & SG_G8_FLAG) {
        for (uint32_t c = c0 + (uint32_t)lane; c < c1; c += 64) {
          u32x16 pc;
          const uint32_t np = decode_chunk8(((const uint4*)ix.postings)[c], pc, 0);
#pragma unroll
          for (int e = 0; e < SG_PPC8; e++) if ((uint32_t)e < np) atomicAdd(cnt + (pc[e] - x0), 1u);
        }
      } else
      for (uint32_t c = c0 + (uint32_t)lane; c < c1; c += 64) {
        u32x8v pc;
        const uint32_t np = decode_chunk(((const uint4*)ix.postings)[c], pc, 0);
#pragma unroll
        for (int e = 0; e < SG_PPC; e++) if ((uint32_t)e < np) atomicAdd(cnt + (pc[e] - x0), 1u);
      }
    }
    __threadfence();
    __syncthreads();
    for (uint32_t j0 = 0; j0 < n_seg; j0 += 64) {
      const uint32_t j = j0 + (uint32_t)lane;
      const uint32_t c = j < n_seg ? ld_l2(cnt + j) : 0u;
      uint64_t m = ballot(c >= (uint32_t)T);
      if (!m) continue;
      // lane <-> candidate: docID and top-k key side by side; once the top-k is full one compare per lane leaves the few that
      // can still enter it (and with them their secondary entries: same docID, at most the same overlap)
      const uint32_t dj = ((m >> lane) & 1ull) ? ix.orig_of[x0 + j] : 0u;
      const uint64_t keyj = a.autocomplete == 1 ? ~(uint64_t)dj : a.autocomplete == 2 ? by_doc_key(dj, (int)c, B) : score_bits(d_score(a.metric, (int)c, (int)A, B, a.mt));
      while (m) {
        if (!a.lm_values && tk.n == k) { m &= ballot(better(keyj, dj, tk.worst_s, tk.worst_id)); if (!m) break; }
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t x = x0 + j0 + (uint32_t)l, ov = readlane(c, l);
        const uint32_t d = readlane(dj, l);
        auto offer = [&](int overlap) {
          if (a.lm_values) topk_insert(tk, (uint64_t)d_lm_count(a.lm_values, lm_from, lm_to, d, lane), d, lane);
          else if (a.autocomplete) { if (d >= a.ac_first) topk_insert(tk, a.autocomplete == 2 ? by_doc_key(d, overlap, B) : ~(uint64_t)d, d, lane); }
          else topk_insert(tk, score_bits(d_score(a.metric, overlap, (int)A, B, a.mt)), d, lane);
        };
        offer((int)ov);
        if (ix.n_dup_docs && ((ix.dup_bits[d >> 5] >> (d & 31u)) & 1u)) {
          // the secondary entries of a document that repeats a term — dup_secondary_overlaps, with the
          // per-list view taken from the store and the tables in HBM
          __syncthreads();
          uint32_t n = 0;
          for (uint32_t base = 0; base < A; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            bool present = false;
            uint32_t len = 0, mult = 0;
            if (i < A && term[i] != kNoTerm) {
              const uint32_t t = term[i];
              const uint32_t c0f = ix.seg_off[(uint64_t)t * S1 + (uint32_t)B], nch = ix.seg_off[(uint64_t)t * S1 + (uint32_t)B + 1u] - c0f;
              const uint32_t c0 = c0f & ~SG_G8_FLAG;
              const bool l8 = (c0f & SG_G8_FLAG) != 0u;
              present = nch != 0u;
              if (present) {
                bool has = false;
                {
                  const uint32_t* p = ix.postings + (uint64_t)c0 * 4;
                  uint32_t lo = 0, hi = nch;
                  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((p[(uint64_t)mid * 4] & (l8 ? SG_X_MASK8 : SG_X_MASK)) <= x) lo = mid + 1; else hi = mid; }
                  if (lo && l8) { u32x16 pc; const uint32_t np = decode_chunk8(((const uint4*)ix.postings)[c0 + lo - 1u], pc, 0); for (int e = 0; e < SG_PPC8; e++) has |= (uint32_t)e < np && pc[e] == x; }
                  else if (lo) { u32x8v pc; const uint32_t np = decode_chunk(((const uint4*)ix.postings)[c0 + lo - 1u], pc, 0); for (int e = 0; e < SG_PPC; e++) has |= (uint32_t)e < np && pc[e] == x; }
                }
                const uint32_t ts = t * (uint32_t)S + (uint32_t)B;
                len = ix.list_len[ts];
                const uint32_t eix = d_lower_bound_u32(ix.extra_ts, ix.n_extra, ts);
                const uint32_t raw = len + ((eix < ix.n_extra && ix.extra_ts[eix] == ts) ? ix.extra_cnt[eix] : 0u);
                mult = has ? 1u : 0u;
                if (raw <= 256u) {
                  len = raw;
                  if (has) {
                    uint32_t lo = d_lower_bound_u32(ix.dup_ts, ix.n_dups, ts);
                    while (lo < ix.n_dups && ix.dup_ts[lo] == ts && ix.dup_doc[lo] < d) lo++;
                    if (lo < ix.n_dups && ix.dup_ts[lo] == ts && ix.dup_doc[lo] == d) mult = ix.dup_mult[lo];
                  }
                }
              }
            }
            const uint64_t pm = ballot(present);
            if (present) { const uint32_t pos = n + popc64(pm & ((1ull << lane) - 1ull)); sk[pos] = len; sv[pos] = mult; }
            n += popc64(pm);
          }
          __threadfence();
          __syncthreads();
          PairSort ps{sk, sv, stk};
          ps.sort((int)n);
          __threadfence();
          __syncthreads();
          const int T0 = a.autocomplete == 1 ? (int)A : d_threshold(a.metric, a.alpha, (int)A, B, a.mt);
          if ((int)n == T0) {
            const int copies = (int)sv[0];
            for (int c2 = 1; c2 < copies; c2++) offer((int)n);
          } else if ((int)n > T0) {
            const int min_q = (int)n - T0 + 1;
            uint32_t maxm = 0;
            int tail = 0;
            for (int pp = 0; pp < (int)n; pp++) { if (pp < min_q) maxm = max(maxm, sv[pp]); else tail += sv[pp] ? 1 : 0; }
            for (uint32_t jj = 2; jj <= maxm; jj++) {
              int c2 = tail;
              for (int pp = 0; pp < min_q; pp++) c2 += sv[pp] >= jj ? 1 : 0;
              if (c2 >= T0) offer(c2);
            }
          }
          __syncthreads();
        }
      }
    }
    __syncthreads();
  }
  // ---- GetCandidates (topk.go:127-147): best first ----
  __syncthreads();
  uint32_t* out_ids = a.out_ids + (uint64_t)qi * k;
  double* out_scores = a.out_scores ? a.out_scores + (uint64_t)qi * k : nullptr;
  const uint32_t n = tk.n;
  for (uint32_t i = lane; i < n; i += 64) {
    const uint64_t s = tk.s[i];
    const uint32_t d = tk.id[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) {
      const uint64_t sj = tk.s[j];
      const uint32_t dj = tk.id[j];
      rank += (better(sj, dj, s, d) || (sj == s && dj == d && j < i)) ? 1u : 0u;
    }
    out_ids[rank] = d;
    if (a.autocomplete == 2) {
      if (out_scores) out_scores[rank] = d_score(a.metric, (int)(s & 0xFFFFu), (int)A, (int)((s >> 16) & 0xFFFFu), a.mt);
      if (a.out_aux) a.out_aux[(uint64_t)qi * k + rank] = (uint32_t)s;
    } else if (out_scores) out_scores[rank] = bits_score(s);
  }
  if (lane == 0) a.out_counts[qi] = n;
}

__global__ __launch_bounds__(64) void sg_long_kernel(const BatchArgs a) {
  const int lane = threadIdx.x;
  const uint32_t n_work = min(__builtin_amdgcn_readfirstlane(a.long_list[0]), a.n_q);   // (usually 0: the workgroup leaves at once)
  int slot = -1;
  for (uint32_t b = blockIdx.x; b < n_work; b += gridDim.x) {
    const uint32_t qi = __builtin_amdgcn_readfirstlane(a.long_list[1u + b]);
    if (slot < 0) {                                              // a slot of the replica's long-query scratch
      if (lane == 0) {
        for (int sidx = (int)(blockIdx.x % SG_LONG_SLOTS);; sidx = (sidx + 1) % SG_LONG_SLOTS) {
          if (atomicCAS(a.long_lock + sidx, 0u, 1u) == 0u) { slot = sidx; break; }
          __builtin_amdgcn_s_sleep(32);
        }
      }
      slot = __builtin_amdgcn_readfirstlane(slot);
      __threadfence();
      __syncthreads();
    }
    long_query(a, qi, a.long_scratch + (uint64_t)slot * a.long_slot_bytes, lane);
    __syncthreads();
  }
  if (slot >= 0) {
    __threadfence();
    __syncthreads();
    if (lane == 0) __hip_atomic_store(a.long_lock + slot, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------
// SpellChecker.Predict on the device (pkg/spellchecker/spellchecker.go:40-92), the steps around the two searches:
//   spell_next_kernel    NGramModel.Next (ngram_model.go:64-98) for every query: the context's word ids are walked down the
//                        levels (one binary search per level in the parent's bucket) to the range of its continuations
//   spell_select_kernel  the queries whose completion list came back short get the fuzzy search (spellchecker.go:66-78)
//   spell_merge_kernel   merge (unique), stable sort by ScoreNext (monotone in the continuation count), candidates[:topK+1] (sic)
// ------------------------------------------------------------------------------------------
struct SpellArgs {
  const uint64_t* values;        // every level's (word << 32 | count), level after level
  const uint32_t* child_begin;   // every level's bucket offsets, level after level
  uint32_t level_base[8];        // first entry of level l in values
  uint32_t cb_base[8];           // first entry of level l in child_begin
  uint32_t n_parents[8];         // entries of level l - 1 (0 for the unigrams): the bucket of orphans is n_parents[l]
  uint32_t order, n_q, top_k;
  const uint32_t* ctx;           // [n_q][8] context word ids, already wrapped / trimmed like LanguageModel.Next
  const uint8_t* ctx_len;        // [n_q] 0: no context (no scorer); 0xFF: the model's Next would return an error
  const uint8_t* has_word;       // [n_q] the query has a last word to complete
  uint32_t* lm_from; uint32_t* lm_to;   // [n_q] out: the continuations (absolute positions in values); from == to: none
  uint8_t* status;               // [n_q] out: 0 scorer, 1 nil scorer, 2 error
  const uint32_t* a_ids; const uint32_t* a_cnt;   // autocomplete rows [n_q][top_k]
  const uint32_t* f_ids; const uint32_t* f_cnt;   // fuzzy rows [n_q][top_k] (rows of the selected queries only)
  uint8_t* sel_flag;                              // [n_q] 1: the query's completion list is short, it gets the fuzzy search
  uint32_t* out_ids; uint32_t* out_counts;        // [n_q][top_k + 1]
  // ---- spell_tokenize_kernel: the word tokeniser and the word ids, on the device ----
  const uint8_t* q_blob; const uint64_t* q_offs;  // the queries
  uint32_t* ctx_w; uint8_t* ctx_len_w; uint8_t* has_word_w;   // out (what ctx / ctx_len / has_word point at)
  uint8_t* w_blob; uint64_t* w_off; uint32_t* w_len;          // out: the last word of query i is w_blob[w_off[i] .. + w_len[i]), somewhere in the query's slot w_blob[2 * q_offs[i] .. 2 * q_offs[i + 1])
  const uint2* alpha_ranges; uint32_t n_alpha_ranges;          // the model alphabet's runes >= 128 as inclusive ranges, ascending
  uint64_t alpha_ascii[2];                                     // ... and below 128 as a bitmap
  const uint32_t* lower_from; const uint32_t* lower_to; uint32_t n_lower;   // simple lower-case pairs (the index replica's)
  const uint4* vocab;                                          // open addressing: {hash lo, hash hi, word id, -}, id 0xFFFFFFFF = empty
  uint32_t vocab_mask;
  const uint8_t* vocab_bytes; const uint32_t* vocab_off;       // the words, id order (a hit is confirmed byte by byte)
  uint32_t start_symbol;
};

__device__ __forceinline__ bool d_lm_alpha_has(const SpellArgs& p, uint32_t r) {
  if (r < 128u) return (p.alpha_ascii[r >> 6] >> (r & 63u)) & 1ull;
  uint32_t lo = 0, hi = p.n_alpha_ranges;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (p.alpha_ranges[mid].y < r) lo = mid + 1; else hi = mid; }
  return lo < p.n_alpha_ranges && p.alpha_ranges[lo].x <= r;
}
__device__ __forceinline__ uint32_t d_lm_lower(const SpellArgs& p, uint32_t r) {
  if (r < 0x80u) return (r - 'A' < 26u) ? r + 32u : r;
  uint32_t lo = 0, hi = p.n_lower;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (p.lower_from[mid] < r) lo = mid + 1; else hi = mid; }
  return (lo < p.n_lower && p.lower_from[lo] == r) ? p.lower_to[lo] : r;
}
__device__ __forceinline__ uint64_t d_word_hash_step(uint64_t h, uint32_t byte) { return (h ^ (uint64_t)byte) * 0x100000001B3ull; }   // FNV-1a
#define SG_WORD_HASH_SEED 0xCBF29CE484222325ull
__device__ uint32_t d_word_id(const SpellArgs& p, uint64_t h, const uint8_t* w, uint32_t n) {
  h = d_mix64(h);
  for (uint32_t s = (uint32_t)h & p.vocab_mask;; s = (s + 1u) & p.vocab_mask) {
    const uint4 e = p.vocab[s];
    if (e.z == 0xFFFFFFFFu) return kUnknownWord;
    if (e.x == (uint32_t)h && e.y == (uint32_t)(h >> 32)) {
      const uint32_t o0 = p.vocab_off[e.z], o1 = p.vocab_off[e.z + 1u];
      bool same = o1 - o0 == n;
      // (sixteen bytes per round, their loads issued together: compared one by one with an early exit, a 7-letter word was seven
      //  dependent memory round trips)
      for (uint32_t i0 = 0; same && i0 < n; i0 += 16u) {
        uint8_t vb[16];
#pragma unroll
        for (uint32_t j = 0; j < 16u; j++) vb[j] = i0 + j < n ? p.vocab_bytes[o0 + i0 + j] : (uint8_t)0;
#pragma unroll
        for (uint32_t j = 0; j < 16u; j++) same = same && (i0 + j >= n || vb[j] == w[i0 + j]);
      }
      if (same) return e.z;
    }
  }
}

// SpellChecker.Predict's host steps (spellchecker.go:40-64,94-107; language_model.go:100-112), one thread per query: the word
// tokeniser (strings.ToLower, strings.Trim(" "), maximal runs of alphabet runes — pkg/analysis word tokenizer as the model
// builds it), the last word, the ids of the words before it (the vocabulary hash, hits confirmed on the bytes), and the
// wrap / trim rules of LanguageModel.Next.  The tokens are written lower-cased into the query's slot of w_blob, one behind the
// other (two bytes of slot per query byte: a lower-case mapping can lengthen a rune's encoding, 2 -> 3 bytes at most).
// [r4] The queries of a block of 256 threads are consecutive in q_blob: the block copies their bytes into LDS with coalesced
// loads, every thread scans ITS query there and writes its tokens into an LDS image of its slot, and the slots go out
// coalesced.  Read and written byte by byte in global memory — 64 lanes at 64 addresses 20-odd bytes apart, a dependent
// load and a store per byte — the kernel took 0.093 ms with four wavefronts per CU resident: a twentieth of a Predict step.
#define SG_STOK_BYTES 12288u     // query bytes a block stages (48 per query on average); a block above it keeps the global path
// One query.  The scan only RECORDS its tokens — where their lower-cased bytes went in the slot, how long they are, their
// hash — for the first eight and, in a ring of eight, the latest ones; the vocabulary lookups of the context words follow in
// lockstep over the wave (token t of every lane together).  Looked up where a token ended — a different byte position in
// every lane — the wave ran d_word_id's chain of dependent loads once per lane and token: most of the kernel's 0.09 ms.
// Tokens are written one behind the other (the slot has two bytes per query byte), so the last word is wherever it was
// written: w_off points there.
__device__ __forceinline__ void spell_tokenize_one(const SpellArgs& p, uint32_t i, const uint8_t* q, uint32_t qlen, uint8_t* slot, uint64_t o0) {
  uint32_t a = 0, b = qlen;
  while (a < b && q[a] == ' ') a++;                            // (U+0020 is the byte 0x20 and nothing else)
  while (b > a && q[b - 1] == ' ') b--;
  const uint32_t N = p.order;
  uint64_t f_h[8], r_h[8];                                     // hash of token t (t < 8) / of the latest token with index = t mod 8
  uint32_t f_o[8], f_l[8], r_o[8], r_l[8];                     // its bytes in the slot: offset, length
  uint32_t n_tok = 0;
  uint32_t out = 0, start = 0;                                 // bytes written to the slot; where the token in progress begins
  uint64_t h = SG_WORD_HASH_SEED;
#pragma unroll
  for (int t = 0; t < 8; t++) { f_h[t] = 0; r_h[t] = 0; f_o[t] = 0; f_l[t] = 0; r_o[t] = 0; r_l[t] = 0; }
  auto end_token = [&]() {
    if (out == start) return;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      if (n_tok == (uint32_t)t) { f_h[t] = h; f_o[t] = start; f_l[t] = out - start; }
      if ((n_tok & 7u) == (uint32_t)t) { r_h[t] = h; r_o[t] = start; r_l[t] = out - start; }
    }
    n_tok++;
    start = out;
  };
  for (uint32_t pos = a; pos < b;) {
    uint32_t adv;
    const uint32_t r = d_lm_lower(p, d_next_rune(q + pos, b - pos, &adv));
    pos += adv;
    if (!d_lm_alpha_has(p, r)) { end_token(); continue; }
    if (out == start) h = SG_WORD_HASH_SEED;                   // a new token
    uint8_t enc[4];
    const uint32_t w = d_width(r);
    if (w == 1u) enc[0] = (uint8_t)r;
    else if (w == 2u) { enc[0] = (uint8_t)(0xC0u | (r >> 6)); enc[1] = (uint8_t)(0x80u | (r & 0x3Fu)); }
    else if (w == 3u) { enc[0] = (uint8_t)(0xE0u | (r >> 12)); enc[1] = (uint8_t)(0x80u | ((r >> 6) & 0x3Fu)); enc[2] = (uint8_t)(0x80u | (r & 0x3Fu)); }
    else { enc[0] = (uint8_t)(0xF0u | (r >> 18)); enc[1] = (uint8_t)(0x80u | ((r >> 12) & 0x3Fu)); enc[2] = (uint8_t)(0x80u | ((r >> 6) & 0x3Fu)); enc[3] = (uint8_t)(0x80u | (r & 0x3Fu)); }
    for (uint32_t j = 0; j < w; j++) { slot[out + j] = enc[j]; h = d_word_hash_step(h, enc[j]); }
    out += w;
  }
  end_token();
  // the last token is the word the searches take; the tokens before it are the context (the id of the last one is never asked for)
  const uint32_t n_ctx = n_tok ? n_tok - 1u : 0u;
  uint32_t lw_o = 0, lw_l = 0;
#pragma unroll
  for (int t = 0; t < 8; t++) if (n_tok && ((n_tok - 1u) & 7u) == (uint32_t)t) { lw_o = r_o[t]; lw_l = r_l[t]; }
  // ids, token t of every lane together.  first[t]: context token t (t < 8).  ring[t]: context token j >= 8 with j mod 8 = t among
  // the last N - 1 <= 7 context tokens (the only ones of index >= 8 that language_model.go:100-112 can ask for) — r_*[t] still
  // holds it: the only later token in that ring slot would be j + 8 > n_ctx.
  uint32_t first[8], ring[8];
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const bool need = (uint32_t)t < n_ctx;
    first[t] = kUnknownWord;
    if (__builtin_amdgcn_ballot_w64(need)) { if (need) first[t] = d_word_id(p, f_h[t], slot + f_o[t], f_l[t]); }
    ring[t] = kUnknownWord;
  }
  if (__builtin_amdgcn_ballot_w64(n_ctx > N && n_ctx > 8u)) {
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const uint32_t j = n_ctx ? (n_ctx - 1u) - (((n_ctx - 1u) - (uint32_t)t) & 7u) : 0u;   // the largest j < n_ctx with j mod 8 = t
      const bool need = n_ctx > N && j >= 8u && j + (N - 1u) >= n_ctx;
      if (__builtin_amdgcn_ballot_w64(need)) { if (need) ring[t] = d_word_id(p, r_h[t], slot + r_o[t], r_l[t]); }
    }
  }
  auto id_at = [&](uint32_t j) -> uint32_t {                   // id of context token j (j < 8, or one of the last N - 1)
    uint32_t v = kUnknownWord;
#pragma unroll
    for (int t = 0; t < 8; t++) if ((j & 7u) == (uint32_t)t) v = j < 8u ? first[t] : ring[t];
    return v;
  };
  p.w_off[i] = 2 * o0 + (uint64_t)lw_o;
  p.has_word_w[i] = n_tok ? 1 : 0;
  uint8_t cl = 0;
  uint32_t seq[8];
  uint32_t n_seq = 0;
  if (n_ctx) {                                                 // spellchecker.go:94-107: no context, no scorer
    // language_model.go:100-112
    if (n_ctx + 1u < N) { seq[n_seq++] = p.start_symbol; for (uint32_t t = 0; t < n_ctx; t++) seq[n_seq++] = id_at(t); }
    else if (n_ctx > N) { for (uint32_t t = n_ctx - (N - 1u); t < n_ctx; t++) seq[n_seq++] = id_at(t); }
    else if (n_ctx == N) { for (uint32_t t = 0; t + 1u < N; t++) seq[n_seq++] = id_at(t); }   // (sic) keeps the FIRST order-1 words
    else { for (uint32_t t = 0; t < n_ctx; t++) seq[n_seq++] = id_at(t); }
    if (n_seq == 0u || n_seq >= N) cl = 0xFF;                  // ngram_model.go:65-67: an error
    else cl = (uint8_t)n_seq;
  }
  p.ctx_len_w[i] = cl;
  for (uint32_t t = 0; t < 8u; t++) p.ctx_w[(uint64_t)i * 8u + t] = (cl != 0xFF && t < n_seq) ? seq[t] : 0u;
  p.w_len[i] = lw_l;
}
__global__ __launch_bounds__(256) void spell_tokenize_kernel(const SpellArgs p) {
  __shared__ __attribute__((aligned(16))) uint8_t s_in[SG_STOK_BYTES + 16];
  __shared__ __attribute__((aligned(16))) uint8_t s_slot[2 * SG_STOK_BYTES];
  const uint32_t tid = threadIdx.x, i0 = blockIdx.x * blockDim.x, i = i0 + tid;
  const uint32_t i1 = min(i0 + (uint32_t)blockDim.x, p.n_q);
  if (i0 >= p.n_q) return;
  const uint64_t B0 = p.q_offs[i0], B1 = p.q_offs[i1];          // the block's bytes: uniform
  const uint32_t nb = (uint32_t)min(B1 - B0, (uint64_t)0xFFFFFFFFu);
  const bool staged = B1 - B0 <= (uint64_t)SG_STOK_BYTES;
  uint32_t shift = 0;
  if (staged) {                                                  // dword loads from the 4-byte boundary at or below the first byte
    const uintptr_t addr0 = (uintptr_t)(p.q_blob + B0);
    shift = (uint32_t)(addr0 & 3u);
    const uint32_t* src = (const uint32_t*)(addr0 - shift);
    const uint32_t n_dw = (nb + shift + 3u) >> 2;
    for (uint32_t k = tid; k < n_dw; k += blockDim.x) ((uint32_t*)s_in)[k] = src[k];
  }
  __syncthreads();
  if (i < p.n_q) {
    const uint64_t o0 = p.q_offs[i], o1 = p.q_offs[i + 1];
    if (staged) spell_tokenize_one(p, i, s_in + shift + (uint32_t)(o0 - B0), (uint32_t)(o1 - o0), s_slot + 2u * (uint32_t)(o0 - B0), o0);
    else spell_tokenize_one(p, i, p.q_blob + o0, (uint32_t)(o1 - o0), p.w_blob + 2 * o0, o0);
  }
  if (!staged) return;                                           // (uniform)
  __syncthreads();
  // the slots out: w_blob + 2 B0 is 2-byte aligned at least (the scratch block is 16-byte aligned)
  uint16_t* dst = (uint16_t*)(p.w_blob + 2 * B0);
  for (uint32_t k = tid; k < nb; k += blockDim.x) dst[k] = ((const uint16_t*)s_slot)[k];
}

__global__ void spell_next_kernel(const SpellArgs p) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n_q) return;
  const uint32_t n = p.ctx_len[i];
  uint32_t from = 0, to = 0, st = 1;
  if (n == 0xFFu) st = 2;                                     // "nGrams length should be less than the nGramModel order"
  else if (n != 0u) {
    uint32_t parent = kNoContext;
    st = 0;
    for (uint32_t l = 0; l < n && st == 0; l++) {
      const uint32_t bucket = parent == kNoContext ? p.n_parents[l] : parent;
      const uint32_t* cb = p.child_begin + p.cb_base[l];
      uint32_t lo = cb[bucket], hi = cb[bucket + 1];
      const uint32_t end = hi, w = p.ctx[(uint64_t)i * 8 + l];
      const uint64_t* v = p.values + p.level_base[l];
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)(v[mid] >> 32) < w) lo = mid + 1; else hi = mid; }
      if (lo >= end || (uint32_t)(v[lo] >> 32) != w || (uint32_t)v[lo] == 0u) st = 1;   // unseen context: no scorer
      parent = lo;
    }
    if (st == 0) {
      const uint32_t* cb = p.child_begin + p.cb_base[n];
      from = cb[parent]; to = cb[parent + 1];
      if (from == to) st = 1;                                 // SubVector(parent) == nil
      from += p.level_base[n]; to += p.level_base[n];
    }
  }
  if (st) { from = 0; to = 0; }
  p.lm_from[i] = from; p.lm_to[i] = to; p.status[i] = (uint8_t)st;
}

// (a flag per query: the fuzzy launch makes its own list of the flagged ones, longest last word first — query_order_*)
__global__ void spell_select_kernel(const SpellArgs p) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n_q) return;
  const uint32_t c = p.a_cnt[i];
  p.sel_flag[i] = (p.has_word[i] && c != SG_COUNT_TOO_LONG && c < p.top_k) ? 1 : 0;
}

// one wavefront per query: at most 2 * top_k candidates
__global__ __launch_bounds__(64) void spell_merge_kernel(const SpellArgs p) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x, k = p.top_k, row = k + 1u;
  extern __shared__ uint32_t sm[];
  uint32_t* cand = sm;                 // [2k]
  uint32_t* cnt = sm + 2 * k;          // [2k] continuation counts
  uint32_t* dst = sm + 4 * k;          // [2k] sorted
  const uint32_t ac = p.a_cnt[i];
  uint32_t out_c = 0xFFFFFFFFu;        // decided below
  if (ac == SG_COUNT_TOO_LONG) out_c = SG_COUNT_TOO_LONG;
  else if (!p.has_word[i]) out_c = 0u;
  else if (p.status[i] == 2) out_c = SG_COUNT_LM_ERROR;
  uint32_t n = 0;
  if (out_c == 0xFFFFFFFFu) {
    n = min(ac, k);
    for (uint32_t j = lane; j < n; j += 64) cand[j] = p.a_ids[(uint64_t)i * k + j];
    __syncthreads();
    if (ac < k) {                                             // the fuzzy search ran for this query
      const uint32_t fc = p.f_cnt[i];
      if (fc >= SG_COUNT_TOO_LONG) out_c = fc;                // the reference panics / dead-locks here (suggester.go:62)
      else {
        for (uint32_t x = 0; x < min(fc, k); x++) {           // merge — spellchecker.go:133-150 (order of the fuzzy list kept)
          const uint32_t y = p.f_ids[(uint64_t)i * k + x];
          bool dup = false;
          for (uint32_t j = lane; j < n; j += 64) dup |= cand[j] == y;
          if (!__builtin_amdgcn_ballot_w64(dup)) { if (lane == 0) cand[n] = y; n++; }
          __syncthreads();
        }
      }
    }
  }
  if (out_c != 0xFFFFFFFFu) { if (lane == 0) p.out_counts[i] = out_c; return; }
  const bool scorer = p.status[i] == 0;
  if (scorer) {                                               // sort.SliceStable by ScoreNext desc — :127-131
    const uint32_t from = p.lm_from[i], to = p.lm_to[i];
    // the continuation counts of all candidates together ([r4]: one wave-wide 64-ary search per candidate, one after the other,
    // was 2 n dependent memory round trips per query): a list of at most 64 entries sits in the lanes and a candidate's count is
    // a compare across the wave; a longer one is searched by every lane for ITS candidate
    if (to - from <= 64u) {
      const uint64_t v = from + lane < to ? p.values[from + lane] : ~0ull;
      for (uint32_t j = 0; j < n; j++) {
        const uint64_t m = __builtin_amdgcn_ballot_w64((uint32_t)(v >> 32) == cand[j] && from + lane < to);
        const uint32_t c = m ? (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, __builtin_ctzll(m)) : 0u;
        if (lane == 0) cnt[j] = c;
      }
    } else {
      for (uint32_t j = lane; j < n; j += 64) {
        const uint32_t w = cand[j];
        uint32_t lo = from, hi = to;
        while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((uint32_t)(p.values[mid] >> 32) < w) lo = mid + 1u; else hi = mid; }
        uint32_t c = 0;
        if (lo < to) { const uint64_t v = p.values[lo]; if ((uint32_t)(v >> 32) == w) c = (uint32_t)v; }
        cnt[j] = c;
      }
    }
    __syncthreads();
    for (uint32_t j = lane; j < n; j += 64) {
      uint32_t r = 0;
      for (uint32_t x = 0; x < n; x++) r += (cnt[x] > cnt[j] || (cnt[x] == cnt[j] && x < j)) ? 1u : 0u;
      dst[r] = cand[j];
    }
    __syncthreads();
  }
  const uint32_t* src = scorer ? dst : cand;
  const uint32_t m = k < n ? row : n;                         // candidates[:topK+1] (sic) — :87-89
  for (uint32_t j = lane; j < m; j += 64) p.out_ids[(uint64_t)i * row + j] = src[j];
  if (lane == 0) p.out_counts[i] = m;
}

// Workgroups start in index order and one wavefront owns one query, so a launch ends when its last-started heavy query
// does.  A query's work grows with its length (more lists, a wider window of segments; for autocomplete it is the short
// prefixes that match the most): a counting sort by byte length gives the launch its queries heaviest first
// (BatchArgs::q_sel).  Two small launches over 1024-query blocks — a single workgroup spent 77 us on the LDS atomics of
// 65 536 queries: (1) lengths histogram, block-local then added to the global one; (2) every block reserves, per length,
// a run behind the lengths before it and scatters its queries there.  Which of two equally long queries comes first is
// left to the atomics; rows are written by query index, so the results do not depend on it.
// ctl: [0] = n_q (BatchArgs::q_sel_n), [4 .. 260) histogram, [260 .. 516) cursors — zeroed by the host before (1).
// `flag` non-null: only the queries it marks are listed (a launch over a subset of the batch: Predict's fuzzy top-up, whose
// subset used to run in the order the selection's atomics left — long words, the heavy ones, anywhere), and ctl[0] = how many.
#define SG_ORDER_CTL_WORDS 516
__device__ __forceinline__ uint32_t d_order_bin(const uint64_t* q_offs, const uint32_t* q_len, uint32_t i, int shortest_first) {
  const uint32_t len = q_len ? min(q_len[i], 255u) : (uint32_t)min((unsigned long long)(q_offs[i + 1] - q_offs[i]), 255ull);
  // (Predict's last words: a query without one has nothing to search and goes last either way)
  if (q_len && len == 0u) return 255u;
  return shortest_first ? len : 255u - len;
}
// blk_hist non-null (batches of at most SG_ORDER_DIRECT_BLOCKS blocks): (1) also leaves every block's own histogram, and (2)
// takes a block's run of a length from the blocks before it — a sum of coalesced loads instead of an atomic per block and
// length on 256 hot words (64 blocks x ~40 lengths: 51 us of a 65 536-query batch's step, now ~10).  Larger batches keep the
// atomics (the sums would grow with the square of the blocks).
#define SG_ORDER_DIRECT_BLOCKS 128u
__global__ __launch_bounds__(1024) void query_order_count_kernel(const uint64_t* q_offs, const uint32_t* q_len, uint32_t n_q, int shortest_first, uint32_t* ctl,
                                                                  const uint8_t* flag, uint32_t* blk_hist, uint32_t* zero0, uint32_t* zero1) {
  __shared__ uint32_t hist[256];
  const uint32_t tid = threadIdx.x, i = blockIdx.x * 1024u + tid;
  if (tid < 256u) hist[tid] = 0u;
  __syncthreads();
  if (i < n_q && (!flag || flag[i])) atomicAdd(&hist[d_order_bin(q_offs, q_len, i, shortest_first)], 1u);
  __syncthreads();
  // (direct path: the blocks' own histograms are all the scatter launch needs — no atomics on the batch's, whose words then
  //  need no memset launch in front of this one)
  if (!blk_hist && tid < 256u && hist[tid]) atomicAdd(ctl + 4 + tid, hist[tid]);
  if (blk_hist && tid < 256u) blk_hist[blockIdx.x * 256u + tid] = hist[tid];
  if (i == 0u) {
    ctl[0] = n_q;
    if (blk_hist) { ctl[1] = 0u; ctl[2] = 0u; ctl[3] = 0u; }     // (ctl[1]: the pipeline's count of queries left to the fused kernel)
    // [r5] the call's other control words (the list of queries for sg_long_kernel, the pipeline's list for the fused kernel): zeroed
    // here, ahead of the launches that append to them, instead of by a memset launch each
    if (zero0) *zero0 = 0u;
    if (zero1) *zero1 = 0u;
  }
}
__global__ __launch_bounds__(1024) void query_order_scatter_kernel(const uint64_t* q_offs, const uint32_t* q_len, uint32_t n_q, int shortest_first, uint32_t* order, uint32_t* ctl,
                                                                    const uint8_t* flag, const uint32_t* blk_hist) {
  __shared__ uint32_t hist[256], start[256];
  const uint32_t tid = threadIdx.x, i = blockIdx.x * 1024u + tid;
  if (tid < 256u) hist[tid] = 0u;
  __syncthreads();
  const bool mine = i < n_q && (!flag || flag[i]);
  const uint32_t bin = mine ? d_order_bin(q_offs, q_len, i, shortest_first) : 0u;
  if (mine) atomicAdd(&hist[bin], 1u);
  // the whole batch's count of every length, and the blocks' before this one: four threads per length, a quarter of the blocks each
  __shared__ uint32_t tot[256], bef[256];
  if (tid < 256u) { tot[tid] = blk_hist ? 0u : ctl[4 + tid]; bef[tid] = 0u; }
  __syncthreads();
  if (blk_hist) {
    uint32_t gp = 0, bp = 0;
    for (uint32_t b = tid >> 8; b < gridDim.x; b += 4u) { const uint32_t c = blk_hist[b * 256u + (tid & 255u)]; gp += c; bp += b < blockIdx.x ? c : 0u; }
    if (gp) atomicAdd(&tot[tid & 255u], gp);
    if (bp) atomicAdd(&bef[tid & 255u], bp);
  }
  __syncthreads();
  const uint32_t g = tid < 256u ? tot[tid] : 0u, before = tid < 256u ? bef[tid] : 0u;
  if (tid < 256u) start[tid] = g;
  __syncthreads();
  if (tid < 64u) {                                             // inclusive scan over the lengths: four per lane, one wave scan
    const uint32_t c0 = start[4u * tid], c1 = start[4u * tid + 1u], c2 = start[4u * tid + 2u], c3 = start[4u * tid + 3u];
    const uint32_t sum = c0 + c1 + c2 + c3, excl = wave_scan_incl(sum, (int)tid) - sum;
    start[4u * tid] = excl + c0; start[4u * tid + 1u] = excl + c0 + c1; start[4u * tid + 2u] = excl + c0 + c1 + c2; start[4u * tid + 3u] = excl + sum;
  }
  __syncthreads();
  if (flag && blockIdx.x == 0u && tid == 255u) ctl[0] = start[255];   // the subset's size (nobody reads ctl[0] before the search launch)
  if (tid < 256u) {
    const uint32_t c = hist[tid], base = start[tid] - g;
    hist[tid] = blk_hist ? base + before : (c ? base + atomicAdd(ctl + 260 + tid, c) : 0u);
  }
  __syncthreads();
  if (mine) order[atomicAdd(&hist[bin], 1u)] = i;
}

// Result rows -> pinned host memory, written by a kernel (the asynchronous host-buffer path, capi.inc): hipMemcpyAsync
// device-to-host was seen to block its calling thread for 5-6 ms every fourth or fifth call — whatever the kernel in front of
// it took — and a pipelined host lost a sixth of the GPU to it; stores from a few wavefronts over PCIe do not involve the
// runtime's copy path at all.  Three segments in one launch.
struct HostStoreArgs { const uint4* src[3]; uint4* dst[3]; uint64_t n16[3]; };
__global__ __launch_bounds__(256) void host_store_kernel(const HostStoreArgs h) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
#pragma unroll
  for (int sgm = 0; sgm < 3; sgm++)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < h.n16[sgm]; i += stride) h.dst[sgm][i] = h.src[sgm][i];
}

// the sampled fill counters (BatchArgs::fill_stat) -> mapped host memory (see launch())
__global__ void fill_stat_copy_kernel(const uint32_t* src, uint32_t* dst_host) {
  if (threadIdx.x < 16u) dst_host[threadIdx.x] = src[threadIdx.x];    // (words 3, 6, 7: what the pipeline left to the fused kernel; 8 .. 12: its sampled volumes)
}

// test hook (sg_debug_pairsort): the device's restatement of Go 1.14 sort.Sort on arbitrary keys — a differential fuzz
// compares it with the oracle's and with a third, independent restatement (tests/gosort.py)
__global__ __launch_bounds__(64) void pairsort_test_kernel(const uint32_t* keys, uint32_t n, uint32_t* out_vals) {
  __shared__ uint32_t k[SG_MAX_A], v[SG_MAX_A];
  __shared__ int stk[64];
  for (uint32_t i = threadIdx.x; i < n; i += 64) { k[i] = keys[i]; v[i] = i; }
  __syncthreads();
  PairSort ps{k, v, stk};
  ps.sort((int)n);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += 64) out_vals[i] = v[i];
}

#include "pipeline.inc"
#include "plan2.inc"

#define sg_search_kernel sg_search_kernel_t<false, false, false, false, false>
#define sg_search_kernel_loop sg_search_kernel_t<false, false, false, false, false, true>
#define sg_search_kernel_slim sg_search_kernel_t<false, false, false, true, false>
#define sg_search_kernel_tight sg_search_kernel_t<false, false, true, false, false>
#define sg_parts_kernel sg_search_kernel_t<true, false, false, false, false>
#define sg_parts_kernel_tight sg_search_kernel_t<true, false, true, false, false>
#define sg_lm_kernel sg_search_kernel_t<false, true, false, false, false>
#define sg_lm_kernel_slim sg_search_kernel_t<false, true, false, true, false>
// indexes with 8-bit-gap terms (DeviceIndex::has_g8): the full LDS layout only
#define sg_search_kernel_g8 sg_search_kernel_t<false, false, false, false, true>
#define sg_search_kernel_tight_g8 sg_search_kernel_t<false, false, true, false, true>
#define sg_parts_kernel_g8 sg_search_kernel_t<true, false, false, false, true>
#define sg_parts_kernel_tight_g8 sg_search_kernel_t<true, false, true, false, true>
#define sg_lm_kernel_g8 sg_search_kernel_t<false, true, false, false, true>

}  // namespace sg

#include "capi.inc"
