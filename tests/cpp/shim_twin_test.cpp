// shim_twin_test.cpp — [r5] a compiled twin of the Go shim's call sequences (go/suggesthip/suggesthip.go: no Go toolchain has ever
// seen it), entry point for entry point and in the shim's order, straight against the C ABI of include/suggest_hip.h:
//
//   dispatcher    slots of pinned memory from sg_host_alloc, two sg_suggest_submit tickets in flight, answers handed back in
//                 order from sg_ticket_wait — every row held against the synchronous sg_suggest_batch (engine.dispatch)
//   any Metric    somebody else's metric.Metric (pkg/metric/metric.go:7-16) as host-built tables: sg_metric_tables_create ->
//                 sg_suggest_batch_tables; the cache's reference and a call's reference (sg_metric_tables_retain / _release)
//                 taken and dropped in the order tablesFor / releaseTables do (engine.tablesFor)
//   any collector every candidate of a query through sg_suggest_batch_from, paged with the header's recipe (resume AT the last
//                 docID, skip what was delivered, double a page that is one docID), sorted into its segments and replayed
//                 through a foreign top-k collector: the rows of sg_suggest_batch come out (engine.suggestAny)
//   k discovery   a fuzzy manager that hides its k: one probe at the remembered k, else doubling + bisection (engine.fuzzyK)
//   Close         callers that retain / submit / wait / release while the creator's reference goes (Index.Close)
//
//   shim_twin_test <golden_dir> [--stress n]
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/suggest_hip.h"

static std::atomic<int> g_failed{0}, g_checks{0};   // (EXPECT runs in the callers' threads too)
#define EXPECT(c, what)                                                                   \
  do {                                                                                    \
    g_checks++;                                                                           \
    if (!(c)) { g_failed++; std::printf("FAIL %s (%s:%d) %s\n", what, __FILE__, __LINE__, sg_last_error()); } \
  } while (0)
#define OK(call) EXPECT((call) == SG_OK, #call)

struct Rows {
  uint32_t k = 0;
  std::vector<uint32_t> ids, cnt;
  std::vector<double> sc;
  bool same_row(uint32_t i, const Rows& o, uint32_t j) const {
    if (cnt[i] != o.cnt[j]) return false;
    const uint32_t n = cnt[i] >= 0xFFFFFFF0u ? 0 : std::min(cnt[i], k);
    for (uint32_t e = 0; e < n; e++)
      if (ids[(size_t)i * k + e] != o.ids[(size_t)j * k + e] || std::memcmp(&sc[(size_t)i * k + e], &o.sc[(size_t)j * k + e], 8)) return false;
    return true;
  }
};
static void pack(const std::vector<std::string>& qs, std::vector<uint8_t>& blob, std::vector<uint64_t>& offs) {
  blob.clear(); offs.assign(1, 0);
  for (auto& q : qs) { blob.insert(blob.end(), q.begin(), q.end()); offs.push_back(blob.size()); }
  if (blob.empty()) blob.push_back(0);
}
static Rows sync_batch(sg_index* ix, const std::vector<std::string>& qs, int metric, double sim, uint32_t k) {
  std::vector<uint8_t> blob; std::vector<uint64_t> offs;
  pack(qs, blob, offs);
  Rows r; r.k = k; r.ids.assign(qs.size() * k, 0); r.sc.assign(qs.size() * k, 0); r.cnt.assign(qs.size(), 0);
  OK(sg_suggest_batch(ix, blob.data(), offs.data(), (uint32_t)qs.size(), metric, sim, k, r.ids.data(), r.sc.data(), r.cnt.data()));
  return r;
}

// ---- the dispatcher's twin: pinned slots, two tickets in flight -----------------------------------------------------------------
struct Flight {
  void* in = nullptr; void* out = nullptr; size_t in_cap = 0, out_cap = 0;
  sg_ticket* t = nullptr; uint32_t n = 0, k = 0; size_t first = 0;
};
static bool grow(void** p, size_t* cap, size_t want) {
  if (want <= *cap) return true;
  if (*p) sg_host_free(*p);
  *p = nullptr; *cap = 0;
  size_t c = 1 << 16;
  while (c < want) c <<= 1;
  if (sg_host_alloc(c, p) != SG_OK) return false;
  *cap = c;
  return true;
}
static void TestDispatcher(sg_index* ix, const std::vector<std::string>& queries) {
  const uint32_t k = 7;
  const Rows want = sync_batch(ix, queries, SG_JACCARD, 0.5, k);
  Rows got; got.k = k; got.ids.assign(queries.size() * k, 0); got.sc.assign(queries.size() * k, 0); got.cnt.assign(queries.size(), 0);
  Flight fl[3];
  std::vector<Flight*> free_ = {&fl[0], &fl[1], &fl[2]}, inflight;
  std::mt19937 rng(7);
  auto finish = [&]() {
    Flight* f = inflight.front(); inflight.erase(inflight.begin());
    OK(sg_ticket_wait(f->t));
    const uint32_t* ids = (const uint32_t*)((char*)f->out + (size_t)f->n * k * 8);
    const uint32_t* cnt = ids + (size_t)f->n * k;
    std::memcpy(&got.sc[f->first * k], f->out, (size_t)f->n * k * 8);
    std::memcpy(&got.ids[f->first * k], ids, (size_t)f->n * k * 4);
    std::memcpy(&got.cnt[f->first], cnt, (size_t)f->n * 4);
    free_.push_back(f);
  };
  size_t at = 0;
  while (at < queries.size()) {
    const uint32_t n = (uint32_t)std::min<size_t>(queries.size() - at, 1 + rng() % 600);   // whatever has arrived: batches of any size
    if (inflight.size() == 2) finish();                                                    // two tickets in flight, the oldest handed back first
    Flight* f = free_.back(); free_.pop_back();
    size_t bytes = 0;
    for (uint32_t i = 0; i < n; i++) bytes += queries[at + i].size();
    const size_t off_bytes = ((size_t)n + 1) * 8;
    bool ok = grow(&f->in, &f->in_cap, off_bytes + bytes + 16) && grow(&f->out, &f->out_cap, (size_t)n * k * 12 + (size_t)n * 4 + 64);
    EXPECT(ok, "sg_host_alloc");
    if (!ok) return;
    uint64_t* offs = (uint64_t*)f->in; uint8_t* blob = (uint8_t*)f->in + off_bytes;
    offs[0] = 0;
    for (uint32_t i = 0; i < n; i++) { std::memcpy(blob + offs[i], queries[at + i].data(), queries[at + i].size()); offs[i + 1] = offs[i] + queries[at + i].size(); }
    f->n = n; f->k = k; f->first = at;
    double* sc = (double*)f->out; uint32_t* ids = (uint32_t*)((char*)f->out + (size_t)n * k * 8); uint32_t* cnt = ids + (size_t)n * k;
    OK(sg_suggest_submit(ix, blob, offs, n, SG_JACCARD, 0.5, k, ids, sc, cnt, &f->t));
    inflight.push_back(f);
    at += n;
  }
  while (!inflight.empty()) finish();
  size_t bad = 0;
  for (uint32_t i = 0; i < queries.size(); i++) bad += !got.same_row(i, want, i);
  EXPECT(bad == 0, "pipelined dispatcher rows == synchronous rows");
  for (auto& f : fl) { if (f.in) sg_host_free(f.in); if (f.out) sg_host_free(f.out); }
}

// ---- somebody else's Metric, tabulated ---------------------------------------------------------------------------------------
struct ForeignJaccard {   // pkg/metric/jaccard.go restated by "a caller": the shim sees only the four methods
  int MinY(double a, int n) const { return (int)std::ceil(a * n); }
  int MaxY(double a, int n) const { return (int)std::floor(n / a); }
  int Threshold(double a, int x, int y) const { return (int)std::ceil(a * (double)(x + y) / (1 + a)); }
  double Distance(int o, int x, int y) const { return 1 - (double)o / (double)(x + y - o); }
};
static sg_metric_tables* tabulate(sg_index* ix, const ForeignJaccard& m, double sim, int terms) {
  sg_stats st; sg_index_stats(ix, &st);
  int aMax = 16;
  while (aMax < terms && aMax < 256) aMax <<= 1;
  const int S = (int)st.n_segments, nA = aMax + 1;
  std::vector<int32_t> minY(nA, 0), maxY(nA, 0), thr((size_t)nA * S, 0);
  std::vector<double> score((size_t)nA * S * nA, 0.0);
  for (int a = 1; a < nA; a++) {
    int lo = m.MinY(sim, a), hi = m.MaxY(sim, a);
    minY[a] = lo; maxY[a] = hi;
    lo = std::max(lo, 0); hi = std::min(hi, S - 1);
    for (int b = lo; b <= hi; b++) {
      int t = m.Threshold(sim, a, b);
      thr[(size_t)a * S + b] = t;
      for (int o = std::max(t, 0); o <= a; o++) score[((size_t)a * S + b) * nA + o] = 1 - m.Distance(o, a, b);
    }
  }
  sg_metric_tables* t = nullptr;
  OK(sg_metric_tables_create(ix, (uint32_t)aMax, minY.data(), maxY.data(), thr.data(), score.data(), &t));
  return t;
}
static void TestForeignMetric(sg_index* ix, const std::vector<std::string>& queries) {
  const uint32_t k = 5;
  const Rows want = sync_batch(ix, queries, SG_JACCARD, 0.6, k);
  sg_metric_tables* t = tabulate(ix, ForeignJaccard(), 0.6, 64);       // tablesFor: the creator's reference is the caller's ...
  EXPECT(t != nullptr, "tables");
  if (!t) return;
  sg_metric_tables_retain(t);                                         // ... and the cache takes its own
  std::vector<uint8_t> blob; std::vector<uint64_t> offs;
  pack(queries, blob, offs);
  Rows got; got.k = k; got.ids.assign(queries.size() * k, 0); got.sc.assign(queries.size() * k, 0); got.cnt.assign(queries.size(), 0);
  OK(sg_suggest_batch_tables(ix, blob.data(), offs.data(), (uint32_t)queries.size(), t, k, got.ids.data(), got.sc.data(), got.cnt.data()));
  sg_metric_tables_release(t);                                        // releaseTables: the call's reference
  size_t bad = 0;
  for (uint32_t i = 0; i < queries.size(); i++) bad += !got.same_row(i, want, i);
  EXPECT(bad == 0, "tabulated foreign metric rows == native rows");
  // a second call finds the set in the cache (retain under the lock), the cache evicts it meanwhile (its release), the call goes on
  sg_metric_tables_retain(t);
  sg_metric_tables_release(t);                                        // eviction / Close: the cache's reference
  Rows again = got; std::fill(again.cnt.begin(), again.cnt.end(), 0u);
  OK(sg_suggest_batch_tables(ix, blob.data(), offs.data(), (uint32_t)queries.size(), t, k, again.ids.data(), again.sc.data(), again.cnt.data()));
  bad = 0;
  for (uint32_t i = 0; i < queries.size(); i++) bad += !again.same_row(i, want, i);
  EXPECT(bad == 0, "a table set evicted while a call holds it still answers");
  sg_metric_tables_release(t);                                        // the last reference: the HBM goes
}

// ---- any CollectorManager: every candidate, paged, replayed segment by segment ------------------------------------------------------
struct Cand { uint32_t doc; double score; uint32_t aux; };
static std::vector<Cand> all_candidates(sg_index* ix, const std::string& q, int metric, double sim, uint32_t page) {
  std::vector<Cand> out;
  std::vector<uint8_t> blob(q.begin(), q.end()); if (blob.empty()) blob.push_back(0);
  uint64_t offs[2] = {0, q.size()};
  uint32_t first = 0, delivered = 0;                                 // entries of docID `first` the pages so far have handed over
  for (;;) {
    std::vector<uint32_t> ids(page), aux(page); std::vector<double> sc(page); uint32_t cnt = 0;
    OK(sg_suggest_batch_from(ix, blob.data(), offs, 1, metric, sim, nullptr, first, page, ids.data(), sc.data(), aux.data(), &cnt));
    if (cnt >= 0xFFFFFFF0u) break;                                    // the reference does not answer this query
    const uint32_t n = std::min(cnt, page);
    uint32_t skipped = 0;
    for (uint32_t e = 0; e < n; e++) {
      if (ids[e] == first && skipped < delivered) { skipped++; continue; }   // (ascending docIDs: they head the page)
      out.push_back(Cand{ids[e], sc[e], aux[e]});
    }
    if (n < page) break;                                              // a short page is the last
    const uint32_t last = ids[n - 1];
    if (last == first) { delivered = n; page *= 2; continue; }        // a whole page of one docID: ask again with twice the limit
    uint32_t run = 0;
    for (uint32_t e = n; e-- > 0 && ids[e] == last;) run++;
    first = last; delivered = run;                                    // resume AT the last docID, past what was delivered of it
  }
  return out;
}
static void TestForeignCollector(sg_index* ix, const std::vector<std::string>& queries) {
  const uint32_t k = 4;
  size_t bad_pages = 0, bad_rows = 0;
  for (size_t qi = 0; qi < queries.size(); qi += 37) {
    const std::string& q = queries[qi];
    const std::vector<Cand> whole = all_candidates(ix, q, SG_COSINE, 0.5, 4096), paged = all_candidates(ix, q, SG_COSINE, 0.5, 3);
    bool same = whole.size() == paged.size();
    for (size_t i = 0; same && i < whole.size(); i++) same = whole[i].doc == paged[i].doc && whole[i].aux == paged[i].aux && !std::memcmp(&whole[i].score, &paged[i].score, 8);
    bad_pages += !same;
    // suggestAny: candidates into their segments, a collector per segment, merged by the manager — here a top-k by (score desc, doc asc)
    std::map<uint32_t, std::vector<Cand>> by_segment;
    for (auto& c : paged) by_segment[c.aux >> 16].push_back(c);
    std::vector<Cand> top;
    for (auto& kv : by_segment) for (auto& c : kv.second) top.push_back(c);
    std::stable_sort(top.begin(), top.end(), [](const Cand& a, const Cand& b) { return a.score > b.score || (a.score == b.score && a.doc < b.doc); });
    if (top.size() > k) top.resize(k);
    const Rows want = sync_batch(ix, {q}, SG_COSINE, 0.5, k);
    bool row = want.cnt[0] >= 0xFFFFFFF0u ? paged.empty() : want.cnt[0] == top.size();
    for (size_t e = 0; row && e < top.size(); e++) row = want.ids[e] == top[e].doc && !std::memcmp(&want.sc[e], &top[e].score, 8);
    bad_rows += !row;
  }
  EXPECT(bad_pages == 0, "pages of 3 through sg_suggest_batch_from deliver what one page does");
  EXPECT(bad_rows == 0, "candidates replayed through a foreign top-k collector == sg_suggest_batch rows");
}

// ---- k discovery: a manager that keeps min(n, k) of n distinct candidates ------------------------------------------------------
static uint32_t discover_k(const std::function<uint32_t(uint32_t)>& kept_of, uint32_t last_k, uint32_t max_k, int* probes) {
  *probes = 0;
  auto keeps = [&](uint32_t n) { (*probes)++; return kept_of(n); };
  if (last_k && keeps(last_k + 1) == last_k) return last_k;         // the steady state: one probe confirms the remembered k
  uint32_t lo = 1, hi = 1;
  while (hi < max_k && keeps(hi + 1) == hi + 1) { lo = hi + 1; hi = std::min(max_k, hi * 2); }
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (keeps(mid + 1) > mid) lo = mid + 1; else hi = mid; }
  return lo;
}
static void TestKDiscovery() {
  for (uint32_t k : {1u, 2u, 5u, 10u, 64u, 1000u, 65536u}) {
    int probes = 0;
    auto mgr = [&](uint32_t n) { return std::min(n, k); };
    EXPECT(discover_k(mgr, 0, SG_MAX_TOPK, &probes) == k, "k by doubling + bisection");
    EXPECT(discover_k(mgr, k, SG_MAX_TOPK, &probes) == k && probes == 1, "the remembered k is confirmed by one probe");
    EXPECT(discover_k(mgr, k == 1 ? 3 : k - 1, SG_MAX_TOPK, &probes) == k, "a stale remembered k is corrected");
  }
}

// ---- Close while callers are in flight ----------------------------------------------------------------------------------------
static void TestCloseDuringLoad(const std::vector<std::string>& lines, const sg_desc& desc, const std::vector<std::string>& queries, int rounds) {
  std::vector<uint8_t> blob; std::vector<uint64_t> offs;
  pack(lines, blob, offs);
  int refused = 0, answered = 0, wrong = 0;
  for (int round = 0; round < rounds; round++) {
    sg_index* ix = nullptr;
    OK(sg_index_build(blob.data(), offs.data(), (uint32_t)lines.size(), &desc, &ix));
    OK(sg_index_upload(ix, 0));
    const Rows want = sync_batch(ix, queries, SG_JACCARD, 0.5, 3);
    std::shared_mutex mu; bool closed = false;                       // engine.mu / engine.closed
    std::atomic<int> a_refused{0}, a_answered{0}, a_wrong{0};
    std::vector<std::thread> callers;
    for (int c = 0; c < 4; c++)
      callers.emplace_back([&, c]() {
        std::mt19937 rng(100 + c);
        for (int it = 0; it < 40; it++) {
          { std::shared_lock<std::shared_mutex> l(mu); if (closed) { a_refused++; continue; } sg_index_retain(ix); }   // engine.retain
          const uint32_t i0 = rng() % (queries.size() - 16);
          std::vector<std::string> qs(queries.begin() + i0, queries.begin() + i0 + 16);
          const Rows got = sync_batch(ix, qs, SG_JACCARD, 0.5, 3);
          for (uint32_t i = 0; i < 16; i++) if (!got.same_row(i, want, i0 + i)) a_wrong++;
          a_answered++;
          sg_index_release(ix);                                       // engine.release
        }
      });
    std::this_thread::sleep_for(std::chrono::milliseconds(3 + round % 5));
    { std::unique_lock<std::shared_mutex> l(mu); closed = true; }
    sg_index_release(ix);                                             // Close: the creator's reference; calls in flight hold theirs
    for (auto& t : callers) t.join();
    refused += a_refused; answered += a_answered; wrong += a_wrong;
  }
  EXPECT(wrong == 0, "every call answered around a Close carries the right rows");
  EXPECT(answered > 0, "some calls were in flight when the index was closed");
  std::printf("close-during-load: %d calls answered, %d refused after Close, %d rounds\n", answered, refused, rounds);
}

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: shim_twin_test <golden_dir> [--stress n]\n"); return 2; }
  const std::string golden = argv[1];
  int rounds = 3;
  for (int i = 2; i + 1 < argc; i++) if (!std::strcmp(argv[i], "--stress")) rounds = std::atoi(argv[i + 1]);
  std::vector<std::string> lines;
  { std::ifstream f(golden + "/cars.dict"); std::string l; while (std::getline(f, l)) if (!l.empty()) lines.push_back(l); }
  if (lines.empty()) { std::printf("no dictionary under %s\n", golden.c_str()); return 2; }
  const char* alphabet[] = {"russian", "english", "numbers", "$"};
  sg_desc desc{}; desc.ngram_size = 3; desc.wrap_start = "$"; desc.wrap_end = "$"; desc.pad = "$"; desc.alphabet = alphabet; desc.n_alphabet = 4;
  std::vector<uint8_t> blob; std::vector<uint64_t> offs;
  pack(lines, blob, offs);
  sg_index* ix = nullptr;
  OK(sg_index_build(blob.data(), offs.data(), (uint32_t)lines.size(), &desc, &ix));
  if (!ix) return 1;
  OK(sg_index_upload(ix, 0));
  std::vector<std::string> queries;                                   // the dictionary's own lines, mangled a little, and odd ones
  std::mt19937 rng(3);
  for (size_t i = 0; i < lines.size(); i++) {
    std::string q = lines[i];
    if (i % 3 == 0 && q.size() > 3) q.erase(rng() % q.size(), 1);
    if (i % 5 == 0 && q.size() > 2) q[rng() % q.size()] = (char)('a' + rng() % 26);
    queries.push_back(q);
  }
  for (const char* odd : {"", "a", "  ", "zz", "BMW", "toyota camry solara convertible", "\xd0\xbb\xd0\xb0\xd0\xb4\xd0\xb0"}) queries.push_back(odd);
  TestDispatcher(ix, queries);
  TestForeignMetric(ix, queries);
  TestForeignCollector(ix, queries);
  TestKDiscovery();
  sg_index_release(ix);
  TestCloseDuringLoad(lines, desc, queries, rounds);
  std::printf("%d checks, %d failed\n", g_checks.load(), g_failed.load());
  return g_failed.load() ? 1 : 0;
}
