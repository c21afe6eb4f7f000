// single_query_load — the calling pattern of the reference's service (one Suggest per goroutine, many goroutines:
// pkg/suggest/service_test.go:36-79) against the C ABI: N threads of blocking sg_suggest_one calls on one index handle.
// Every answer is checked against the rows of one sg_suggest_batch call over the same queries.
//   usage: single_query_load <dict_size> <threads> <seconds>   -> one JSON line
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/suggest_hip.h"

static uint64_t mix(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}

int main(int argc, char** argv) {
  const uint32_t n_docs = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u;
  const int n_thr = argc > 2 ? atoi(argv[2]) : 256;
  const double secs = argc > 3 ? atof(argv[3]) : 3.0;
  const uint32_t n_q = 16384, k = 10;
  const char* sym = "abcdefghijklmnopqrstuvwxyz0123456789";
  std::string blob;
  std::vector<uint64_t> offs(1, 0);
  for (uint32_t d = 0; d < n_docs; d++) {
    const uint32_t len = 8 + (uint32_t)(mix(d * 2654435761ull + 1) % 25);
    for (uint32_t c = 0; c < len; c++) blob.push_back(sym[mix(((uint64_t)d << 8) + c + 77) % 36]);
    offs.push_back(blob.size());
  }
  std::string qblob;
  std::vector<uint64_t> qoffs(1, 0);
  for (uint32_t q = 0; q < n_q; q++) {                       // a dictionary string with one substitution
    const uint32_t d = (uint32_t)(mix(q + 0x1234567ull) % n_docs);
    std::string s = blob.substr(offs[d], offs[d + 1] - offs[d]);
    s[mix(q * 31 + 5) % s.size()] = sym[mix(q * 17 + 3) % 36];
    qblob += s;
    qoffs.push_back(qblob.size());
  }
  const char* alpha[] = {"english", "numbers", "$"};
  sg_desc desc{3, "$", "$", "$", alpha, 3};
  sg_index* ix = nullptr;
  if (sg_index_build_device((const uint8_t*)blob.data(), offs.data(), n_docs, &desc, 0, &ix) || sg_index_upload(ix, 0)) {
    fprintf(stderr, "build/upload failed: %s\n", sg_last_error());
    return 2;
  }
  std::vector<uint32_t> ids((size_t)n_q * k), cnt(n_q);
  std::vector<double> sc((size_t)n_q * k);
  if (sg_suggest_batch(ix, (const uint8_t*)qblob.data(), qoffs.data(), n_q, SG_JACCARD, 0.5, k, ids.data(), sc.data(), cnt.data())) {
    fprintf(stderr, "batch failed: %s\n", sg_last_error());
    return 2;
  }
  std::atomic<uint64_t> done{0}, bad{0};
  std::atomic<bool> stop{false};
  std::vector<std::thread> pool;
  const auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < n_thr; t++)
    pool.emplace_back([&, t] {
      uint32_t my_ids[16]; double my_sc[16]; uint32_t my_cnt = 0;
      uint64_t n = 0, wrong = 0;
      for (uint32_t q = (uint32_t)t; !stop.load(std::memory_order_relaxed); q = (q + (uint32_t)n_thr) % n_q) {
        const int rc = sg_suggest_one(ix, (const uint8_t*)qblob.data() + qoffs[q], (uint32_t)(qoffs[q + 1] - qoffs[q]), SG_JACCARD, 0.5, k,
                                      my_ids, my_sc, &my_cnt);
        bool ok = rc == 0 && my_cnt == cnt[q];
        for (uint32_t j = 0; ok && j < my_cnt && j < k; j++)
          ok = my_ids[j] == ids[(size_t)q * k + j] && memcmp(&my_sc[j], &sc[(size_t)q * k + j], 8) == 0;
        wrong += !ok;
        n++;
      }
      done += n; bad += wrong;
    });
  std::this_thread::sleep_for(std::chrono::duration<double>(secs));
  stop = true;
  for (auto& th : pool) th.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("{\"dict\": %u, \"threads\": %d, \"seconds\": %.3f, \"queries\": %llu, \"qps\": %.0f, \"mismatches\": %llu}\n", n_docs, n_thr, dt,
         (unsigned long long)done.load(), (double)done.load() / dt, (unsigned long long)bad.load());
  sg_index_release(ix);
  return bad.load() ? 1 : 0;
}
