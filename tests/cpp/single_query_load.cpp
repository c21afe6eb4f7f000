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
  const int direct = argc > 4 ? atoi(argv[4]) : 0;   // 1: every thread calls sg_suggest_batch with one query (no coalescing)
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
        int rc;
        if (direct > 1) {          // batches of `direct` consecutive queries through sg_suggest_batch, every row checked
          const uint32_t m = (uint32_t)direct, q0 = q % (n_q - m);
          std::vector<uint64_t> o2(m + 1);
          for (uint32_t i = 0; i <= m; i++) o2[i] = qoffs[q0 + i] - qoffs[q0];
          std::vector<uint32_t> bi((size_t)m * k), bc(m);
          std::vector<double> bs((size_t)m * k);
          rc = sg_suggest_batch(ix, (const uint8_t*)qblob.data() + qoffs[q0], o2.data(), m, SG_JACCARD, 0.5, k, bi.data(), bs.data(), bc.data());
          for (uint32_t i = 0; i < m; i++) {
            bool ok = rc == 0 && bc[i] == cnt[q0 + i];
            for (uint32_t j = 0; ok && j < bc[i] && j < k; j++)
              ok = bi[(size_t)i * k + j] == ids[(size_t)(q0 + i) * k + j] && memcmp(&bs[(size_t)i * k + j], &sc[(size_t)(q0 + i) * k + j], 8) == 0;
            if (!ok && bad.load() + wrong < 6) fprintf(stderr, "direct batch mismatch: row %u of %u (q=%u) score %.6f expected %.6f\n", i, m, q0 + i, bs[(size_t)i * k], sc[(size_t)(q0 + i) * k]);
            wrong += !ok; n++;
          }
          continue;
        } else if (direct) {
          const uint64_t o2[2] = {0, qoffs[q + 1] - qoffs[q]};
          rc = sg_suggest_batch(ix, (const uint8_t*)qblob.data() + qoffs[q], o2, 1, SG_JACCARD, 0.5, k, my_ids, my_sc, &my_cnt);
        } else
          rc = sg_suggest_one(ix, (const uint8_t*)qblob.data() + qoffs[q], (uint32_t)(qoffs[q + 1] - qoffs[q]), SG_JACCARD, 0.5, k,
                              my_ids, my_sc, &my_cnt);
        bool ok = rc == 0 && my_cnt == cnt[q];
        for (uint32_t j = 0; ok && j < my_cnt && j < k; j++)
          ok = my_ids[j] == ids[(size_t)q * k + j] && memcmp(&my_sc[j], &sc[(size_t)q * k + j], 8) == 0;
        if (!ok && bad.load() + wrong < 6) {
          {   // the truth, from the strings: distinct 3-grams of "$" + s + "$" (no character is outside the alphabet here)
            auto grams = [](const std::string& t) { std::vector<std::string> g; const std::string w = "$" + t + "$";
              for (size_t i = 0; i + 3 <= w.size(); i++) { const std::string x = w.substr(i, 3); bool dup = false; for (auto& y : g) dup |= y == x; if (!dup) g.push_back(x); } return g; };
            const auto gq = grams(qblob.substr(qoffs[q], qoffs[q + 1] - qoffs[q]));
            const uint32_t d0 = my_cnt ? my_ids[0] : ids[(size_t)q * k];
            const auto gd = grams(blob.substr(offs[d0], offs[d0 + 1] - offs[d0]));
            int o = 0; for (auto& x : gq) for (auto& y : gd) o += x == y;
            fprintf(stderr, "[truth for doc %u: A=%zu B=%zu overlap=%d jaccard=%.6f] ", d0, gq.size(), gd.size(), o, 1.0 - (1.0 - (double)o / (double)(gq.size() + gd.size() - o)));
          }
          fprintf(stderr, "mismatch q=%u rc=%d cnt %u vs %u:", q, rc, my_cnt, cnt[q]);
          for (uint32_t j = 0; j < my_cnt && j < k; j++) fprintf(stderr, " %u:%.6f", my_ids[j], my_sc[j]);
          fprintf(stderr, " [dbg A=%u qlen=%u qi=%u qb=%u; my len %u]", my_ids[k - 1], my_ids[k - 2], my_ids[k - 3], my_ids[k - 4], (uint32_t)(qoffs[q + 1] - qoffs[q]));
          fprintf(stderr, " | expected");
          for (uint32_t j = 0; j < cnt[q] && j < k; j++) fprintf(stderr, " %u:%.6f", ids[(size_t)q * k + j], sc[(size_t)q * k + j]);
          fprintf(stderr, "\n");
        }
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
