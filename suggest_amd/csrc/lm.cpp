// lm.cpp — host side of the spellchecker caller's language model: n-gram counts with stupid backoff.
//
// What it mirrors (behaviour; the layout is this project's own):
//   pkg/lm/ngram_reader.go:38-98        Google n-gram format: <dir>/<k>-gm, lines "w1 .. wk\tcount"
//   pkg/lm/indexer.go:50-114            word ids = line numbers of 1-gm; unknown word = 0xFFFFFFFF
//   pkg/lm/ngram_vector_builder.go + packed_array.go   counts keyed by (parent offset, word), repeated lines accumulate
//   pkg/lm/ngram_model.go:44-98,163-175 Score / Next / calcScore (alpha = 0.4, unknown = -100)
//   pkg/lm/language_model.go:78-112     ScoreWordIDs, Next (wrap / trim rules)
//   pkg/lm/tokenizer.go + pkg/analysis/word_tokenizer.go
// Layout: one LmLevel per order, entries sorted by (parent, word); instead of the reference's binary-searched list of
// (context, from) containers the children of parent p are child_begin[p] .. child_begin[p+1] — a direct index, which is
// also what the GPU kernel gets (a [from, to) range per query into one flat array of word<<32|count).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>

#include "sg_internal.h"

namespace sg {

namespace {

// entry of `word` under parent `parent` (an offset into the previous level, or kNoContext): its offset or kNoContext
uint32_t level_find(const LmLevel& lv, uint32_t n_parents, uint32_t word, uint32_t parent) {
  uint32_t bucket;
  if (parent == kNoContext) bucket = n_parents;
  else if (parent < n_parents) bucket = parent;
  else return kNoContext;
  const uint32_t from = lv.child_begin[bucket], to = lv.child_begin[bucket + 1];
  const auto b = lv.word.begin();
  const auto it = std::lower_bound(b + from, b + to, word);
  if (it == b + to || *it != word) return kNoContext;
  return (uint32_t)(it - b);
}

uint32_t parents_of(const HostLM& lm, size_t level) { return level == 0 ? 0u : (uint32_t)lm.level[level - 1].word.size(); }

double calc_score(const uint32_t* counts, size_t n) {   // ngram_model.go:163-175; counts[0] may be the corpus total
  double factor = 1;
  for (size_t i = n - 1; i >= 1; i--) {
    if (counts[i] > 0) return std::log(factor * (double)counts[i] / (double)counts[i - 1]);
    factor *= 0.4;
  }
  return -100.0;
}

}  // namespace

uint32_t lm_word_id(const HostLM& lm, const std::string& token) {
  auto it = lm.id_of.find(token);
  return it == lm.id_of.end() ? kUnknownWord : it->second;
}

int lm_load_google(const char* dir, uint32_t order, const char* start_symbol, const char* end_symbol, const std::vector<std::string>& alphabet,
                   int id_order, HostLM& lm, std::string& err) {
  if (order < 1 || order > 8) { err = "nGramOrder should be >= 1"; return SG_E_INVALID; }
  lm.order = order;
  lm.alphabet = alphabet;
  {
    std::ifstream f(std::string(dir) + "/1-gm");
    if (!f) { err = std::string("failed to open ") + dir + "/1-gm"; return SG_E_INVALID; }
    std::string line;
    if (id_order == 0) {                                       // buildIndexerWithInMemoryDictionary (indexer.go:88-114): line order
      while (std::getline(f, line)) {
        const std::string w = line.substr(0, line.find('\t'));
        lm.id_of.emplace(w, (uint32_t)lm.words.size());
        lm.words.push_back(w);
      }
    } else {                                                   // buildDictionary (binary.go:101-199): ids by (count desc, word asc)
      std::vector<std::pair<uint32_t, std::string>> items;
      while (std::getline(f, line)) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) { err = "strconv.ParseUint: parsing \"\": invalid syntax"; return SG_E_INVALID; }
        char* endp = nullptr;
        const unsigned long long c = strtoull(line.c_str() + tab + 1, &endp, 10);
        if (endp == line.c_str() + tab + 1 || *endp || c > 0xFFFFFFFFull) { err = "strconv.ParseUint: parsing the count of a 1-gm line"; return SG_E_INVALID; }
        if (tab == 0) continue;                                // an empty word is skipped (binary.go:167-169)
        items.emplace_back((uint32_t)c, line.substr(0, tab));
      }
      std::sort(items.begin(), items.end(), [](const auto& x, const auto& y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
      items.erase(std::unique(items.begin(), items.end()), items.end());      // an equal item is not inserted twice
      for (auto& it : items) { lm.id_of.emplace(it.second, (uint32_t)lm.words.size()); lm.words.push_back(it.second); }
    }
  }
  for (uint32_t k = 1; k <= order; k++) {
    std::ifstream f(std::string(dir) + "/" + std::to_string(k) + "-gm");
    if (!f) { err = "failed to open a ngram input: " + std::to_string(k) + "-gm"; return SG_E_INVALID; }
    std::map<std::pair<uint32_t, uint32_t>, uint64_t> acc;   // (parent, word) -> count; kNoContext sorts last
    std::string line;
    std::vector<uint32_t> ids;
    while (std::getline(f, line)) {
      const size_t tab = line.find('\t');
      if (tab == std::string::npos) { err = "ngram file is corrupted, expected number"; return SG_E_INVALID; }
      ids.clear();
      for (size_t p = 0;;) {
        const size_t sp = line.find(' ', p);
        const size_t end = (sp == std::string::npos || sp > tab) ? tab : sp;
        ids.push_back(lm_word_id(lm, line.substr(p, end - p)));
        if (end == tab) break;
        p = end + 1;
      }
      if (ids.size() != k) { err = "failed to add nGrams to a builder: nGrams order is out of range"; return SG_E_INVALID; }
      uint32_t parent = kNoContext;
      for (uint32_t i = 0; i + 1 < k; i++) parent = level_find(lm.level[i], parents_of(lm, i), ids[i], parent);
      char* endp = nullptr;
      const unsigned long long c = strtoull(line.c_str() + tab + 1, &endp, 10);
      if (endp == line.c_str() + tab + 1) { err = "ngram file is corrupted, expected number"; return SG_E_INVALID; }
      acc[{parent, ids.back()}] += c;
    }
    LmLevel lv;
    const uint32_t n_parents = parents_of(lm, k - 1);
    lv.child_begin.assign((size_t)n_parents + 2, 0);
    for (const auto& kv : acc) {
      const uint32_t bucket = kv.first.first == kNoContext ? n_parents : kv.first.first;
      lv.child_begin[bucket + 1]++;
      lv.word.push_back(kv.first.second);
      lv.count.push_back((uint32_t)kv.second);
      lv.total = (uint32_t)(lv.total + kv.second);           // WordCount is uint32: wraps like the reference's
    }
    for (size_t b = 0; b + 1 < lv.child_begin.size(); b++) lv.child_begin[b + 1] += lv.child_begin[b];
    lm.level.push_back(std::move(lv));
  }
  lm.start_symbol = lm_word_id(lm, start_symbol ? start_symbol : "");
  lm.end_symbol = lm_word_id(lm, end_symbol ? end_symbol : "");
  return SG_OK;
}

// dictionary.OpenCDBDictionary (pkg/dictionary/cdb_dictionary.go over alldroll/cdb): D. J. Bernstein's constant database as
// BuildCDBDictionary writes it (helpers.go:52-100) — key = docID as 4 bytes little endian, value = the word.  Records start at
// byte 2048 and run to the first hash table.
static bool read_cdb_words(const char* path, std::vector<std::string>& words, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = std::string("failed to open cdb dictionary file: ") + path; return false; }
  std::string d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  auto u32 = [&](size_t o) { return (uint32_t)(uint8_t)d[o] | ((uint32_t)(uint8_t)d[o + 1] << 8) | ((uint32_t)(uint8_t)d[o + 2] << 16) | ((uint32_t)(uint8_t)d[o + 3] << 24); };
  if (d.size() < 2048) { err = "cdb dictionary is truncated"; return false; }
  size_t end = d.size();
  for (int i = 0; i < 256; i++) {
    if (!u32((size_t)i * 8 + 4)) continue;                   // (an empty table has no position)
    const size_t tpos = u32((size_t)i * 8);
    if (tpos < 2048 || tpos > d.size()) { err = "cdb dictionary is corrupted: hash table inside the header or past the end"; return false; }
    end = std::min<size_t>(end, tpos);
  }
  std::map<uint32_t, std::string> by_id;
  for (size_t pos = 2048; pos + 8 <= end;) {
    const size_t klen = u32(pos), dlen = u32(pos + 4);
    if (pos + 8 + klen + dlen > end) { err = "cdb dictionary is corrupted"; return false; }
    if (klen == 4) by_id[u32(pos + 8)] = d.substr(pos + 8 + klen, dlen);
    pos += 8 + klen + dlen;
  }
  words.clear();
  for (const auto& kv : by_id) { if (kv.first != words.size()) { err = "cdb dictionary has a hole in its ids"; return false; } words.push_back(kv.second); }
  return true;
}

// RetrieveLMFromBinary (pkg/lm/binary.go:59-98): <name>.lm = "0.0.2", the order, then per level "containers values total\n" +
// containers (context << 32 | from, u64 LE) + values (word << 32 | count) (nGramModel.Load ngram_model.go:123-160,
// packedArray.Load packed_array.go:118-160), then the MPH table — not needed here: the dictionary (<name>.cdb) gives the
// words in id order and lookups are exact.
int lm_load_binary(const char* lm_path, const char* cdb_path, const char* start_symbol, const char* end_symbol,
                   const std::vector<std::string>& alphabet, HostLM& lm, std::string& err) {
  lm.words.clear(); lm.id_of.clear(); lm.level.clear();
  if (!read_cdb_words(cdb_path, lm.words, err)) { lm.words.clear(); return SG_E_INVALID; }
  for (uint32_t i = 0; i < lm.words.size(); i++) lm.id_of.emplace(lm.words[i], i);
  std::ifstream f(lm_path, std::ios::binary);
  if (!f) { err = std::string("failed to open the lm binary file: ") + lm_path; return SG_E_INVALID; }
  const std::string d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (d.size() < 6 || d.compare(0, 5, "0.0.2") != 0) { err = "Version mismatch, expected 0.0.2, got " + d.substr(0, std::min<size_t>(5, d.size())); return SG_E_INVALID; }
  lm.order = (uint8_t)d[5];
  if (lm.order < 1 || lm.order > 8) { err = "unsupported nGramOrder in the binary model"; return SG_E_INVALID; }
  lm.alphabet = alphabet;
  size_t pos = 6;
  for (uint32_t k = 0; k < lm.order; k++) {
    const size_t nl = d.find('\n', pos);
    if (nl == std::string::npos) { err = "unexpected end of the binary model"; return SG_E_INVALID; }
    unsigned long long cs = 0, vs = 0, total = 0;
    if (sscanf(d.substr(pos, nl - pos).c_str(), "%llu %llu %llu", &cs, &vs, &total) != 3 || cs % 8 || vs % 8) { err = "malformed packed array header"; return SG_E_INVALID; }
    pos = nl + 1;
    if (cs > d.size() - pos || vs > d.size() - pos - cs) { err = "unexpected end of the binary model"; return SG_E_INVALID; }
    auto u64 = [&](size_t o) { uint64_t v; memcpy(&v, d.data() + o, 8); return v; };
    const size_t n_c = (size_t)cs / 8, n_v = (size_t)vs / 8;
    LmLevel lv;
    lv.total = total;
    const uint32_t n_parents = parents_of(lm, k);
    std::vector<uint32_t> start((size_t)n_parents + 2, 0xFFFFFFFFu);
    start[(size_t)n_parents + 1] = (uint32_t)n_v;
    uint64_t prev = 0;
    uint32_t prev_from = 0;
    for (size_t i = 0; i < n_c; i++) {
      const uint64_t c = u64(pos + i * 8);
      const uint32_t ctx = (uint32_t)(c >> 32), from = (uint32_t)c;
      // contexts ascend AND so do their offsets: a bucket is [from, next from) — an offset that steps back would give a
      // bucket with from > to, and the searches (host lower_bound, device binary / 64-ary search) would leave the array
      if ((i && c <= prev) || from > n_v || from < prev_from) { err = "packed array containers are not ascending"; return SG_E_INVALID; }
      prev = c; prev_from = from;
      const uint32_t bucket = ctx == kNoContext ? n_parents : ctx;
      if (bucket > n_parents) { err = "packed array context outside the previous level"; return SG_E_INVALID; }
      start[bucket] = from;
    }
    for (size_t b = (size_t)n_parents + 1; b-- > 0;) if (start[b] == 0xFFFFFFFFu) start[b] = start[b + 1];
    lv.child_begin = std::move(start);
    pos += (size_t)cs;
    lv.word.resize(n_v); lv.count.resize(n_v);
    for (size_t i = 0; i < n_v; i++) {
      const uint64_t v = u64(pos + i * 8);
      lv.word[i] = (uint32_t)(v >> 32); lv.count[i] = (uint32_t)v;
      if (lv.word[i] >= lm.words.size() && lv.word[i] != kUnknownWord) { err = "packed array holds a word id outside the dictionary"; return SG_E_INVALID; }
    }
    for (size_t b = 0; b + 1 < lv.child_begin.size(); b++)       // a bucket is searched by word: strictly ascending inside
      for (uint32_t i = lv.child_begin[b] + 1; i < lv.child_begin[b + 1]; i++)
        if (lv.word[i] <= lv.word[i - 1]) { err = "packed array values of a context are not ascending by word"; return SG_E_INVALID; }
    if (total > 0xFFFFFFFFull) { err = "packed array total does not fit 32 bits"; return SG_E_INVALID; }
    pos += (size_t)vs;
    lm.level.push_back(std::move(lv));
  }
  lm.start_symbol = lm_word_id(lm, start_symbol ? start_symbol : "");
  lm.end_symbol = lm_word_id(lm, end_symbol ? end_symbol : "");
  return SG_OK;
}

// One level in the reference's packed form (packed_array.go:12-16): containers (context << 32 | from) and values
// (word << 32 | count) — what packedArray.Store would write; lets a test compare a model with the bytes of a .lm file.
void lm_level_packed(const HostLM& lm, uint32_t level, std::vector<uint64_t>& containers, std::vector<uint64_t>& values, uint32_t* total) {
  containers.clear(); values.clear();
  const LmLevel& lv = lm.level[level];
  const uint32_t n_parents = parents_of(lm, level);
  for (uint32_t b = 0; b <= n_parents; b++)
    if (lv.child_begin[b + 1] > lv.child_begin[b]) containers.push_back(((uint64_t)(b == n_parents ? kNoContext : b) << 32) | lv.child_begin[b]);
  for (size_t i = 0; i < lv.word.size(); i++) values.push_back(((uint64_t)lv.word[i] << 32) | lv.count[i]);
  *total = (uint32_t)lv.total;
}

double lm_model_score(const HostLM& lm, const uint32_t* ids, size_t n) {   // NGramModel.Score
  const size_t order = std::min<size_t>(lm.order, n);
  uint32_t counts[10] = {0};
  uint32_t parent = kNoContext;
  for (size_t i = 0; i < order; i++) {
    if (i == 0) counts[0] = (uint32_t)lm.level[0].total;
    const uint32_t at = level_find(lm.level[i], parents_of(lm, i), ids[i], parent);
    counts[i + 1] = at == kNoContext ? 0u : lm.level[i].count[at];
    parent = at;
  }
  return calc_score(counts, order + 1);
}

double lm_score_word_ids(const HostLM& lm, const uint32_t* ids, size_t n) {   // LanguageModel.ScoreWordIDs
  std::vector<uint32_t> seq;
  seq.push_back(lm.start_symbol);
  seq.insert(seq.end(), ids, ids + n);
  seq.push_back(lm.end_symbol);
  double score = 0;
  for (size_t i = 0; i + lm.order <= seq.size(); i++) score += lm_model_score(lm, seq.data() + i, lm.order);
  return score;
}

LmNext lm_model_next(const HostLM& lm, const uint32_t* ids, size_t n) {   // NGramModel.Next
  LmNext nx;
  if (lm.order <= n || n == 0) { nx.status = 2; return nx; }   // "nGrams length should be less than the nGramModel order"
  uint32_t parent = kNoContext;
  for (size_t i = 0; i < n; i++) {
    const uint32_t at = level_find(lm.level[i], parents_of(lm, i), ids[i], parent);
    if (at == kNoContext || lm.level[i].count[at] == 0) { nx.status = 1; return nx; }
    nx.context_count = lm.level[i].count[at];
    parent = at;
  }
  const LmLevel& lv = lm.level[n];
  nx.level = (uint32_t)n;
  nx.from = lv.child_begin[parent];
  nx.to = lv.child_begin[parent + 1];
  if (nx.from == nx.to) nx.status = 1;                          // SubVector(parent) == nil
  return nx;
}

LmNext lm_next(const HostLM& lm, const uint32_t* ids, size_t n) {   // LanguageModel.Next
  std::vector<uint32_t> seq(ids, ids + n);
  const size_t N = lm.order;
  if (seq.size() + 1 < N) seq.insert(seq.begin(), lm.start_symbol);
  else if (seq.size() > N) seq.erase(seq.begin(), seq.end() - (N - 1));
  else if (seq.size() == N) seq.resize(N - 1);                   // (sic) keeps the FIRST order-1 words
  return lm_model_next(lm, seq.data(), seq.size());
}

uint32_t lm_next_count(const HostLM& lm, const LmNext& nx, uint32_t word) {
  if (nx.status) return 0;
  const LmLevel& lv = lm.level[nx.level];
  const auto b = lv.word.begin();
  const auto it = std::lower_bound(b + nx.from, b + nx.to, word);
  return (it == b + nx.to || *it != word) ? 0u : lv.count[it - b];
}

double lm_next_score(const HostLM& lm, const LmNext& nx, uint32_t word) {   // scorerNext.ScoreNext
  const uint32_t c = lm_next_count(lm, nx, word);
  if (c == 0) return -100.0;
  return std::log(1.0 * (double)c / (double)nx.context_count);
}

// NGramBuilder.Build + googleNGramFormatWriter.Write (pkg/lm/ngram_builder.go:16-64, ngram_writer.go:32-76) over
// NewSentenceRetriever (sentence_retriever.go:17-81): the text is cut into sentences at the runes of `separators`,
// every sentence is tokenised (lm tokenizer), wrapped in start/end symbols and all its k-grams, k = 1..order, are
// counted; <dir>/<k>-gm gets one "w1 .. wk\tcount" line per distinct k-gram.  The reference walks a trie of Go maps
// (random line order); here lines come in order of first appearance, so word ids (lines of 1-gm) are deterministic.
int lm_build_google_files(const uint8_t* text, size_t n, uint32_t order, const char* start_symbol, const char* end_symbol,
                          const std::vector<std::string>& alphabet, const std::vector<std::string>& separators, const char* out_dir,
                          std::string& err) {
  if (order < 1 || order > 8) { err = "nGramOrder should be >= 1"; return SG_E_INVALID; }
  HostLM tok;                                                  // only its alphabet is used (by lm_tokenize)
  tok.alphabet = alphabet;
  std::vector<std::map<std::vector<std::string>, std::pair<uint64_t, uint64_t>>> counts(order);   // gram -> (first seen, count)
  uint64_t clock = 0;
  std::vector<std::string> words, sentence;
  auto flush = [&](size_t from, size_t to) {
    if (to <= from) return;
    lm_tokenize(tok, text + from, to - from, words);
    if (words.empty()) return;                                 // ngram_builder.go:51-53
    sentence.clear();
    sentence.push_back(start_symbol);
    sentence.insert(sentence.end(), words.begin(), words.end());
    sentence.push_back(end_symbol);
    for (uint32_t k = 1; k <= order; k++)
      for (size_t i = 0; i + k <= sentence.size(); i++) {
        auto& slot = counts[k - 1][std::vector<std::string>(sentence.begin() + i, sentence.begin() + i + k)];
        if (slot.second++ == 0) slot.first = clock++;
      }
  };
  size_t start = 0;
  for (size_t i = 0; i < n;) {
    size_t adv;
    const uint32_t r = host_next_rune(text + i, n - i, &adv);
    if (host_alphabet_has(separators, r)) { flush(start, i); start = i + adv; }
    i += adv;
  }
  flush(start, n);
  for (uint32_t k = 1; k <= order; k++) {
    std::vector<std::pair<uint64_t, const std::vector<std::string>*>> lines;
    for (const auto& kv : counts[k - 1]) lines.emplace_back(kv.second.first, &kv.first);
    std::sort(lines.begin(), lines.end());
    std::ofstream f(std::string(out_dir) + "/" + std::to_string(k) + "-gm", std::ios::binary);
    if (!f) { err = "failed to create an output: " + std::string(out_dir) + "/" + std::to_string(k) + "-gm"; return SG_E_INVALID; }
    for (const auto& ln : lines) {
      const auto& gram = *ln.second;
      for (size_t i = 0; i < gram.size(); i++) { if (i) f << ' '; f << gram[i]; }
      f << '\t' << counts[k - 1][gram].second << '\n';
    }
  }
  return SG_OK;
}

void lm_tokenize(const HostLM& lm, const uint8_t* text, size_t n, std::vector<std::string>& out) {
  // strings.ToLower, strings.Trim(" "), then maximal runs of alphabet runes
  // (ASCII membership from a table made on first use: the alphabet of a model does not change once it is loaded)
  struct AsciiAlpha { uint64_t key = 0; uint8_t has[128]; };
  static thread_local AsciiAlpha aa;
  uint64_t key = 0xCBF29CE484222325ull ^ lm.alphabet.size();      // FNV-1a over the alphabet specification
  for (const auto& part : lm.alphabet) { for (unsigned char c : part) key = (key ^ c) * 0x100000001B3ull; key = (key ^ 0xFF) * 0x100000001B3ull; }
  if (aa.key != key) {
    for (uint32_t r = 0; r < 128; r++) aa.has[r] = host_alphabet_has(lm.alphabet, r) ? 1 : 0;
    aa.key = key;
  }
  static thread_local std::vector<uint32_t> runes;
  runes.clear();
  for (size_t i = 0; i < n;) {
    if (text[i] < 0x80) { const uint32_t c = text[i++]; runes.push_back(c >= 'A' && c <= 'Z' ? c + 32 : c); continue; }
    size_t adv;
    runes.push_back(host_lower_rune(host_next_rune(text + i, n - i, &adv)));
    i += adv;
  }
  size_t a = 0, b = runes.size();
  while (a < b && runes[a] == ' ') a++;
  while (b > a && runes[b - 1] == ' ') b--;
  out.clear();
  std::string cur;
  auto flush = [&] { if (!cur.empty()) { out.push_back(cur); cur.clear(); } };
  for (size_t i = a; i < b; i++) {
    const uint32_t r = runes[i];
    if (!(r < 128 ? aa.has[r] != 0 : host_alphabet_has(lm.alphabet, r))) { flush(); continue; }
    if (r < 0x80) cur.push_back((char)r);
    else if (r < 0x800) { cur.push_back((char)(0xC0 | (r >> 6))); cur.push_back((char)(0x80 | (r & 0x3F))); }
    else if (r < 0x10000) { cur.push_back((char)(0xE0 | (r >> 12))); cur.push_back((char)(0x80 | ((r >> 6) & 0x3F))); cur.push_back((char)(0x80 | (r & 0x3F))); }
    else { cur.push_back((char)(0xF0 | (r >> 18))); cur.push_back((char)(0x80 | ((r >> 12) & 0x3F))); cur.push_back((char)(0x80 | ((r >> 6) & 0x3F))); cur.push_back((char)(0x80 | (r & 0x3F))); }
  }
  flush();
}

}  // namespace sg
