// suggest_hip.hpp — the reference's public Go API for the fuzzy-search path, mirrored in C++ over the
// C ABI of libsuggest_hip.so (include/suggest_hip.h).  Header only; link with -lsuggest_hip.
//
// Same names, argument meaning and error behaviour as the Go types they mirror:
//   suggest::Service            pkg/suggest/service.go:20-173
//   suggest::NGramIndex         pkg/suggest/ngram_index.go:7-33      (Suggester + Autocomplete)
//   suggest::Builder            pkg/suggest/ngram_index_builder.go:14-83 (NewRAMBuilder / NewFSBuilder)
//   suggest::IndexDescription   pkg/suggest/config.go:25-112          (+ ReadConfigs)
//   suggest::SearchConfig       pkg/suggest/search.go:10-35
//   suggest::Candidate          pkg/suggest/collector.go:12-17
//   suggest::ResultItem         pkg/suggest/service.go:12-17
//   suggest::metric::*          pkg/metric/{jaccard,cosine,dice,exact,overlap}.go (constructors only: the maths runs on the GPU)
//   suggest::dictionary::*      pkg/dictionary/{dictionary,memory_dictionary,cdb_dictionary,helpers}.go
//   suggest::lm::LanguageModel, suggest::spellchecker::SpellChecker   pkg/lm/language_model.go, pkg/spellchecker/spellchecker.go
// Go `error` returns become exceptions (suggest::Error) carrying the reference's message.  The Go seam hides `k` in
// a CollectorManagerFactory closure (collector.go:143-149); here it is an explicit argument, as in the C ABI, and
// *Batch methods are additive (the GPU earns its keep on batches).
#ifndef SUGGEST_HIP_HPP
#define SUGGEST_HIP_HPP

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "suggest_hip.h"

namespace suggest {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------------------------------------
// metric — pkg/metric/metric.go:4-13.  The values are tags: MinY/MaxY/Threshold/Distance are evaluated on the device
// in IEEE double with the reference's evaluation order (engine.hip d_min_y .. d_score).
// ---------------------------------------------------------------------------------------------
namespace metric {
struct Metric {
  int id;
  const char* name;
};
inline Metric JaccardMetric() { return {SG_JACCARD, "jaccard"}; }
inline Metric CosineMetric() { return {SG_COSINE, "cosine"}; }
inline Metric DiceMetric() { return {SG_DICE, "dice"}; }
inline Metric ExactMetric() { return {SG_EXACT, "exact"}; }
inline Metric OverlapMetric() { return {SG_OVERLAP, "overlap"}; }
}  // namespace metric

// ---------------------------------------------------------------------------------------------
// dictionary — docID (uint32, dictionary order) -> word
// ---------------------------------------------------------------------------------------------
namespace dictionary {
using Key = uint32_t;
using Value = std::string;
using Iterator = std::function<void(Key, const Value&)>;

class Dictionary {  // pkg/dictionary/dictionary.go:18-25
 public:
  virtual ~Dictionary() = default;
  virtual Value Get(Key key) const = 0;  // throws Error("key is not exists") — memory_dictionary.go:19-25
  virtual size_t Size() const = 0;
  virtual void Iterate(const Iterator& it) const = 0;
};

class InMemoryDictionary : public Dictionary {  // memory_dictionary.go:4-43
 public:
  explicit InMemoryDictionary(std::vector<std::string> words) : words_(std::move(words)) {}
  Value Get(Key key) const override {
    if (key >= words_.size()) throw Error("key is not exists");
    return words_[key];
  }
  size_t Size() const override { return words_.size(); }
  void Iterate(const Iterator& it) const override {
    for (size_t i = 0; i < words_.size(); i++) it((Key)i, words_[i]);
  }
  const std::vector<std::string>& Words() const { return words_; }

 private:
  std::vector<std::string> words_;
};

inline std::shared_ptr<Dictionary> NewInMemoryDictionary(std::vector<std::string> words) {
  return std::make_shared<InMemoryDictionary>(std::move(words));
}

inline std::string ReadFile(const std::string& path, const char* what) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw Error(std::string(what) + ": open " + path + ": no such file or directory");
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// OpenRAMDictionary — helpers.go:25-48: bufio.Scanner lines (a trailing '\r' is dropped), docID = line number
inline std::shared_ptr<Dictionary> OpenRAMDictionary(const std::string& path) {
  const std::string data = ReadFile(path, "failed to open dictionary file");   // helpers.go:29
  std::vector<std::string> words;
  size_t pos = 0;
  while (pos < data.size()) {
    size_t nl = data.find('\n', pos);
    if (nl == std::string::npos) nl = data.size();
    size_t end = nl;
    if (end > pos && data[end - 1] == '\r') end--;
    words.emplace_back(data.substr(pos, end - pos));
    pos = nl + 1;
  }
  return NewInMemoryDictionary(std::move(words));
}

// OpenCDBDictionary — helpers.go:14-22 + cdb_dictionary.go:17-60: D. J. Bernstein's constant database as
// BuildCDBDictionary writes it (helpers.go:52-100): key = docID as 4 bytes little endian, value = the word.
// Records start at byte 2048 and run up to the first hash table.
inline std::shared_ptr<Dictionary> OpenCDBDictionary(const std::string& path) {
  const std::string data = ReadFile(path, "failed to open cdb dictionary file");   // helpers.go:18
  if (data.size() < 2048) throw Error("fail to create cdb dictionary: short file " + path);   // cdb_dictionary.go:22
  auto u32 = [&](size_t off) {
    uint32_t v;
    memcpy(&v, data.data() + off, 4);
    return v;
  };
  uint32_t end = 0xFFFFFFFFu;
  for (int i = 0; i < 256; i++) if (u32(8 * (size_t)i + 4)) end = std::min(end, u32(8 * (size_t)i));   // (an empty table carries no position)
  std::map<uint32_t, std::string> rec;
  size_t pos = 2048;
  while (pos + 8 <= end && pos + 8 <= data.size()) {
    const uint32_t klen = u32(pos), dlen = u32(pos + 4);
    if (pos + 8 + (size_t)klen + dlen > data.size()) break;
    if (klen == 4) rec[u32(pos + 8)] = data.substr(pos + 8 + klen, dlen);
    pos += 8 + (size_t)klen + dlen;
  }
  std::vector<std::string> words(rec.size());
  for (auto& kv : rec)
    if (kv.first < words.size()) words[kv.first] = kv.second;
  return NewInMemoryDictionary(std::move(words));
}
}  // namespace dictionary

// ---------------------------------------------------------------------------------------------
// a small JSON reader (config.json and the golden vectors of the tests): objects, arrays, strings with the
// usual escapes and \uXXXX, numbers, true/false/null
// ---------------------------------------------------------------------------------------------
struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json& at(const std::string& key) const {
    for (auto& kv : obj)
      if (kv.first == key) return kv.second;
    throw Error("json: no key " + key);
  }
  bool has(const std::string& key) const {
    for (auto& kv : obj)
      if (kv.first == key) return true;
    return false;
  }
  const Json& at(size_t i) const { return arr.at(i); }
  size_t size() const { return kind == Array ? arr.size() : obj.size(); }

  static Json Parse(const std::string& text) {
    size_t p = 0;
    Json j = ParseValue(text, p);
    SkipWs(text, p);
    if (p != text.size()) throw Error("json: trailing characters");
    return j;
  }

 private:
  static void SkipWs(const std::string& s, size_t& p) {
    while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) p++;
  }
  static void AppendUtf8(std::string& out, uint32_t r) {
    if (r < 0x80) out.push_back((char)r);
    else if (r < 0x800) { out.push_back((char)(0xC0 | (r >> 6))); out.push_back((char)(0x80 | (r & 0x3F))); }
    else if (r < 0x10000) {
      out.push_back((char)(0xE0 | (r >> 12))); out.push_back((char)(0x80 | ((r >> 6) & 0x3F))); out.push_back((char)(0x80 | (r & 0x3F)));
    } else {
      out.push_back((char)(0xF0 | (r >> 18))); out.push_back((char)(0x80 | ((r >> 12) & 0x3F)));
      out.push_back((char)(0x80 | ((r >> 6) & 0x3F))); out.push_back((char)(0x80 | (r & 0x3F)));
    }
  }
  static uint32_t Hex4(const std::string& s, size_t p) {
    if (p + 4 > s.size()) throw Error("json: bad \\u escape");
    return (uint32_t)std::stoul(s.substr(p, 4), nullptr, 16);
  }
  static std::string ParseString(const std::string& s, size_t& p) {
    std::string out;
    p++;  // opening quote
    while (p < s.size() && s[p] != '"') {
      if (s[p] != '\\') { out.push_back(s[p++]); continue; }
      if (++p >= s.size()) break;
      const char c = s[p++];
      switch (c) {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'r': out.push_back('\r'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'u': {
          uint32_t r = Hex4(s, p);
          p += 4;
          if (r >= 0xD800 && r < 0xDC00 && p + 6 <= s.size() && s[p] == '\\' && s[p + 1] == 'u') {
            const uint32_t lo = Hex4(s, p + 2);
            if (lo >= 0xDC00 && lo < 0xE000) { r = 0x10000 + ((r - 0xD800) << 10) + (lo - 0xDC00); p += 6; }
          }
          AppendUtf8(out, r);
          break;
        }
        default: out.push_back(c);
      }
    }
    if (p >= s.size()) throw Error("json: unterminated string");
    p++;
    return out;
  }
  static Json ParseValue(const std::string& s, size_t& p) {
    SkipWs(s, p);
    if (p >= s.size()) throw Error("json: unexpected end");
    Json j;
    const char c = s[p];
    if (c == '{') {
      j.kind = Object;
      p++;
      SkipWs(s, p);
      if (p < s.size() && s[p] == '}') { p++; return j; }
      for (;;) {
        SkipWs(s, p);
        if (p >= s.size() || s[p] != '"') throw Error("json: expected a key");
        std::string key = ParseString(s, p);
        SkipWs(s, p);
        if (p >= s.size() || s[p] != ':') throw Error("json: expected ':'");
        p++;
        j.obj.emplace_back(std::move(key), ParseValue(s, p));
        SkipWs(s, p);
        if (p < s.size() && s[p] == ',') { p++; continue; }
        if (p < s.size() && s[p] == '}') { p++; return j; }
        throw Error("json: expected ',' or '}'");
      }
    }
    if (c == '[') {
      j.kind = Array;
      p++;
      SkipWs(s, p);
      if (p < s.size() && s[p] == ']') { p++; return j; }
      for (;;) {
        j.arr.push_back(ParseValue(s, p));
        SkipWs(s, p);
        if (p < s.size() && s[p] == ',') { p++; continue; }
        if (p < s.size() && s[p] == ']') { p++; return j; }
        throw Error("json: expected ',' or ']'");
      }
    }
    if (c == '"') { j.kind = String; j.str = ParseString(s, p); return j; }
    if (s.compare(p, 4, "true") == 0) { j.kind = Bool; j.b = true; p += 4; return j; }
    if (s.compare(p, 5, "false") == 0) { j.kind = Bool; p += 5; return j; }
    if (s.compare(p, 4, "null") == 0) { p += 4; return j; }
    size_t used = 0;
    j.kind = Number;
    try { j.num = std::stod(s.substr(p, 64), &used); } catch (...) { throw Error("json: bad value"); }
    p += used;
    return j;
  }
};

// ---------------------------------------------------------------------------------------------
// IndexDescription — pkg/suggest/config.go:25-112
// ---------------------------------------------------------------------------------------------
using Driver = std::string;
static const char* const RAMDriver = "RAM";    // config.go:19
static const char* const DiscDriver = "DISC";  // config.go:21

struct IndexDescription {
  Driver driver = RAMDriver;
  std::string Name;
  int NGramSize = 3;
  std::string SourcePath, OutputPath;
  std::vector<std::string> Alphabet;
  std::string Pad;
  std::string Wrap[2];
  std::string basePath;

  static std::string Join(const std::string& base, const std::string& p) {
    if (!p.empty() && p[0] == '/') return p;  // path.IsAbs
    return base + "/" + p;
  }
  std::string GetIndexPath() const { return Join(basePath, OutputPath); }                      // config.go:43-49
  std::string GetSourcePath() const { return Join(basePath, SourcePath); }                     // config.go:52-58
  std::string GetDictionaryFile() const { return GetIndexPath() + "/" + Name + ".cdb"; }       // config.go:38-40
  std::string GetHeaderFile() const { return GetIndexPath() + "/" + Name + ".hd"; }            // config.go:74-76
  std::string GetDocumentListFile() const { return GetIndexPath() + "/" + Name + ".dl"; }      // config.go:79-81
};

// ReadConfigs — config.go:84-112: a JSON array of descriptions; relative paths resolve against the config's directory
inline std::vector<IndexDescription> ReadConfigs(const std::string& configPath) {
  std::string text;
  try {
    text = dictionary::ReadFile(configPath, "invalid config file");
  } catch (const Error& e) {
    throw Error(std::string("invalid config file format ") + e.what());
  }
  Json j;
  try {
    j = Json::Parse(text);
  } catch (const Error& e) {
    throw Error(std::string("invalid config file format ") + e.what());
  }
  if (j.kind != Json::Array) throw Error("invalid config file format: expected an array");
  const size_t slash = configPath.find_last_of('/');
  const std::string base = slash == std::string::npos ? "." : configPath.substr(0, slash);
  std::vector<IndexDescription> out;
  for (const Json& d : j.arr) {
    IndexDescription x;
    if (d.has("driver")) x.driver = d.at("driver").str;
    if (d.has("name")) x.Name = d.at("name").str;
    if (d.has("nGramSize")) x.NGramSize = (int)d.at("nGramSize").num;
    if (d.has("source")) x.SourcePath = d.at("source").str;
    if (d.has("output")) x.OutputPath = d.at("output").str;
    if (d.has("pad")) x.Pad = d.at("pad").str;
    if (d.has("alphabet"))
      for (const Json& a : d.at("alphabet").arr) x.Alphabet.push_back(a.str);
    if (d.has("wrap")) {
      const Json& w = d.at("wrap");
      for (size_t i = 0; i < 2 && i < w.arr.size(); i++) x.Wrap[i] = w.arr[i].str;
    }
    x.basePath = base;
    out.push_back(std::move(x));
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// Candidate, SearchConfig, ResultItem
// ---------------------------------------------------------------------------------------------
struct Candidate {  // collector.go:12-17
  uint32_t Key;
  double Score;
  bool Less(const Candidate& o) const {  // collector.go:20-26
    if (Score == o.Score) return Key > o.Key;
    return Score < o.Score;
  }
};

struct ResultItem {  // service.go:12-17
  double Score;
  std::string Value;
};

struct SearchConfig {  // search.go:10-15
  std::string query;
  int topK;
  metric::Metric metric;
  double similarity;
};

inline SearchConfig NewSearchConfig(const std::string& query, int topK, metric::Metric m, double similarity) {  // search.go:18-35
  if (topK < 1) throw Error("topK should be greater or equal to 1");
  if (!(similarity > 0 && similarity <= 1)) throw Error("similarity shouble be in (0.0, 1.0]");
  return SearchConfig{query, topK, m, similarity};
}

// ---------------------------------------------------------------------------------------------
// NGramIndex — ngram_index.go:7-33 over an sg_index handle (reference counted on the C side, so an index swapped
// out of the Service stays valid for the queries still running on it — service_test.go:36-79)
// ---------------------------------------------------------------------------------------------
class NGramIndex {
 public:
  explicit NGramIndex(sg_index* h) : h_(h) {}
  ~NGramIndex() { if (h_) sg_index_release(h_); }
  NGramIndex(const NGramIndex&) = delete;
  NGramIndex& operator=(const NGramIndex&) = delete;
  sg_index* Handle() const { return h_; }

  // Suggester.Suggest — suggester.go:46-131 with newFuzzyCollectorManager(topK)
  // One query per call, as the reference's callers do it from many goroutines: sg_suggest_one coalesces the concurrent
  // callers of this handle into shared launches.
  std::vector<Candidate> Suggest(const std::string& query, double similarity, metric::Metric m, int topK) const {
    const size_t k = (size_t)(topK > 0 ? topK : 0);
    std::vector<uint32_t> ids(k ? k : 1);
    std::vector<double> scores(k ? k : 1);
    uint32_t count = 0;
    Check(sg_suggest_one(h_, (const uint8_t*)query.data(), (uint32_t)query.size(), m.id, similarity, (uint32_t)k, ids.data(), scores.data(), &count));
    if (count == SG_COUNT_REF_PANIC) throw Error("reference behaviour: panic: makechan: size out of range");
    if (count == SG_COUNT_REF_DEADLOCK) throw Error("reference behaviour: deadlock (unbuffered channel, suggester.go:62)");
    if (count == SG_COUNT_TOO_LONG) throw Error("query has more than SG_MAX_QUERY_TERMS n-grams");
    std::vector<Candidate> out;
    for (uint32_t j = 0; j < count; j++) out.push_back(Candidate{ids[j], scores[j]});
    return out;
  }
  // Autocomplete.Autocomplete — autocomplete.go:40-77 with newFirstKCollectorManager(limit)
  std::vector<Candidate> Autocomplete(const std::string& query, int limit) const {
    const size_t k = (size_t)(limit > 0 ? limit : 0);
    std::vector<uint32_t> ids(k ? k : 1);
    uint32_t count = 0;
    Check(sg_autocomplete_one(h_, (const uint8_t*)query.data(), (uint32_t)query.size(), (uint32_t)k, ids.data(), &count));
    if (count == SG_COUNT_TOO_LONG) throw Error("query has more than SG_MAX_QUERY_TERMS n-grams");
    std::vector<Candidate> out;
    for (uint32_t j = 0; j < count; j++) out.push_back(Candidate{ids[j], 0.0});   // collector.go:104-106 -> service.go:165
    return out;
  }

  std::vector<std::vector<Candidate>> SuggestBatch(const std::vector<std::string>& queries, double similarity, metric::Metric m,
                                                   int topK) const {
    std::vector<uint64_t> offs;
    const std::string blob = Pack(queries, offs);
    const size_t n = queries.size(), k = (size_t)(topK > 0 ? topK : 0);
    std::vector<uint32_t> ids(n * k), counts(n);
    std::vector<double> scores(n * k);
    Check(sg_suggest_batch(h_, (const uint8_t*)blob.data(), offs.data(), (uint32_t)n, m.id, similarity, (uint32_t)k, ids.data(),
                           scores.data(), counts.data()));
    std::vector<std::vector<Candidate>> out(n);
    for (size_t i = 0; i < n; i++) {
      // suggester.go:62 — an empty clipped window panics (negative channel size) or blocks for ever in the reference
      if (counts[i] == SG_COUNT_REF_PANIC) throw Error("reference behaviour: panic: makechan: size out of range");
      if (counts[i] == SG_COUNT_REF_DEADLOCK) throw Error("reference behaviour: deadlock (unbuffered channel, suggester.go:62)");
      if (counts[i] == SG_COUNT_TOO_LONG) throw Error("query has more than SG_MAX_QUERY_TERMS n-grams");
      for (uint32_t j = 0; j < counts[i]; j++) out[i].push_back(Candidate{ids[i * k + j], scores[i * k + j]});
    }
    return out;
  }

  std::vector<std::vector<Candidate>> AutocompleteBatch(const std::vector<std::string>& queries, int limit) const {
    std::vector<uint64_t> offs;
    const std::string blob = Pack(queries, offs);
    const size_t n = queries.size(), k = (size_t)(limit > 0 ? limit : 0);
    std::vector<uint32_t> ids(n * k), counts(n);
    Check(sg_autocomplete_batch(h_, (const uint8_t*)blob.data(), offs.data(), (uint32_t)n, (uint32_t)k, ids.data(), counts.data()));
    std::vector<std::vector<Candidate>> out(n);
    for (size_t i = 0; i < n; i++) {
      if (counts[i] == SG_COUNT_TOO_LONG) throw Error("query has more than SG_MAX_QUERY_TERMS n-grams");
      for (uint32_t j = 0; j < counts[i]; j++) out[i].push_back(Candidate{ids[i * k + j], 0.0});   // collector.go:104-106 -> service.go:165
    }
    return out;
  }

  static void Check(int rc) {
    if (rc != SG_OK) throw Error(sg_last_error());
  }

 private:
  static std::string Pack(const std::vector<std::string>& qs, std::vector<uint64_t>& offs) {
    std::string blob;
    offs.assign(1, 0);
    for (auto& q : qs) {
      blob += q;
      offs.push_back(blob.size());
    }
    return blob;
  }
  sg_index* h_;
};

// ---------------------------------------------------------------------------------------------
// Builder — ngram_index_builder.go:14-83
// ---------------------------------------------------------------------------------------------
class Builder {
 public:
  virtual ~Builder() = default;
  virtual std::shared_ptr<NGramIndex> Build() = 0;
};

namespace detail {
struct DescC {  // an sg_desc whose strings live as long as this object
  std::vector<const char*> alpha;
  sg_desc d{};
  explicit DescC(const IndexDescription& x) {
    for (auto& a : x.Alphabet) alpha.push_back(a.c_str());
    d.ngram_size = (uint32_t)x.NGramSize;
    d.wrap_start = x.Wrap[0].c_str();
    d.wrap_end = x.Wrap[1].c_str();
    d.pad = x.Pad.c_str();
    d.alphabet = alpha.data();
    d.n_alphabet = (uint32_t)alpha.size();
  }
};

class RAMBuilder : public Builder {  // NewRAMBuilder: index the dictionary in memory (indexer.go:14-45), upload to HBM
 public:
  RAMBuilder(std::shared_ptr<dictionary::Dictionary> dict, IndexDescription desc, int device)
      : dict_(std::move(dict)), desc_(std::move(desc)), device_(device) {}
  std::shared_ptr<NGramIndex> Build() override {
    std::string blob;
    std::vector<uint64_t> offs(1, 0);
    dict_->Iterate([&](dictionary::Key, const dictionary::Value& w) {
      blob += w;
      offs.push_back(blob.size());
    });
    DescC dc(desc_);
    sg_index* h = nullptr;
    NGramIndex::Check(sg_index_build((const uint8_t*)blob.data(), offs.data(), (uint32_t)(offs.size() - 1), &dc.d, &h));
    auto ix = std::make_shared<NGramIndex>(h);
    NGramIndex::Check(sg_index_upload(h, device_));
    return ix;
  }

 private:
  std::shared_ptr<dictionary::Dictionary> dict_;
  IndexDescription desc_;
  int device_;
};

class FSBuilder : public Builder {  // NewFSBuilder: <output>/<name>.hd + .dl written by the reference's indexer
 public:
  FSBuilder(IndexDescription desc, int device) : desc_(std::move(desc)), device_(device) {}
  std::shared_ptr<NGramIndex> Build() override {
    DescC dc(desc_);
    sg_index* h = nullptr;
    NGramIndex::Check(sg_index_load_reference(desc_.GetHeaderFile().c_str(), desc_.GetDocumentListFile().c_str(), &dc.d, &h));
    auto ix = std::make_shared<NGramIndex>(h);
    NGramIndex::Check(sg_index_upload(h, device_));
    return ix;
  }

 private:
  IndexDescription desc_;
  int device_;
};
}  // namespace detail

inline std::shared_ptr<Builder> NewRAMBuilder(std::shared_ptr<dictionary::Dictionary> dict, const IndexDescription& d, int device = 0) {
  return std::make_shared<detail::RAMBuilder>(std::move(dict), d, device);
}
inline std::shared_ptr<Builder> NewFSBuilder(const IndexDescription& d, int device = 0) {
  return std::make_shared<detail::FSBuilder>(d, device);
}

// ---------------------------------------------------------------------------------------------
// Service — service.go:20-173
// ---------------------------------------------------------------------------------------------
class Service {
 public:
  explicit Service(int device = 0) : device_(device) {}

  void AddIndexByDescription(const IndexDescription& d) {  // service.go:35-41
    if (d.driver == RAMDriver) return AddRunTimeIndex(d);
    if (d.driver == DiscDriver) return AddOnDiscIndex(d);
    throw Error("unsupported driver " + d.driver);
  }
  void AddRunTimeIndex(const IndexDescription& d) {  // service.go:44-58
    std::shared_ptr<dictionary::Dictionary> dict;
    try {
      dict = dictionary::OpenRAMDictionary(d.GetSourcePath());
    } catch (const Error& e) {
      throw Error(std::string("failed to create RAMDriver builder: ") + e.what());
    }
    AddIndex(d.Name, dict, NewRAMBuilder(dict, d, device_));
  }
  void AddOnDiscIndex(const IndexDescription& d) {  // service.go:61-75
    std::shared_ptr<dictionary::Dictionary> dict;
    try {
      dict = dictionary::OpenCDBDictionary(d.GetDictionaryFile());
    } catch (const Error& e) {
      throw Error(std::string("failed to create CDB dictionary: ") + e.what());
    }
    AddIndex(d.Name, dict, NewFSBuilder(d, device_));
  }
  void AddIndex(const std::string& name, std::shared_ptr<dictionary::Dictionary> dict, const std::shared_ptr<Builder>& builder) {
    std::shared_ptr<NGramIndex> ix;
    try {
      ix = builder->Build();
    } catch (const Error& e) {
      throw Error(std::string("failed to build NGramIndex: ") + e.what());  // service.go:80-82
    }
    std::unique_lock<std::shared_mutex> lock(mu_);  // service.go:85-88
    indexes_[name] = std::move(ix);
    dictionaries_[name] = std::move(dict);
  }
  std::vector<std::string> GetDictionaries() const {  // service.go:94-103
    std::shared_lock<std::shared_mutex> lock(mu_);
    std::vector<std::string> names;
    for (auto& kv : dictionaries_) names.push_back(kv.first);
    return names;
  }
  std::vector<ResultItem> Suggest(const std::string& dictName, const SearchConfig& config) const {  // service.go:105-139
    std::shared_ptr<NGramIndex> ix;
    std::shared_ptr<dictionary::Dictionary> dict;
    Lookup(dictName, ix, dict);
    std::vector<ResultItem> out;
    for (const Candidate& c : ix->Suggest(config.query, config.similarity, config.metric, config.topK))
      out.push_back(ResultItem{c.Score, dict->Get(c.Key)});
    return out;
  }
  std::vector<ResultItem> Autocomplete(const std::string& dictName, const std::string& query, int limit) const {  // service.go:142-173
    std::shared_ptr<NGramIndex> ix;
    std::shared_ptr<dictionary::Dictionary> dict;
    Lookup(dictName, ix, dict);
    std::vector<ResultItem> out;
    for (const Candidate& c : ix->Autocomplete(query, limit)) out.push_back(ResultItem{c.Score, dict->Get(c.Key)});
    return out;
  }
  // additive: one launch for many queries of one dictionary
  std::vector<std::vector<ResultItem>> SuggestBatch(const std::string& dictName, const std::vector<std::string>& queries, int topK,
                                                    metric::Metric m, double similarity) const {
    NewSearchConfig("", topK, m, similarity);
    std::shared_ptr<NGramIndex> ix;
    std::shared_ptr<dictionary::Dictionary> dict;
    Lookup(dictName, ix, dict);
    std::vector<std::vector<ResultItem>> out;
    for (auto& row : ix->SuggestBatch(queries, similarity, m, topK)) {
      out.emplace_back();
      for (const Candidate& c : row) out.back().push_back(ResultItem{c.Score, dict->Get(c.Key)});
    }
    return out;
  }

 private:
  void Lookup(const std::string& name, std::shared_ptr<NGramIndex>& ix, std::shared_ptr<dictionary::Dictionary>& dict) const {
    std::shared_lock<std::shared_mutex> lock(mu_);
    auto i = indexes_.find(name);
    auto d = dictionaries_.find(name);
    if (i == indexes_.end() || d == dictionaries_.end())
      throw Error("given dictionary " + name + " is not exists");  // service.go:111-113
    ix = i->second;
    dict = d->second;
  }
  mutable std::shared_mutex mu_;
  std::map<std::string, std::shared_ptr<NGramIndex>> indexes_;
  std::map<std::string, std::shared_ptr<dictionary::Dictionary>> dictionaries_;
  int device_;
};

// ---------------------------------------------------------------------------------------------
// lm / spellchecker — pkg/lm/language_model.go, pkg/spellchecker/spellchecker.go
// ---------------------------------------------------------------------------------------------
namespace lm {
using WordID = uint32_t;
static const WordID UnknownWordID = 0xFFFFFFFFu;   // indexer.go:16
static const double UnknownWordScore = -100.0;     // ngram_model.go:23

struct Config {  // pkg/lm/config.go:13-23 (the fields the model and the tokenizer use)
  std::string Name;
  uint8_t NGramOrder = 3;
  std::string OutputPath;                           // directory of the <k>-gm count files
  std::vector<std::string> Alphabet;
  std::string StartSymbol = "<S>", EndSymbol = "</S>";
};

class LanguageModel {  // language_model.go:8-14 over NewGoogleNGramReader(order, indexer, dir).Read (ngram_reader.go:38-98)
 public:
  explicit LanguageModel(const Config& c) {
    std::vector<const char*> alpha;
    for (auto& a : c.Alphabet) alpha.push_back(a.c_str());
    NGramIndex::Check(sg_lm_load_google(c.OutputPath.c_str(), c.NGramOrder, c.StartSymbol.c_str(), c.EndSymbol.c_str(), alpha.data(),
                                        (uint32_t)alpha.size(), &h_));
  }
  ~LanguageModel() { if (h_) sg_lm_release(h_); }
  LanguageModel(const LanguageModel&) = delete;
  LanguageModel& operator=(const LanguageModel&) = delete;
  sg_lm* Handle() const { return h_; }

  WordID GetWordID(const std::string& token) const { return sg_lm_word_id(h_, (const uint8_t*)token.data(), (uint32_t)token.size()); }
  std::string Find(WordID id) const {                // Indexer.Find
    char buf[512];
    const int n = sg_lm_word(h_, id, buf, sizeof buf);
    if (n < 0) return "<UNK>";
    return std::string(buf, (size_t)std::min<int>(n, (int)sizeof buf));
  }
  double ScoreWordIDs(const std::vector<WordID>& ids) const { return sg_lm_score_word_ids(h_, ids.data(), (uint32_t)ids.size()); }
  double ScoreSentence(const std::vector<std::string>& sentence) const {   // language_model.go:66-76
    std::vector<WordID> ids;
    for (auto& w : sentence) ids.push_back(GetWordID(w));
    return ScoreWordIDs(ids);
  }
  size_t Size() const { return sg_lm_num_words(h_); }

 private:
  sg_lm* h_ = nullptr;
};
}  // namespace lm

namespace spellchecker {
class SpellChecker {  // spellchecker.go:14-37, wired like internal/spellchecker/dep/spellchecker.go:14-53
 public:
  SpellChecker(std::shared_ptr<lm::LanguageModel> model, const IndexDescription& indexDescription, int device = 0) : model_(std::move(model)) {
    detail::DescC dc(indexDescription);
    sg_index* h = nullptr;
    NGramIndex::Check(sg_spell_index_build(model_->Handle(), &dc.d, device, &h));
    index_ = std::make_shared<NGramIndex>(h);
  }
  // Predict — spellchecker.go:40-92
  std::vector<std::string> Predict(const std::string& query, int topK, double similarity) const {
    const uint64_t offs[2] = {0, query.size()};
    std::vector<uint32_t> ids((size_t)topK + 1);
    uint32_t count = 0;
    NGramIndex::Check(sg_spell_predict_batch(index_->Handle(), model_->Handle(), (const uint8_t*)query.data(), offs, 1, (uint32_t)topK, similarity,
                                             ids.data(), &count));
    if (count == SG_COUNT_REF_PANIC) throw Error("reference behaviour: panic: makechan: size out of range");
    if (count == SG_COUNT_REF_DEADLOCK) throw Error("reference behaviour: deadlock (unbuffered channel, suggester.go:62)");
    if (count == SG_COUNT_LM_ERROR) throw Error("nGrams length should be less than the nGramModel order");   // ngram_model.go:66
    if (count == SG_COUNT_TOO_LONG) throw Error("query has more than SG_MAX_QUERY_TERMS n-grams");
    std::vector<std::string> out;
    for (uint32_t i = 0; i < count; i++) out.push_back(model_->Find(ids[i]));
    return out;
  }

 private:
  std::shared_ptr<lm::LanguageModel> model_;
  std::shared_ptr<NGramIndex> index_;
};
}  // namespace spellchecker

}  // namespace suggest
#endif  // SUGGEST_HIP_HPP
