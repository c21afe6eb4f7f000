// service_test.cpp — the reference's own tests for the fuzzy-search path, written against the C++ mirror of its API
// (include/suggest_hip.hpp).  Expected values come from tests/golden/reference_tests.json (transcribed from the Go tests,
// each entry cites its source lines).
//
//   service_test --cpu <golden_dir>   host-side logic only (no device calls): ReadConfigs, dictionaries, NewSearchConfig,
//                                     Candidate.Less, error messages
//   service_test <golden_dir>         + everything that searches: needs an MI355X
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../include/suggest_hip.hpp"

using namespace suggest;

static int g_failed = 0, g_checks = 0;
#define EXPECT(cond, what)                                               \
  do {                                                                   \
    g_checks++;                                                          \
    if (!(cond)) {                                                       \
      g_failed++;                                                        \
      std::fprintf(stderr, "FAIL %s:%d  %s\n", __FILE__, __LINE__, what); \
    }                                                                    \
  } while (0)

template <class F>
static std::string ErrorOf(F f) {
  try {
    f();
  } catch (const Error& e) {
    return e.what();
  }
  return "";
}

static IndexDescription DescriptionOf(const Json& d, const std::string& name) {
  IndexDescription x;
  x.Name = name;
  x.NGramSize = (int)d.at("nGramSize").num;
  x.Pad = d.at("pad").str;
  x.Wrap[0] = d.at("wrap").at(0).str;
  x.Wrap[1] = d.at("wrap").at(1).str;
  for (const Json& a : d.at("alphabet").arr) x.Alphabet.push_back(a.str);
  return x;
}

static metric::Metric MetricOf(const std::string& name) {
  if (name == "jaccard") return metric::JaccardMetric();
  if (name == "cosine") return metric::CosineMetric();
  if (name == "dice") return metric::DiceMetric();
  if (name == "exact") return metric::ExactMetric();
  return metric::OverlapMetric();
}

static std::vector<std::string> Strings(const Json& a) {
  std::vector<std::string> out;
  for (const Json& x : a.arr) out.push_back(x.str);
  return out;
}

// ---- host-side logic ---------------------------------------------------------------------------------------------
static void TestHost(const std::string& golden) {
  // search.go:18-35
  EXPECT(ErrorOf([] { NewSearchConfig("q", 0, metric::CosineMetric(), 0.5); }) == "topK should be greater or equal to 1", "topK check");
  EXPECT(ErrorOf([] { NewSearchConfig("q", 5, metric::CosineMetric(), 0.0); }) == "similarity shouble be in (0.0, 1.0]", "similarity 0");
  EXPECT(ErrorOf([] { NewSearchConfig("q", 5, metric::CosineMetric(), 1.5); }) == "similarity shouble be in (0.0, 1.0]", "similarity 1.5");
  EXPECT(ErrorOf([] { NewSearchConfig("q", 5, metric::CosineMetric(), 1.0); }).empty(), "similarity 1 is valid");

  // collector.go:20-26: lower score is less; on ties the higher key is less
  EXPECT((Candidate{1, 0.5}.Less(Candidate{2, 0.6})), "less by score");
  EXPECT((Candidate{7, 0.5}.Less(Candidate{3, 0.5})), "tie: higher key is less");
  EXPECT(!(Candidate{3, 0.5}.Less(Candidate{7, 0.5})), "tie: lower key is not less");

  // config.go:84-112 on the reference's own testdata/config.json
  const std::vector<IndexDescription> configs = ReadConfigs(golden + "/config.json");
  EXPECT(configs.size() >= 2, "two descriptions");
  EXPECT(configs[0].Name == "cars" && configs[0].NGramSize == 3 && configs[0].driver == DiscDriver, "cars description");
  EXPECT(configs[0].Alphabet.size() == 4 && configs[0].Alphabet[3] == "$", "cars alphabet");
  EXPECT(configs[0].Wrap[0] == "$" && configs[0].Wrap[1] == "$" && configs[0].Pad == "$", "cars wrap/pad");
  EXPECT(configs[0].GetSourcePath() == golden + "/cars.dict", "source path is relative to the config");
  EXPECT(configs[0].GetDictionaryFile() == golden + "/db/cars.cdb", "dictionary file");
  EXPECT(configs[0].GetHeaderFile() == golden + "/db/cars.hd", "header file");
  EXPECT(ErrorOf([&] { ReadConfigs(golden + "/missing.json"); }).rfind("invalid config file format", 0) == 0, "missing config");

  // dictionaries: the RAM dictionary (line order) and the reference-built CDB hold the same words under the same ids
  auto ram = dictionary::OpenRAMDictionary(configs[0].GetSourcePath());
  auto cdb = dictionary::OpenCDBDictionary(configs[0].GetDictionaryFile());
  EXPECT(ram->Size() == 5066 && cdb->Size() == ram->Size(), "cars has 5066 entries in both dictionaries");
  bool same = ram->Size() == cdb->Size();
  for (uint32_t i = 0; same && i < ram->Size(); i++) same = ram->Get(i) == cdb->Get(i);
  EXPECT(same, "cdb == lines");
  EXPECT(ErrorOf([&] { ram->Get(999999); }) == "key is not exists", "Get out of range");
  EXPECT(ErrorOf([&] { dictionary::OpenRAMDictionary(golden + "/nope.dict"); }).rfind("failed to open dictionary file", 0) == 0, "missing dict");

  // service.go:111-113
  Service empty;
  EXPECT(ErrorOf([&] { empty.Suggest("cars", NewSearchConfig("x", 1, metric::CosineMetric(), 0.5)); }) == "given dictionary cars is not exists",
         "unknown dictionary");
  EXPECT(ErrorOf([&] { empty.Autocomplete("cars", "x", 1); }) == "given dictionary cars is not exists", "unknown dictionary (autocomplete)");
  IndexDescription bad = configs[0];
  bad.driver = RAMDriver;
  bad.SourcePath = "nope.dict";
  EXPECT(ErrorOf([&] { empty.AddIndexByDescription(bad); }).rfind("failed to create RAMDriver builder: ", 0) == 0, "missing source");
}

// ---- searches (GPU) ----------------------------------------------------------------------------------------------
static void TestDevice(const std::string& golden) {
  const Json ref = Json::Parse(dictionary::ReadFile(golden + "/reference_tests.json", "golden"));
  const std::vector<std::string> collection = Strings(ref.at("small_collection"));

  {  // ngram_index_test.go:15-40 TestSuggestAuto
    const Json& t = ref.at("suggest_auto");
    auto dict = dictionary::NewInMemoryDictionary(collection);
    auto index = NewRAMBuilder(dict, DescriptionOf(t.at("description"), "index"))->Build();
    auto cands = index->Suggest(t.at("query").str, t.at("similarity").num, MetricOf(t.at("metric").str), (int)t.at("topK").num);
    bool ok = cands.size() == t.at("expected_ids").size();
    for (size_t i = 0; ok && i < cands.size(); i++) ok = cands[i].Key == (uint32_t)t.at("expected_ids").at(i).num;
    EXPECT(ok, "TestSuggestAuto ids");
  }
  {  // ngram_index_test.go:42-67 TestAutoComplete
    const Json& t = ref.at("autocomplete");
    auto dict = dictionary::NewInMemoryDictionary(collection);
    auto index = NewRAMBuilder(dict, DescriptionOf(t.at("description"), "index"))->Build();
    auto cands = index->Autocomplete(t.at("query").str, (int)t.at("limit").num);
    bool ok = cands.size() == t.at("expected_ids").size();
    for (size_t i = 0; ok && i < cands.size(); i++) ok = cands[i].Key == (uint32_t)t.at("expected_ids").at(i).num && cands[i].Score == 0;
    EXPECT(ok, "TestAutoComplete ids");
  }
  {  // example_test.go:14-72
    const Json& t = ref.at("example");
    Service service;
    auto dict = dictionary::NewInMemoryDictionary(collection);
    service.AddIndex("cars", dict, NewRAMBuilder(dict, DescriptionOf(t.at("description"), "cars")));
    auto res = service.Suggest("cars", NewSearchConfig(t.at("query").str, (int)t.at("topK").num, MetricOf(t.at("metric").str), t.at("similarity").num));
    bool ok = res.size() == t.at("expected_values").size();
    for (size_t i = 0; ok && i < res.size(); i++) ok = res[i].Value == t.at("expected_values").at(i).str;
    EXPECT(ok, "Example values");
  }

  const std::vector<IndexDescription> configs = ReadConfigs(golden + "/config.json");
  const Json& t = ref.at("service_cars");
  for (int pass = 0; pass < 2; pass++) {  // service_test.go:11-80, RAM driver and the reference's own DISC files
    Service service;
    IndexDescription d = configs[0];
    d.driver = pass == 0 ? RAMDriver : DiscDriver;
    service.AddIndexByDescription(d);
    EXPECT(service.GetDictionaries() == std::vector<std::string>{"cars"}, "GetDictionaries");
    for (size_t q = 0; q < t.at("queries").size(); q++) {
      auto res = service.Suggest("cars", NewSearchConfig(t.at("queries").at(q).str, (int)t.at("topK").num, MetricOf(t.at("metric").str),
                                                         t.at("similarity").num));
      const Json& exp = t.at("expected_values").at(q);
      bool ok = res.size() == exp.size();
      for (size_t i = 0; ok && i < res.size(); i++) ok = res[i].Value == exp.at(i).str;
      EXPECT(ok, ("service_test cars query " + t.at("queries").at(q).str + (pass ? " (DISC)" : " (RAM)")).c_str());
    }
    // the batch entry point returns the same rows as the single-query one
    auto rows = service.SuggestBatch("cars", Strings(t.at("queries")), (int)t.at("topK").num, MetricOf(t.at("metric").str), t.at("similarity").num);
    bool ok = rows.size() == t.at("queries").size();
    for (size_t q = 0; ok && q < rows.size(); q++) {
      auto one = service.Suggest("cars", NewSearchConfig(t.at("queries").at(q).str, (int)t.at("topK").num, MetricOf(t.at("metric").str),
                                                         t.at("similarity").num));
      ok = one.size() == rows[q].size();
      for (size_t i = 0; ok && i < one.size(); i++) ok = one[i].Value == rows[q][i].Value && one[i].Score == rows[q][i].Score;
    }
    EXPECT(ok, "SuggestBatch == Suggest");
  }

  // service_test.go:36-79: queries run while the index is swapped; a replaced index stays valid until released
  // (SG_STRESS=n repeats the scenario n times)
  for (int rep = 0, reps = getenv("SG_STRESS") ? atoi(getenv("SG_STRESS")) : 1; rep < reps; rep++) {
    Service service;
    IndexDescription d = configs[0];
    d.driver = RAMDriver;
    service.AddIndexByDescription(d);
    std::vector<std::thread> workers;
    std::vector<int> bad(4, 0);
    for (int w = 0; w < 4; w++)
      workers.emplace_back([&, w] {
        for (int it = 0; it < 8; it++)
          for (size_t q = 0; q < t.at("queries").size(); q++) {
            auto res = service.Suggest("cars", NewSearchConfig(t.at("queries").at(q).str, 5, metric::CosineMetric(), 0.7));
            const Json& exp = t.at("expected_values").at(q);
            bool ok = res.size() == exp.size();
            for (size_t i = 0; ok && i < res.size(); i++) ok = res[i].Value == exp.at(i).str;
            if (!ok && !bad[w]) {
              std::string got;
              for (auto& r : res) got += " [" + r.Value + " " + std::to_string(r.Score) + "]";
              fprintf(stderr, "worker %d pass %d query '%s': %zu results (want %zu):%s\n", w, it, t.at("queries").at(q).str.c_str(), res.size(), exp.size(), got.c_str());
            }
            bad[w] += !ok;
          }
      });
    for (int it = 0; it < 3; it++) service.AddIndexByDescription(d);   // reindex under load
    for (auto& th : workers) th.join();
    EXPECT(bad[0] + bad[1] + bad[2] + bad[3] == 0, "concurrent Suggest during AddIndex");
  }

  {  // the reference panics / blocks for ever on an empty clipped window (suggester.go:62): surfaced as an error
    Service service;
    auto dict = dictionary::NewInMemoryDictionary(collection);
    IndexDescription d = DescriptionOf(ref.at("example").at("description"), "cars");
    service.AddIndex("cars", dict, NewRAMBuilder(dict, d));
    const std::string long_query(60, 'x');   // MinY(0.9, 62) is past the largest cardinality in the index
    std::string msg = ErrorOf([&] { service.Suggest("cars", NewSearchConfig(long_query + "abcdefghijklmnopqrstuvwxyz", 5, metric::JaccardMetric(), 0.9)); });
    EXPECT(msg.rfind("reference behaviour", 0) == 0, "empty window is reported, not answered");
    EXPECT(service.Suggest("cars", NewSearchConfig("", 5, metric::JaccardMetric(), 0.5)).empty(), "empty query -> no candidates, no error");   // suggester.go:49-51
  }
}

// ---- language model + spellchecker ---------------------------------------------------------------------------------
static void TestSpell(const std::string& golden, bool device) {
  const Json ref = Json::Parse(dictionary::ReadFile(golden + "/reference_tests.json", "golden"));
  const Json& g = ref.at("lm");
  lm::Config config;
  config.NGramOrder = (uint8_t)g.at("order").num;
  config.OutputPath = golden + "/lm";
  config.Alphabet = {"english", "russian", "numbers", "-."};
  config.StartSymbol = g.at("startSymbol").str;
  config.EndSymbol = g.at("endSymbol").str;
  auto model = std::make_shared<lm::LanguageModel>(config);
  for (const Json& c : g.at("score_sentence").arr)              // language_model_test.go:52-70
    EXPECT(std::fabs(model->ScoreSentence(Strings(c.at(0))) - c.at(1).num) < g.at("tolerance").num, "ScoreSentence");
  EXPECT(model->GetWordID("sam") == 1 && model->GetWordID("nope") == lm::UnknownWordID && model->Find(1) == "sam", "indexer");
  if (!device) return;
  IndexDescription d;                                            // cmd/spellchecker/cmd/eval.go:16-23
  d.Name = "words"; d.NGramSize = 3; d.Wrap[0] = "^"; d.Wrap[1] = "$"; d.Pad = "$";
  d.Alphabet = {"english", "russian", "numbers", "$^'"};
  spellchecker::SpellChecker checker(model, d);
  EXPECT(checker.Predict("i am sa", 5, 0.3) == std::vector<std::string>{"sam"}, "Predict completes the word");
  EXPECT((checker.Predict("<s> i am", 5, 0.3) == std::vector<std::string>{"am", "sam", "ham"}), "Predict tops up with the fuzzy search");
  EXPECT(checker.Predict("gren egs", 5, 0.3) == std::vector<std::string>{"eggs"}, "Predict corrects a typo");
  EXPECT(checker.Predict("", 5, 0.3).empty(), "empty query");
}

int main(int argc, char** argv) {
  bool cpu_only = false;
  std::string golden;
  for (int i = 1; i < argc; i++) {
    if (std::string(argv[i]) == "--cpu") cpu_only = true;
    else golden = argv[i];
  }
  if (golden.empty()) {
    std::fprintf(stderr, "usage: service_test [--cpu] <tests/golden>\n");
    return 2;
  }
  try {
    TestHost(golden);
    TestSpell(golden, !cpu_only);
    if (!cpu_only) TestDevice(golden);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FAIL uncaught: %s\n", e.what());
    return 1;
  }
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
