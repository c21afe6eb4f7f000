// sg_internal.h — structures shared by the host index builder (host_index.cpp) and the
// HIP engine (engine.hip).  Not part of the public ABI (include/suggest_hip.h).
#pragma once

#include <atomic>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/suggest_hip.h"

namespace sg {

constexpr uint32_t kNoTerm = 0xFFFFFFFFu;
constexpr uint32_t kRuneError = 0xFFFD;

// Symbol table: every rune that can appear in a normalised term (alphabet runes + the runes of
// the pad string) gets an id 1..255; a term is at most 8 symbols and is packed little-endian
// into a u64 (byte i = symbol i, 0 = end).  Key equality == reference term-string equality.
struct Symbols {
  uint8_t ascii_sym[128];    // symbol id of an ASCII rune, 0 = not a symbol
  uint8_t ascii_alpha[128];  // 1 = Alphabet.Has(rune)
  std::vector<uint32_t> na_rune;  // non-ASCII symbol runes, ascending
  std::vector<uint8_t> na_sym, na_alpha;
  uint8_t pad_sym[8];
  uint32_t n_pad = 0;
  std::vector<uint32_t> sym_rune;  // id -> rune (index 0 unused)
};

struct TermSlot {  // open-addressing hash table slot (16 B)
  uint64_t key;
  uint32_t term;  // kNoTerm = empty
  uint32_t pad;
};

struct DupEntry {  // a (term, segment, doc) whose doc repeats the term
  uint32_t term, segment, doc, mult;
};

struct HostIndex {
  // description
  uint32_t q = 3;
  std::vector<uint32_t> wrap0, wrap1;  // runes
  std::string wrap0_s, wrap1_s, pad_s;
  std::vector<std::string> alphabet_spec;
  Symbols sym;

  uint64_t n_docs = 0;
  uint32_t n_segments = 0;  // == header.Indices of the reference (max cardinality + 1)
  uint32_t min_segments = 0;  // builders: at least this many segments (docID-sharded indexes agree on the global number)
  std::vector<uint64_t> term_key;                    // termID -> key
  std::unordered_map<uint64_t, uint32_t> term_of;    // key -> termID
  std::vector<uint32_t> seg_off;                     // [n_terms*(S+1)] chunk (16 B) offsets, term-major
  std::vector<uint32_t> list_len;                    // [n_terms*S] stored (de-duplicated) lengths
  std::vector<uint32_t> postings;                    // padded chunks, ascending docIDs per (term, segment)
  std::vector<DupEntry> dups;                        // docs with repeated terms, ascending (term, segment, doc)
  std::vector<TermSlot> slots;                       // hash table, size = power of two
  uint64_t n_lists = 0, n_postings = 0, n_postings_raw = 0;
};

// host tokenizer (NewSuggestTokenizer / NewAutocompleteTokenizer, pkg/suggest/tokenizer.go:9-34)
// -> packed term keys, repeats included, first-occurrence order.  Returns false if a term does
// not fit the 8-symbol key.
bool tokenize_keys(const HostIndex& ix, const uint8_t* s, size_t n, bool autocomplete, std::vector<uint64_t>& out);

int build_host_index(const uint8_t* utf8, const uint64_t* offs, uint32_t n_docs, const sg_desc* desc, HostIndex& ix,
                     std::string& err);

int init_description(const sg_desc* desc, HostIndex& ix, std::string& err);
bool term_string_key(const HostIndex& ix, const std::string& term, uint64_t* key);
void build_term_table(HostIndex& ix);
// loads reference-built <name>.hd / <name>.dl (ref_index_reader.cpp)
int load_reference_index(const char* hd_path, const char* dl_path, const sg_desc* desc, HostIndex& ix, std::string& err);

// metric maths in IEEE double, evaluation order of pkg/metric/*.go (host copies; engine.hip has
// the device twins)
int metric_min_y(int m, double alpha, int size);
int metric_max_y(int m, double alpha, int size);
int metric_threshold(int m, double alpha, int a, int b);

uint64_t mix64(uint64_t k);

uint32_t host_next_rune(const uint8_t* s, size_t n, size_t* adv);   // Go range decoding (invalid byte -> U+FFFD, width 1)
uint32_t host_lower_rune(uint32_t r);                                // unicode.ToLower (simple mappings)
uint32_t host_utf8_width(uint32_t r);
bool host_alphabet_has(const std::vector<std::string>& spec, uint32_t r);   // alphabet.CreateAlphabet(spec).Has(r)

// ---- language model of the spellchecker caller (lm.cpp) ----
constexpr uint32_t kUnknownWord = 0xFFFFFFFFu;      // pkg/lm/indexer.go:16
constexpr uint32_t kNoContext = 0xFFFFFFFDu;        // InvalidContextOffset, pkg/lm/ngram_vector.go:25-27

struct LmLevel {   // n-grams of one order, sorted by (parent, word); the index of an entry is its "context offset"
  std::vector<uint32_t> word, count;
  std::vector<uint32_t> child_begin;   // [n_parents + 2]: entries whose parent is p are child_begin[p] .. child_begin[p + 1];
                                       // the last bucket (p = n_parents) holds entries without a parent (kNoContext)
  uint64_t total = 0;                  // CorpusCount
};

struct HostLM {
  std::vector<std::string> words;      // id order (lines of 1-gm)
  std::unordered_map<std::string, uint32_t> id_of;
  std::vector<LmLevel> level;          // order 1 .. N
  uint32_t order = 0, start_symbol = kUnknownWord, end_symbol = kUnknownWord;
  std::vector<std::string> alphabet;
};

struct LmNext {    // NGramModel.Next: the continuations of a context = one bucket of one level
  int status = 0;  // 0 scorer, 1 nil scorer, 2 error (pkg/lm/ngram_model.go:64-98)
  uint32_t level = 0, from = 0, to = 0;
  uint32_t context_count = 0;          // count of the context itself (the denominator of ScoreNext)
};

int lm_load_google(const char* dir, uint32_t order, const char* start_symbol, const char* end_symbol, const std::vector<std::string>& alphabet,
                   int id_order, HostLM& lm, std::string& err);
int lm_load_binary(const char* lm_path, const char* cdb_path, const char* start_symbol, const char* end_symbol,
                   const std::vector<std::string>& alphabet, HostLM& lm, std::string& err);
void lm_level_packed(const HostLM& lm, uint32_t level, std::vector<uint64_t>& containers, std::vector<uint64_t>& values, uint32_t* total);
int lm_build_google_files(const uint8_t* text, size_t n, uint32_t order, const char* start_symbol, const char* end_symbol,
                          const std::vector<std::string>& alphabet, const std::vector<std::string>& separators, const char* out_dir,
                          std::string& err);
uint32_t lm_word_id(const HostLM& lm, const std::string& token);
double lm_model_score(const HostLM& lm, const uint32_t* ids, size_t n);
double lm_score_word_ids(const HostLM& lm, const uint32_t* ids, size_t n);
LmNext lm_model_next(const HostLM& lm, const uint32_t* ids, size_t n);
LmNext lm_next(const HostLM& lm, const uint32_t* ids, size_t n);
uint32_t lm_next_count(const HostLM& lm, const LmNext& nx, uint32_t word);
double lm_next_score(const HostLM& lm, const LmNext& nx, uint32_t word);
void lm_tokenize(const HostLM& lm, const uint8_t* text, size_t n, std::vector<std::string>& out);
void set_error(const std::string& msg);

}  // namespace sg
