/* suggest_hip.h — C ABI of libsuggest_hip.so, the MI355X (gfx950) engine behind suggest-go's
 * n-gram fuzzy-search path.
 *
 * The reference has no FFI; its extension seam is the pair of Go interfaces
 *   suggest.Builder{Build() (NGramIndex, error)}          pkg/suggest/ngram_index_builder.go:14-17
 *   suggest.NGramIndex = Suggester + Autocomplete          pkg/suggest/ngram_index.go:7-10
 * consumed by Service.AddIndex / Suggest / Autocomplete    pkg/suggest/service.go:78-173.
 * A cgo shim implementing those two interfaces binds exactly the entry points below
 * (see INTEGRATION.md for the shim).  Every function cites the reference code it replaces.
 *
 * Conventions: return 0 on success, a negative SG_E_* code on failure (sg_last_error() gives a
 * thread-local message); the caller owns every host buffer; the library owns device memory;
 * no callbacks; every entry point may be called concurrently from any thread on the same handle
 * (the handle is immutable after sg_index_upload and reference counted).
 */
#ifndef SUGGEST_HIP_H
#define SUGGEST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sg_index sg_index;

/* IndexDescription — pkg/suggest/config.go:25-35 (the fields the tokenizer and index use) */
typedef struct sg_desc {
  uint32_t ngram_size;          /* NGramSize, 1..8 (pkg/analysis/ngram_tokenizer.go:3) */
  const char* wrap_start;       /* Wrap[0], UTF-8, NUL terminated */
  const char* wrap_end;         /* Wrap[1] */
  const char* pad;              /* Pad */
  const char* const* alphabet;  /* Alphabet: "english" | "russian" | "numbers" | literal rune set
                                   (pkg/alphabet/alphabet.go:23-36) */
  uint32_t n_alphabet;
} sg_desc;

/* metric.Metric implementations — pkg/metric/{jaccard,cosine,dice,exact,overlap}.go */
enum sg_metric { SG_JACCARD = 0, SG_COSINE = 1, SG_DICE = 2, SG_EXACT = 3, SG_OVERLAP = 4 };

enum sg_error {
  SG_OK = 0,
  SG_E_INVALID = -1,      /* bad argument (k == 0, similarity outside (0,1], q outside 1..8 ...) —
                             the checks of NewSearchConfig, pkg/suggest/search.go:18-35 */
  SG_E_UNSUPPORTED = -2,  /* description outside what the packed 64-bit term key can hold */
  SG_E_NOT_UPLOADED = -3,
  SG_E_HIP = -4,          /* HIP runtime error; message has the hipError string */
  SG_E_NOMEM = -5
};

/* Values stored in out_counts[i] instead of a count, for queries the reference itself does not
 * answer (pkg/suggest/suggester.go:62: the clipped window [bMin,bMax] is empty) and for queries
 * beyond the device path's limits.  Rows of such queries are left zeroed. */
#define SG_COUNT_REF_PANIC 0xFFFFFFFFu     /* reference: make(chan, negative) panics */
#define SG_COUNT_REF_DEADLOCK 0xFFFFFFFEu  /* reference: capacity-0 channel, send blocks forever */
#define SG_COUNT_TOO_LONG 0xFFFFFFFDu      /* more than SG_MAX_QUERY_TERMS n-grams (queries above 128 n-grams take a slower kernel) */
#define SG_COUNT_LM_ERROR 0xFFFFFFFCu       /* LanguageModel.Next returned an error (sg_spell_predict_batch) */
#define SG_MAX_QUERY_TERMS 65536u
#define SG_MAX_TOPK 65536u

/* suggest.Index + Writer.AddDocument/Commit + Reader.Read
 * (pkg/suggest/indexer.go:14-45, pkg/index/indexer_writer.go:66-145, index_reader.go:29-120):
 * tokenises docs[i] = utf8[offs[i]..offs[i+1]) (docID = i, dictionary order) and builds the
 * cardinality-segmented inverted index as term-major CSR in host memory. */
int sg_index_build(const uint8_t* utf8, const uint64_t* offs, uint32_t n_docs, const sg_desc* desc, sg_index** out);

/* Same index, built on the GPU `device`: documents are tokenised by the kernel-side tokenizer, terms
 * interned in a device hash table, postings radix-sorted and laid out as the same CSR — array for array identical to
 * sg_index_build's (sg_index_digest).  Fails with SG_E_UNSUPPORTED when a document has more than 65 536 bytes or the
 * dictionary 2^26 documents or more (use sg_index_build).  The index still has to be uploaded with sg_index_upload. */
int sg_index_build_device(const uint8_t* utf8, const uint64_t* offs, uint32_t n_docs, const sg_desc* desc, int device,
                          sg_index** out);

/* Either builder (device < 0: host) with at least `min_segments` cardinality segments: the shards of a dictionary split by
 * docID range (suggest_amd/distributed.py) agree on the global number, so that every shard clips the
 * window [MinY, MaxY] at the same segment (suggester.go:57-59) as the unsharded index would. */
int sg_index_build_ex(const uint8_t* utf8, const uint64_t* offs, uint32_t n_docs, const sg_desc* desc, uint32_t min_segments,
                      int device, sg_index** out);

/* NewFSBuilder + Reader.Read (pkg/suggest/ngram_index_builder.go:44-83, pkg/index/index_reader.go:29-120): loads an
 * index the reference itself built — <name>.hd (gob header) and <name>.dl (VB / skip-VB / roaring posting lists,
 * pkg/index/codec.go:39-51) — into the same CSR.  `desc` must be the IndexDescription the files were built with. */
int sg_index_load_reference(const char* hd_path, const char* dl_path, const sg_desc* desc, sg_index** out);

/* Copies the CSR index into the HBM of `device` (one replica per GPU; per-process).  The first upload makes the primary
 * replica (the one sg_suggest_batch / sg_autocomplete_batch run on); uploading to a device that already holds a replica is
 * a no-op.  A device-built index (sg_index_build_device) whose first upload goes to the building device keeps the posting
 * store where the build left it (no D2H + H2D round trip).  Safe to call concurrently.  SG_E_UNSUPPORTED: the packed
 * posting store numbers documents below 2^29 (shard larger dictionaries by docID range: sg_index_build_ex). */
int sg_index_upload(sg_index* index, int device);

/* Multi-GPU (BASELINE north_star: "query batches shard naturally across the 8 GPUs of one node"): ONE host
 * build, one replica per listed device.  A device listed j times ends up with j replicas (only useful to exercise the
 * multi-replica paths on a one-GPU box).  Replicas already present are kept. */
int sg_index_replicate(sg_index* index, const int* devices, uint32_t n_devices);
/* Devices of the replicas in upload order (primary first): fills up to cap entries, returns their number. */
uint32_t sg_index_replicas(sg_index* index, int* out_devices, uint32_t cap);

/* nGramSuggester.Suggest for a batch of queries — pkg/suggest/suggester.go:46-131 with
 * newFuzzyCollectorManager(k) (collector.go:143-149): tokenise, window [MinY,MaxY], per segment
 * T-occurrence merge (pkg/index/searcher.go:28-78, pkg/merger/cp_merge.go:19-120 result set),
 * score 1-Distance (scorer.go:29-31), top-k by (score desc, docID asc) (collector.go:20-26,
 * topk.go:82-147).  Row i of out_ids/out_scores (k entries each) holds the candidates of query i
 * best first; out_counts[i] = how many (or an SG_COUNT_* flag).  Host buffers; synchronous. */
int sg_suggest_batch(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q, int metric,
                     double similarity, uint32_t k, uint32_t* out_ids, double* out_scores, uint32_t* out_counts);

/* Same, on buffers already resident in HBM; enqueued on `stream` (a hipStream_t of that device, NULL = the legacy default
 * stream) and asynchronous w.r.t. the host.  With several replicas the one on the device that owns d_q_offs runs it. */
int sg_suggest_batch_device(sg_index* index, const void* d_q_utf8, const void* d_q_offs, uint32_t n_q, int metric,
                            double similarity, uint32_t k, void* d_out_ids, void* d_out_scores, void* d_out_counts,
                            void* stream);

/* nGramAutocomplete.Autocomplete with newFirstKCollectorManager(limit) —
 * pkg/suggest/autocomplete.go:40-77, collector.go:48-115: the `limit` smallest docIDs whose
 * n-gram set contains every n-gram of the (tail-unwrapped) query. */
int sg_autocomplete_batch(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q,
                          uint32_t limit, uint32_t* out_ids, uint32_t* out_counts);
int sg_autocomplete_batch_device(sg_index* index, const void* d_q_utf8, const void* d_q_offs, uint32_t n_q,
                                 uint32_t limit, void* d_out_ids, void* d_out_counts, void* stream);

/* sg_suggest_batch / sg_autocomplete_batch over every replica of the index: the batch is cut into contiguous slices, one per
 * replica, every slice is in flight (copy in, launch, copy out on that device's stream) before the first is awaited, and
 * rows land in caller order.  No collective: queries are independent and the index is read-only.  With
 * one replica these are the plain calls. */
int sg_suggest_batch_multi(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q, int metric,
                           double similarity, uint32_t k, uint32_t* out_ids, double* out_scores, uint32_t* out_counts);
int sg_autocomplete_batch_multi(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q,
                                uint32_t limit, uint32_t* out_ids, uint32_t* out_counts);

/* Suggester.Suggest / Autocomplete.Autocomplete as the reference calls them: ONE query per call, from many goroutines at
 * once (pkg/suggest/suggester.go:46, autocomplete.go:40, service_test.go:36-79).  Blocking; concurrent callers are
 * coalesced: the request is queued and dispatcher threads (SG_COALESCE_LANES per replica, default 1) run whatever is
 * pending with the same (metric, similarity, k) as one launch — no timer, an idle engine serves a lone request at once.
 * out_ids / out_scores hold k entries, *out_count the number written (or an SG_COUNT_* flag, nothing written). */
int sg_suggest_one(sg_index* index, const uint8_t* q_utf8, uint32_t len, int metric, double similarity, uint32_t k,
                   uint32_t* out_ids, double* out_scores, uint32_t* out_count);
int sg_autocomplete_one(sg_index* index, const uint8_t* q_utf8, uint32_t len, uint32_t limit, uint32_t* out_ids,
                        uint32_t* out_count);
/* Autocomplete restricted to documents with docID >= first_doc (the `limit` smallest of those).  The reference hands EVERY
 * match to the caller's collector (pkg/suggest/autocomplete.go:40-77; pkg/spellchecker/collector.go:61-79 ranks them all):
 * a binding that must do the same pages through them — first_doc = 0, then the last docID of a page + 1, until a page comes
 * back with fewer than `limit` entries. */
int sg_autocomplete_one_from(sg_index* index, const uint8_t* q_utf8, uint32_t len, uint32_t first_doc, uint32_t limit,
                             uint32_t* out_ids, uint32_t* out_count);
int sg_autocomplete_batch_from(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q, uint32_t first_doc,
                               uint32_t limit, uint32_t* out_ids, uint32_t* out_counts);

/* ---- the Suggester seam for ANY Metric and ANY collector (pkg/suggest/suggester.go:17-20) -------------------------
 * metric.Metric is an interface (pkg/metric/metric.go:7-16: MinY, MaxY, Threshold, Distance); the five implementations
 * above have device twins, any other one reaches the engine as tables its binding fills by calling the four methods:
 *   min_y[a], max_y[a]                      a = 0 .. a_max          Metric.MinY / MaxY(alpha, a)
 *   threshold[a * S + b]                    b = 0 .. S - 1          Metric.Threshold(alpha, a, b)
 *   score[(a * S + b) * (a_max + 1) + o]    o = 0 .. a_max          1 - Metric.Distance(o, a, b)   (scorer.go:29-31)
 * S = sg_stats.n_segments of `index`.  A query with more than a_max n-grams comes back SG_COUNT_TOO_LONG.  The tables are
 * copied to the HBM of the index's primary replica and live until released (they retain the index). */
typedef struct sg_metric_tables sg_metric_tables;
int sg_metric_tables_create(sg_index* index, uint32_t a_max, const int32_t* min_y, const int32_t* max_y,
                            const int32_t* threshold, const double* score, sg_metric_tables** out);
void sg_metric_tables_retain(sg_metric_tables* tables);   /* one reference per user: a cache entry, a call in flight ([r5]) */
void sg_metric_tables_release(sg_metric_tables* tables);
/* sg_suggest_batch under a tabulated metric: same rows, same order. */
int sg_suggest_batch_tables(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q,
                            const sg_metric_tables* tables, uint32_t k, uint32_t* out_ids, double* out_scores,
                            uint32_t* out_counts);
/* nGramSuggester.Suggest hands EVERY document whose overlap reaches the segment's threshold to the caller's collector
 * (suggester.go:78-99; the fuzzy top-k manager is only the usual one).  For a binding that must serve another collector:
 * row i = the `limit` smallest docIDs >= first_doc among query i's candidates over all admissible segments, ascending,
 * with their scores and (out_aux, may be NULL) segment << 16 | overlap — enough to rebuild
 * merger.MergeCandidate{Position, Overlap} and the segment's scorer.
 * Paging: a document that repeats a term has SEVERAL entries (same docID, one per secondary candidate), and a full page may
 * end inside such a run.  Resume with first_doc = the LAST docID of the full page (not + 1) and drop, from the head of the next
 * page, as many entries of that docID as the pages so far have already delivered; if a whole page consists of one docID, ask
 * again with twice the limit.  (first_doc = last docID + 1 loses the rest of the run.)  A page that comes back short is the last.
 * `tables` non-NULL replaces (metric, similarity). */
int sg_suggest_batch_from(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q, int metric,
                          double similarity, const sg_metric_tables* tables, uint32_t first_doc, uint32_t limit,
                          uint32_t* out_ids, double* out_scores, uint32_t* out_aux, uint32_t* out_counts);

/* ---- asynchronous host-buffer calls -----------------------------------------------------------------
 * Service.Suggest (pkg/suggest/service.go:105-139) is called with host strings and returns host rows; a Go host behind
 * this ABI therefore lives on host buffers.  The synchronous calls above copy in, run and copy out one after the other;
 * these let the copies of one batch run beside the kernel of another: submit returns once everything is enqueued (copy
 * in -> search launch -> copy out, on three streams of the primary replica tied by events; all search launches of the
 * replica stay serialised on one stream), sg_ticket_wait blocks until the rows are in the caller's buffers and frees the
 * ticket.  Up to 8 tickets may be in flight per replica; two are enough to hide PCIe behind the kernel.  Buffers from
 * sg_host_alloc (pinned) are read / written by the DMA engine directly; other buffers are staged through pinned memory
 * (a memcpy at submit, one at wait).  ONLY sg_host_alloc blocks count as pinned, and only when the whole range lies inside one:
 * memory the caller pinned itself (hipHostMalloc, torch pin_memory) is correct but staged like pageable memory — the library
 * keeps a registry of its own blocks instead of asking the driver about every pointer (tens of microseconds per call).  The caller's buffers must stay valid and untouched until the wait returns.  Submit
 * and wait may be called from different threads. */
typedef struct sg_ticket sg_ticket;
int sg_host_alloc(uint64_t bytes, void** out);
void sg_host_free(void* p);
int sg_suggest_submit(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q, int metric,
                      double similarity, uint32_t k, uint32_t* out_ids, double* out_scores, uint32_t* out_counts,
                      sg_ticket** out_ticket);
/* ... on replica number `replica` of the index (0 = the primary, in the order of sg_index_replicas): a slice of a batch per GPU
 * from ONE host thread, no staging copy when the buffers are pinned (the north star's "query batches shard naturally across
 * the 8 GPUs of one node"; sg_suggest_batch_multi is the synchronous form over pageable buffers). */
int sg_suggest_submit_on(sg_index* index, uint32_t replica, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q,
                         int metric, double similarity, uint32_t k, uint32_t* out_ids, double* out_scores,
                         uint32_t* out_counts, sg_ticket** out_ticket);
int sg_autocomplete_submit(sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q, uint32_t first_doc,
                           uint32_t limit, uint32_t* out_ids, uint32_t* out_counts, sg_ticket** out_ticket);
int sg_ticket_wait(sg_ticket* ticket);

/* Reference counting: the Go shim retains while a query is in flight and releases from a
 * finalizer, mirroring the reference's mmap release (pkg/index/index_reader.go:49-51). */
void sg_index_retain(sg_index* index);
void sg_index_release(sg_index* index);

const char* sg_last_error(void);

/* ---- the spellchecker caller of the path --------------------------------------
 * pkg/lm: n-gram language model with stupid backoff; pkg/spellchecker: SpellChecker.Predict. */
typedef struct sg_lm sg_lm;

/* NewGoogleNGramReader(order, indexer, dir).Read + NewLanguageModel (pkg/lm/ngram_reader.go:38-98, indexer.go:86-114,
 * language_model.go:24-49): <dir>/1-gm .. <dir>/<order>-gm hold "w1 .. wk\tcount" lines; word ids are the line numbers
 * of 1-gm; `alphabet` is the words alphabet of the lm config (pkg/lm/config.go:25-27) used by the query tokenizer. */
int sg_lm_load_google(const char* dir, uint32_t order, const char* start_symbol, const char* end_symbol,
                      const char* const* alphabet, uint32_t n_alphabet, sg_lm** out);
/* The same files with the word ids the production path gives them: id_order 1 = buildDictionary (pkg/lm/binary.go:101-199),
 * words ordered by (count desc, word asc) — what `lm build-lm` stores and RetrieveLMFromBinary serves; id_order 0 = the line
 * order of 1-gm (buildIndexerWithInMemoryDictionary, indexer.go:88-114, used by the reference's tests).  The ids matter:
 * SpellChecker.Predict breaks ties by docID = word id. */
int sg_lm_load_google_ex(const char* dir, uint32_t order, const char* start_symbol, const char* end_symbol,
                         const char* const* alphabet, uint32_t n_alphabet, int id_order, sg_lm** out);
/* RetrieveLMFromBinary (pkg/lm/binary.go:59-98) — what BuildSpellChecker loads (internal/spellchecker/dep/spellchecker.go:13-53):
 * <name>.lm (nGramModel.Load ngram_model.go:123-160: "0.0.2", order, per level packedArray.Load packed_array.go:118-160)
 * and the dictionary <name>.cdb (word of every id; pkg/dictionary/cdb_dictionary.go).  The MPH table behind the model in the
 * file is not read: lookups here are exact. */
int sg_lm_load_binary(const char* lm_path, const char* cdb_path, const char* start_symbol, const char* end_symbol,
                      const char* const* alphabet, uint32_t n_alphabet, sg_lm** out);
/* One level of the model in the reference's packed form (packed_array.go:12-16: containers context<<32|from, values
 * word<<32|count, total) — what packedArray.Store writes.  Introspection (tests compare with the bytes of a .lm file). */
int sg_lm_level(const sg_lm* lm, uint32_t level, uint64_t* containers, uint32_t cap_containers, uint32_t* n_containers,
                uint64_t* values, uint32_t cap_values, uint32_t* n_values, uint32_t* total);
uint32_t sg_lm_order(const sg_lm* lm);
/* NGramBuilder.Build over NewSentenceRetriever + googleNGramFormatWriter.Write (pkg/lm/ngram_builder.go:16-64,
 * sentence_retriever.go:17-81, ngram_writer.go:32-76) — what `lm build-lm` does to a corpus: sentences are cut at the
 * runes of `separators`, tokenised, wrapped in start/end symbols and their k-grams (k = 1..order) counted into
 * <out_dir>/<k>-gm.  Lines come in order of first appearance (the reference's order is Go-map random). */
int sg_lm_build_google(const uint8_t* text, uint64_t len, uint32_t order, const char* start_symbol, const char* end_symbol,
                       const char* const* alphabet, uint32_t n_alphabet, const char* const* separators, uint32_t n_separators,
                       const char* out_dir);
void sg_lm_retain(sg_lm* lm);
void sg_lm_release(sg_lm* lm);
uint32_t sg_lm_num_words(const sg_lm* lm);
int sg_lm_word(const sg_lm* lm, uint32_t id, char* out, uint32_t cap);             /* Indexer.Find; returns the length */
uint32_t sg_lm_word_id(const sg_lm* lm, const uint8_t* word, uint32_t len);          /* Indexer.Get; 0xFFFFFFFF = unknown */
double sg_lm_score(const sg_lm* lm, const uint32_t* ids, uint32_t n);                /* NGramModel.Score, ngram_model.go:44-62 */
double sg_lm_score_word_ids(const sg_lm* lm, const uint32_t* ids, uint32_t n);       /* LanguageModel.ScoreWordIDs, language_model.go:78-86 */
/* Next(context).ScoreNext(word): returns 0 (score written), 1 (nil scorer) or 2 (error); model_level != 0 calls
 * NGramModel.Next (ngram_model.go:64-98) instead of LanguageModel.Next (language_model.go:100-112). */
int sg_lm_next_score(const sg_lm* lm, const uint32_t* context, uint32_t n, uint32_t word, int model_level, double* score);
/* lm.NewTokenizer(alphabet).Tokenize (pkg/lm/tokenizer.go:26-31): tokens joined by '\n' into out; returns their number */
int sg_lm_tokenize(const sg_lm* lm, const uint8_t* text, uint32_t len, char* out, uint32_t cap);

/* The fuzzy index of the spellchecker: built over the model's vocabulary (docID = word id) and uploaded —
 * internal/spellchecker/dep/spellchecker.go:33-45 (NewRAMBuilder over the LM dictionary). */
int sg_spell_index_build(const sg_lm* lm, const sg_desc* desc, int device, sg_index** out);

/* SpellChecker.Predict (pkg/spellchecker/spellchecker.go:40-92) for a batch of queries: the last word of a query is
 * completed (Autocomplete with the LM collector, collector.go / scorer.go: top_k by ScoreNext of the preceding words),
 * topped up by the Cosine fuzzy search when fewer than top_k complete it, re-ranked by ScoreNext (stable) and cut to
 * top_k + 1 entries (sic, :87-89).  Row i of out_ids (top_k + 1 word ids) holds the prediction of query i,
 * out_counts[i] how many (or SG_COUNT_REF_PANIC / _DEADLOCK / _TOO_LONG / _LM_ERROR).  `index` must come from
 * sg_spell_index_build (or be any uploaded index over the model's vocabulary in id order). */
int sg_spell_predict_batch(sg_index* index, sg_lm* lm, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q,
                           uint32_t top_k, double similarity, uint32_t* out_ids, uint32_t* out_counts);
/* The same with every buffer in the HBM of the GPU that holds `index`'s primary replica, asynchronous on `stream`: d_q =
 * the queries (q_bytes bytes), d_offs = n_q + 1 uint64 offsets starting at 0, d_out_ids = [n_q][top_k + 1] uint32,
 * d_out_counts = [n_q] uint32.  The word tokeniser, the word ids and LanguageModel.Next's wrap / trim rules run on the
 * device too (spellchecker.go:40-64,94-107; language_model.go:100-112): nothing of a query touches the host. */
int sg_spell_predict_batch_device(sg_index* index, sg_lm* lm, const void* d_q, const void* d_offs, uint32_t n_q, uint64_t q_bytes,
                                  uint32_t top_k, double similarity, void* d_out_ids, void* d_out_counts, void* stream);

/* ---- introspection (tests, bench.py) ---------------------------------------------------- */
typedef struct sg_stats {
  uint64_t n_docs, n_segments, n_terms, n_lists, n_postings /* (term,doc) pairs stored */,
      n_postings_raw /* incl. repeated terms of a doc, as the reference stores them */, posting_bytes /* padded */,
      table_bytes, device_bytes;
} sg_stats;
int sg_index_stats(const sg_index* index, sg_stats* out);

/* Sampled counters the fuzzy launches of the primary replica leave (cumulative; [0..2] wrap at 2^32, [3] is 64 bits wide): out[0] sampled queries
 * whose top-k ended full, [1] sampled queries, [2] their results, [3] the 16-byte chunks of the packed posting store they
 * streamed — what bench.py's roofline.model_bytes is made of.  Synchronises the device. */
int sg_index_launch_stats(sg_index* index, uint64_t out[4]);

/* [r5] Queries the three-launch pipeline of ordinary fuzzy batches (plan -> stream -> verify) handed to the fused
 * kernel, cumulative (wrapping at 2^32): out[0] the plan could not express them (more than 64 n-grams, a window of more than 63
 * segments, a segment that needs docID-range passes, more groups / lists than a slot record holds), [1] their candidates
 * overflowed the slots, [2] a matching document repeats a term (the secondary entries of SURVEY.md §A.3); [3] the queries
 * of all launches that took the pipeline so far (not wrapping).
 * Results never depend on which path answered.  Synchronises the device. */
int sg_index_pipe_stats(sg_index* index, uint64_t out[4]);

/* [r6] The pipeline's sampled volumes, cumulative (wrapping at 2^32): out[0] sampled queries the plan expressed (one in 256 of a batch
 * above 1 024 queries, else every one), [1] their groups of cardinality segments, [2] streamed lists, [3] rows of 64 lanes,
 * [4] candidates pushed for the sampled queries that reached the verify launch, [5] 16-byte chunks of the packed posting store,
 * [6] 1 when the stream launch uses 8-byte sub-row descriptors (stores of 2^26 chunks and more), [7] the stream workgroup of the replica's latest
 * launch: 0 / 1 / 2 = 2 / 4 / 8 wavefronts on 2^11 / 2^12 / 2^13 counters (chosen per launch from the index's expected query volume
 * and the metric's threshold), 3 = fixed by SG_PIPE_NW / SG_PIPE_LOG2_CNT / SG_PIPE_DT_BYTES.  Introspection for bench.py and the
 * tests; no reference counterpart.  Synchronises the device. */
int sg_index_pipe_volumes(sg_index* index, uint64_t out[8]);

/* The forward index (doc -> distinct terms) of the primary replica, copied back for documents
 * first .. first+n-1: out_card[i] = cardinality, out_n[i] = number of distinct terms, out_keys[i*cap ..] their term keys. */
int sg_index_forward(sg_index* index, uint32_t first, uint32_t n, uint32_t cap, uint32_t* out_card, uint32_t* out_n,
                     uint64_t* out_keys);

/* Test hook: the permutation the device's restatement of Go 1.14 sort.Sort (it orders equal-length posting lists in
 * cpMerge.Merge, cp_merge.go:24) gives n <= 128 keys; out[i] = original index of the element that ends at position i. */
int sg_debug_pairsort(int device, const uint32_t* keys, uint32_t n, uint32_t* out);
/* [r5] The auto-tuner's choices (no GPU needed): out = {log2 of the LDS counter words, filter level, 1 if the plan -> stream ->
 * verify pipeline pays} for an index whose queries are expected to stream est_query_chunks 16-byte chunks of u32 postings and
 * whose longest term holds max_term_chunks; and the same for a built index together with its two statistics. */
int sg_debug_tune_choice(double est_query_chunks, double max_term_chunks, int32_t out[6]);
/* [r6] Test hook: the stream workgroup a launch under `metric` / `similarity` starts from on an index with these statistics
 * (expected query volume in 16-byte chunks, n-grams per document, SG_T_FLOOR): *out_shape = 0 / 1 / 2 for 2 / 4 / 8 wavefronts on
 * 2^11 / 2^12 / 2^13 counters.  No GPU needed; no reference counterpart. */
int sg_debug_pipe_shape(double est_query_chunks, double terms_per_doc, int32_t t_floor, int32_t metric, double similarity, int32_t* out_shape);
int sg_debug_tune_index(sg_index* index, double out_stats[2], int32_t out[6]);
/* [r6] Test hook: out[0] = the device ordinal of replica number `replica`, out[1..7] = the device its posting store, seg_off, orig_of,
 * forward-index records and terms, term table and counter block are resident on (-1: null).  All must equal out[0]. */
int sg_debug_replica_devices(sg_index* index, uint32_t replica, int32_t out[8]);

/* Sets a tuning knob of the index (names and ranges of the SG_* environment variables in DESIGN.md: SG_LOG2_CNT, SG_T_FLOOR,
 * SG_FILTER_LEVEL, SG_TIGHTEN, SG_ROOMY, SG_ORDER, SG_PRETOK, SG_SPLIT_CHUNKS, SG_PARTS_CNT_BONUS).  Results never depend on the knobs; for parameter sweeps. */
int sg_index_tune(sg_index* index, const char* knob, int value);

/* Tokens of `text` as the index sees them, one packed 64-bit term key each (DESIGN.md §Term keys);
 * returns the token count (may exceed cap; only cap are written).  NewSuggestTokenizer /
 * NewAutocompleteTokenizer, pkg/suggest/tokenizer.go:9-34. */
int sg_tokenize(const sg_index* index, const uint8_t* text, uint32_t len, int autocomplete, uint64_t* out_keys,
                uint32_t cap);
/* Renders a term key back to the reference's term string (UTF-8); returns its byte length. */
int sg_term_string(const sg_index* index, uint64_t key, char* out, uint32_t cap);
/* Posting list of (segment, term key) from the host CSR: returns the stored (de-duplicated)
 * length, writes up to cap docIDs; *raw_len = length incl. a doc's repeated terms. -1 if absent. */
int64_t sg_index_list(const sg_index* index, uint32_t segment, uint64_t key, uint32_t* out, uint64_t cap,
                      uint64_t* raw_len);
/* Enumerates the non-empty (segment, term) lists: fills up to cap entries, returns the total. */
uint64_t sg_index_lists(const sg_index* index, uint32_t* out_segments, uint64_t* out_keys, uint64_t cap);

/* 64-bit digests of the host CSR: postings, seg_off, list lengths, term keys (+ repeated-term table). */
int sg_index_digest(const sg_index* index, uint64_t out[4]);

/* Algorithmic bytes of a suggest batch: per query 4*sum of |postings| over every
 * admissible segment and present term + len(query) + 12*k.  Host-side accounting for bench.py. */
int sg_suggest_algorithmic_bytes(const sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q,
                                 int metric, double similarity, uint32_t k, uint64_t* out_total);

/* Same accounting for sg_autocomplete_batch: 4 * |postings| of every query term in every segment >= |terms| that holds them all. */
int sg_autocomplete_algorithmic_bytes(const sg_index* index, const uint8_t* q_utf8, const uint64_t* q_offs, uint32_t n_q,
                                      uint32_t limit, uint64_t* out_total);

#ifdef __cplusplus
}
#endif
#endif /* SUGGEST_HIP_H */
