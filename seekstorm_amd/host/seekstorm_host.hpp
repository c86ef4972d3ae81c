// C++ host side of the MI355X query hot path: the planner / merge layer that sits ABOVE the C ABI
// (include/seekstorm_hip.h), mirroring the `seekstorm` crate's interface for this path.  The reference is Rust; this
// image has no Rust toolchain, so the host mirror is C++ (INTEGRATION.md shows the Rust binding).  Citations are
// relative to /root/reference/seekstorm/src.
//
//   QueryType / ResultType / SearchMode / ResultSource      search.rs:59, 168, 73; min_heap.rs:17-40
//   Result / ResultObject                                   min_heap.rs:17-40 (feature `vb`), search.rs:186-213
//   Shard::search_lexical_shard                             search.rs:2427-2442 (dispatch block 3374-3560 -> ss_bm25_search)
//   Shard::search_vector_shard                              vector.rs:1105-1115 (-> ss_vec_search)
//   Index::search                                           <IndexArc as Search>::search, search.rs:1134-1150 / 1154-2131
//   (batch coalescing)                                      behind the C ABI: the reference has no batched entry point (one query per
//                                                           request, http_server.rs:218-289); concurrent callers are
//                                                           coalesced into one C-ABI batch per device pass
//
// No compute happens here: every score, set operation, top-k and merge is behind the C ABI.  Like the reference's
// search path (search.rs:2461-2463, vector.rs:1222-1224) a failing shard degrades to an empty ResultObject; the
// C-ABI code is kept in `last_error` for diagnosis -- SS_ENOTSUP there means "the host's own dispatch answers this one"
// (ResultObject::cpu_dispatch), never "no hits".  Out of scope (stays in the Rust host):
// tokenizer / term hashing -- a lexical query arrives as resolved term ids -- facets, filters, query rewriting.
#pragma once
#include <stdint.h>

#include <condition_variable>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/seekstorm_hip.h"

namespace seekstorm {

// Phrase: the query terms are the words of the phrase, in order (a word may repeat)
enum class QueryType : uint32_t { Union = SS_OP_UNION, Intersection = SS_OP_INTERSECTION, Phrase = SS_OP_PHRASE };  // search.rs:59
enum class ResultType : uint32_t { Count = SS_RT_COUNT, Topk = SS_RT_TOPK, TopkCount = SS_RT_TOPKCOUNT };  // search.rs:168
enum class SearchMode : int { Lexical = SS_MODE_LEXICAL, Vector = SS_MODE_VECTOR, Hybrid = SS_MODE_HYBRID };  // search.rs:73
// search.rs AnnMode (used at vector.rs:1300-1307): All | Similaritythreshold(t) | Nprobe(n) | NprobeSimilaritythreshold(n, t);
// t is the normalised similarity the reference's callers pass (TopK::new converts it, vector.rs:388-397)
struct AnnMode {
  enum class Kind { All, Similaritythreshold, Nprobe, NprobeSimilaritythreshold } kind = Kind::All;
  size_t n_probe = 0;
  float similarity_threshold = 0.f;
  static AnnMode All() { return AnnMode(); }
  static AnnMode Similaritythreshold(float t) { return AnnMode{Kind::Similaritythreshold, 0, t}; }
  static AnnMode Nprobe(size_t n) { return AnnMode{Kind::Nprobe, n, 0.f}; }
  static AnnMode NprobeSimilaritythreshold(size_t n, float t) { return AnnMode{Kind::NprobeSimilaritythreshold, n, t}; }
};
enum class ResultSource : uint8_t { Lexical = SS_SRC_LEXICAL, Vector = SS_SRC_VECTOR, Hybrid = SS_SRC_HYBRID };

// min_heap.rs:17-40 with the default feature `vb`
struct Result {
  uint64_t doc_id = 0;  // usize: shard-local from the per-shard executors, global (local * S + shard) from Index::search
  float score = 0.f;
  uint32_t field_id = 0, chunk_id = 0, level_id = 0, shard_id = 0, cluster_id = 0;
  float cluster_score = 0.f, vector_score = 0.f, lexical_score = 0.f;
  ResultSource source = ResultSource::Lexical;
};

// search.rs:186-213 (fields of this path)
struct ResultObject {
  std::vector<Result> results;
  uint64_t result_count = 0;        // = results.len()
  uint64_t result_count_total = 0;  // exact match count for Count / TopkCount; accepted pushes for vectors
  uint64_t observed_vector_count = 0;
  uint64_t observed_cluster_count = 0;
  int last_error = 0;               // SS_OK or the C-ABI code that emptied this object
  // last_error == SS_ENOTSUP is NOT a failing shard: the query is one the device path leaves to the host's own dispatch
  // (search.rs:3374-3560 stands in the same function as the seam; INTEGRATION.md section 4 lists the shapes).  This mirror has no CPU
  // path by design -- the product never computes on the host -- so it says so instead of looking like "no hits".
  bool cpu_dispatch() const { return last_error == SS_ENOTSUP; }
};

// ---- host-side scalar pieces of the reference algorithm
float idf(uint64_t indexed_doc_count, uint64_t posting_count);  // search.rs:3225-3230, all f32
void normalize_f32(float* v, size_t n);                         // vector_similarity.rs:70-74 (search.rs:1464-1475)
void quantize_f32_to_i8(const float* v, size_t n, int8_t* out);  // vector_similarity.rs:1226-1232 (query side: search.rs:1487-1490)
// Quantization::TurboQuantI8, query side (search.rs:1545-1594): sign mask x FWHT x scalar quantisation; the index's seed mask
// (TurboQuant::seed_mask, vector_similarity.rs:1845-1858) is handed in.  out: dim = turboquant_dim(n) values.
size_t turboquant_dim(size_t n);
void turboquant_f32_to_i8(const float* v, size_t n, const float* seed_mask, size_t dim, bool avx2, int8_t* out, float* scale_out,
                          float* norm_out);
float threshold_raw(const float* similarity_threshold, bool euclidean = false);  // TopK::new, vector.rs:388-398; nullptr = none
float vector_score_of(float raw_dot);                           // vector.rs:1495-1499: ((dot / 16129) + 1) / 2

// search.rs ResultSort: one sort field of a lexical search -- a numeric facet (its offset inside the facet.bin record and its
// SS_FACET_* type) ascending or descending, or a Point facet (type SS_FACET_POINT) by its distance to `base` (lat, lon).
struct ResultSort {
  uint32_t facet_offset = 0;
  uint32_t facet_type = SS_FACET_U32;
  bool descending = false;
  double base[2] = {0.0, 0.0};
};
// Result sort by a String16 / String32 facet (min_heap.rs:860-897, 939-976: the STRINGS of the two docs' value ids are compared,
// Rust String order = byte-wise UTF-8).  The device sorts numbers: append one derived u32 column to the facet.bin records before
// upload_facets -- rank[id] of the id's string in that order, equal strings sharing a rank -- and sort by it as SS_FACET_U32.
// strings[id] = the facet value of id (facet.json).  Returns the records with the column appended ([n_docs][record_size + 4]);
// *rank_offset = the column's offset.
std::vector<uint8_t> string_facet_rank_column(const uint8_t* records, uint64_t n_docs, uint32_t record_size, uint32_t facet_offset,
                                              uint32_t facet_type, const std::vector<std::string>& strings, uint32_t* rank_offset);
// FacetFilter::Point (search.rs:852-859) as an ss_facet_filter: the distance to base inside [lo, hi), unit SS_POINT_KM / _MILES
ss_facet_filter point_facet_filter(uint32_t facet_offset, const double base[2], double lo, double hi, uint32_t unit, uint32_t flags = 0);


// One shard image on one MI355X.
class Shard {
 public:
  Shard(int device, uint32_t shard_id);  // failure (no device, no library) leaves an unusable shard: searches return empty
  // a view of a shard image that somebody else built and owns (its handle outlives this object): the planner and the seams over
  // an existing ss_shard -- single indexed field, SingleTerm keys; image sizes are read back from the handle
  Shard(ss_shard* borrowed, int device, uint32_t shard_id);
  ~Shard();
  Shard(const Shard&) = delete;
  Shard& operator=(const Shard&) = delete;

  bool ok() const { return h_ != nullptr; }
  int create_error() const { return create_rc_; }
  ss_shard* handle() const { return h_; }
  uint32_t shard_id() const { return shard_id_; }
  int device() const { return device_; }

  // (re)build of the device image: end of open_shard (index.rs:3796) / after a commit (commit.rs:142-148)
  // positions (optional): every posting's tf positions in CSR order -- what phrase queries walk
  int upload_lexical(uint64_t n_docs, const uint8_t* doclen_bytes, uint32_t n_terms, const uint64_t* term_offsets,
                     const uint32_t* doc_ids, const uint16_t* tfs, const uint16_t* positions = nullptr, uint64_t n_positions = 0);
  // several indexed fields (BM25F): doclen [n_fields][n_docs], postings (doc, field, tf) sorted by (doc, field) per term
  // positions (optional): every (term, doc, field) entry's tf positions inside the field -- phrase queries over several fields
  int upload_lexical_fields(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen_bytes, const float* boost, uint32_t n_terms,
                            const uint64_t* term_offsets, const uint32_t* doc_ids, const uint8_t* field_ids, const uint16_t* tfs,
                            const uint16_t* positions = nullptr, uint64_t n_positions = 0);
  // one commit (commit.rs:142-148 commit -> warmup; the re-commit of an incomplete level, 204-206, is the same call with the same
  // level): the level's length bytes and its postings of ALL known terms in id order -- ids below n_dense_terms belong to the dense
  // image, the others are the sparse tier's lists, new rare terms extend the id range (n_dense_terms == n_terms: no tier).
  // positions (optional): every posting's, tf of them each or npos[i]
  int commit_level(uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms, uint32_t n_dense_terms,
                   const uint64_t* term_offsets, const uint32_t* doc_ids, const uint16_t* tfs, const uint16_t* positions = nullptr,
                   uint64_t n_positions = 0, const uint16_t* npos = nullptr);
  // VectorSimilarity of the image (index-wide in the reference): Dot / Cosine (default) or Euclidean; BEFORE the upload
  int set_vector_similarity(bool euclidean);
  int upload_vectors(uint64_t n_rows, uint32_t dim, const float* rows, const uint32_t* row_doc_ids);
  // device-side synthetic images (bench / tests: the generators of ss_bm25_synth / ss_vec_synth), shard `shard_id` of n_shards
  int synth_lexical(uint64_t seed, uint64_t n_docs, uint32_t n_terms, const uint32_t* thresh32, const uint8_t* len_table1024, uint32_t n_shards = 1);
  int synth_vectors(uint64_t seed, uint64_t n_rows, uint32_t dim, uint32_t n_shards = 1);
  // shard files as the reference writes them: index.bin (single indexed field), vector.bin (f32), delete.bin.
  // term_keys: key_hash of every term id, ascending; an n-gram key (key_hash & 7 != 0) holds one id per component term,
  // consecutive -- a query term that resolved to it is passed as those ids (make_query applies idf_ngram_i)
  // with_positions: the positions of every posting are decoded as well (QueryType::Phrase on an opened index; one indexed
  // field; an n-gram key's own positions stand behind its first component term: in a phrase the key is one entry, make_query)
  int open_index_bin(const uint8_t* bytes, uint64_t len, uint32_t key_head_size, std::vector<uint64_t>* term_keys,
                     bool with_positions = false);
  int open_vector_bin(const uint8_t* bytes, uint64_t len, uint32_t dim, bool i8 = false, bool use_record_scale = false);
  // Precision::I8 records; queries given as f32 are quantised with quantize_f32_to_i8 like the reference's
  int upload_vectors_i8(uint64_t n_rows, uint32_t dim, const int8_t* rows, const float* row_scale, const uint32_t* row_doc_ids);
  bool vectors_are_i8() const { return i8_; }
  bool euclidean() const { return euclidean_; }
  // delete_hashset (index.rs:1594): replaces the tombstone set; delete_document (index.rs:5110) re-sends it
  int set_deleted(const uint64_t* doc_ids, uint64_t n);
  int synth_lexical(uint64_t seed, uint64_t n_docs, uint32_t n_terms, const uint32_t* thresh32, const uint8_t* len_table1024);
  int synth_vectors(uint64_t seed, uint64_t n_rows, uint32_t dim);

  uint64_t indexed_doc_count() const { return n_docs_; }
  uint64_t vector_count() const { return n_rows_; }
  uint32_t dim() const { return dim_; }

  // term resolution result -> device query (idf from shard-local N and posting_count, search.rs:3225-3230)
  // field_filter: indexed field ids every term must occur in one of (several indexed fields, search.rs:2483-2492)
  int make_query(const std::vector<uint32_t>& terms, QueryType qt, ss_bm25_query* out,
                 const std::vector<uint32_t>& not_terms = {}, const std::vector<uint16_t>& field_filter = {});

  // the reference's per-shard seams (one query)
  // not_terms: the "-term" operands; field_filter: indexed field ids (a union of several terms under it is answered through
  // the reference's own sub-queries, union.rs:1168-1479); result_sort: Vec<ResultSort> over numeric and Point facets
  ResultObject search_lexical_shard(const std::vector<uint32_t>& query_terms, QueryType query_type_default, size_t offset,
                                    size_t length, ResultType result_type, const std::vector<ss_facet_filter>& facet_filter = {},
                                    const std::vector<uint32_t>& not_terms = {}, const std::vector<uint16_t>& field_filter = {},
                                    const std::vector<ResultSort>& result_sort = {});
  // query_facets of one query (facet_count, add_result.rs:484-640): counts [n_buckets + 1], the last slot = outside the buckets.
  // String facets: n_buckets ids (bounds empty); numeric facets: the ranges' ascending lower bounds as the value's bits;
  // Point facets (facet_type SS_FACET_POINT): base + the lower bounds of the distance ranges as f64 bits.
  // order keys of docs under one sort field: larger = better (the facet's stored value mapped to an unsigned integer that orders
  // like it, complemented for an ascending sort; Point facets: by simplified_distance to the field's base)
  int sort_keys(const std::vector<uint32_t>& doc_ids, const ResultSort& field, std::vector<uint64_t>* keys);
  int facet_count(const ss_bm25_query& query, uint32_t facet_offset, uint32_t facet_type, uint32_t n_buckets,
                  const std::vector<uint64_t>& range_lower_bounds, std::vector<uint64_t>* counts, uint64_t* total,
                  const std::vector<ss_facet_filter>& facet_filter = {}, const ss_facet_point* base = nullptr);
  // field_filter: indexed field ids to search (the reference resolves the names through schema_map, vector.rs:1225-1237);
  // empty = every field
  ResultObject search_vector_shard(const float* query_vector /* normalised, dim() floats */, size_t length,
                                   const float* similarity_threshold, const AnnMode& ann_mode = AnnMode(),
                                   const std::vector<uint16_t>& field_filter = {});
  // batched forms used by the coalescer (results sorted by score desc, shard-local ids)
  // facet_filter (search.rs FacetFilter, add_result.rs:341-482): shared by the queries of the call; needs upload_facets
  // mark_frequent: evaluate the all_terms_frequent shortcut per query here (false: the caller already marked its queries)
  std::vector<ResultObject> search_lexical_batch(const std::vector<ss_bm25_query>& queries, size_t k, ResultType result_type,
                                                 const std::vector<ss_facet_filter>& facet_filter = {}, bool mark_frequent = true);
  // sets SS_OP_ALL_TERMS_FREQUENT on *q when the reference's condition holds for this shard and top_k (intersection.rs:198-209)
  bool mark_all_terms_frequent(ss_bm25_query* q, size_t top_k) const;
  int upload_facets(uint64_t n_docs, uint32_t record_size, const uint8_t* records);  // facet.bin records
  std::vector<ResultObject> search_vector_batch(const float* query_vectors, size_t n_queries, size_t k,
                                                const float* similarity_threshold, const AnnMode& ann_mode = AnnMode(),
                                                const std::vector<uint16_t>& field_filter = {});
  int set_fields(const std::vector<uint16_t>& row_field);  // VectorHeader.field_id per record of an array upload
  // cluster structure of rows uploaded in file order (vector.rs:1066-1094); open_vector_bin keeps the file's own
  int set_clusters(const std::vector<uint32_t>& level_clusters, const std::vector<uint32_t>& child_count);

 private:
  int sorted_topk(const ss_bm25_query& q, const ResultSort* sorts, size_t n_sorts, size_t k, std::vector<ss_facet_filter> filters,
                  std::vector<Result>* out, uint64_t* total, bool* have_total);
  ss_shard* h_ = nullptr;
  bool owns_ = true;
  uint32_t shard_id_ = 0;
  int device_ = 0;
  int create_rc_ = SS_OK;
  uint64_t n_docs_ = 0, n_rows_ = 0, n_deleted_ = 0;
  uint32_t dim_ = 0;
  bool i8_ = false;
  bool euclidean_ = false;
  uint32_t lexical_fields_ = 1;                // indexed fields of the lexical image
  std::vector<uint8_t> ngram_components_;      // per term id of an opened index.bin: components of its key (1 = SingleTerm)
  std::vector<uint8_t> ngram_component_;       // ... and which of them the term is (0 = a SingleTerm key or a key's first component)
  std::vector<uint32_t> ngram_component_df_;   // posting count of the component term (n-gram components)
};

// In-process multi-shard index: doc g lives in shard g % S with local id g / S (index.rs:5284).
class Index {
 public:
  explicit Index(std::vector<std::shared_ptr<Shard>> shards) : shards_(std::move(shards)) {}
  ~Index();
  size_t shard_number() const { return shards_.size(); }
  // Shards on DIFFERENT GPUs: form one RCCL communicator over them (ss_comm_create_all, rank = shard id).  From then on
  // search_lexical_batch exchanges and merges the per-shard lists on the devices over xGMI (ss_bm25_search_sharded) instead
  // of gathering them on the host.  SS_EINVAL when two shards share a device (a communicator holds one rank per GPU).
  int enable_device_exchange();
  bool device_exchange() const { return !comms_.empty(); }
  // A batch of lexical queries over all shards, merged: per query the top `k` of the whole index (global ids) and the
  // summed totals -- the batched form of search() for SearchMode::Lexical, offset 0.  One host thread per shard.
  std::vector<ResultObject> search_lexical_batch(const std::vector<std::vector<uint32_t>>& query_terms, QueryType query_type_default,
                                                 size_t k, ResultType result_type);
  Shard& shard(size_t i) { return *shards_[i]; }
  // search() for SearchMode::Lexical with result_sort: every shard returns its best offset + length under the sort
  // (Shard::search_lexical_shard with result_sort), the lists are merged under the same order across shards -- the facet
  // values of the two docs, each read from its own shard, then the score (result_ordering_root, min_heap.rs:56-300;
  // search.rs:2088) --, then offset / length.  Global ids, totals summed.
  ResultObject search_lexical_sorted(const std::vector<uint32_t>& query_terms, QueryType query_type_default, size_t offset, size_t length,
                                     const std::vector<ResultSort>& result_sort, const std::vector<ss_facet_filter>& facet_filter = {},
                                     const std::vector<uint32_t>& not_terms = {});

  // <IndexArc as Search>::search for this path.  query_terms: resolved term ids (empty = no lexical part);
  // query_vector: dim floats or nullptr; Cosine with external inference -> normalised here when normalize_query.
  // One host thread per shard (the reference spawns one task per shard, search.rs:1637-1743); each shard is asked for
  // (offset 0, length offset+length) (search.rs:1658-1659); ids become local * S + shard (search.rs:1671);
  // totals are summed, Hybrid takes max(lexical, vector) per shard (search.rs:1919-1921); merge / RRF / sort /
  // offset / truncate through ss_merge_results (search.rs:1875-2119).
  ResultObject search(const std::vector<uint32_t>& query_terms, const float* query_vector, QueryType query_type_default,
                      SearchMode search_mode, size_t offset, size_t length, ResultType result_type,
                      const float* similarity_threshold = nullptr, bool normalize_query = true,
                      const AnnMode& ann_mode = AnnMode(), const std::vector<uint16_t>& vector_field_filter = {},
                      const std::vector<ss_facet_filter>& facet_filter = {}, const std::vector<uint32_t>& not_terms = {},
                      const std::vector<uint16_t>& lexical_field_filter = {});

 private:
  std::vector<std::shared_ptr<Shard>> shards_;
  std::vector<ss_comm*> comms_;  // one per shard once enable_device_exchange succeeded
};

// Concurrent single-query callers need no helper here: the reference's calling pattern -- many runtime workers, each inside
// Search::search with ONE query -- is coalesced into device batches BEHIND the C ABI (ss_shard_set_coalescing,
// include/seekstorm_hip.h), so Shard::search_lexical_shard / search_vector_shard and Index::search are simply called from as
// many threads as the host has.  (Rounds 1-2 kept batch coalescer classes in this mirror; a Rust host does not link the mirror.)

}  // namespace seekstorm
