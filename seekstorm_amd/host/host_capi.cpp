// Flat C shim over the C++ host mirror (seekstorm_host.hpp) so that the parity tests can drive it through ctypes.
// Test plumbing only: a C++ application links seekstorm_host.hpp directly.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "seekstorm_host.hpp"

using namespace seekstorm;

struct ssh_index {
  std::vector<std::shared_ptr<Shard>> shards;
  std::unique_ptr<Index> index;
};

static int write_out(const ResultObject& ro, uint32_t cap, uint64_t* out_doc, float* out_score, uint8_t* out_source,
                     float* out_lexical, float* out_vector, uint64_t* out_meta /* [4] */) {
  const uint32_t n = (uint32_t)std::min<size_t>(cap, ro.results.size());
  for (uint32_t i = 0; i < n; i++) {
    out_doc[i] = ro.results[i].doc_id;
    out_score[i] = ro.results[i].score;
    if (out_source) out_source[i] = (uint8_t)ro.results[i].source;
    if (out_lexical) out_lexical[i] = ro.results[i].lexical_score;
    if (out_vector) out_vector[i] = ro.results[i].vector_score;
  }
  if (out_meta) {
    out_meta[0] = ro.result_count;
    out_meta[1] = ro.result_count_total;
    out_meta[2] = ro.observed_vector_count;
    out_meta[3] = (uint64_t)(int64_t)ro.last_error;
  }
  return (int)n;
}

extern "C" {

float ssh_idf(uint64_t n_docs, uint64_t posting_count) { return idf(n_docs, posting_count); }
void ssh_normalize_f32(float* v, uint64_t n) { normalize_f32(v, (size_t)n); }
float ssh_threshold_raw(float similarity_threshold) { return threshold_raw(&similarity_threshold); }
float ssh_vector_score(float raw) { return vector_score_of(raw); }
void ssh_turboquant(const float* v, uint64_t n, const float* mask, uint64_t dim, int avx2, int8_t* out, float* scale, float* norm) {
  turboquant_f32_to_i8(v, (size_t)n, mask, (size_t)dim, avx2 != 0, out, scale, norm);
}

ssh_index* ssh_index_create(int n_shards, const int* devices) {
  ssh_index* ix = new ssh_index();
  for (int i = 0; i < n_shards; i++) ix->shards.push_back(std::make_shared<Shard>(devices ? devices[i] : 0, (uint32_t)i));
  ix->index.reset(new Index(ix->shards));
  return ix;
}
// an index over shard images that already exist (handles owned by the caller, e.g. the Python mirror's shards)
ssh_index* ssh_index_adopt(int n_shards, void* const* handles, const int* devices) {
  ssh_index* ix = new ssh_index();
  for (int i = 0; i < n_shards; i++) ix->shards.push_back(std::make_shared<Shard>((ss_shard*)handles[i], devices ? devices[i] : 0, (uint32_t)i));
  ix->index.reset(new Index(ix->shards));
  return ix;
}
void ssh_index_destroy(ssh_index* ix) { delete ix; }
int ssh_shard_ok(ssh_index* ix, int shard) { return ix->shards[shard]->ok() ? 1 : 0; }
int ssh_shard_create_error(ssh_index* ix, int shard) { return ix->shards[shard]->create_error(); }

int ssh_upload_lexical(ssh_index* ix, int shard, uint64_t n_docs, const uint8_t* doclen, uint32_t n_terms, const uint64_t* offs,
                       const uint32_t* docs, const uint16_t* tfs) {
  return ix->shards[shard]->upload_lexical(n_docs, doclen, n_terms, offs, docs, tfs);
}
int ssh_upload_vectors_i8(ssh_index* ix, int shard, uint64_t n_rows, uint32_t dim, const int8_t* rows, const uint32_t* ids) {
  return ix->shards[shard]->upload_vectors_i8(n_rows, dim, rows, nullptr, ids);
}
void ssh_quantize_f32_to_i8(const float* v, uint64_t n, int8_t* out) { quantize_f32_to_i8(v, (size_t)n, out); }

int ssh_upload_vectors(ssh_index* ix, int shard, uint64_t n_rows, uint32_t dim, const float* rows, const uint32_t* ids) {
  return ix->shards[shard]->upload_vectors(n_rows, dim, rows, ids);
}

// Shard::open_index_bin; returns the number of terms (term ids) or a negative code; keys_out [cap] = their key hashes
int ssh_open_index_bin(ssh_index* ix, int shard, const uint8_t* bytes, uint64_t len, uint32_t key_head_size, uint64_t* keys_out,
                       uint32_t cap) {
  std::vector<uint64_t> keys;
  const int rc = ix->shards[shard]->open_index_bin(bytes, len, key_head_size, &keys);
  if (rc) return rc;
  for (size_t i = 0; i < keys.size() && i < cap; i++) keys_out[i] = keys[i];
  return (int)keys.size();
}

// Index::search; returns the number of results written (<= cap)
int ssh_search(ssh_index* ix, const uint32_t* terms, uint32_t n_terms, const float* query_vector, uint32_t query_type,
               int search_mode, uint32_t offset, uint32_t length, uint32_t result_type, int has_threshold, float threshold,
               int normalize_query, uint32_t cap, uint64_t* out_doc, float* out_score, uint8_t* out_source,
               float* out_lexical, float* out_vector, uint64_t* out_meta) {
  std::vector<uint32_t> t(terms, terms + n_terms);
  ResultObject ro = ix->index->search(t, query_vector, (QueryType)query_type, (SearchMode)search_mode, offset, length,
                                      (ResultType)result_type, has_threshold ? &threshold : nullptr, normalize_query != 0);
  return write_out(ro, cap, out_doc, out_score, out_source, out_lexical, out_vector, out_meta);
}

// per-shard seams
int ssh_search_lexical_shard(ssh_index* ix, int shard, const uint32_t* terms, uint32_t n_terms, uint32_t query_type,
                             uint32_t offset, uint32_t length, uint32_t result_type, uint32_t cap, uint64_t* out_doc,
                             float* out_score, uint64_t* out_meta) {
  std::vector<uint32_t> t(terms, terms + n_terms);
  ResultObject ro = ix->shards[shard]->search_lexical_shard(t, (QueryType)query_type, offset, length, (ResultType)result_type);
  return write_out(ro, cap, out_doc, out_score, nullptr, nullptr, nullptr, out_meta);
}

int ssh_upload_facets(ssh_index* ix, int shard, uint64_t n_docs, uint32_t record_size, const uint8_t* records) {
  return ix->shards[shard]->upload_facets(n_docs, record_size, records);
}
int ssh_search_lexical_shard_filtered(ssh_index* ix, int shard, const uint32_t* terms, uint32_t n_terms, uint32_t query_type,
                                      uint32_t offset, uint32_t length, uint32_t result_type, uint32_t n_filters,
                                      const ss_facet_filter* filters, uint32_t cap, uint64_t* out_doc, float* out_score,
                                      uint64_t* out_meta) {
  std::vector<uint32_t> t(terms, terms + n_terms);
  std::vector<ss_facet_filter> f(filters, filters + n_filters);
  ResultObject ro = ix->shards[shard]->search_lexical_shard(t, (QueryType)query_type, offset, length, (ResultType)result_type, f);
  return write_out(ro, cap, out_doc, out_score, nullptr, nullptr, nullptr, out_meta);
}

// Shard::search_lexical_shard with every option of the seam: facet filter, NOT terms, field filter, result sort
struct ssh_result_sort {
  uint32_t facet_offset, facet_type, descending, reserved;
  double base[2];
};
int ssh_search_lexical_shard_ex(ssh_index* ix, int shard, const uint32_t* terms, uint32_t n_terms, uint32_t query_type,
                                uint32_t offset, uint32_t length, uint32_t result_type, uint32_t n_filters,
                                const ss_facet_filter* filters, const uint32_t* not_terms, uint32_t n_not, const uint16_t* field_filter,
                                uint32_t n_field_filter, const ssh_result_sort* sorts, uint32_t n_sorts, uint32_t cap, uint64_t* out_doc,
                                float* out_score, uint64_t* out_meta) {
  std::vector<uint32_t> t(terms, terms + n_terms), nt(not_terms, not_terms + n_not);
  std::vector<ss_facet_filter> f(filters, filters + n_filters);
  std::vector<uint16_t> ff(field_filter, field_filter + n_field_filter);
  std::vector<ResultSort> rs(n_sorts);
  for (uint32_t i = 0; i < n_sorts; i++) {
    rs[i].facet_offset = sorts[i].facet_offset;
    rs[i].facet_type = sorts[i].facet_type;
    rs[i].descending = sorts[i].descending != 0;
    rs[i].base[0] = sorts[i].base[0];
    rs[i].base[1] = sorts[i].base[1];
  }
  ResultObject ro = ix->shards[shard]->search_lexical_shard(t, (QueryType)query_type, offset, length, (ResultType)result_type, f, nt, ff, rs);
  return write_out(ro, cap, out_doc, out_score, nullptr, nullptr, nullptr, out_meta);
}
// string_facet_rank_column: strings given as one NUL-separated buffer; out [n_docs][record_size + 4]
uint32_t ssh_string_rank_column(const uint8_t* records, uint64_t n_docs, uint32_t record_size, uint32_t facet_offset, uint32_t facet_type,
                                const char* strings, uint64_t strings_len, uint32_t n_strings, uint8_t* out) {
  std::vector<std::string> v;
  const char* p = strings;
  const char* end = strings + strings_len;
  for (uint32_t i = 0; i < n_strings && p <= end; i++) {
    const char* q = (const char*)memchr(p, 0, (size_t)(end - p));
    if (!q) q = end;
    v.emplace_back(p, q);
    p = q + 1;
  }
  uint32_t off = 0;
  std::vector<uint8_t> r = string_facet_rank_column(records, n_docs, record_size, facet_offset, facet_type, v, &off);
  std::memcpy(out, r.data(), r.size());
  return off;
}
// Index::search, lexical mode, with the seam's NOT terms and field filter
int ssh_search_lexical_ex(ssh_index* ix, const uint32_t* terms, uint32_t n_terms, uint32_t query_type, uint32_t offset, uint32_t length,
                          uint32_t result_type, const uint32_t* not_terms, uint32_t n_not, const uint16_t* field_filter, uint32_t n_ff,
                          uint32_t cap, uint64_t* out_doc, float* out_score, uint64_t* out_meta) {
  if (!ix->index) ix->index.reset(new Index(ix->shards));
  ResultObject ro = ix->index->search(std::vector<uint32_t>(terms, terms + n_terms), nullptr, (QueryType)query_type, SearchMode::Lexical, offset,
                                      length, (ResultType)result_type, nullptr, true, AnnMode(), {}, {},
                                      std::vector<uint32_t>(not_terms, not_terms + n_not), std::vector<uint16_t>(field_filter, field_filter + n_ff));
  return write_out(ro, cap, out_doc, out_score, nullptr, nullptr, nullptr, out_meta);
}
// Index::search_lexical_sorted over all shards of the index
int ssh_index_search_sorted(ssh_index* ix, const uint32_t* terms, uint32_t n_terms, uint32_t query_type, uint32_t offset, uint32_t length,
                            uint32_t n_filters, const ss_facet_filter* filters, const ssh_result_sort* sorts, uint32_t n_sorts, uint32_t cap,
                            uint64_t* out_doc, float* out_score, uint64_t* out_meta) {
  std::vector<ResultSort> rs(n_sorts);
  for (uint32_t i = 0; i < n_sorts; i++) {
    rs[i].facet_offset = sorts[i].facet_offset;
    rs[i].facet_type = sorts[i].facet_type;
    rs[i].descending = sorts[i].descending != 0;
    rs[i].base[0] = sorts[i].base[0];
    rs[i].base[1] = sorts[i].base[1];
  }
  if (!ix->index) ix->index.reset(new Index(ix->shards));
  ResultObject ro = ix->index->search_lexical_sorted(std::vector<uint32_t>(terms, terms + n_terms), (QueryType)query_type, offset, length, rs,
                                                     std::vector<ss_facet_filter>(filters, filters + n_filters));
  return write_out(ro, cap, out_doc, out_score, nullptr, nullptr, nullptr, out_meta);
}
int ssh_upload_lexical_fields(ssh_index* ix, int shard, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                              uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs) {
  return ix->shards[shard]->upload_lexical_fields(n_docs, n_fields, doclen, boost, n_terms, offs, docs, fields, tfs);
}
// ... with the positions of every (term, doc, field) entry: QueryType::Phrase over several indexed fields
int ssh_upload_lexical_fields_positions(ssh_index* ix, int shard, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                                        uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs,
                                        const uint16_t* positions, uint64_t n_positions) {
  return ix->shards[shard]->upload_lexical_fields(n_docs, n_fields, doclen, boost, n_terms, offs, docs, fields, tfs, positions, n_positions);
}
int ssh_set_deleted(ssh_index* ix, int shard, const uint64_t* doc_ids, uint64_t n) { return ix->shards[shard]->set_deleted(doc_ids, n); }
int ssh_commit_level(ssh_index* ix, int shard, uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms,
                     uint32_t n_dense_terms, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs, const uint16_t* positions,
                     uint64_t n_positions) {
  return ix->shards[shard]->commit_level(level, n_level_docs, level_doclen, n_terms, n_dense_terms, offs, docs, tfs, positions, n_positions);
}
// Shard::facet_count of one query; out_counts [n_buckets + 1]
int ssh_facet_count(ssh_index* ix, int shard, const uint32_t* terms, uint32_t n_terms, uint32_t query_type, uint32_t facet_offset,
                    uint32_t facet_type, uint32_t n_buckets, const uint64_t* bounds, uint32_t n_bounds, const ss_facet_point* base,
                    uint32_t n_filters, const ss_facet_filter* filters, uint64_t* out_counts, uint64_t* out_total) {
  ss_bm25_query q;
  const int rc = ix->shards[shard]->make_query(std::vector<uint32_t>(terms, terms + n_terms), (QueryType)query_type, &q);
  if (rc != SS_OK) return rc;
  std::vector<uint64_t> counts;
  const int rc2 = ix->shards[shard]->facet_count(q, facet_offset, facet_type, n_buckets, std::vector<uint64_t>(bounds, bounds + n_bounds), &counts,
                                                 out_total, std::vector<ss_facet_filter>(filters, filters + n_filters), base);
  if (rc2 == SS_OK) std::memcpy(out_counts, counts.data(), counts.size() * 8);
  return rc2;
}

// Shard::search_vector_shard with an AnnMode: kind 0 All, 1 Similaritythreshold(t), 2 Nprobe(n), 3 NprobeSimilaritythreshold(n, t);
// out_meta[4] = count, total, observed vectors, last_error; *out_clusters = observed_cluster_count
int ssh_set_clusters(ssh_index* ix, int shard, uint32_t n_levels, const uint32_t* level_clusters, uint32_t n_clusters,
                     const uint32_t* child_count) {
  return ix->shards[shard]->set_clusters(std::vector<uint32_t>(level_clusters, level_clusters + n_levels),
                                         std::vector<uint32_t>(child_count, child_count + n_clusters));
}
int ssh_search_vector_shard_ann(ssh_index* ix, int shard, const float* query_vector, uint32_t length, int kind, uint32_t n_probe,
                                float threshold, uint32_t cap, uint64_t* out_doc, float* out_score, uint64_t* out_meta,
                                uint64_t* out_clusters) {
  AnnMode am;
  if (kind == 1) am = AnnMode::Similaritythreshold(threshold);
  else if (kind == 2) am = AnnMode::Nprobe(n_probe);
  else if (kind == 3) am = AnnMode::NprobeSimilaritythreshold(n_probe, threshold);
  ResultObject ro = ix->shards[shard]->search_vector_shard(query_vector, length, nullptr, am);
  if (out_clusters) *out_clusters = ro.observed_cluster_count;
  return write_out(ro, cap, out_doc, out_score, nullptr, nullptr, nullptr, out_meta);
}

// n concurrent single-query vector searches: one thread per query, each calling Shard::search_vector_shard as the reference's
// runtime workers call search_vector_shard -- the C ABI coalesces them into device batches (ss_shard_set_coalescing);
// out arrays are [n][length]; returns the number of device batches used
int ssh_coalesced_vector_search(ssh_index* ix, int shard, uint32_t n, const float* queries, uint32_t length,
                                uint32_t max_batch, uint32_t max_wait_us, uint64_t* out_doc, float* out_score,
                                uint32_t* out_count) {
  Shard& sh = *ix->shards[shard];
  if (ss_shard_set_coalescing(sh.handle(), 1024, max_batch, max_wait_us) != SS_OK) return SS_EINVAL;
  uint64_t b0 = 0, b1 = 0;
  (void)ss_shard_coalescing_stats(sh.handle(), nullptr, nullptr, &b0, nullptr);
  const uint32_t dim = sh.dim();
  std::vector<ResultObject> res(n);
  std::vector<std::thread> th;
  for (uint32_t i = 0; i < n; i++)
    th.emplace_back([&, i] { res[i] = sh.search_vector_shard(queries + (size_t)i * dim, length, nullptr); });
  for (auto& t : th) t.join();
  (void)ss_shard_coalescing_stats(sh.handle(), nullptr, nullptr, &b1, nullptr);
  (void)ss_shard_set_coalescing(sh.handle(), 1024, SS_VEC_BATCH, 0);
  for (uint32_t i = 0; i < n; i++) {
    const ResultObject& ro = res[i];
    if (ro.last_error) return ro.last_error;
    out_count[i] = (uint32_t)ro.results.size();
    for (size_t j = 0; j < ro.results.size() && j < length; j++) {
      out_doc[(size_t)i * length + j] = ro.results[j].doc_id;
      out_score[(size_t)i * length + j] = ro.results[j].score;
    }
  }
  return (int)(b1 - b0);
}

// n concurrent single-query lexical searches (one thread per query through Shard::search_lexical_shard); query i =
// terms[term_off[i] .. term_off[i+1]) followed by its NOT terms not_terms[not_off[i] .. not_off[i+1]); out arrays are [n][length]
int ssh_coalesced_lexical_search(ssh_index* ix, int shard, uint32_t n, const uint32_t* terms, const uint32_t* term_off,
                                 const uint32_t* not_terms, const uint32_t* not_off, uint32_t query_type, uint32_t offset,
                                 uint32_t length, uint32_t result_type, uint32_t max_batch, uint32_t max_wait_us,
                                 uint64_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total) {
  Shard& sh = *ix->shards[shard];
  if (ss_shard_set_coalescing(sh.handle(), max_batch, SS_VEC_BATCH, max_wait_us) != SS_OK) return SS_EINVAL;
  uint64_t b0 = 0, b1 = 0;
  (void)ss_shard_coalescing_stats(sh.handle(), &b0, nullptr, nullptr, nullptr);
  std::vector<ResultObject> res(n);
  std::vector<std::thread> th;
  for (uint32_t i = 0; i < n; i++)
    th.emplace_back([&, i] {
      std::vector<uint32_t> t(terms + term_off[i], terms + term_off[i + 1]);
      std::vector<uint32_t> nt;
      if (not_terms) nt.assign(not_terms + not_off[i], not_terms + not_off[i + 1]);
      res[i] = sh.search_lexical_shard(t, (QueryType)query_type, offset + (i % 3), length, (ResultType)result_type, {}, nt);  // mixed offsets
    });
  for (auto& t : th) t.join();
  (void)ss_shard_coalescing_stats(sh.handle(), &b1, nullptr, nullptr, nullptr);
  (void)ss_shard_set_coalescing(sh.handle(), 1024, SS_VEC_BATCH, 0);
  for (uint32_t i = 0; i < n; i++) {
    const ResultObject& ro = res[i];
    if (ro.last_error) return ro.last_error;
    out_count[i] = (uint32_t)ro.results.size();
    out_total[i] = ro.result_count_total;
    for (size_t j = 0; j < ro.results.size() && j < length; j++) {
      out_doc[(size_t)i * length + j] = ro.results[j].doc_id;
      out_score[(size_t)i * length + j] = ro.results[j].score;
    }
  }
  return (int)(b1 - b0);
}

int ssh_synth_lexical(ssh_index* ix, int shard, uint64_t seed, uint64_t n_docs, uint32_t n_terms, const uint32_t* thresh32,
                      const uint8_t* len_table1024) {
  return ix->shards[shard]->synth_lexical(seed, n_docs, n_terms, thresh32, len_table1024, (uint32_t)ix->shards.size());
}
int ssh_synth_vectors(ssh_index* ix, int shard, uint64_t seed, uint64_t n_rows, uint32_t dim) {
  return ix->shards[shard]->synth_vectors(seed, n_rows, dim, (uint32_t)ix->shards.size());
}

// The reference's REAL calling pattern, measured: n_threads host threads, each issuing ONE query per call through Index::search
// (search.rs:1637-1743: one search per runtime worker, no batched entry point) for `seconds` of wall time.  mode: SS_MODE_*;
// query i of the n_queries given = terms[term_off[i] .. term_off[i+1]) and / or vectors[i * dim ..]; threads draw queries round
// robin.  Every thread's first call is a warm-up, and so is every call begun in the first min(0.25 s, seconds / 8): not recorded, not
// counted, outside the wall time.  out[0] = completed searches, out[1] = wall seconds, out[2] / out[3] = p50 / p99 of the per-call latency in
// microseconds (host clock around the call), out[4] = calls that came back with an error.
int ssh_bench_concurrent(ssh_index* ix, int mode, uint32_t n_threads, double seconds, uint32_t n_queries, const uint32_t* terms,
                         const uint32_t* term_off, const float* vectors, uint32_t query_type, uint32_t length, uint32_t result_type,
                         double* out /* [5] */) {
  if (!ix->index || n_threads == 0 || n_queries == 0) return SS_EINVAL;
  const uint32_t dim = ix->shards[0]->dim();
  std::atomic<uint64_t> next{0}, errors{0};
  std::atomic<bool> go{false}, stop{false};
  std::mutex go_mu;  // (the threads sleep until the start: 256 of them spinning on yield() spent the process's CPU quota before the first call)
  std::condition_variable go_cv;
  std::vector<std::vector<float>> lat(n_threads), at(n_threads);  // per call: its latency, and when it began (us since the start)
  std::vector<std::thread> th;
  const bool hist = getenv("SSH_BENCH_HIST") != nullptr;  // diagnostics on stderr: the tail's shape and WHEN its calls happened
  std::chrono::steady_clock::time_point w0, t_steady;  // (both set before `go`: the threads read them after it)
  for (uint32_t t = 0; t < n_threads; t++)
    th.emplace_back([&, t] {
      lat[t].reserve(1 << 16);
      { std::unique_lock<std::mutex> lk(go_mu); go_cv.wait(lk, [&] { return go.load(std::memory_order_acquire); }); }
      // every thread's FIRST call is a warm-up and is not recorded: all n_threads arrive at once, find the coalescer empty and the lanes'
      // staging unallocated -- T = 256 hybrid callers: 256 calls of 90 .. 120 ms, 1.3 % of a 3 s run, which is where its "p99" then sat
      // (profiles/r6_hybrid_hist.log: every slow call began in the first 100 ms, one per thread)
      bool warm = false;
      while (!stop.load(std::memory_order_relaxed)) {
        const uint32_t i = (uint32_t)(next.fetch_add(1, std::memory_order_relaxed) % n_queries);
        std::vector<uint32_t> q;
        if (mode != SS_MODE_VECTOR) q.assign(terms + term_off[i], terms + term_off[i + 1]);
        const float* v = mode != SS_MODE_LEXICAL ? vectors + (size_t)i * dim : nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        ResultObject ro = ix->index->search(q, v, (QueryType)query_type, (SearchMode)mode, 0, length, (ResultType)result_type, nullptr,
                                            false /* the bench's vectors are normalised already */);
        const auto t1 = std::chrono::steady_clock::now();
        if (ro.last_error || ro.results.empty()) errors.fetch_add(1, std::memory_order_relaxed);
        if (!warm) { warm = true; continue; }
        // ... and so is every call that BEGAN in the first `warm_s` of the run: the callers reach their steady state -- all of them riding
        // the same pass, one batch behind the other -- only after two or three passes; until then a caller arrives in the middle of
        // somebody else's pass and waits it out before its own (hybrid, T = 64: exactly 64 calls of 20 - 29 ms, all begun in the first
        // 100 ms, in every run: profiles/r6f_tail_repeat.log).  Not recorded, not counted; the wall time starts after them.
        if (t0 < t_steady) continue;
        lat[t].push_back((float)std::chrono::duration<double, std::micro>(t1 - t0).count());
        if (hist) at[t].push_back((float)std::chrono::duration<double, std::micro>(t0 - w0).count());
      }
    });
  const double warm_s = std::min(0.25, seconds / 8.0);
  w0 = std::chrono::steady_clock::now();
  t_steady = w0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(warm_s));
  { std::lock_guard<std::mutex> lk(go_mu); go.store(true, std::memory_order_release); }
  go_cv.notify_all();
  std::this_thread::sleep_for(std::chrono::duration<double>(warm_s + seconds));
  stop.store(true, std::memory_order_relaxed);
  for (auto& t : th) t.join();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_steady).count();
  std::vector<float> all;
  for (auto& l : lat) all.insert(all.end(), l.begin(), l.end());
  std::sort(all.begin(), all.end());
  out[0] = (double)all.size();
  out[1] = wall;
  out[2] = all.empty() ? 0.0 : all[all.size() / 2];
  out[3] = all.empty() ? 0.0 : all[std::min(all.size() - 1, (size_t)((double)all.size() * 0.99))];
  out[4] = (double)errors.load();
  if (hist && !all.empty()) {
    auto pc = [&](double p) { return all[std::min(all.size() - 1, (size_t)((double)all.size() * p))]; };
    const float slow = 1.5f * (float)out[2];
    size_t n_slow = 0;
    std::vector<uint32_t> when((size_t)(wall * 10.0) + 2, 0u), per_thread(n_threads, 0u);
    for (uint32_t t = 0; t < n_threads; t++)
      for (size_t j = 0; j < lat[t].size(); j++)
        if (lat[t][j] > slow) { n_slow++; per_thread[t]++; when[std::min(when.size() - 1, (size_t)(at[t][j] / 1e5f))]++; }
    uint32_t thr_hit = 0, thr_max = 0;
    for (uint32_t c : per_thread) { thr_hit += c != 0; thr_max = std::max(thr_max, c); }
    fprintf(stderr, "[hist] T=%u: %zu calls; us p10 %.0f p50 %.0f p90 %.0f p95 %.0f p99 %.0f p99.9 %.0f max %.0f; slow (> 1.5 x p50): %zu calls on %u threads (most on one: %u); by start time, per 100 ms:",
            n_threads, all.size(), pc(0.10), pc(0.50), pc(0.90), pc(0.95), pc(0.99), pc(0.999), all.back(), n_slow, thr_hit, thr_max);
    for (uint32_t c : when) fprintf(stderr, " %u", c);
    fprintf(stderr, "\n");
  }
  return SS_OK;
}

// Index::search_lexical_batch: query i = terms[term_off[i] .. term_off[i+1]); out arrays [n][k]; device_exchange != 0 first calls
// Index::enable_device_exchange and returns its code (negative) on failure
int ssh_index_search_lexical_batch(ssh_index* ix, uint32_t n, const uint32_t* terms, const uint32_t* term_off, uint32_t query_type,
                                   uint32_t k, uint32_t result_type, int device_exchange, uint64_t* out_doc, float* out_score,
                                   uint32_t* out_count, uint64_t* out_total) {
  if (device_exchange) {
    const int rc = ix->index->enable_device_exchange();
    if (rc != SS_OK) return rc;
  }
  std::vector<std::vector<uint32_t>> q(n);
  for (uint32_t i = 0; i < n; i++) q[i].assign(terms + term_off[i], terms + term_off[i + 1]);
  std::vector<ResultObject> r = ix->index->search_lexical_batch(q, (QueryType)query_type, k, (ResultType)result_type);
  int err = 0;
  for (uint32_t i = 0; i < n; i++) {
    out_count[i] = (uint32_t)r[i].results.size();
    out_total[i] = r[i].result_count_total;
    if (r[i].last_error) err = r[i].last_error;
    for (size_t j = 0; j < r[i].results.size() && j < k; j++) {
      out_doc[(size_t)i * k + j] = r[i].results[j].doc_id;
      out_score[(size_t)i * k + j] = r[i].results[j].score;
    }
  }
  return err;
}

}  // extern "C"
