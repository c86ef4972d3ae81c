// Flat C shim over the C++ host mirror (seekstorm_host.hpp) so that the parity tests can drive it through ctypes.
// Test plumbing only: a C++ application links seekstorm_host.hpp directly.
#include <cstring>

#include "seekstorm_host.hpp"

using namespace seekstorm;

struct ssh_index {
  std::vector<std::shared_ptr<Shard>> shards;
  std::unique_ptr<Index> index;
  std::vector<std::unique_ptr<VectorBatchCoalescer>> coalescers;
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

// n concurrent single-query vector searches through a VectorBatchCoalescer (one submitting thread per query);
// out arrays are [n][length]; returns the number of device batches used
int ssh_coalesced_vector_search(ssh_index* ix, int shard, uint32_t n, const float* queries, uint32_t length,
                                uint32_t max_batch, uint32_t max_wait_us, uint64_t* out_doc, float* out_score,
                                uint32_t* out_count) {
  Shard& sh = *ix->shards[shard];
  VectorBatchCoalescer co(ix->shards[shard], max_batch, max_wait_us);
  const uint32_t dim = sh.dim();
  std::vector<std::future<ResultObject>> fut(n);
  std::vector<std::thread> th;
  for (uint32_t i = 0; i < n; i++)
    th.emplace_back([&, i] { fut[i] = co.submit(std::vector<float>(queries + (size_t)i * dim, queries + (size_t)(i + 1) * dim), length, nullptr); });
  for (auto& t : th) t.join();
  for (uint32_t i = 0; i < n; i++) {
    ResultObject ro = fut[i].get();
    out_count[i] = (uint32_t)ro.results.size();
    for (size_t j = 0; j < ro.results.size() && j < length; j++) {
      out_doc[(size_t)i * length + j] = ro.results[j].doc_id;
      out_score[(size_t)i * length + j] = ro.results[j].score;
    }
  }
  return (int)co.batches_submitted();
}

// n concurrent single-query lexical searches through a LexicalBatchCoalescer; query i = terms[term_off[i] .. term_off[i+1])
// followed by its NOT terms not_terms[not_off[i] .. not_off[i+1]); out arrays are [n][length]
int ssh_coalesced_lexical_search(ssh_index* ix, int shard, uint32_t n, const uint32_t* terms, const uint32_t* term_off,
                                 const uint32_t* not_terms, const uint32_t* not_off, uint32_t query_type, uint32_t offset,
                                 uint32_t length, uint32_t result_type, uint32_t max_batch, uint32_t max_wait_us,
                                 uint64_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total) {
  LexicalBatchCoalescer co(ix->shards[shard], max_batch, max_wait_us);
  std::vector<std::future<ResultObject>> fut(n);
  std::vector<std::thread> th;
  for (uint32_t i = 0; i < n; i++)
    th.emplace_back([&, i] {
      std::vector<uint32_t> t(terms + term_off[i], terms + term_off[i + 1]);
      std::vector<uint32_t> nt;
      if (not_terms) nt.assign(not_terms + not_off[i], not_terms + not_off[i + 1]);
      fut[i] = co.submit(t, (QueryType)query_type, offset + (i % 3), length, (ResultType)result_type, nt);  // mixed offsets
    });
  for (auto& t : th) t.join();
  for (uint32_t i = 0; i < n; i++) {
    ResultObject ro = fut[i].get();
    out_count[i] = (uint32_t)ro.results.size();
    out_total[i] = ro.result_count_total;
    for (size_t j = 0; j < ro.results.size() && j < length; j++) {
      out_doc[(size_t)i * length + j] = ro.results[j].doc_id;
      out_score[(size_t)i * length + j] = ro.results[j].score;
    }
  }
  return (int)co.batches_submitted();
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
