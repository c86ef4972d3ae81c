// C++ host mirror of the reference's search interface for the query hot path (see seekstorm_host.hpp).
#include "seekstorm_host.hpp"

#include <algorithm>
#include <string>
#include <limits>
#include <unordered_map>
#include <chrono>
#include <cmath>
#include <cstring>

namespace seekstorm {

// ------------------------------------------------------------------ scalar pieces
float idf(uint64_t indexed_doc_count, uint64_t posting_count) {
  // search.rs:3225-3230: (((N - n + 0.5) / (n + 0.5)) + 1).ln(), every operand f32
  const float N = (float)indexed_doc_count, n = (float)posting_count;
  return std::log(((N - n + 0.5f) / (n + 0.5f)) + 1.0f);
}

void normalize_f32(float* v, size_t n) {
  // vector_similarity.rs:70-74: sequential f32 sum of squares, then multiply by 1 / sqrt(sum)
  float s = 0.f;
  for (size_t i = 0; i < n; i++) s += v[i] * v[i];
  const float f = 1.0f / std::sqrt(s);
  for (size_t i = 0; i < n; i++) v[i] *= f;
}

void quantize_f32_to_i8(const float* v, size_t n, int8_t* out) {
  // vector_similarity.rs:1226-1232: (v * 127.0).round().clamp(-127.0, 127.0) as i8; f32::round = half away from zero
  for (size_t i = 0; i < n; i++) out[i] = (int8_t)std::fmin(std::fmax(std::round(v[i] * 127.0f), -127.0f), 127.0f);
}

size_t turboquant_dim(size_t n) {  // TurboQuant::next_power_of_two, vector_similarity.rs:1836-1842
  size_t d = 1;
  while (d < n) d <<= 1;
  return d;
}

void turboquant_f32_to_i8(const float* v, size_t n, const float* seed_mask, size_t dim, bool avx2, int8_t* out, float* scale_out,
                          float* norm_out) {
  // TurboQuant::quantize_f32_i8 / _avx2 (vector_similarity.rs:1927-1983): pad, sign mask, FWHT (1861-1925), scale (2011-2039),
  // round to i8, norm = sum q^2 * scale^2
  std::vector<float> a(dim, 0.0f);
  for (size_t i = 0; i < std::min(n, dim); i++) a[i] = v[i];
  for (size_t i = 0; i < dim; i++) a[i] *= seed_mask[i];
  for (size_t h = 1; h < dim; h *= 2)
    for (size_t i = 0; i < dim; i += 2 * h)
      for (size_t j = i; j < i + h; j++) {
        const float x = a[j], y = a[j + h];
        a[j] = x + y;
        a[j + h] = x - y;
      }
  const float nrm = std::sqrt((float)dim);
  for (size_t i = 0; i < dim; i++) a[i] /= nrm;
  float sum_sq = 0.0f;
  if (avx2 && dim >= 8) {  // eight lanes (mul, then add), folded 4+4, 2+2, 1+1 (horizontal_sum_avx2)
    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i + 8 <= dim; i += 8)
      for (int j = 0; j < 8; j++) {
        const float p = a[i + j] * a[i + j];
        l[j] = l[j] + p;
      }
    const float x0 = l[4] + l[0], x1 = l[5] + l[1], x2 = l[6] + l[2], x3 = l[7] + l[3];
    const float y0 = x0 + x2, y1 = x1 + x3;
    sum_sq = y0 + y1;
  } else {
    for (size_t i = 0; i < dim; i++) sum_sq += a[i] * a[i];
  }
  float scale = std::sqrt(sum_sq) / std::sqrt((float)dim) / 32.0f;
  if (!(scale > 1e-8f)) scale = 1e-8f;
  const float inv = 1.0f / scale;
  int32_t sq = 0;
  for (size_t i = 0; i < dim; i++) {
    int32_t q;
    if (avx2 && dim >= 16) {  // quantize_avx2 (1252-1289): reciprocal, + copysign(0.5), truncate, saturating packs
      const float s = a[i] * inv;
      q = (int32_t)(s + (std::signbit(s) ? -0.5f : 0.5f));
      q = std::min(127, std::max(-128, q));
    } else {
      q = (int32_t)std::fmin(std::fmax(std::round(a[i] / scale), -127.0f), 127.0f);
    }
    out[i] = (int8_t)q;
    sq += q * q;
  }
  *scale_out = scale;
  *norm_out = (float)sq * scale * scale;
}

static const float kSimilarityNormalization64I8 = 1.0f / 16129.0f;  // vector.rs:29

float threshold_raw(const float* similarity_threshold, bool euclidean) {
  if (!similarity_threshold) return -3.4028234663852886e38f;
  if (euclidean) return -*similarity_threshold;                                    // vector.rs:398
  return ((*similarity_threshold * 2.0f) - 1.0f) / kSimilarityNormalization64I8;  // vector.rs:388-397
}

float vector_score_of(float raw_dot) { return ((raw_dot * kSimilarityNormalization64I8) + 1.0f) / 2.0f; }  // vector.rs:1495-1499

// ------------------------------------------------------------------ Shard
Shard::Shard(int device, uint32_t shard_id) : shard_id_(shard_id), device_(device) {
  create_rc_ = ss_shard_create(device, &h_);
  if (create_rc_ != SS_OK) h_ = nullptr;
}

Shard::Shard(ss_shard* borrowed, int device, uint32_t shard_id) : h_(borrowed), owns_(false), shard_id_(shard_id), device_(device) {
  if (!h_) { create_rc_ = SS_EINVAL; return; }
  uint64_t n = 0;
  uint32_t d = 0;
  if (ss_bm25_info(h_, &n, nullptr, nullptr, nullptr) == SS_OK) n_docs_ = n;
  if (ss_vec_info(h_, &n, &d) == SS_OK) { n_rows_ = n; dim_ = d; }
}

Shard::~Shard() {
  if (h_ && owns_) ss_shard_destroy(h_);
}

int Shard::upload_lexical(uint64_t n_docs, const uint8_t* doclen_bytes, uint32_t n_terms, const uint64_t* term_offsets,
                          const uint32_t* doc_ids, const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  lexical_fields_ = 1; ngram_components_.clear(); ngram_component_.clear(); ngram_component_df_.clear();  // a new image: no index.bin n-gram keys behind the term ids
  const int rc = positions ? ss_bm25_upload_positions(h_, n_docs, doclen_bytes, n_terms, term_offsets, doc_ids, tfs, positions, n_positions)
                           : ss_bm25_upload(h_, n_docs, doclen_bytes, n_terms, term_offsets, doc_ids, tfs);
  n_docs_ = rc == SS_OK ? n_docs : 0;
  return rc;
}

int Shard::upload_lexical_fields(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen_bytes, const float* boost, uint32_t n_terms,
                                 const uint64_t* term_offsets, const uint32_t* doc_ids, const uint8_t* field_ids,
                                 const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  lexical_fields_ = n_fields; ngram_components_.clear(); ngram_component_.clear(); ngram_component_df_.clear();  // a new image: no index.bin n-gram keys behind the term ids
  const int rc = positions ? ss_bm25_upload_fields_positions(h_, n_docs, n_fields, doclen_bytes, boost, n_terms, term_offsets, doc_ids,
                                                             field_ids, tfs, positions, n_positions)
                           : ss_bm25_upload_fields(h_, n_docs, n_fields, doclen_bytes, boost, n_terms, term_offsets, doc_ids, field_ids, tfs);
  n_docs_ = rc == SS_OK ? n_docs : 0;
  return rc;
}

int Shard::synth_lexical(uint64_t seed, uint64_t n_docs, uint32_t n_terms, const uint32_t* thresh32, const uint8_t* len_table1024, uint32_t n_shards) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  lexical_fields_ = 1; ngram_components_.clear(); ngram_component_.clear(); ngram_component_df_.clear();
  int rc = ss_synth_set_partition(h_, shard_id_, n_shards);
  if (rc == SS_OK) rc = ss_bm25_synth(h_, seed, n_docs, n_terms, thresh32, len_table1024);
  n_docs_ = rc == SS_OK ? n_docs : 0;
  return rc;
}

int Shard::synth_vectors(uint64_t seed, uint64_t n_rows, uint32_t dim, uint32_t n_shards) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  i8_ = false;
  int rc = ss_synth_set_partition(h_, shard_id_, n_shards);
  if (rc == SS_OK) rc = ss_vec_synth(h_, seed, n_rows, dim);
  n_rows_ = rc == SS_OK ? n_rows : 0;
  dim_ = rc == SS_OK ? dim : 0;
  return rc;
}

int Shard::set_vector_similarity(bool euclidean) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  const int rc = ss_vec_set_similarity(h_, euclidean ? SS_SIM_EUCLIDEAN : SS_SIM_DOT);
  if (rc == SS_OK) euclidean_ = euclidean;
  return rc;
}

int Shard::upload_vectors(uint64_t n_rows, uint32_t dim, const float* rows, const uint32_t* row_doc_ids) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  i8_ = false;
  const int rc = ss_vec_upload(h_, n_rows, dim, rows, row_doc_ids);
  n_rows_ = rc == SS_OK ? n_rows : 0;
  dim_ = rc == SS_OK ? dim : 0;
  return rc;
}

int Shard::open_index_bin(const uint8_t* bytes, uint64_t len, uint32_t key_head_size, std::vector<uint64_t>* term_keys,
                          bool with_positions) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  lexical_fields_ = 1;
  ss_index_bin* ix = nullptr;
  int rc = ss_index_bin_open(bytes, len, 1, key_head_size, 11, &ix);  // open_index always uses 2048 segments (index.rs:3285)
  if (rc) return rc;
  uint64_t n_docs = 0;
  uint32_t n_terms = 0;
  ss_index_bin_info(ix, &n_docs, nullptr, nullptr, &n_terms, nullptr);
  if (term_keys) {
    term_keys->assign(n_terms, 0);
    if (n_terms) ss_index_bin_term_keys(ix, term_keys->data());
  }
  // n-gram keys: one term id per component; a component's idf comes from the component TERM's posting count in the key
  // head, not from the n-gram's own list (search.rs:3231-3262) -- remembered here, applied by make_query
  ngram_components_.assign(n_terms, 1);
  ngram_component_.assign(n_terms, 0);
  ngram_component_df_.assign(n_terms, 0);
  if (n_terms) ss_index_bin_term_ngram(ix, ngram_components_.data(), ngram_component_.data(), ngram_component_df_.data());
  rc = with_positions ? ss_bm25_upload_index_bin_positions(h_, ix) : ss_bm25_upload_index_bin(h_, ix);
  ss_index_bin_close(ix);
  n_docs_ = rc == SS_OK ? n_docs : 0;
  if (rc != SS_OK) { ngram_components_.clear(); ngram_component_.clear(); ngram_component_df_.clear(); }
  return rc;
}

int Shard::upload_vectors_i8(uint64_t n_rows, uint32_t dim, const int8_t* rows, const float* row_scale, const uint32_t* row_doc_ids) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  const int rc = ss_vec_upload_i8(h_, n_rows, dim, rows, row_scale, row_doc_ids);
  n_rows_ = rc == SS_OK ? n_rows : 0;
  dim_ = rc == SS_OK ? dim : 0;
  i8_ = rc == SS_OK;
  return rc;
}

int Shard::open_vector_bin(const uint8_t* bytes, uint64_t len, uint32_t dim, bool i8, bool use_record_scale) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  const int rc = i8 ? ss_vec_upload_vector_bin_i8(h_, bytes, len, dim, use_record_scale ? 1 : 0) : ss_vec_upload_vector_bin(h_, bytes, len, dim);
  i8_ = rc == SS_OK && i8;
  uint64_t n = 0;
  uint32_t d = 0;
  if (rc == SS_OK) ss_vec_info(h_, &n, &d);
  n_rows_ = n;
  dim_ = d;
  return rc;
}

int Shard::commit_level(uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms, uint32_t n_dense_terms,
                        const uint64_t* term_offsets, const uint32_t* doc_ids, const uint16_t* tfs, const uint16_t* positions,
                        uint64_t n_positions, const uint16_t* npos) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  if (!term_offsets || n_dense_terms == 0 || n_dense_terms > n_terms) return SS_EINVAL;
  // the CSR is in id order: the dense terms' part and the rare terms' part are its two ends, the positions pool splits where the
  // dense postings' positions end
  const uint64_t cut = term_offsets[n_dense_terms];
  uint64_t p_cut = 0;
  if (positions)
    for (uint64_t j = term_offsets[0]; j < cut; j++) p_cut += npos ? npos[j] : tfs[j];
  if (positions && p_cut > n_positions) return SS_EINVAL;
  int rc = positions ? ss_bm25_append_level_positions(h_, level, n_level_docs, level_doclen, n_dense_terms, term_offsets, doc_ids, tfs, npos, positions, p_cut)
                     : ss_bm25_append_level(h_, level, n_level_docs, level_doclen, n_dense_terms, term_offsets, doc_ids, tfs);
  if (rc != SS_OK) return rc;
  lexical_fields_ = 1;
  n_docs_ = (uint64_t)level * 65536u + n_level_docs;
  if (n_dense_terms < n_terms)
    rc = ss_bm25_append_sparse_level(h_, level, n_terms - n_dense_terms, term_offsets + n_dense_terms, doc_ids, tfs, npos,
                                     positions ? positions + p_cut : nullptr, positions ? n_positions - p_cut : 0);
  return rc;
}

int Shard::set_deleted(const uint64_t* doc_ids, uint64_t n) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  const int rc = ss_set_deleted(h_, doc_ids, n);
  if (rc == SS_OK) n_deleted_ = n;
  return rc;
}

int Shard::synth_lexical(uint64_t seed, uint64_t n_docs, uint32_t n_terms, const uint32_t* thresh32,
                         const uint8_t* len_table1024) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  lexical_fields_ = 1; ngram_components_.clear(); ngram_component_.clear(); ngram_component_df_.clear();  // a new image: no index.bin n-gram keys behind the term ids
  const int rc = ss_bm25_synth(h_, seed, n_docs, n_terms, thresh32, len_table1024);
  n_docs_ = rc == SS_OK ? n_docs : 0;
  return rc;
}

int Shard::synth_vectors(uint64_t seed, uint64_t n_rows, uint32_t dim) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  const int rc = ss_vec_synth(h_, seed, n_rows, dim);
  n_rows_ = rc == SS_OK ? n_rows : 0;
  dim_ = rc == SS_OK ? dim : 0;
  return rc;
}

int Shard::make_query(const std::vector<uint32_t>& terms, QueryType qt, ss_bm25_query* out,
                      const std::vector<uint32_t>& not_terms, const std::vector<uint16_t>& field_filter) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  std::vector<uint32_t> uniq, nots;  // unique_terms in first-seen order (search.rs:3023)
  for (uint32_t t : terms)
    if (std::find(uniq.begin(), uniq.end(), t) == uniq.end()) uniq.push_back(t);
  for (uint32_t t : not_terms)  // not_query_list: the "-term" operands (add_result.rs:3440-3497)
    if (std::find(uniq.begin(), uniq.end(), t) == uniq.end() && std::find(nots.begin(), nots.end(), t) == nots.end())
      nots.push_back(t);
  if (uniq.empty()) return SS_EINVAL;
  // more than 32 unique terms (NOT terms included): the crate answers such a query -- union_scan_32 ranks, block by block, the 32 lists
  // with the largest block maxima and union_count recounts (union.rs:233-259, 617-624) -- this library's query record does not hold it:
  // the host's own dispatch keeps it (ResultObject::cpu_dispatch), it is not an invalid query
  if (uniq.size() + nots.size() > SS_MAX_QUERY_TERMS) return SS_ENOTSUP;
  uint64_t df[SS_MAX_QUERY_TERMS];
  const int rc = ss_bm25_term_df(h_, (uint32_t)uniq.size(), uniq.data(), df);
  if (rc != SS_OK) return rc;
  std::memset(out, 0, sizeof(*out));
  out->n_terms = (uint32_t)uniq.size();
  if (qt == QueryType::Phrase) {  // non_unique_query_list: the entries in order, each naming its unique term (search.rs:3304-3331)
    // an n-gram key of an opened index.bin arrives as its component term ids, consecutive: ONE entry, at the place of its first
    // word (its first component carries the key's positions), spanning 2 / 3 places -- the places of its other words carry no
    // entry (term_index_nonunique = entries before + preceding_ngram_count, search.rs:3305-3328)
    size_t entries = 0;
    for (size_t i = 0; i < terms.size(); i++)
      if (!(terms[i] < ngram_component_.size() && ngram_component_[terms[i]] != 0)) entries++;
    if (entries < 2) qt = QueryType::Intersection;  // a one-entry phrase is a term query
    else if (terms.size() > SS_MAX_PHRASE) return SS_EINVAL;
    else {
      out->phrase_len = (uint32_t)terms.size();
      for (size_t i = 0; i < terms.size(); i++) {
        const bool inner = terms[i] < ngram_component_.size() && ngram_component_[terms[i]] != 0;  // 2nd / 3rd component of a key
        if (inner && i == 0) return SS_EINVAL;
        out->phrase_seq[i] = inner ? (uint8_t)SS_PHRASE_SKIP : (uint8_t)(std::find(uniq.begin(), uniq.end(), terms[i]) - uniq.begin());
      }
    }
  }
  uint32_t fmask = 0;
  for (uint16_t f : field_filter) {
    if (f >= 15) return SS_EINVAL;
    fmask |= 1u << f;
  }
  out->op = (uint32_t)qt | SS_OP_NOT_TERMS(nots.size()) | SS_OP_FIELD_FILTER(fmask);
  for (size_t i = 0; i < uniq.size(); i++) {
    out->term[i] = uniq[i];
    const bool ngram = uniq[i] < ngram_components_.size() && ngram_components_[uniq[i]] > 1;
    out->idf[i] = idf(n_docs_, ngram ? ngram_component_df_[uniq[i]] : df[i]);  // idf_ngram_i for a component of an n-gram key
  }
  for (size_t i = 0; i < nots.size(); i++) out->term[uniq.size() + i] = nots[i];
  return SS_OK;
}

int Shard::upload_facets(uint64_t n_docs, uint32_t record_size, const uint8_t* records) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  return ss_facet_upload(h_, n_docs, record_size, records);
}

// all_terms_frequent (intersection.rs:198-209), evaluated where the reference evaluates it -- on the host, per query:
// N > top_k << 8 and posting_count / N >= 0.5 (f32) for every term of an intersection of several terms -> the query is
// marked and ranks only docs whose every tf >= 10 (several indexed fields: the tf in the doc's lowest field).  Not under a field filter
// (add_result.rs:3545) -- nor under a facet filter (add_result.rs:2096-2100), which the caller knows about.
bool Shard::mark_all_terms_frequent(ss_bm25_query* q, size_t top_k) const {
  if (!h_ || !(n_docs_ > ((uint64_t)top_k << 8))) return false;
  if (lexical_fields_ != 1) {  // several indexed fields: over the image's merged lists (add_result.rs:1595-1607)
    uint32_t merged = 0;
    if (ss_bm25_fields_info(h_, nullptr, &merged, nullptr) != SS_OK || !merged) return false;
  }
  if ((q->op & 0xFFu) != SS_OP_INTERSECTION || q->n_terms < 2 || ((q->op >> 16) & 0x7FFFu)) return false;
  uint64_t df[SS_MAX_QUERY_TERMS];
  if (ss_bm25_term_df(h_, q->n_terms, q->term, df) != SS_OK) return false;
  for (uint32_t t = 0; t < q->n_terms; t++)
    if (!((float)df[t] / (float)n_docs_ >= 0.5f)) return false;
  q->op |= SS_OP_ALL_TERMS_FREQUENT;
  return true;
}

std::vector<ResultObject> Shard::search_lexical_batch(const std::vector<ss_bm25_query>& queries, size_t k,
                                                      ResultType result_type, const std::vector<ss_facet_filter>& facet_filter,
                                                      bool mark_frequent) {
  const size_t nq = queries.size();
  std::vector<ResultObject> out(nq);
  if (nq == 0) return out;
  const size_t kk = std::max<size_t>(k, 1);
  std::vector<uint32_t> doc(nq * kk), cnt(nq);
  std::vector<float> score(nq * kk);
  std::vector<uint64_t> tot(nq);
  // all_terms_frequent: marked per query unless the caller already did (the coalescer marks every request with its OWN
  // offset + length) or the call carries a facet filter, which disables the shortcut (add_result.rs:2096-2100)
  std::vector<ss_bm25_query> marked;
  const ss_bm25_query* qp = queries.data();
  if (mark_frequent && facet_filter.empty() && result_type != ResultType::Count) {
    for (size_t q = 0; q < nq; q++) {
      ss_bm25_query m = queries[q];
      if (!mark_all_terms_frequent(&m, k)) continue;
      if (marked.empty()) marked = queries;
      marked[q] = m;
    }
    if (!marked.empty()) qp = marked.data();
  }
  const int rc = h_ ? ss_bm25_search_filtered(h_, (uint32_t)nq, qp, (uint32_t)k, (uint32_t)result_type,
                                              (uint32_t)facet_filter.size(), facet_filter.empty() ? nullptr : facet_filter.data(),
                                              doc.data(), score.data(), cnt.data(), tot.data())
                    : (create_rc_ ? create_rc_ : SS_ESTATE);
  for (size_t q = 0; q < nq; q++) {
    ResultObject& ro = out[q];
    if (rc != SS_OK) { ro.last_error = rc; continue; }  // degrade to empty (search.rs:2461-2463)
    const size_t n = result_type == ResultType::Count ? 0 : cnt[q];
    ro.results.resize(n);
    for (size_t i = 0; i < n; i++) {
      Result& r = ro.results[i];
      r.doc_id = doc[q * kk + i];
      r.score = score[q * kk + i];
      r.lexical_score = r.score;
      r.shard_id = shard_id_;
      r.level_id = (uint32_t)(r.doc_id >> 16);  // 65 536-doc levels (index.rs:115)
      r.source = ResultSource::Lexical;
    }
    ro.result_count = n;
    ro.result_count_total = tot[q];
  }
  return out;
}

int Shard::set_clusters(const std::vector<uint32_t>& level_clusters, const std::vector<uint32_t>& child_count) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  uint64_t nc = 0;
  for (uint32_t c : level_clusters) nc += c;
  if (nc != child_count.size()) return SS_EINVAL;
  return ss_vec_set_clusters(h_, (uint32_t)level_clusters.size(), level_clusters.data(), child_count.data());
}

int Shard::set_fields(const std::vector<uint16_t>& row_field) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  return ss_vec_set_fields(h_, row_field.size(), row_field.data());
}

std::vector<ResultObject> Shard::search_vector_batch(const float* query_vectors, size_t n_queries, size_t k,
                                                     const float* similarity_threshold, const AnnMode& ann_mode,
                                                     const std::vector<uint16_t>& field_filter) {
  std::vector<ResultObject> out(n_queries);
  if (n_queries == 0) return out;
  const size_t kk = std::max<size_t>(k, 1);
  std::vector<uint32_t> doc(n_queries * kk), cnt(n_queries);
  std::vector<float> score(n_queries * kk);
  std::vector<uint64_t> tot(n_queries);
  std::vector<uint32_t> ncl(n_queries * 3, 0);  // SS_ANN_REPORT_OBSERVED: clusters, observed records (low, high) per query
  int rc = create_rc_ ? create_rc_ : SS_ESTATE;
  // vector.rs:1300-1307: (n_probe, cluster threshold) of the mode; the threshold goes through TopK::new like the record one
  ss_ann_mode am{0u, threshold_raw(nullptr), 0ull, 0u, 0u};
  const bool euc = euclidean_;
  const bool ann = ann_mode.kind != AnnMode::Kind::All;
  bool bad_field = false;
  for (uint16_t f : field_filter) {
    if (f < 64) am.field_mask |= 1ull << f;
    else bad_field = true;
  }
  // observed_vector_count is counted on the device wherever it is not simply the record count: ANN modes, a field filter,
  // tombstones (TopK::push is only handed live records of listed fields, vector.rs:1397-1400, 1450-1452).  AnnMode::All on a
  // shard without tombstones keeps mode == NULL: those calls coalesce behind the ABI.
  const bool opts = ann || am.field_mask != 0 || n_deleted_ != 0;
  if (opts) am.flags = SS_ANN_REPORT_OBSERVED;
  if (ann_mode.kind == AnnMode::Kind::Nprobe || ann_mode.kind == AnnMode::Kind::NprobeSimilaritythreshold)
    am.n_probe = (uint32_t)std::min<size_t>(ann_mode.n_probe, 0xFFFFFFFFu);
  if (ann_mode.kind == AnnMode::Kind::Similaritythreshold || ann_mode.kind == AnnMode::Kind::NprobeSimilaritythreshold)
    am.cluster_threshold_raw = threshold_raw(&ann_mode.similarity_threshold, euc);
  if ((ann && ann_mode.kind != AnnMode::Kind::Similaritythreshold && am.n_probe == 0) || bad_field) {
    rc = bad_field ? SS_ENOTSUP : SS_EINVAL;  // Nprobe(0): TopK::new(0, ..) has no slot to push into
  } else if (h_ && i8_) {  // the query is quantised like the records (search.rs:1487-1490); score = raw integer dot
    std::vector<int8_t> q8(n_queries * dim_);
    quantize_f32_to_i8(query_vectors, q8.size(), q8.data());
    rc = ss_vec_search_i8_ann(h_, (uint32_t)n_queries, q8.data(), nullptr, (uint32_t)k, threshold_raw(similarity_threshold, euc),
                              opts ? &am : nullptr, doc.data(), score.data(), cnt.data(), tot.data(), ncl.data());
  } else if (h_) {
    rc = ss_vec_search_ann(h_, (uint32_t)n_queries, query_vectors, (uint32_t)k, threshold_raw(similarity_threshold, euc),
                           opts ? &am : nullptr, doc.data(), score.data(), cnt.data(), tot.data(), ncl.data());
  }
  uint32_t all_clusters = 0;
  if (h_ && !ann) (void)ss_vec_cluster_info(h_, nullptr, &all_clusters);
  for (size_t q = 0; q < n_queries; q++) {
    ResultObject& ro = out[q];
    if (rc != SS_OK) { ro.last_error = rc; continue; }  // degrade to empty (vector.rs:1222-1224)
    const size_t n = cnt[q];
    ro.results.resize(n);
    for (size_t i = 0; i < n; i++) {
      Result& r = ro.results[i];
      r.doc_id = doc[q * kk + i];
      r.score = score[q * kk + i];              // raw similarity: dot, or -distance^2 under Euclidean (vector.rs:1474-1506)
      r.vector_score = euc ? -r.score : vector_score_of(r.score);  // vector.rs:1495-1499
      r.shard_id = shard_id_;
      r.level_id = (uint32_t)(r.doc_id >> 16);
      r.source = ResultSource::Vector;
    }
    ro.result_count = n;
    ro.result_count_total = tot[q];
    ro.observed_vector_count = opts ? ((uint64_t)ncl[3 * q + 1] | ((uint64_t)ncl[3 * q + 2] << 32))
                                    : n_rows_;  // AnnMode::All, nothing filtered: every record is pushed (vector.rs:421)
    if (!ann) ro.observed_cluster_count = all_clusters ? all_clusters : 1;
    else ro.observed_cluster_count = ncl[3 * q];  // vector.rs:1394
  }
  return out;
}

std::vector<uint8_t> string_facet_rank_column(const uint8_t* records, uint64_t n_docs, uint32_t record_size, uint32_t facet_offset,
                                              uint32_t facet_type, const std::vector<std::string>& strings, uint32_t* rank_offset) {
  const uint32_t width = facet_type == SS_FACET_STRING16 ? 2u : 4u;
  std::vector<std::string> order(strings);
  std::sort(order.begin(), order.end());  // std::string compares bytes as unsigned chars: Rust's String order
  order.erase(std::unique(order.begin(), order.end()), order.end());
  std::vector<uint32_t> rank(strings.size());
  for (size_t i = 0; i < strings.size(); i++)
    rank[i] = (uint32_t)(std::lower_bound(order.begin(), order.end(), strings[i]) - order.begin());
  std::vector<uint8_t> out((size_t)n_docs * (record_size + 4u));
  for (uint64_t d = 0; d < n_docs; d++) {
    const uint8_t* r = records + d * record_size;
    uint32_t id = 0;
    for (uint32_t b = 0; b < width; b++) id |= (uint32_t)r[facet_offset + b] << (8u * b);
    const uint32_t rk = rank.empty() ? 0u : rank[std::min<size_t>(id, rank.size() - 1)];
    uint8_t* o = out.data() + d * (record_size + 4u);
    std::memcpy(o, r, record_size);
    for (uint32_t b = 0; b < 4; b++) o[record_size + b] = (uint8_t)(rk >> (8u * b));
  }
  if (rank_offset) *rank_offset = record_size;
  return out;
}

ss_facet_filter point_facet_filter(uint32_t facet_offset, const double base[2], double lo, double hi, uint32_t unit, uint32_t flags) {
  ss_facet_filter f;
  std::memset(&f, 0, sizeof(f));
  f.offset = facet_offset;
  f.type = SS_FACET_POINT;
  std::memcpy(&f.lo, &lo, 8);
  std::memcpy(&f.hi, &hi, 8);
  std::memcpy(&f.values[0], &base[0], 8);
  std::memcpy(&f.values[2], &base[1], 8);
  f.n_values = unit;
  f.reserved = flags;
  return f;
}

int Shard::facet_count(const ss_bm25_query& query, uint32_t facet_offset, uint32_t facet_type, uint32_t n_buckets,
                       const std::vector<uint64_t>& range_lower_bounds, std::vector<uint64_t>* counts, uint64_t* total,
                       const std::vector<ss_facet_filter>& facet_filter, const ss_facet_point* base) {
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  const bool strings = facet_type == SS_FACET_STRING16 || facet_type == SS_FACET_STRING32;
  const uint32_t nb = strings ? n_buckets : (uint32_t)range_lower_bounds.size();
  if (!counts || nb == 0) return SS_EINVAL;
  counts->assign((size_t)nb + 1, 0);
  const ss_facet_filter* fp = facet_filter.empty() ? nullptr : facet_filter.data();
  if (facet_type == SS_FACET_POINT)
    return ss_bm25_facet_count_point(h_, &query, (uint32_t)facet_filter.size(), fp, facet_offset, base, nb, range_lower_bounds.data(),
                                     counts->data(), total);
  return ss_bm25_facet_count(h_, &query, (uint32_t)facet_filter.size(), fp, facet_offset, facet_type, nb,
                             strings ? nullptr : range_lower_bounds.data(), counts->data(), total);
}

// ---- result sort (search.rs ResultSort; ordering result_ordering_shard, min_heap.rs:574-1050): the k best matches under
// (field 1, field 2, ..., score).  Composed from ordinary searches around the sort's pivot -- the k-th best value of the first
// field among the matches (ss_bm25_facet_kth[_point]): "strictly better" holds fewer than k docs, which are fetched by a
// search filtered to that range and ordered here by their values; the "equal" ones recurse on the remaining fields.
namespace {
uint64_t order_key(uint64_t bits, uint32_t type, bool descending) {  // larger = better under the sort
  static const uint32_t width[] = {8, 16, 32, 64, 8, 16, 32, 64, 32, 64};
  const uint32_t nb = width[type];
  const uint64_t mask = nb == 64 ? ~0ull : ((1ull << nb) - 1ull), top = 1ull << (nb - 1);
  uint64_t k = bits & mask;
  if (type >= SS_FACET_I8 && type <= SS_FACET_I64) k ^= top;
  else if (type == SS_FACET_F32 || type == SS_FACET_F64) k = (k & top) ? (~k & mask) : (k | top);
  return descending ? k : (~k & mask);
}
uint64_t filter_bits(uint64_t v, uint32_t type) {  // stored bits -> ss_facet_filter's form (signed integers sign-extended)
  switch (type) {
    case SS_FACET_I8: return (uint64_t)(int64_t)(int8_t)v;
    case SS_FACET_I16: return (uint64_t)(int64_t)(int16_t)v;
    case SS_FACET_I32: return (uint64_t)(int64_t)(int32_t)v;
    default: return v;
  }
}
void type_range(uint32_t type, uint64_t* lo, uint64_t* hi) {  // smallest / largest value of the type in the filter's form
  static const uint32_t width[] = {8, 16, 32, 64, 8, 16, 32, 64, 32, 64};
  const uint32_t nb = width[type];
  if (type <= SS_FACET_U64) { *lo = 0; *hi = nb == 64 ? ~0ull : ((1ull << nb) - 1ull); }
  else if (type <= SS_FACET_I64) { *lo = (uint64_t)(-(int64_t)(1ull << (nb - 1))); *hi = (1ull << (nb - 1)) - 1ull; }
  else if (type == SS_FACET_F32) { *lo = 0xFF800000ull; *hi = 0x7F800000ull; }             // -inf, +inf
  else { *lo = 0xFFF0000000000000ull; *hi = 0x7FF0000000000000ull; }
}
ss_facet_filter bits_filter(uint32_t offset, uint32_t type, uint64_t lo, uint64_t hi, uint32_t flags) {
  ss_facet_filter f;
  std::memset(&f, 0, sizeof(f));
  f.offset = offset; f.type = type; f.lo = lo; f.hi = hi; f.reserved = flags;
  return f;
}
}  // namespace

int Shard::sort_keys(const std::vector<uint32_t>& doc_ids, const ResultSort& sf, std::vector<uint64_t>* keys) {
  keys->assign(doc_ids.size(), 0);
  if (doc_ids.empty()) return SS_OK;
  if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
  int rc;
  if (sf.facet_type == SS_FACET_POINT) {
    const ss_facet_point base{sf.base[0], sf.base[1], SS_POINT_SORTKEY, 0};
    rc = ss_facet_point_distances(h_, (uint32_t)doc_ids.size(), doc_ids.data(), sf.facet_offset, &base, keys->data());
  } else {
    if (sf.facet_type > SS_FACET_F64) return SS_ENOTSUP;
    rc = ss_facet_values(h_, (uint32_t)doc_ids.size(), doc_ids.data(), sf.facet_offset, sf.facet_type, keys->data());
  }
  if (rc != SS_OK) return rc;
  const uint32_t kt = sf.facet_type == SS_FACET_POINT ? (uint32_t)SS_FACET_F64 : sf.facet_type;
  for (uint64_t& x : *keys) x = order_key(x, kt, sf.descending);
  return SS_OK;
}

int Shard::sorted_topk(const ss_bm25_query& q, const ResultSort* sorts, size_t n_sorts, size_t k, std::vector<ss_facet_filter> filters,
                       std::vector<Result>* out, uint64_t* total, bool* have_total) {
  if (k == 0) return SS_OK;
  // round 4: one ABI call -- the pivots of every sort field are found on the device (ss_bm25_search_sorted); the composition below
  // remains for more sort fields than the call takes and as the recursion's own by-score leaf
  if (n_sorts >= 1 && n_sorts <= SS_MAX_SORT_FIELDS && !*have_total && out->empty()) {
    if (!h_) return create_rc_ ? create_rc_ : SS_ESTATE;
    std::vector<ss_result_sort> rs(n_sorts);
    for (size_t f = 0; f < n_sorts; f++) {
      rs[f].facet_offset = sorts[f].facet_offset; rs[f].facet_type = sorts[f].facet_type; rs[f].descending = sorts[f].descending ? 1u : 0u;
      rs[f].reserved = 0; rs[f].base_lat = sorts[f].base[0]; rs[f].base_lon = sorts[f].base[1];
    }
    std::vector<uint32_t> doc(k);
    std::vector<float> score(k);
    uint32_t cnt = 0;
    uint64_t tot = 0;
    const int rc = ss_bm25_search_sorted(h_, 1, &q, (uint32_t)n_sorts, rs.data(), (uint32_t)k, (uint32_t)filters.size(),
                                         filters.empty() ? nullptr : filters.data(), doc.data(), score.data(), &cnt, &tot);
    if (rc != SS_OK) return rc;
    *total = tot; *have_total = true;
    for (uint32_t i = 0; i < cnt; i++) { Result r; r.doc_id = doc[i]; r.score = r.lexical_score = score[i]; out->push_back(r); }
    return SS_OK;
  }
  if (n_sorts == 0) {  // by score
    ResultObject ro = std::move(search_lexical_batch({q}, k, ResultType::TopkCount, filters)[0]);
    if (ro.last_error != SS_OK) return ro.last_error;
    if (!*have_total) { *total = ro.result_count_total; *have_total = true; }
    out->insert(out->end(), ro.results.begin(), ro.results.end());
    return SS_OK;
  }
  if (filters.size() + 1 > SS_MAX_FACET_FILTERS) return SS_EINVAL;
  const ResultSort& s0 = sorts[0];
  const bool point = s0.facet_type == SS_FACET_POINT;
  const ss_facet_filter* fp = filters.empty() ? nullptr : filters.data();
  uint64_t v = 0, n_better = 0, n_equal = 0, tot = 0;
  int rc;
  if (point) {
    const ss_facet_point base{s0.base[0], s0.base[1], SS_POINT_SORTKEY, 0};
    rc = ss_bm25_facet_kth_point(h_, &q, (uint32_t)filters.size(), fp, s0.facet_offset, &base, s0.descending ? 1u : 0u, k, &v, &n_better,
                                 &n_equal, &tot);
  } else {
    rc = ss_bm25_facet_kth(h_, &q, (uint32_t)filters.size(), fp, s0.facet_offset, s0.facet_type, s0.descending ? 1u : 0u, k, &v,
                           &n_better, &n_equal, &tot);
  }
  if (rc != SS_OK) return rc;
  if (!*have_total) { *total = tot; *have_total = true; }
  if (tot == 0) return SS_OK;
  ss_facet_filter better, equal;
  if (point) {
    double pv;
    std::memcpy(&pv, &v, 8);
    const double inf = std::numeric_limits<double>::infinity();
    better = s0.descending ? point_facet_filter(s0.facet_offset, s0.base, pv, inf, SS_POINT_SORTKEY, SS_FACET_LO_EXCLUSIVE | SS_FACET_HI_INCLUSIVE)
                           : point_facet_filter(s0.facet_offset, s0.base, -inf, pv, SS_POINT_SORTKEY, 0);
    equal = point_facet_filter(s0.facet_offset, s0.base, pv, pv, SS_POINT_SORTKEY, SS_FACET_HI_INCLUSIVE);
  } else {
    const uint64_t vb = filter_bits(v, s0.facet_type);
    uint64_t lo_all, hi_all;
    type_range(s0.facet_type, &lo_all, &hi_all);
    better = s0.descending ? bits_filter(s0.facet_offset, s0.facet_type, vb, hi_all, SS_FACET_LO_EXCLUSIVE | SS_FACET_HI_INCLUSIVE)
                           : bits_filter(s0.facet_offset, s0.facet_type, lo_all, vb, 0);
    equal = bits_filter(s0.facet_offset, s0.facet_type, vb, vb, SS_FACET_HI_INCLUSIVE);
  }
  if (n_better) {
    std::vector<ss_facet_filter> fb = filters;
    fb.push_back(better);
    std::vector<Result> got;
    rc = sorted_topk(q, nullptr, 0, (size_t)n_better, fb, &got, total, have_total);
    if (rc != SS_OK) return rc;
    std::vector<uint32_t> docs(got.size());
    for (size_t i = 0; i < got.size(); i++) docs[i] = (uint32_t)got[i].doc_id;
    std::vector<std::vector<uint64_t>> keys(n_sorts, std::vector<uint64_t>(got.size()));
    for (size_t f = 0; f < n_sorts && !got.empty(); f++) {
      const ResultSort& sf = sorts[f];
      if (sf.facet_type == SS_FACET_POINT) {
        const ss_facet_point base{sf.base[0], sf.base[1], SS_POINT_SORTKEY, 0};
        rc = ss_facet_point_distances(h_, (uint32_t)docs.size(), docs.data(), sf.facet_offset, &base, keys[f].data());
      } else {
        rc = ss_facet_values(h_, (uint32_t)docs.size(), docs.data(), sf.facet_offset, sf.facet_type, keys[f].data());
      }
      if (rc != SS_OK) return rc;
      const uint32_t kt = sf.facet_type == SS_FACET_POINT ? (uint32_t)SS_FACET_F64 : sf.facet_type;
      for (uint64_t& x : keys[f]) x = order_key(x, kt, sf.descending);
    }
    std::vector<size_t> order(got.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
      for (size_t f = 0; f < n_sorts; f++)
        if (keys[f][a] != keys[f][b]) return keys[f][a] > keys[f][b];
      if (got[a].score != got[b].score) return got[a].score > got[b].score;
      return got[a].doc_id < got[b].doc_id;
    });
    for (size_t i : order) out->push_back(got[i]);
  }
  if (n_better < k && n_equal) {
    filters.push_back(equal);
    return sorted_topk(q, sorts + 1, n_sorts - 1, k - (size_t)n_better, filters, out, total, have_total);
  }
  return SS_OK;
}

ResultObject Shard::search_lexical_shard(const std::vector<uint32_t>& query_terms, QueryType query_type_default, size_t offset,
                                         size_t length, ResultType result_type, const std::vector<ss_facet_filter>& facet_filter,
                                         const std::vector<uint32_t>& not_terms, const std::vector<uint16_t>& field_filter,
                                         const std::vector<ResultSort>& result_sort) {
  ResultObject ro;
  // a union of several terms under a field filter goes down like any other query: per-term gating inside the scan kernels (<= 7
  // terms, round 3) or the reference's own sub-queries composed BEHIND the ABI (8 .. 10 terms, a sparse-tier term; round 6).
  {
    ss_bm25_query q;
    const int rc = make_query(query_terms, query_type_default, &q, not_terms, lexical_fields_ > 1 ? field_filter : std::vector<uint16_t>());
    if (rc != SS_OK) { ro.last_error = rc; return ro; }
    if (!result_sort.empty() && result_type != ResultType::Count) {
      uint64_t total = 0;
      bool have_total = false;
      const int rs = sorted_topk(q, result_sort.data(), result_sort.size(), offset + length, facet_filter, &ro.results, &total, &have_total);
      if (rs != SS_OK) { ro = ResultObject(); ro.last_error = rs; return ro; }
      ro.result_count_total = total;
      ro.result_count = ro.results.size();
    } else {
      ro = std::move(search_lexical_batch({q}, offset + length, result_type, facet_filter)[0]);
    }
  }
  if (offset) {  // drain offset (search.rs:3585-3593)
    ro.results.erase(ro.results.begin(), ro.results.begin() + std::min(offset, ro.results.size()));
    ro.result_count = ro.results.size();
  }
  return ro;
}

ResultObject Shard::search_vector_shard(const float* query_vector, size_t length, const float* similarity_threshold,
                                        const AnnMode& ann_mode, const std::vector<uint16_t>& field_filter) {
  if (!query_vector) return ResultObject();
  return std::move(search_vector_batch(query_vector, 1, length, similarity_threshold, ann_mode, field_filter)[0]);
}

ResultObject Index::search_lexical_sorted(const std::vector<uint32_t>& query_terms, QueryType query_type_default, size_t offset, size_t length,
                                          const std::vector<ResultSort>& result_sort, const std::vector<ss_facet_filter>& facet_filter,
                                          const std::vector<uint32_t>& not_terms) {
  ResultObject ro;
  const size_t S = shards_.size();
  if (S == 0 || result_sort.empty()) { ro.last_error = SS_EINVAL; return ro; }
  struct Row { std::vector<uint64_t> keys; Result r; };
  std::vector<Row> rows;
  for (size_t i = 0; i < S; i++) {
    Shard& sh = *shards_[i];
    ResultObject part = sh.search_lexical_shard(query_terms, query_type_default, 0, offset + length, ResultType::TopkCount, facet_filter,
                                                not_terms, {}, result_sort);
    if (part.last_error != SS_OK) { ro.last_error = part.last_error; continue; }  // a failing shard degrades to empty
    ro.result_count_total += part.result_count_total;
    std::vector<uint32_t> docs(part.results.size());
    for (size_t j = 0; j < docs.size(); j++) docs[j] = (uint32_t)part.results[j].doc_id;
    std::vector<std::vector<uint64_t>> keys(result_sort.size());
    bool ok = true;
    for (size_t f = 0; f < result_sort.size() && ok; f++) ok = sh.sort_keys(docs, result_sort[f], &keys[f]) == SS_OK;
    if (!ok) { ro.last_error = SS_EDEVICE; continue; }
    for (size_t j = 0; j < docs.size(); j++) {
      Row row;
      for (size_t f = 0; f < result_sort.size(); f++) row.keys.push_back(keys[f][j]);
      row.r = part.results[j];
      row.r.doc_id = row.r.doc_id * S + sh.shard_id();  // search.rs:1671
      rows.push_back(std::move(row));
    }
  }
  std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) {
    for (size_t f = 0; f < a.keys.size(); f++)
      if (a.keys[f] != b.keys[f]) return a.keys[f] > b.keys[f];
    if (a.r.score != b.r.score) return a.r.score > b.r.score;
    return a.r.doc_id < b.r.doc_id;
  });
  for (size_t j = offset; j < rows.size() && j < offset + length; j++) ro.results.push_back(rows[j].r);
  ro.result_count = ro.results.size();
  return ro;
}

// ------------------------------------------------------------------ Index::search
ResultObject Index::search(const std::vector<uint32_t>& query_terms, const float* query_vector, QueryType query_type_default,
                           SearchMode search_mode, size_t offset, size_t length, ResultType result_type,
                           const float* similarity_threshold, bool normalize_query, const AnnMode& ann_mode,
                           const std::vector<uint16_t>& vector_field_filter, const std::vector<ss_facet_filter>& facet_filter,
                           const std::vector<uint32_t>& not_terms, const std::vector<uint16_t>& lexical_field_filter) {
  ResultObject ro;
  const size_t S = shards_.size();
  if (S == 0) return ro;
  const bool want_lex = (search_mode == SearchMode::Lexical || search_mode == SearchMode::Hybrid) && !query_terms.empty();
  const bool want_vec = (search_mode == SearchMode::Vector || search_mode == SearchMode::Hybrid) && query_vector != nullptr;
  if (!comms_.empty() && want_lex && !want_vec && facet_filter.empty() && not_terms.empty() && lexical_field_filter.empty()) {
    // shards on different GPUs (enable_device_exchange): every shard task searches its shard and the lists are exchanged and
    // merged on the devices over xGMI (ss_bm25_search_sharded) -- offset / length are applied to the merged list, as
    // search.rs:2109-2119 applies them after the gather
    std::vector<ResultObject> r = search_lexical_batch({query_terms}, query_type_default, offset + length, result_type);
    ro = std::move(r[0]);
    if (offset) ro.results.erase(ro.results.begin(), ro.results.begin() + (ptrdiff_t)std::min(offset, ro.results.size()));
    if (ro.results.size() > length) ro.results.resize(length);
    ro.result_count = ro.results.size();
    return ro;
  }
  std::vector<float> qv;
  if (want_vec) {
    qv.assign(query_vector, query_vector + shards_[0]->dim());
    if (normalize_query) normalize_f32(qv.data(), qv.size());  // search.rs:1464-1475
  }
  bool f32_images = true;
  for (const auto& sh : shards_) f32_images = f32_images && !sh->vectors_are_i8() && sh->dim() == qv.size();
  if (!comms_.empty() && want_vec && (want_lex || search_mode == SearchMode::Vector) && f32_images && ann_mode.kind == AnnMode::Kind::All &&
      vector_field_filter.empty() && facet_filter.empty() && not_terms.empty() && lexical_field_filter.empty() &&
      result_type != ResultType::Count && length > 0) {
    // Vector / Hybrid over shards on different GPUs: every shard task runs its searches at (0, offset + length), the lists travel
    // in ONE all-gather and are merged -- Hybrid: fused by RRF over the cross-shard concatenations -- on the devices
    // (ss_vec_search_sharded / ss_hybrid_search_sharded; search.rs:1680-1689, 1723-1732, 1962-2035, 2098-2119).  A shard whose own
    // search fails still enters the exchange; every task then reports an error (SS_EPEER on the healthy ones).
    const bool hybrid = want_lex;
    const uint32_t k = (uint32_t)(offset + length);
    const size_t out_len = hybrid ? length : k;
    std::vector<uint64_t> doc(out_len);
    std::vector<float> score(out_len);
    std::vector<uint8_t> src(out_len);
    uint32_t cnt = 0;
    uint64_t tot = 0;
    std::vector<int> rcs(S, SS_OK);
    std::vector<ss_bm25_query> q(S);
    for (size_t i = 0; i < S && hybrid; i++) {
      const int rc = shards_[i]->make_query(query_terms, query_type_default, &q[i]);
      if (rc != SS_OK) { ro.last_error = rc; return ro; }
      shards_[i]->mark_all_terms_frequent(&q[i], k);
    }
    std::vector<std::thread> th;
    for (size_t i = 0; i < S; i++)
      th.emplace_back([&, i] {
        std::vector<uint64_t> d(i ? out_len : 0);
        std::vector<float> sc(i ? out_len : 0);
        std::vector<uint8_t> so(i ? out_len : 0);
        uint32_t c = 0;
        uint64_t t = 0;
        const float thr = threshold_raw(similarity_threshold, shards_[i]->euclidean());
        if (hybrid)
          rcs[i] = ss_hybrid_search_sharded(shards_[i]->handle(), comms_[i], 1, &q[i], (uint32_t)result_type, qv.data(), thr, k, (uint32_t)offset,
                                            (uint32_t)length, i ? d.data() : doc.data(), i ? sc.data() : score.data(), i ? so.data() : src.data(),
                                            i ? &c : &cnt, i ? &t : &tot);
        else
          rcs[i] = ss_vec_search_sharded(shards_[i]->handle(), comms_[i], 1, qv.data(), k, thr, i ? d.data() : doc.data(),
                                         i ? sc.data() : score.data(), i ? &c : &cnt, i ? &t : &tot);
      });
    for (auto& t : th) t.join();
    for (int r : rcs) if (r != SS_OK) ro.last_error = r;
    if (ro.last_error != SS_OK) return ro;  // degrade to empty
    ro.result_count_total = tot;
    const size_t first = hybrid ? 0 : std::min<size_t>(offset, cnt);  // the fused list is already cut; a vector list is cut here
    for (size_t x = first; x < cnt && ro.results.size() < length; x++) {
      Result r;
      r.doc_id = doc[x];
      r.score = score[x];
      r.source = hybrid ? (ResultSource)src[x] : ResultSource::Vector;
      if (!hybrid) r.vector_score = vector_score_of(score[x]);
      r.shard_id = (uint32_t)(r.doc_id % S);
      r.level_id = (uint32_t)((r.doc_id / S) >> 16);
      ro.results.push_back(r);
    }
    ro.result_count = ro.results.size();
    return ro;
  }
  // one host thread per shard; lexical then vector sequentially inside the task for Hybrid (search.rs:1698-1740)
  std::vector<ResultObject> lex(S), vec(S);
  auto task = [&](size_t i) {
    Shard& sh = *shards_[i];
    if (want_lex) lex[i] = sh.search_lexical_shard(query_terms, query_type_default, 0, offset + length, result_type, facet_filter, not_terms,
                                                   lexical_field_filter);
    if (want_vec && qv.size() == sh.dim()) vec[i] = sh.search_vector_shard(qv.data(), offset + length, similarity_threshold, ann_mode, vector_field_filter);
  };
  if (S == 1) {
    task(0);  // shard_number == 1: called directly, no spawn (search.rs:1434)
  } else {
    std::vector<std::thread> th;
    for (size_t i = 0; i < S; i++) th.emplace_back(task, i);
    for (auto& t : th) t.join();
  }
  std::vector<uint64_t> ld, vd;
  std::vector<float> ls, vs;
  for (size_t i = 0; i < S; i++) {
    const uint64_t sid = shards_[i]->shard_id();
    for (const Result& r : lex[i].results) { ld.push_back(r.doc_id * S + sid); ls.push_back(r.score); }  // search.rs:1671
    for (const Result& r : vec[i].results) { vd.push_back(r.doc_id * S + sid); vs.push_back(r.score); }  // search.rs:1693
    const uint64_t lt = lex[i].result_count_total, vt = vec[i].result_count_total;
    ro.result_count_total += search_mode == SearchMode::Hybrid ? std::max(lt, vt) : (lt + vt);  // search.rs:1884-1921
    ro.observed_vector_count += vec[i].observed_vector_count;
    ro.observed_cluster_count += vec[i].observed_cluster_count;
    if (lex[i].last_error) ro.last_error = lex[i].last_error;
    if (vec[i].last_error) ro.last_error = vec[i].last_error;
  }
  if (result_type != ResultType::Count && length > 0) {
    std::vector<uint64_t> od(length);
    std::vector<float> os(length);
    std::vector<uint8_t> src(length);
    const int n = ss_merge_results((int)search_mode, ld.data(), ls.data(), (uint32_t)ld.size(), vd.data(), vs.data(),
                                   (uint32_t)vd.size(), (uint32_t)offset, (uint32_t)length, od.data(), os.data(), src.data());
    if (n < 0) {
      ro.last_error = n;
    } else {
      ro.results.resize((size_t)n);
      for (int i = 0; i < n; i++) {
        Result& r = ro.results[i];
        r.doc_id = od[i];
        r.score = os[i];
        r.source = (ResultSource)src[i];
        r.shard_id = (uint32_t)(od[i] % S);
        r.level_id = (uint32_t)((od[i] / S) >> 16);
        // per-list component scores of a fused result (min_heap.rs:17-40)
        for (size_t j = 0; j < ld.size(); j++)
          if (ld[j] == od[i]) { r.lexical_score = ls[j]; break; }
        for (size_t j = 0; j < vd.size(); j++)
          if (vd[j] == od[i]) { r.vector_score = vector_score_of(vs[j]); break; }
      }
    }
  }
  ro.result_count = ro.results.size();
  return ro;
}

Index::~Index() {
  for (ss_comm* c : comms_) ss_comm_destroy(c);
}

int Index::enable_device_exchange() {
  if (!comms_.empty()) return SS_OK;
  const size_t S = shards_.size();
  if (S == 0) return SS_ESTATE;
  std::vector<int> devices(S);
  for (size_t i = 0; i < S; i++) {
    if (!shards_[i]->ok() || shards_[i]->shard_id() != i) return SS_ESTATE;  // rank = shard id: global id = local * S + rank
    devices[i] = shards_[i]->device();
    for (size_t j = 0; j < i; j++)
      if (devices[j] == devices[i]) return SS_EINVAL;
  }
  std::vector<ss_comm*> c(S, nullptr);
  const int rc = ss_comm_create_all((int)S, devices.data(), c.data());
  if (rc != SS_OK) return rc;
  comms_ = std::move(c);
  return SS_OK;
}

std::vector<ResultObject> Index::search_lexical_batch(const std::vector<std::vector<uint32_t>>& query_terms, QueryType query_type_default,
                                                      size_t k, ResultType result_type) {
  const size_t S = shards_.size(), nq = query_terms.size();
  std::vector<ResultObject> out(nq);
  if (S == 0 || nq == 0) return out;
  const size_t kk = std::max<size_t>(k, 1);
  // every shard resolves the batch with its OWN idf (shard-local N and posting counts, search.rs:3225-3230)
  std::vector<std::vector<ss_bm25_query>> q(S, std::vector<ss_bm25_query>(nq));
  int bad = SS_OK;
  for (size_t i = 0; i < S; i++)
    for (size_t j = 0; j < nq; j++) {
      const int rc = shards_[i]->make_query(query_terms[j], query_type_default, &q[i][j]);
      if (rc != SS_OK) bad = rc;
      else if (result_type != ResultType::Count) shards_[i]->mark_all_terms_frequent(&q[i][j], k);
    }
  if (bad != SS_OK) {  // decided BEFORE any shard task starts: a collective must be entered by every rank or by none
    for (ResultObject& ro : out) ro.last_error = bad;
    return out;
  }
  if (!comms_.empty()) {
    // device exchange: every shard task ends with the SAME merged lists; the task of shard 0 fills the answer
    std::vector<uint64_t> doc(nq * kk), tot(nq);
    std::vector<float> score(nq * kk);
    std::vector<uint32_t> cnt(nq);
    std::vector<int> rcs(S, SS_OK);
    std::vector<std::thread> th;
    for (size_t i = 0; i < S; i++)
      th.emplace_back([&, i] {
        std::vector<uint64_t> d(i ? nq * kk : 0), t(i ? nq : 0);
        std::vector<float> sc(i ? nq * kk : 0);
        std::vector<uint32_t> c(i ? nq : 0);
        rcs[i] = ss_bm25_search_sharded(shards_[i]->handle(), comms_[i], (uint32_t)nq, q[i].data(), (uint32_t)k, (uint32_t)result_type,
                                        i ? d.data() : doc.data(), i ? sc.data() : score.data(), i ? c.data() : cnt.data(),
                                        i ? t.data() : tot.data());
      });
    for (auto& t : th) t.join();
    int rc = SS_OK;
    for (int r : rcs) if (r != SS_OK) rc = r;
    for (size_t j = 0; j < nq; j++) {
      ResultObject& ro = out[j];
      if (rc != SS_OK) { ro.last_error = rc; continue; }
      const size_t n = result_type == ResultType::Count ? 0 : cnt[j];
      ro.results.resize(n);
      for (size_t x = 0; x < n; x++) {
        Result& r = ro.results[x];
        r.doc_id = doc[j * kk + x];
        r.score = r.lexical_score = score[j * kk + x];
        r.shard_id = (uint32_t)(r.doc_id % S);
        r.level_id = (uint32_t)((r.doc_id / S) >> 16);
        r.source = ResultSource::Lexical;
      }
      ro.result_count = n;
      ro.result_count_total = tot[j];
    }
    return out;
  }
  // host gather: per-shard batches on one thread each, then ss_merge_results per query (search.rs:1875-2119)
  std::vector<std::vector<ResultObject>> part(S);
  std::vector<std::thread> th;
  for (size_t i = 0; i < S; i++)
    th.emplace_back([&, i] { part[i] = shards_[i]->search_lexical_batch(q[i], k, result_type, {}, false); });
  for (auto& t : th) t.join();
  std::vector<uint64_t> ld, od(kk);
  std::vector<float> ls, os(kk);
  std::vector<uint8_t> src(kk);
  for (size_t j = 0; j < nq; j++) {
    ResultObject& ro = out[j];
    ld.clear(); ls.clear();
    for (size_t i = 0; i < S; i++) {
      const ResultObject& p = part[i][j];
      for (const Result& r : p.results) { ld.push_back(r.doc_id * S + shards_[i]->shard_id()); ls.push_back(r.score); }
      ro.result_count_total += p.result_count_total;
      if (p.last_error) ro.last_error = p.last_error;
    }
    if (result_type == ResultType::Count || k == 0) continue;
    const int n = ss_merge_results(SS_MODE_LEXICAL, ld.data(), ls.data(), (uint32_t)ld.size(), nullptr, nullptr, 0, 0, (uint32_t)k,
                                   od.data(), os.data(), src.data());
    if (n < 0) { ro.last_error = n; continue; }
    ro.results.resize((size_t)n);
    for (int x = 0; x < n; x++) {
      Result& r = ro.results[(size_t)x];
      r.doc_id = od[(size_t)x];
      r.score = r.lexical_score = os[(size_t)x];
      r.shard_id = (uint32_t)(r.doc_id % S);
      r.level_id = (uint32_t)((r.doc_id / S) >> 16);
      r.source = ResultSource::Lexical;
    }
    ro.result_count = ro.results.size();
  }
  return out;
}

}  // namespace seekstorm
