/*
 * ss_oracle.h -- CPU restatement ("oracle") of SeekStorm's query hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported CPU baseline.  The product (seekstorm_amd/) never links it.
 *
 * Parity status: the reference (Rust, ~50 un-vendored crates, no Cargo.lock, no toolchain
 * in this image) cannot be built here, and its own tests pin only result COUNTS
 * (tests/test.rs:150-208, 676-744).  This restatement is therefore pinned by
 *   (1) those count assertions, re-enacted in tests/test_oracle_kat.py,
 *   (2) the formula KATs of SURVEY.md section 8(c) (SmallFloat, idf, BM25, RRF),
 *   (3) an independent naive numpy restatement (oracle/naive.py) that must agree.
 * Score/rank parity against the real Rust binary is "parity unpinned" (see DESIGN.md).
 *
 * Every function cites the reference file:line (relative to /root/reference/seekstorm/src)
 * it follows.
 */
#ifndef SS_ORACLE_H
#define SS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- scoring constants: add_result.rs:20-22 ---- */
#define SO_K 1.2f
#define SO_B 0.75f
#define SO_SIGMA 0.0f
#define SO_BLOCK 65536u /* ROARING_BLOCK_SIZE index.rs:115 */

/* ---- SmallFloat: index.rs:4237-4279 ---- */
uint8_t so_int_to_byte4(uint32_t i);
uint32_t so_byte4_to_int(uint8_t b);

/* bm25_component_cache: commit.rs:318-325 (same at index.rs:3786-3792) */
float so_avgdl(uint64_t positions_sum_normalized, uint64_t indexed_doc_count);
void so_bm25_component_cache(float avgdl, float* out256);
/* idf: search.rs:3225-3230 */
float so_idf(uint64_t indexed_doc_count, uint64_t posting_count);
/* one term of get_bm25f_multiterm_singlefield: add_result.rs:1444-1449 */
float so_bm25_term(float idf, uint32_t tf, float comp);

/* ---- synthetic corpus generator (integer-exact variant) ---- */
uint64_t so_splitmix64(uint64_t x);
uint64_t so_h(uint64_t seed, uint64_t a, uint64_t b);
/* doc-length bytes for docs [d0, d0+n): len_table1024 holds SmallFloat bytes per quantile */
void so_lex_doclen(uint64_t seed, uint64_t d0, uint64_t n, const uint8_t* len_table1024, uint8_t* out);
/* postings of term t over docs [0,n_docs): posting present iff (h(seed,t+1,d)>>32) < thresh32;
 * tf = 1 + min(ctz(low32), 31).  Returns the count; out arrays may be NULL to count only. */
uint32_t so_lex_cluster_thresh(uint64_t seed, uint32_t term, uint64_t doc, uint32_t thresh32); /* clustered corpora: seeds with bit 63 set */
uint64_t so_lex_term_postings(uint64_t seed, uint32_t term, uint32_t thresh32, uint64_t n_docs,
                              uint32_t* out_docs, uint16_t* out_tfs, uint64_t cap);
/* vector rows [r0,r0+n) x dim, uniform(-1,1) then normalize_f32 semantics */
void so_vec_gen(uint64_t seed, uint64_t r0, uint64_t n, uint32_t dim, int normalize, float* out);
void so_vec_gen_strided(uint64_t seed, uint64_t r0, uint64_t stride, uint64_t n, uint32_t dim, int normalize, float* out);
uint32_t so_geom06(uint32_t u); /* tf - 1 of a synthetic posting: geometric p = 0.6 from 32 hash bits */

/* ---- lexical shard model (index.rs:770-860, compress_postinglist.rs:240-332) ---- */
typedef struct so_shard so_shard;
enum { SO_CT_ARRAY = 1, SO_CT_BITMAP = 2, SO_CT_RLE = 3 };
enum { SO_OP_AND = 0, SO_OP_OR = 1 };
enum { SO_RT_COUNT = 0, SO_RT_TOPK = 1, SO_RT_TOPKCOUNT = 2 };

/* Build a shard from decoded postings (CSR by term; doc ids shard-local, ascending per term).
 * Containers are chosen per 65536-doc block by the reference rule. */
so_shard* so_shard_build(uint64_t n_docs, const uint8_t* doclen_bytes, uint32_t n_terms,
                         const uint64_t* term_offsets, const uint32_t* doc_ids, const uint16_t* tfs);
void so_shard_free(so_shard*);
/* delete_hashset of the shard (index.rs:1594): replaces the set; searches skip these docs (add_result.rs:3435) */
void so_shard_set_deleted(so_shard*, const uint64_t* doc_ids, uint64_t n);
float so_shard_avgdl(const so_shard*);
uint64_t so_shard_posting_count(const so_shard*, uint32_t term);
/* container kind chosen for (term, block ordinal); returns 0 if out of range */
int so_shard_container(const so_shard*, uint32_t term, uint32_t block_ordinal, uint32_t* block_id,
                       uint32_t* count, float* max_block_part);
/* decode a container back to doc ids (tests the three formats round-trip) */
uint32_t so_shard_decode_block(const so_shard*, uint32_t term, uint32_t block_ordinal, uint16_t* out);

/* search_lexical_shard dispatch (search.rs:3374-3560) restricted to single-field AND / OR,
 * Topk / TopkCount / Count, no filters.  Results sorted by score desc (search.rs:3565-3593).
 * out_* sized k; returns number of results. */
uint32_t so_search_lex(const so_shard*, uint32_t n_q_terms, const uint32_t* q_terms, int op, uint32_t k,
                       int result_type, uint32_t* out_doc, float* out_score, uint64_t* out_total);
/* same with a not_query_list (the "-term" operands): a doc of a NOT list neither counts nor ranks, add_result.rs:3440-3497 */
uint32_t so_search_lex_not(const so_shard*, uint32_t n_q_terms, const uint32_t* q_terms, uint32_t n_not,
                           const uint32_t* not_terms, int op, uint32_t k, int result_type, uint32_t* out_doc,
                           float* out_score, uint64_t* out_total);
/* the same dispatch as the reference structures it: single_blockid (single.rs:292-417), union_docid_2 / union_docid_3 with
 * their sub-query queue and MinHeap::add_topk's docid_hashset arm (union.rs:1168-1479, min_heap.rs:1193-1260),
 * intersection_blockid, union_blockid for counts.  Must return what so_search_lex_not returns (tests assert it); it is the
 * function the CPU baseline times. */
uint32_t so_search_lex_ref(const so_shard*, uint32_t n_q_terms, const uint32_t* q_terms, uint32_t n_not,
                           const uint32_t* not_terms, int op, uint32_t k, int result_type, uint32_t* out_doc,
                           float* out_score, uint64_t* out_total);
/* CPU baseline harness: S document-partitioned shards, one task per shard and query (search.rs:1637-1743), gather + sort
 * (1875-1940, 2098-2119).  mode 0 = throughput (`threads` workers, independent queries), mode 1 = latency (one query at a
 * time over S worker threads).  Returns queries/s. */
double so_bench_lex(so_shard* const* shards, uint32_t S, const uint32_t* q_terms, uint32_t nq, uint32_t nt, int op, uint32_t k,
                    int rt, int mode, uint32_t threads, double seconds, uint64_t* out_queries, double* out_lat_us,
                    uint32_t lat_cap, uint32_t* out_nlat);
uint32_t so_search_lex_exhaustive_not(const so_shard*, uint32_t n_q_terms, const uint32_t* q_terms, uint32_t n_not,
                                      const uint32_t* not_terms, int op, uint32_t k, uint32_t* out_doc, float* out_score,
                                      uint64_t* out_total);
uint32_t so_search_lex_exhaustive_idf(const so_shard* s, uint32_t n_query_terms, const uint32_t* query_terms, const float* idf,
                                      uint32_t n_not, const uint32_t* not_terms, int op, uint32_t k, uint32_t* out_doc,
                                      float* out_score, uint64_t* total); /* idf given per term (n-gram components) */
/* the reference's all_terms_frequent condition (intersection.rs:198-209) and the exhaustive search under it (shortcut != 0:
 * a doc with some tf < 10 is counted but not ranked; add_result.rs:2091-2104, 3541-3556).  idf may be NULL. */
int so_all_terms_frequent(const so_shard* s, uint32_t n_query_terms, const uint32_t* query_terms, uint32_t top_k);
uint32_t so_search_lex_exhaustive_opt(const so_shard* s, uint32_t n_query_terms, const uint32_t* query_terms, const float* idf,
                                      uint32_t n_not, const uint32_t* not_terms, int op, uint32_t k, int shortcut,
                                      uint32_t* out_doc, float* out_score, uint64_t* total);
/* brute-force ground truth (independent code path): exhaustive scoring + exact top-k by
 * (score desc, doc asc); also returns the exact match count. */
uint32_t so_search_lex_exhaustive(const so_shard*, uint32_t n_q_terms, const uint32_t* q_terms, int op,
                                  uint32_t k, uint32_t* out_doc, float* out_score, uint64_t* out_total);
/* several indexed fields (BM25F, get_bm25f_multiterm_multifield add_result.rs:1171-1426): brute-force ground truth.
 * doclen [n_fields][n_docs]; postings of a term sorted by (doc, field); boost NULL = 1; deleted = doc ids or NULL */
uint32_t so_search_fields_exhaustive(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                                     const uint64_t* off, const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs,
                                     uint32_t n_q_terms, const uint32_t* q_terms, uint32_t n_not, const uint32_t* not_terms,
                                     int op, uint32_t k, const uint64_t* deleted, uint64_t n_deleted, uint32_t* out_doc,
                                     float* out_score, uint64_t* out_total, float* out_avgdl);
/* with the query's field_filter (bit f = indexed field f; 0 = none): every term must occur in a listed field of the doc
 * (add_result.rs:3124-3136); intersections and single-term queries only */
uint32_t so_search_fields_filtered(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                                   const uint64_t* off, const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs,
                                   uint32_t n_q_terms, const uint32_t* q_terms, uint32_t n_not, const uint32_t* not_terms,
                                   int op, uint32_t k, const uint64_t* deleted, uint64_t n_deleted, uint32_t field_mask,
                                   uint32_t* out_doc, float* out_score, uint64_t* out_total, float* out_avgdl);
/* an intersection under the all_terms_frequent shortcut over several indexed fields (add_result.rs:1595-1607, 3111-3122): every
 * matching doc is counted, a doc is ranked only if every term has >= 10 positions in the lowest field that holds the doc */
uint32_t so_search_fields_shortcut(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost, const uint64_t* off,
                                   const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs, uint32_t n_q_terms,
                                   const uint32_t* q_terms, uint32_t k, const uint64_t* deleted, uint64_t n_deleted, uint32_t* out_doc,
                                   float* out_score, uint64_t* out_total);
/* ---- phrase queries (QueryType::Phrase; phrase check add_result.rs:3586-3684, positions add_result.rs:38-59) ----
 * positions: for every posting in CSR order its tf positions (ascending, < 65 536 per field: index.rs:5343) */
void so_shard_set_positions(so_shard*, const uint16_t* positions, uint64_t n_positions);
void so_shard_set_positions_counts(so_shard*, const uint16_t* positions, uint64_t n_positions, const uint16_t* counts /* per posting, NULL = tf */);
int so_phrase_match_places(uint32_t n_seq, const uint16_t* const* pos, const uint32_t* cnt, const uint32_t* place, int reference_loop);
uint32_t so_search_phrase_items(const so_shard*, uint32_t n_q_terms, const uint32_t* q_terms, const float* idf /* NULL = from df */, uint32_t n_seq,
                                const uint8_t* seq, const uint8_t* place /* NULL = 0, 1, ... */, uint32_t k, int reference_loop, uint32_t* out_doc,
                                float* out_score, uint64_t* total);
/* does the phrase match?  pos[i] / cnt[i]: positions of the i-th word of the phrase.  reference_loop != 0: the reference's
 * merge loop restated; 0: the definition (some start carries word i at start + i) */
int so_phrase_match(uint32_t n_seq, const uint16_t* const* pos, const uint32_t* cnt, int reference_loop);
/* q_terms: the unique terms; seq[n_seq]: index into q_terms of every word of the phrase */
uint32_t so_search_phrase(const so_shard*, uint32_t n_q_terms, const uint32_t* q_terms, uint32_t n_seq, const uint8_t* seq, uint32_t k,
                          int reference_loop, uint32_t* out_doc, float* out_score, uint64_t* out_total);
/* phrase over SEVERAL indexed fields (add_result.rs:2964-3414): the phrase must stand inside ONE field (a listed one under
 * field_mask != 0); score = BM25F over all fields of the unique terms.  positions: per (term, doc, field) entry in CSR order its tf
 * positions inside the field */
uint32_t so_search_fields_phrase(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost, const uint64_t* off,
                                 const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs, const uint16_t* positions,
                                 uint32_t n_q_terms, const uint32_t* q_terms, uint32_t n_seq, const uint8_t* seq, uint32_t k,
                                 const uint64_t* deleted, uint64_t n_deleted, uint32_t field_mask, int reference_loop,
                                 uint32_t* out_doc, float* out_score, uint64_t* out_total);
uint32_t so_search_fields_phrase_items(uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost, const uint64_t* off,
                                       const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs, const uint16_t* counts /* NULL = tfs */,
                                       const uint16_t* positions, uint32_t nq, const uint32_t* qt, const float* idf /* NULL = from the lists */,
                                       uint32_t n_seq, const uint8_t* seq, const uint8_t* place /* NULL = 0, 1, ... */, uint32_t k,
                                       const uint64_t* deleted, uint64_t n_deleted, uint32_t field_mask, int reference_loop, uint32_t* od, float* os,
                                       uint64_t* total);
/* statistics for the roofline's algorithmic bytes: sum df, #blocks touched */
void so_query_stats(const so_shard*, uint32_t n_q_terms, const uint32_t* q_terms, uint64_t* sum_df,
                    uint64_t* sum_blocks);

/* ---- vector path (vector.rs:356-497, 1202-1515; vector_similarity.rs:70-74,1006-1008,1118-1142) ---- */
void so_normalize_f32(float* v, uint32_t dim);
float so_dot_f32(const float* a, const float* b, uint32_t dim);       /* scalar order  */
float so_dot_f32_lanes8(const float* q, const float* e, uint32_t dim); /* dot_f32_avx2 order */
/* search_vector_shard, AnnMode::All, F32 dot/cosine.  row_doc_ids may be NULL (= row index).
 * threshold_raw: compared as `score < threshold_raw` (pass -FLT_MAX for none).  Returns count. */
uint32_t so_vec_search(const float* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc_ids,
                       const float* query, uint32_t k, float threshold_raw, int simd_order,
                       uint32_t* out_doc, float* out_score, uint64_t* out_total, uint64_t* out_observed);
/* same with a delete_hashset (ascending doc ids): a deleted record is scored but not pushed, vector.rs:1450-1452 */
uint32_t so_vec_search_del(const float* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc_ids,
                           const float* query, uint32_t k, float threshold_raw, int simd_order,
                           const uint64_t* deleted_sorted, uint64_t n_deleted, uint32_t* out_doc, float* out_score,
                           uint64_t* out_total, uint64_t* out_observed);
/* i8 embeddings: quantize_f32_to_i8 (vector_similarity.rs:1226-1232); record score = dot_i8 as f32 (1011-1016) or, when
 * scaled, dot_i8_quantized = dot as f32 * q_scale * row_scale[r] (1754-1758; row_scale NULL = 1).  Same TopK. */
void so_quantize_f32_to_i8(const float* v, uint32_t n, int8_t* out);
int32_t so_dot_i8(const int8_t* a, const int8_t* b, uint32_t dim);
uint32_t so_vec_search_i8(const int8_t* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc_ids,
                          const float* row_scale, const int8_t* query, int scaled, float q_scale, uint32_t k,
                          float threshold_raw, const uint64_t* deleted_sorted, uint64_t n_deleted, uint32_t* out_doc,
                          float* out_score, uint64_t* out_total, uint64_t* out_observed);
/* AnnMode::Nprobe / Similaritythreshold / NprobeSimilaritythreshold (vector.rs:1300-1392): rows are laid out level after
 * level, cluster after cluster (level_clusters[n_levels] cluster counts, child_count[sum] records per cluster); per level the
 * medoid (first record) of every cluster is pushed into TopK::new(min(n_probe, clusters), cluster_thr) and the records of the
 * surviving clusters are visited in medoid-score order.  n_probe = UINT32_MAX and cluster_thr = -FLT_MAX give the two
 * single-criterion modes.  *out_clusters = observed_cluster_count.  n_levels = 0: AnnMode::All.
 * row_field / field_mask (bit f = indexed field f is searched; NULL / 0 = no filter): field_filter, vector.rs:1397-1400. */
uint32_t so_vec_search_ann(const float* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc_ids, const float* query,
                           uint32_t k, float threshold_raw, int simd_order, uint32_t n_levels, const uint32_t* level_clusters,
                           const uint32_t* child_count, uint32_t n_probe, float cluster_threshold_raw,
                           const uint64_t* deleted_sorted, uint64_t n_deleted, const uint16_t* row_field, uint64_t field_mask,
                           uint32_t* out_doc, float* out_score, uint64_t* out_total, uint64_t* out_observed,
                           uint64_t* out_clusters);
uint32_t so_vec_search_i8_ann(const int8_t* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc_ids,
                              const float* row_scale, const int8_t* query, int scaled, float query_scale, uint32_t k,
                              float threshold_raw, uint32_t n_levels, const uint32_t* level_clusters, const uint32_t* child_count,
                              uint32_t n_probe, float cluster_threshold_raw, const uint64_t* deleted_sorted, uint64_t n_deleted,
                              const uint16_t* row_field, uint64_t field_mask, uint32_t* out_doc, float* out_score,
                              uint64_t* out_total, uint64_t* out_observed, uint64_t* out_clusters);
/* VectorSimilarity::Euclidean: similarity = MINUS the squared distance (vector_similarity.rs:257-345, 905-907; euclidean_f32 912,
 * euclidean_f32_avx2 938, euclidean_i8 921, euclidean_i8_quantized 1721); threshold_raw = -similarity_threshold (vector.rs:398) */
float so_euclidean_f32(const float* a, const float* b, uint32_t dim);
float so_euclidean_f32_lanes8(const float* q, const float* e, uint32_t dim);
uint32_t so_vec_search_euclid(const float* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc_ids, const float* query,
                              uint32_t k, float threshold_raw, int simd_order, uint32_t n_levels, const uint32_t* level_clusters,
                              const uint32_t* child_count, uint32_t n_probe, float cluster_threshold_raw,
                              const uint64_t* deleted_sorted, uint64_t n_deleted, const uint16_t* row_field, uint64_t field_mask,
                              uint32_t* out_doc, float* out_score, uint64_t* out_total, uint64_t* out_observed,
                              uint64_t* out_clusters);
uint32_t so_vec_search_i8_euclid(const int8_t* rows, uint64_t n_rows, uint32_t dim, const uint32_t* row_doc_ids,
                                 const float* row_scale, const float* row_norm, const int8_t* query, int quantized,
                                 float query_scale, float query_norm, uint32_t k, float threshold_raw, uint32_t n_levels,
                                 const uint32_t* level_clusters, const uint32_t* child_count, uint32_t n_probe,
                                 float cluster_threshold_raw, const uint64_t* deleted_sorted, uint64_t n_deleted,
                                 const uint16_t* row_field, uint64_t field_mask, uint32_t* out_doc, float* out_score,
                                 uint64_t* out_total, uint64_t* out_observed, uint64_t* out_clusters);
/* CPU baseline harness of the vector path: AnnMode::All scans (dot_f32_avx2 order + TopK::push) on `threads` workers; mode 0 =
 * throughput (independent queries per worker, all rows each), mode 1 = latency (one query at a time, rows split over the
 * workers as the reference splits them over shards).  Returns queries/s. */
double so_bench_vec(const float* rows, uint64_t n_rows, uint32_t dim, const float* queries, uint32_t nq, uint32_t k, int mode,
                    uint32_t threads, double seconds, uint64_t* out_queries, double* out_lat_us, uint32_t lat_cap, uint32_t* out_nlat);
/* TurboQuant::quantize_f32_i8 (vector_similarity.rs:1927-1983): sign mask, FWHT, scale, i8 + norm; dim = so_turboquant_dim(n) */
uint32_t so_turboquant_dim(uint32_t n);
void so_turboquant_i8(const float* v, uint32_t n, const float* seed_mask, uint32_t dim, int avx2, int8_t* out, float* scale_out,
                      float* norm_out);
/* vector_score field: vector.rs:1495-1499 */
float so_vector_score_field(float dot);
/* TopK threshold transform: vector.rs:388-397 */
float so_threshold_raw(float similarity_threshold);

/* ---- Point (geo) facets: geo_search.rs.  The reference's tests hold no vectors for these (parity unpinned: the
 * restatement is checked against the interleaving's defining properties only).
 * so_morton_encode: encode_morton_2_d (27-41): (deg * 1e7) as i32 as u32 -- Rust's saturating, truncating cast --, latitude
 * into the even bits, longitude into the odd ones; so_morton_decode: decode_morton_2_d (58-79).
 * so_geo_distances: unit 1 = km / 2 = miles: euclidian_distance(base, doc) (115-124); 0: simplified_distance(doc, base)
 * (82-87, the comparison key of morton_ordering 90-108).  so_geo_morton_range: point_distance_to_morton_range (128-144). */
uint64_t so_morton_encode(double lat, double lon);
void so_morton_decode(uint64_t code, double* lat, double* lon);
void so_geo_distances(uint64_t n, const uint64_t* codes, double base_lat, double base_lon, int unit, double* out);
void so_geo_morton_range(double lat, double lon, double distance, int unit, uint64_t out[2]);

/* ---- fusion / merge (search.rs:1669-1673, 1875-2035, 2098-2119) ---- */
/* lists are concatenations over shards of per-shard top-(offset+length); ids already global.
 * mode 0 = lexical only, 1 = vector only, 2 = hybrid RRF(k=0.6).  Output sorted desc, offset
 * dropped, truncated to length.  Ties broken by doc id asc (reference: hash order). */
uint32_t so_merge(int mode, const uint64_t* lex_doc, const float* lex_score, uint32_t n_lex,
                  const uint64_t* vec_doc, const float* vec_score, uint32_t n_vec, uint32_t offset,
                  uint32_t length, uint64_t* out_doc, float* out_score, uint8_t* out_source);

/* ================================================================== text-shaped corpus + the reference's indexing path (ss_textindex.c)
 * A mini indexer as test infrastructure: Zipf tokens with topic clusters, SingleTerm postings with real positions, NgramFF /
 * NgramFFF keys over the frequent ranks (tokenizer.rs:674-782, index_posting.rs:666-741), index.bin as commit.rs:264-552 writes it. */
typedef struct so_text so_text;
so_text* so_text_build(uint64_t seed, uint64_t n_docs, uint32_t vocab, uint32_t n_frequent, int ngrams /* NgramSet bits: 1 FF, 8 FFF */,
                       double topic_share, double mean_len);
/* ... the docs' tokens cut into n_fields (<= 4) consecutive spans indexed as separate fields: positions restart in every field, n-grams
 * stay inside one, the writer emits the multi-field records (field vectors, index_posting.rs:433-940) */
so_text* so_text_build_fields(uint64_t seed, uint64_t n_docs, uint32_t vocab, uint32_t n_frequent, int ngrams, double topic_share, double mean_len,
                              uint32_t n_fields, uint32_t longest_field_id);
const uint8_t* so_text_doclen_fields(const so_text*);   /* [n_fields][n_docs] */
uint32_t so_text_fields(const so_text*, uint32_t* longest_field);
uint32_t so_text_doc_field_tokens(const so_text*, uint64_t doc, uint32_t field, uint32_t cap, uint32_t* out);
uint64_t so_text_key_entries(const so_text*, uint32_t key, uint32_t component, uint32_t* docs, uint8_t* fields, uint16_t* tfs, uint16_t* counts,
                             uint16_t* positions, uint64_t pos_cap, uint64_t* n_pos_out);
void so_text_free(so_text*);
void so_text_info(const so_text*, uint64_t* n_tokens, uint32_t* n_keys, uint32_t* n_keys_nonempty, uint64_t* n_postings, uint32_t* n_ngram_keys);
const uint8_t* so_text_doclen(const so_text*);
uint32_t so_text_doc_tokens(const so_text*, uint64_t doc, uint32_t cap, uint32_t* out);
uint32_t so_text_ngram_key(const so_text*, uint32_t n, const uint32_t* component_ranks);
uint64_t so_text_key_hash(const so_text*, uint32_t key);
uint64_t so_text_key_df(const so_text*, uint32_t key);
uint64_t so_text_key_postings(const so_text*, uint32_t key, uint32_t component, uint32_t* docs, uint16_t* tfs, uint16_t* counts,
                              uint16_t* positions, uint64_t pos_cap, uint64_t* n_pos_out);
int so_text_write_index_bin(const so_text*, uint32_t segment_number_bits, uint32_t key_head_size, uint32_t positions_limit, uint8_t** out,
                            uint64_t* out_len);
void so_text_free_bytes(uint8_t*);

#ifdef __cplusplus
}
#endif
#endif
