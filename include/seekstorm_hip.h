/*
 * seekstorm_hip.h -- C ABI of the MI355X-native SeekStorm query hot path.
 *
 * This is the drop-in boundary.  The reference (Rust crate `seekstorm`) has NO FFI / plugin
 * interface; the seams this ABI replaces are the two per-shard executors and the cross-shard
 * merge (citations relative to /root/reference/seekstorm/src):
 *
 *   ss_bm25_search   <- the dispatch block of search_lexical_shard, search.rs:3374-3560
 *                       (single_blockid / union_docid_2|3 / union_blockid / intersection_blockid)
 *                       inputs already resolved by the host: term -> posting list, idf (search.rs:3225-3230)
 *   ss_vec_search    <- SearchVectorShard::search_vector_shard, vector.rs:1105-1115 / 1202-1515
 *                       (AnnMode::All, F32 dot/cosine; query already normalised, search.rs:1464-1475)
 *   ss_vec_search_ann, ss_vec_search_i8[_ann]
 *                    <- the same seam with its other arguments: AnnMode::Nprobe / Similaritythreshold (vector.rs:1300-1392),
 *                       field_filter (1225-1237, 1397-1400), Precision::I8 records (vector_similarity.rs:1011-1016, 1754-1758)
 *   ss_bm25_search_filtered, ss_bm25_facet_count
 *                    <- facet_filter / query_facets of search_lexical_shard (add_result.rs:341-640)
 *   ss_index_bin_*, ss_ref_decode_block*, ss_vec_upload_vector_bin*
 *                    <- readers of the shard files and in-RAM blocks (index.rs:3263-3740, vector.rs:1066-1094,
 *                       add_result.rs:2036-2293)
 *   ss_*_upload      <- (re)build of the device image at the end of open_shard (index.rs:3796)
 *                       and after each commit (commit.rs:142-148)
 *   ss_merge_results, ss_topk_merge_dev*, ss_rrf_merge_dev
 *                    <- cross-shard gather + RRF + sort/offset/length, search.rs:1875-2119
 *
 * Conventions (mirroring the reference's search path, search.rs:2461-2463 / vector.rs:1222-1224):
 *   - every function returns 0 on success or a negative SS_E* code; never throws, never aborts;
 *   - results come back sorted by score descending, per-shard LOCAL doc ids, caller-owned buffers;
 *   - unused result slots hold doc id SS_NO_DOC and score 0;
 *   - per-shard request size is offset+length with offset 0 (search.rs:1658-1659): `k` below;
 *   - thread-safe: the host side of concurrent calls on one handle is serialised (a mutex around validation and launch
 *     queuing); concurrent SMALL host-pointer searches (the reference's calling pattern: one query per runtime worker,
 *     search.rs:1637-1743) do not queue up one behind the other on that mutex but are COALESCED into device batches
 *     (ss_shard_set_coalescing below; on by default); the DEVICE side of the *_dev searches is not serialised -- BM25 and
 *     vector (AnnMode::All) searches queued on different
 *     streams of one shard run concurrently, each stream with its own workspace (searches sharing a stream run in order).
 *     The ANN modes and the facet-filtered BM25 search keep one workspace per shard: queue those on one stream per shard.
 *     An image upload / rebuild must not race with searches still queued on foreign streams.  Destroy must not race.
 * Plain C types only: no torch / HIP types in any signature (streams cross as void*).
 */
#ifndef SEEKSTORM_HIP_H
#define SEEKSTORM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 7
#define SS_NO_DOC 0xFFFFFFFFu
/* scored + NOT terms of one query: union_docid_3 takes unions of <= 10 terms (union.rs:1308), union_blockid -> union_scan_32 those of
 * 11..32 (search.rs:3497-3520, union.rs:598-805; its 32-bit match mask is the limit) */
#define SS_MAX_QUERY_TERMS 32
/* results one PASS over the index holds.  The host-pointer searches (ss_bm25_search[_filtered], ss_vec_search*, ss_vec_search_i8*) take
 * any k -- the crate's offset + length is unbounded (search.rs:1658-1659) -- and answer k > SS_MAX_K in passes of SS_MAX_K, every pass
 * the ordinary search under (tombstones or the facet filter's bitmap) | (the docs of the earlier passes); totals are the first pass's.
 * ss_bm25_search_sharded / ss_vec_search_sharded do the same before their exchange, ss_bm25_search_sorted peels a deep page the same way
 * (its order -- the sort fields, then score descending, then doc ascending -- is total as well).  The device-pointer entries (ss_*_dev)
 * answer a deep page too, SYNCHRONOUSLY: the passes are steered from the host, so the queries make one trip there; the page is left in
 * the caller's device arrays.  ss_hybrid_search_sharded runs both shard tasks in passes and fuses the gathered lists on the host
 * (ss_merge_results) when the page is deep or n_ranks * k * 2 exceeds the fusion kernel's 8192 entries. */
#define SS_MAX_K 1024
#define SS_VEC_BATCH 64 /* queries scanned per pass over the matrix */

enum {
  SS_OK = 0,
  SS_EINVAL = -1,   /* bad argument */
  SS_ENOMEM = -2,   /* device or host allocation failed */
  SS_EDEVICE = -3,  /* HIP runtime error (no device, launch failure, ...) */
  SS_ENOTSUP = -4,  /* a valid request the device path does not answer: the caller runs its own (CPU) path for it -- for queries the
                     * complete list is INTEGRATION.md section 4 "CPU fall-through", pinned by tests/test_gpu_shape_sweep.py */
  SS_ESTATE = -5,   /* image not uploaded yet */
  SS_EPEER = -6     /* a collective call (ss_*_search_sharded) failed on ANOTHER rank: every rank returns an error, none blocks */
};

/* QueryType (search.rs:59): the two set operations and Phrase (an intersection whose docs must carry the words at
 * consecutive positions, add_result.rs:3586-3684); ResultType (search.rs:168) */
enum { SS_OP_INTERSECTION = 0, SS_OP_UNION = 1, SS_OP_PHRASE = 2 };
#define SS_MAX_PHRASE 12 /* words of a phrase query */
#define SS_PHRASE_SKIP 0xFFu /* phrase_seq entry of a word place that lies inside an n-gram key (its entry stands at the key's first word) */
enum { SS_RT_COUNT = 0, SS_RT_TOPK = 1, SS_RT_TOPKCOUNT = 2 };
/* SearchMode (search.rs:73) for ss_merge_results */
enum { SS_MODE_LEXICAL = 0, SS_MODE_VECTOR = 1, SS_MODE_HYBRID = 2 };
/* ResultSource (min_heap.rs:17-40) */
enum { SS_SRC_LEXICAL = 0, SS_SRC_VECTOR = 1, SS_SRC_HYBRID = 2 };

typedef struct ss_shard ss_shard; /* opaque: the HBM image of ONE shard on ONE device */

int ss_abi_version(void);
const char* ss_strerror(int code);
int ss_device_count(int* out);
int ss_shard_create(int device, ss_shard** out);
/* Coalescing of concurrent callers (the reference has no batched search entry point: Search::search takes one query, and many
 * tokio workers call it at once holding only the shard's read lock -- search.rs:1637-1743).  While one batch of a
 * shard runs on the device, the plain host-pointer searches that arrive (ss_bm25_search / ss_bm25_search_filtered without
 * filters, ss_vec_search, ss_vec_search_i8; at most SS_COALESCE_MAX_REQUEST queries per call, same k / result type /
 * threshold) wait in a queue; the first of them becomes the leader of the next batch, runs ALL queued requests as one device
 * batch on behalf of their callers and hands every caller its own rows.  A lone caller runs at once (nothing is ever delayed
 * to wait for company unless max_wait_us > 0); answers are those of separate calls, bit for bit (queries of a batch are
 * independent); a request that makes its batch fail (an invalid query) is re-run alone so that only its own caller sees the
 * error.  max_lexical_batch / max_vector_batch: queries per merged batch (0 = coalescing off for that kind; defaults 1024
 * and SS_VEC_BATCH).  ss_shard_coalescing_stats: batches run and queries served through the coalescer so far. */
#define SS_COALESCE_MAX_REQUEST 64
int ss_shard_set_coalescing(ss_shard* s, uint32_t max_lexical_batch, uint32_t max_vector_batch, uint32_t max_wait_us);
int ss_shard_coalescing_stats(ss_shard* s, uint64_t* lexical_batches, uint64_t* lexical_queries, uint64_t* vector_batches,
                              uint64_t* vector_queries);
/* Small host-pointer lexical batches -- the reference's call shape is ONE query per call (search.rs:1637-1743: one task per shard and
 * query) -- take a one-launch path (csrc/bm25_small.hip: <= 64 queries of <= 4 scored and <= 4 NOT terms each, k <= 128, one list per
 * term -- one indexed field, or several with merged lists and no field filter --, every list with a probe row, no facet filter): same answers bit for bit, a third of the latency.  ss_bm25_path_stats: how
 * many ss_bm25_search[_filtered] batches (coalesced ones included) took it so far. */
int ss_bm25_path_stats(ss_shard* s, uint64_t* one_launch_batches);
/* Query shapes (ABI v6).  Each query of a host-pointer batch is classified behind the call and the batch run as sub-batches per
 * kernel family, answers back in the callers' order: what the specialised kernels serve goes to them; intersections, filtered terms
 * and phrases beyond them -- the all_terms_frequent shortcut over more than 7 terms, more than 8 terms or 32 (term, field) lists over
 * per-field lists, phrases of 7 .. SS_MAX_PHRASE unique terms or with k > 128 -- to the generic galloping kernels
 * (csrc/bm25_gallop.hip: the shortest list drives, the others are looked up by binary search, intersection.rs:352-362); unions of
 * 8 .. 10 terms under a field filter, or naming a sparse-tier term, are composed from the reference's own sub-queries
 * (union.rs:1330-1425).  ss_bm25_shape_stats: sub-batches the generic kernels answered so far. */
int ss_bm25_shape_stats(ss_shard* s, uint64_t* generic_batches);
int ss_shard_destroy(ss_shard* s);
/* block until all work queued on the shard's stream is done */
int ss_shard_sync(ss_shard* s);

/* ------------------------------------------------------------------ BM25 image
 * Host passes DECODED postings (CSR by term, shard-local doc ids ascending per term, tf = positions_count
 * of the single indexed field) and the per-doc SmallFloat length bytes (index.rs:5397); the library builds
 * the HBM image (sub-block CSR of packed postings, bm25_component_cache per commit.rs:318-325). */
int ss_bm25_upload(ss_shard* s, uint64_t n_docs, const uint8_t* doclen_bytes, uint32_t n_terms,
                   const uint64_t* term_offsets, const uint32_t* doc_ids, const uint16_t* tfs);
/* Several indexed fields (BM25F, get_bm25f_multiterm_multifield, add_result.rs:1171-1426): a posting is (term, doc, field, tf)
 * and a doc's score sums  boost[field] * idf * tf (K+1) / (tf + comp[len_byte(doc, field)])  over the query terms and the
 * fields they occur in; idf from the docs containing the term in any field (ss_bm25_term_df returns that); an
 * intersection needs every term in at least one field.  doclen_bytes is [n_fields][n_docs] (level_index
 * document_length_compressed_array[field], index.rs:770-776); the postings of a term are sorted by (doc, field);
 * boost = schema boost per field (NULL = 1).  Queries are the same ss_bm25_query.  n_fields <= 8.
 * Two or more fields (boosts > 0): the score is additive per (term, field), so the image also carries one MERGED list per term
 * -- every doc holding the term in any field, with the weight sum_f boost[f] * w_f scaled into the weight code's range (the
 * scale returns through idf) -- and a query WITHOUT a field filter reads only those: it is a single-field query to every kernel
 * (pruned strategy, 16-bit scan, plain intersections, up to SS_MAX_QUERY_TERMS terms).  Its scores equal the per-field sums up to the code's
 * rounding (2^-16 relative per term, inside the 1e-4 tolerance); the image holds the postings twice.  A query WITH a field
 * filter reads the (term, field) lists: at most 32 / n_fields terms (NOT terms included), intersections of at most 8 terms.
 * (Boosts that leave a merged weight outside the code's range build the image without merged lists: ss_bm25_fields_info.) */
int ss_bm25_upload_fields(ss_shard* s, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen_bytes, const float* boost,
                          uint32_t n_terms, const uint64_t* term_offsets, const uint32_t* doc_ids, const uint8_t* field_ids,
                          const uint16_t* tfs);

/* ss_bm25_upload_fields plus the POSITIONS of every (term, doc, field) entry, in entry order: tf positions inside the field,
 * ascending, each < 65 536 -- what decode_positions_multiterm_multifield / get_next_position_multifield hand the phrase check of
 * add_result_multiterm_multifield (add_result.rs:1485-2034, 2964-3414).  Phrase queries then run over the image's merged
 * lists (SS_ENOTSUP when the boosts kept them from being built): the phrase must stand inside ONE field -- a listed one under
 * SS_OP_FIELD_FILTER --, the score is the BM25F sum over all fields of the unique terms.  n_positions = sum of tfs.
 * 4 bytes per position + 4 bytes per image slot of HBM. */
int ss_bm25_upload_fields_positions(ss_shard* s, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen_bytes, const float* boost,
                                    uint32_t n_terms, const uint64_t* term_offsets, const uint32_t* doc_ids, const uint8_t* field_ids,
                                    const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions);

/* ss_bm25_upload plus the POSITIONS of every posting (one indexed field): positions = for every posting in CSR order its tf
 * positions inside the field, ascending, each < 65 536 (token_per_field_max, index.rs:5343) -- what the reference decodes from
 * the position records / embedded pointers (add_result.rs:38-59, 2036-2197; compress_postinglist.rs:949-977).  n_positions must
 * be the sum of tfs.  2 bytes per position + 4 bytes per posting of HBM; only phrase queries read them. */
int ss_bm25_upload_positions(ss_shard* s, uint64_t n_docs, const uint8_t* doclen_bytes, uint32_t n_terms, const uint64_t* term_offsets,
                             const uint32_t* doc_ids, const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions);

/* Tombstones = the shard's delete_hashset (index.rs:1594): shard-local doc ids, as delete.bin stores them (a plain
 * stream of u64, index.rs:3798-3809 -- the file's bytes can be passed as they are) or as delete_document adds them
 * (index.rs:5110).  The call replaces the set (n = 0 clears it); it applies to lexical and vector searches alike: a
 * deleted doc is skipped before it counts or ranks (add_result.rs:3435, union.rs:975, vector.rs:1450-1452). */
int ss_set_deleted(ss_shard* s, const uint64_t* doc_ids, uint64_t n);

/* Same image from the reference's own in-RAM block format (first step of SURVEY section 8 f-1): one ss_ref_block per
 * (term, 65 536-doc block) = BlockObjectIndex (index.rs:781-789) + the key-body byte array of the segment the posting
 * list lives in (index.rs:991-995).  Doc-id containers (Array / Bitmap / Rle, compress_postinglist.rs:694-946) and the
 * rank/position pointers (tf = positions_count, add_result.rs:2036-2197) are decoded on the host.  Single indexed
 * field, SingleTerm keys; blocks of a term ascending by block_id. */
typedef struct {
  uint32_t block_id;                 /* docs block_id * 65536 + local */
  uint32_t compression_type_pointer; /* CompressionType << 30 | rank_position_pointer_range */
  uint16_t posting_count_m1;         /* posting_count - 1, as stored */
  uint16_t pointer_pivot_p_docid;    /* ranks below it have 2-byte pointers, the others 3-byte */
  const uint8_t* byte_array;         /* key bodies of the segment */
  uint64_t byte_array_len;
} ss_ref_block;
int ss_ref_decode_block(const ss_ref_block* block, uint16_t* docs_out /*[65536]*/, uint16_t* tfs_out /*[65536]*/);
/* the same with the positions of every posting (sum(tf) absolute positions in posting order; SingleTerm keys): returns the
 * posting count, *n_pos_out = positions written (SS_EINVAL with the needed number when pos_cap is too small) */
int ss_ref_decode_block_positions(const ss_ref_block* block, uint16_t* docs_out, uint16_t* tfs_out, uint16_t* pos_out,
                                  uint64_t pos_cap, uint64_t* n_pos_out);
/* a block of an N-GRAM key (key_hash & 7 = NgramType != 0, index.rs:1854-1872; one indexed field): never embedded, every
 * record starts with the tf of each component term (2: bigram types 1-3, 3: trigram types 4-7) before the positions count
 * (add_result.rs:2074-2089).  tfs_out = tf of component `component` -- what the n-gram arms of
 * get_bm25f_multiterm_singlefield (add_result.rs:1454-1477) weigh with idf_ngram{1,2,3}. */
int ss_ref_decode_block_ngram(const ss_ref_block* block, uint32_t n_components, uint32_t component, uint16_t* docs_out,
                              uint16_t* tfs_out);
/* ... with the key's OWN positions: the positions_count and the positions that follow the component tfs in every record -- the
 * places of the n-gram's FIRST word (tokenizer.rs:699), which the phrase check walks for a query entry that resolved to the key
 * (add_result.rs:2074-2089, 3596-3684).  npos_out [65536] = positions per posting, pos_out their concatenation (absolute,
 * ascending per posting); tfs_out = tf of component 0.  SS_EINVAL with *n_pos_out = needed size when pos_cap is too small. */
int ss_ref_decode_block_ngram_positions(const ss_ref_block* block, uint32_t n_components, uint16_t* docs_out, uint16_t* tfs_out,
                                        uint16_t* npos_out, uint16_t* pos_out, uint64_t pos_cap, uint64_t* n_pos_out);
int ss_bm25_upload_ref_blocks(ss_shard* s, uint64_t n_docs, const uint8_t* doclen_bytes, uint32_t n_terms,
                              const uint64_t* term_block_offsets /*[n_terms+1]*/, const ss_ref_block* blocks);
/* ... and from a shard's index.bin as it lies on disk / in the mmap (writer commit.rs:264-369 and
 * 467-552, reader index.rs:3263-3740).  ss_index_bin_open only walks the levels, segment head tables and key heads
 * (host, no device needed) and borrows `bytes`, which must outlive the handle.  Term id = rank of the key_hash among
 * the keys, ascending: the Rust side keeps translating term -> key_hash (ahash / gxhash stay in Rust) and
 * binary-searches ss_index_bin_term_keys.  An N-GRAM key (the default index has NgramFF | NgramFFF keys,
 * index.rs:1422-1424) occupies one term id PER COMPONENT, consecutive and in order (same key_hash, same docs, the
 * component's tf): a query term that resolved to the n-gram becomes its 2 or 3 component terms with idf_ngram_i each
 * (from ss_index_bin_term_ngram's component df, search.rs:3231-3262) -- the same documents match, and the scores add up
 * to the n-gram arm of get_bm25f_multiterm_singlefield (add_result.rs:1454-1477); with several indexed fields each component's
 * field vector is decoded the same way (add_result.rs:1524-1600).  In a PHRASE the key is ONE entry whose positions are those of
 * its first word and which spans 2 / 3 places (search.rs:3305-3328): ss_bm25_upload_index_bin_positions puts the key's positions
 * behind its FIRST component term; the query names that term at the key's first place and SS_PHRASE_SKIP at its other places, the
 * other component terms are scored only.  indexed_field_count / key_head_size (20 | 22 | 23, index.rs:2806-2812) come from
 * schema.json / index.json; segment_number_bits is 11 for every index opened by the reference (index.rs:3285). */
typedef struct ss_index_bin ss_index_bin;
int ss_index_bin_open(const uint8_t* bytes, uint64_t len, uint32_t indexed_field_count, uint32_t key_head_size,
                      uint32_t segment_number_bits, ss_index_bin** out);
/* Drops the keys with fewer than min_posting_count postings (term ids are re-ranked among the kept keys): the device
 * image is built for the frequent terms that dominate query cost; a query touching a dropped term (ss_index_bin_term_keys
 * has no entry for it) is answered by the host's own path. */
int ss_index_bin_filter(ss_index_bin* ix, uint64_t min_posting_count, uint32_t* n_terms_kept);
/* Host only: decodes every key as the uploads do (worker threads) and reports the postings ((doc, field) entries with several indexed
 * fields) and positions that came out -- what an open will have to move, and the decoder's speed on its own. */
int ss_index_bin_decode_stats(const ss_index_bin* ix, int with_positions, uint64_t* n_postings_out, uint64_t* n_positions_out);
/* ... and hands the decoded postings over (one indexed field; sizes from ss_index_bin_decode_stats): offs [terms + 1], doc_ids / tfs
 * [postings] in term-id order; positions_out != NULL: npos [postings] = positions per posting (the tf for a SingleTerm key, the key's
 * own count behind the FIRST component of an n-gram key, 0 behind the others) and their concatenation.  SS_EINVAL when a capacity is
 * too small.  What the image builders consume -- for hosts that assemble levels themselves (ss_bm25_append_level) and for tests. */
int ss_index_bin_decode_all(const ss_index_bin* ix, uint64_t* offs_out, uint32_t* doc_ids_out, uint16_t* tfs_out, uint64_t postings_cap,
                            uint16_t* npos_out, uint16_t* positions_out, uint64_t positions_cap);
/* Two tiers instead of dropping the tail: keys with at least dense_min_posting_count postings come first (ascending key hash) and
 * become the dense image's terms, the others follow (ascending key hash) and go to the SPARSE tier (ss_bm25_append_sparse) when
 * the index is uploaded with ss_bm25_upload_index_bin -- a real vocabulary's millions of rare keys then cost 8 bytes per posting,
 * not a directory row each, and every key of the index stays searchable on the device.  Term id = position in that order
 * (ss_index_bin_term_keys): the host looks a key hash up with one binary search per tier, *n_dense_out = first sparse term id.
 * One or several indexed fields (several: the rare keys' MERGED lists go to the tier, ss_bm25_append_sparse_fields).  With
 * ss_bm25_upload_index_bin[_fields]_positions both tiers carry positions: phrases may name terms of either. */
int ss_index_bin_tier(ss_index_bin* ix, uint64_t dense_min_posting_count, uint32_t* n_dense_out);
int ss_index_bin_close(ss_index_bin* ix);
int ss_index_bin_info(const ss_index_bin* ix, uint64_t* n_docs, uint64_t* positions_sum_normalized, uint32_t* n_levels,
                      uint32_t* n_terms, uint32_t* n_ngram_keys_skipped);
int ss_index_bin_term_keys(const ss_index_bin* ix, uint64_t* keys_out /*[n_terms], ascending*/);
/* per term (any pointer may be NULL): components of its key (1 = SingleTerm key), which component the term is (0-based),
 * and for n-gram components the posting count of the component TERM = DOCUMENT_LENGTH_COMPRESSION[
 * posting_count_ngram_i_compressed] from the key head (compress_postinglist.rs:105-106, 339-409); 0 for SingleTerm */
int ss_index_bin_term_ngram(const ss_index_bin* ix, uint8_t* n_components_out, uint8_t* component_out,
                            uint32_t* component_df_out);
/* decoded postings of one term (tooling / tests); SS_EINVAL with *n_out = needed capacity when cap is too small */
int ss_index_bin_term_postings(const ss_index_bin* ix, uint32_t term, uint64_t cap, uint32_t* docs_out, uint16_t* tfs_out,
                               uint64_t* n_out);
int ss_bm25_upload_index_bin(ss_shard* s, const ss_index_bin* ix);
/* The same plus the positions of every posting, decoded from the rank/position pointers (embedded forms) and the VINT records
 * (decode_positions_multiterm_singlefield, add_result.rs:2036-2197; stored as first position, then gap - 1): phrase queries
 * (SS_OP_PHRASE) then work on an image built from the file -- also on the reference's DEFAULT index (NgramFF | NgramFFF keys,
 * 22 / 23-byte key heads): an n-gram key's positions go to its first component term (see ss_index_bin_open).  SS_ENOTSUP for a
 * position beyond 65 535.  Several indexed fields: see ss_bm25_upload_index_bin_fields_positions (n-gram keys alike: the key's own
 * field vector and positions behind its first component term). */
int ss_bm25_upload_index_bin_positions(ss_shard* s, const ss_index_bin* ix);
/* the same for an index with several indexed fields: position records carry a field vector per posting
 * (decode_positions_multiterm_multifield, add_result.rs:1485-2034; read_multifield_vec 2200-2293) -> ss_bm25_upload_fields.
 * boost = schema boost per field (schema.json; NULL = 1).  ss_bm25_upload_index_bin calls it with NULL. */
int ss_bm25_upload_index_bin_fields(ss_shard* s, const ss_index_bin* ix, const float* boost);
/* ... plus the positions of every (posting, field) entry -- the VINT positions behind a record's field vector and the bit-packed
 * positions of the embedded pointers (add_result.rs:1606-2017) -- for phrase queries over several indexed fields
 * (ss_bm25_upload_fields_positions).  ss_bm25_upload_index_bin_positions calls it with boost NULL for such an index. */
int ss_bm25_upload_index_bin_fields_positions(ss_shard* s, const ss_index_bin* ix, const float* boost);
/* one block of a multi-field index: docs_out [65536], first_out [65537] = CSR of the field entries per posting,
 * field_out / tf_out [65536 * n_fields] */
int ss_ref_decode_block_fields(const ss_ref_block* block, uint32_t n_fields, uint32_t longest_field_id, uint16_t* docs_out,
                               uint32_t* first_out, uint8_t* field_out, uint16_t* tf_out);
/* the same with the positions of every (posting, field) entry in entry order (sum of tf_out; SingleTerm keys): returns the posting
 * count, *n_pos_out = positions written (SS_EINVAL with the needed number when pos_cap is too small) */
int ss_ref_decode_block_fields_positions(const ss_ref_block* block, uint32_t n_fields, uint32_t longest_field_id, uint16_t* docs_out,
                                         uint32_t* first_out, uint8_t* field_out, uint16_t* tf_out, uint16_t* pos_out, uint64_t pos_cap,
                                         uint64_t* n_pos_out);
/* an N-GRAM key's block in a multi-field index: every record starts with the field vector of each component term (2 or 3,
 * index_posting.rs:664-722 / add_result.rs:1524-1600) before the n-gram's own; output = component `component`'s vector */
int ss_ref_decode_block_fields_ngram(const ss_ref_block* block, uint32_t n_fields, uint32_t longest_field_id, uint32_t n_components,
                                     uint32_t component, uint16_t* docs_out, uint32_t* first_out, uint8_t* field_out,
                                     uint16_t* tf_out);
/* ... with the key's OWN positions (several indexed fields): the key's field vector and positions follow the components' vectors in the
 * record.  Entries = those of component 0; npos_out [65536 * n_fields] = the key's positions behind every entry (0 where the key does not
 * stand in that field), pos_out their concatenation.  SS_EINVAL with *n_pos_out = needed size when pos_cap is too small. */
int ss_ref_decode_block_fields_ngram_positions(const ss_ref_block* block, uint32_t n_fields, uint32_t longest_field_id, uint32_t n_components,
                                               uint16_t* docs_out, uint32_t* first_out, uint8_t* field_out, uint16_t* tf_out,
                                               uint16_t* npos_out, uint16_t* pos_out, uint64_t pos_cap, uint64_t* n_pos_out);

/* Device-side synthetic corpus (bench/test utility; generator = oracle so_lex_*):
 * posting (t,d) iff (h(seed,t+1,d)>>32) < thresh32[t]; bit-identical to ss_bm25_upload of the same corpus. */
int ss_bm25_synth(ss_shard* s, uint64_t seed, uint64_t n_docs, uint32_t n_terms, const uint32_t* thresh32,
                  const uint8_t* len_table1024);
/* The generators above and below build shard `shard_id` of `n_shards` of ONE synthetic corpus, partitioned as the reference
 * partitions documents (doc g -> shard g % S with local id g / S, index.rs:5284): local doc / row d is global d * S + id of
 * the generator stream.  Applies to the following ss_bm25_synth / ss_vec_synth[_i8] calls; default 0 of 1. */
int ss_synth_set_partition(ss_shard* s, uint32_t shard_id, uint32_t n_shards);
int ss_bm25_info(ss_shard* s, uint64_t* n_docs, float* avgdl, uint32_t* n_terms, uint64_t* n_postings);
/* indexed fields of the image; merged_lists = 1 when an image with several fields carries one merged list per term (always,
 * unless the boosts are so far apart that the merged weights leave the weight code's range, or SS_BM25_MERGED=0): phrase queries
 * and SS_OP_ALL_TERMS_FREQUENT over several fields need them; positions = 1 when phrase queries can be answered */
int ss_bm25_fields_info(ss_shard* s, uint32_t* n_fields, uint32_t* merged_lists, uint32_t* positions);
/* SPARSE TIER: the posting lists of RARE terms as plain sorted arrays -- no per-sub-block directory row (4 B per 4096 docs and
 * list: 9.8 KB at 10 M docs, whatever the list's length), no probe row; 8 bytes per posting.  A real vocabulary holds millions of
 * keys, almost all rare (key_count per segment, index.rs:3419-3740): they go here, the lists that cost query time stay in the
 * dense image.  Appends n_lists lists (CSR over offs[n_lists + 1]: ascending doc ids, tf >= 1) to an image with ONE indexed
 * field (several: ss_bm25_append_sparse_fields); sparse list i of the call becomes term *first_term_id_out + i (ids continue behind the dense terms and earlier appends).
 * A query may mix dense and sparse terms through the host-pointer entry points (ss_bm25_search[_filtered without filters],
 * ss_bm25_search_sharded, the coalesced single-query calls): unions -- the dense terms through the ordinary kernels, every doc of a
 * sparse list scored in full by binary-search probes of the query's other lists (north_star's galloping, intersection.rs:352-362),
 * the two lists merged per query; intersections -- the shortest sparse list drives.  Exact counts, tombstones, NOT terms
 * (a union that excludes a SPARSE term is answered on its own, under an exclusion bitmap = tombstones | the list's docs), facet
 * filters, any k (beyond SS_MAX_K in passes), phrases (ss_bm25_append_sparse_positions), field filters of intersections, single terms,
 * phrases and unions (a UNION of several terms under a field filter that names a sparse term is composed from the reference's own
 * sub-queries up to 10 terms and follows its union_scan rule beyond).  The device-pointer entry points take
 * sparse terms when ops_mask bit 28 says so (one host round trip).  ss_bm25_term_df covers the sparse ids. */
int ss_bm25_append_sparse(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                          uint32_t* first_term_id_out);
/* ... on an image with SEVERAL indexed fields (and merged lists, ss_bm25_fields_info): the entries (doc, field, tf) of every rare
 * term, sorted by (doc, field) as ss_bm25_upload_fields takes them.  The tier keeps a term's MERGED list -- every doc once, weighted
 * sum_f boost_f * tf (K + 1) / (tf + comp[len_f]) -- which is what a query without a field filter reads of a dense term as well;
 * beside the weight a posting records the fields that hold the term, which is what a field filter asks of it (intersections, single
 * terms; the score of a filtered query stays the merged weight, within the weight code's 1.5e-5 of the per-field sum).  SS_ENOTSUP
 * when a weight falls outside the range the dense merged lists fixed the weight code to (boosts / lengths unlike anything in the
 * dense image). */
int ss_bm25_append_sparse_fields(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                 const uint16_t* tfs, uint32_t* first_term_id_out);
/* ... with POSITIONS, for phrase queries naming a sparse term (a quoted rare word: the tier's typical phrase).  positions: for every
 * posting (ss_bm25_append_sparse_positions) / every (doc, field) entry (..._fields_positions) in order its positions, ascending --
 * tf of them, or npos[i] where that is not the tf (npos may be NULL): the component terms of an n-gram key, whose own positions
 * stand behind its FIRST component's postings as in the dense tier (ss_bm25_upload_index_bin_positions).  Such a phrase is an
 * intersection driven by its shortest sparse list; every other word is found by binary search, its positions with it -- a dense
 * word's in the image's pool, so the image must carry positions as well (SS_ESTATE otherwise).  Several indexed fields: a phrase's
 * field filter is honoured (a test on the start position's field tag).  Appends without positions to a tier that has some leave
 * those postings without any (a phrase then never matches there). */
int ss_bm25_append_sparse_positions(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                                    const uint16_t* positions, uint64_t n_positions, const uint16_t* npos, uint32_t* first_term_id_out);
int ss_bm25_append_sparse_fields_positions(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                           const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions, const uint16_t* npos,
                                           uint32_t* first_term_id_out);
int ss_bm25_sparse_info(ss_shard* s, uint32_t* n_lists, uint64_t* n_postings, uint64_t* bytes);
/* INCREMENTAL COMMIT (commit.rs:142-148 commit -> warmup, 264-369; index.rs:3796 -- the "(re)build device image" seam after a commit).
 * The reference commits one 65 536-doc level at a time.  ss_bm25_append_level hands over the decoded postings of ONE level -- the
 * length bytes of its docs and, per term, (shard-local doc id, tf) ascending, doc ids inside [level * 65536, level * 65536 +
 * n_level_docs) -- and the image then covers every level committed so far.  level = the number of levels committed (append) or the
 * last committed level (replace: the re-commit of a partial level); only the last level may hold fewer than 65 536 docs.  n_terms may
 * grow from call to call (term ids are the caller's and stable: new terms get the next ids).  The first call on an empty shard creates
 * the image; a shard whose image came from another builder answers SS_ESTATE.  One indexed field; positions and a sparse tier: below.
 * Cost: the level's bytes over PCIe + a device-side rebuild of the image from the levels' postings kept in HBM (6 bytes per posting):
 * BM25 weights depend on avgdl, which every commit moves (commit.rs:318-325), so the reference too refreshes every block's scores.
 * Searches keep running on the previous image until the new one is swapped in (the call then waits for the searches in flight). */
int ss_bm25_append_level(ss_shard* s, uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms,
                         const uint64_t* term_offsets /*[n_terms+1]*/, const uint32_t* doc_ids, const uint16_t* tfs);
/* ... of an image with SEVERAL indexed fields (ABI v7; commit.rs:142-148): the level's entries (term, doc, field, tf) sorted by (doc, field)
 * inside a term, doc ids inside the level, and level_doclen [n_fields][n_level_docs] as ss_bm25_upload_fields takes them.  The levels are
 * kept on the host and the image is rebuilt by the multi-field builder -- a commit costs what an upload of the shard costs (the
 * one-field form rebuilds on the device); the answers afterwards are those of ONE ss_bm25_upload_fields of all levels, bit for bit.
 * n_fields and boost must stay the same from level to level.  No sparse tier beside it (SS_ENOTSUP), no positions yet. */
int ss_bm25_append_level_fields(ss_shard* s, uint32_t level, uint32_t n_level_docs, uint32_t n_fields, const uint8_t* level_doclen,
                                const float* boost, uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                const uint16_t* tfs);
/* ... with the postings' POSITIONS, so that phrase queries work on an image that grows by commits: positions = every posting's in CSR
 * order (ascending inside a posting), tf of them each, or npos[i] where that is not the tf (npos may be NULL) -- the component terms of
 * an n-gram key, whose own positions stand behind its FIRST component's postings (ss_bm25_upload_index_bin_positions).  Every level of
 * an image brings positions, or none does (SS_EINVAL).  The position arrays of the image are rebuilt on the device with the rest. */
int ss_bm25_append_level_positions(ss_shard* s, uint32_t level, uint32_t n_level_docs, const uint8_t* level_doclen, uint32_t n_terms,
                                   const uint64_t* term_offsets, const uint32_t* doc_ids, const uint16_t* tfs, const uint16_t* npos,
                                   const uint16_t* positions, uint64_t n_positions);
/* ... on an image WITH A SPARSE TIER (one indexed field): after ss_bm25_append_level[_positions] of the level's dense terms (whose
 * number then stays what it was: n_terms as before, SS_ENOTSUP otherwise -- new terms start rare), the level's postings of the RARE
 * terms in that level (level = the one just committed): list i continues sparse list i (term id = dense terms + i), doc ids inside the
 * level; n_lists >= the tier's current lists -- the further ones are new terms.  A level the tier has seen already is REPLACED by the
 * call (the re-commit of a level that was incomplete, commit.rs:204-206: first ss_bm25_append_level with the same level, then this);
 * levels without rare postings may be skipped.  The first call creates the tier.  The tier keeps every posting's tf (2 bytes
 * beside its 8) and re-codes all its postings on the device after every commit, as the average length moves (commit.rs:318-325); with
 * positions (every level brings them, or none does; same meaning as above) phrases may name its terms.  A tier filled by whole lists
 * (ss_bm25_append_sparse) takes no levels, and this one no whole lists (SS_ESTATE).  SS_EINVAL leaves the tier as it was. */
int ss_bm25_append_sparse_level(ss_shard* s, uint32_t level, uint32_t n_lists, const uint64_t* offs /*[n_lists+1]*/, const uint32_t* doc_ids,
                                const uint16_t* tfs, const uint16_t* npos, const uint16_t* positions, uint64_t n_positions);
int ss_bm25_incremental_info(ss_shard* s, uint32_t* n_levels, uint64_t* raw_bytes, double* last_append_ms, double* last_rebuild_ms);
/* Search strategy.  AUTO: requests with <= 4 scored terms and k <= 128 take the PRUNED path (the reference's block-max /
 * sub-query pruning, intersection.rs:2224-2233, union.rs:1355-1405, as MaxScore over a probe index: only essential /
 * shortest lists are read; exact union counts are popcounts over the index's bit records like union_count,
 * union.rs:807-); everything else, and every request when the probe index did not fit in device memory, takes the
 * EXHAUSTIVE scan.  Both return identical results.  SS_BM25_PRUNED fails with SS_ENOTSUP where pruning cannot serve
 * the request (> 4 scored terms, k > 128, no probe index).  SS_BM25_EXHAUSTIVE_F32 holds the exhaustive strategy on the
 * f32-tile scan kernel that the 16-bit-tile kernel replaced for unions of <= 6 lists / intersections of 2-3 terms at k <= 64
 * (same answers; kept selectable for measurements and cross-checks). */
enum { SS_BM25_AUTO = 0, SS_BM25_EXHAUSTIVE = 1, SS_BM25_PRUNED = 2, SS_BM25_EXHAUSTIVE_F32 = 3 };
/* The probe index costs 12 bytes per 64 docs and posting list (1.9 MB per list at 10 M docs).  Its rows go to the longest
 * lists first until the budget of the NEXT image build is spent (bytes; 0 = half of the free device memory); queries
 * touching a list without a row are ranked by the scan kernels (exact counts then come from the scan as well).
 * When the rows are rationed, a quarter of them (at most 8192; none below 32 rows) form a POOL: a host-pointer batch
 * (ss_bm25_search[_filtered], ss_bm25_search_sharded, the facet routines) first builds rows for the row-less lists it touches
 * from their postings -- free pool rows first, then the least recently used ones -- so that its queries still take the pruned
 * strategy; lists that found no row leave their queries to the scan kernels, and a mixed batch is run as two.
 * ss_bm25_term_probed: per term 0 = some list without a row, 1 = all lists have rows, 2 = all have rows and one of them is a
 * pool row, which a later host-pointer batch may take away.  A caller of ss_bm25_search_dev reports "every term has rows" in
 * ops_mask bit 2 (a vouch gone stale is caught on the device: the query fails loudly). */
int ss_bm25_set_probe_budget(ss_shard* s, uint64_t max_bytes);
int ss_bm25_term_probed(ss_shard* s, uint32_t n, const uint32_t* terms, uint8_t* out);
int ss_bm25_set_strategy(ss_shard* s, int strategy);
/* posting_count per term (the df the host needs for idf, search.rs:3225-3230) */
int ss_bm25_term_df(ss_shard* s, uint32_t n, const uint32_t* terms, uint64_t* df_out);

/* query_list + not_query_list of the dispatch block (search.rs:3374-3560).  NOT terms ("-term", not_query_list): a doc
 * found in one of their posting lists neither counts nor ranks (add_result.rs:3440-3497).  They are stored after the
 * n_terms query terms, their number in bits 8..15 of op:  op = SS_OP_* | SS_OP_NOT_TERMS(n);  n_terms + n <= SS_MAX_QUERY_TERMS. */
#define SS_OP_NOT_TERMS(n) ((uint32_t)(n) << 8)
/* field_filter of search_lexical_shard for an image with several indexed fields (search.rs:2483-2492, add_result.rs:3124-
 * 3136): bits 16..31 of op, bit f = indexed field f is listed; 0 = no filter.  Intersections and single-term queries: a doc is
 * kept only if EVERY query term occurs in at least one listed field; its score still sums all fields.  A UNION of several terms
 * (<= 10, the range of union_docid_3; MORE than 10: the reference runs union_scan_32 with add_result's per-doc filter, another rule --
 * a doc answers iff EVERY term it holds stands in a listed field, and scores with all of them; answered by that rule on both tiers
 * over merged lists, SS_ENOTSUP only when a dense list of such a union has no probe row -- a rationed vocabulary):
 * the reference filters inside union_docid_3's sub-queries (union.rs:1330-1425, 1168-1305), which comes to -- a doc's
 * score is the sum over its terms that occur in a listed field (all fields of those terms counted), a doc none of whose terms passes
 * is no result; exact count: two terms |pass(X) u pass(Y)|, more terms the UNFILTERED union (union_scan counts a doc before the
 * filter sees it, union.rs:552-553).  Answered by the scan kernels with per-term gating of a doc's score (<= 7 terms); one of 8 .. 10
 * terms, or one that names a term of the sparse tier, by the reference's own sub-queries, behind this call: every subset of its terms
 * as a filtered intersection through both tiers, a doc keeps its best.  Ignored by an image with
 * one indexed field.  For ss_bm25_search_dev such a query counts as an intersection in ops_mask (bit 0) and a filtered union of
 * several terms sets bit 7 as well. */
#define SS_OP_FIELD_FILTER(mask) (((uint32_t)(mask) & 0x7FFFu) << 16)
/* The reference's all_terms_frequent shortcut (intersection.rs:198-209): when the shard holds more than 256 x top_k docs
 * and EVERY term of an intersection occurs in at least half of them, a doc in which some term has an embedded position
 * pointer or fewer than 10 positions is counted but never scored (decode_positions_multiterm_singlefield returns true,
 * add_result.rs:2091-2104, 3541-3556).  An embedded pointer holds at most 4 positions, so the rule is "ranked only if every
 * term has tf >= 10".  The caller evaluates the condition (it knows N, top_k and the posting counts) and sets this bit on
 * the query; the host mirrors do -- and the library CHECKS the posting-count half of the condition again (it holds the counts): a bit
 * set on a query for which it does not hold is ignored, as the reference would not have set it.  Intersections of 2 or more terms
 * (more than 7: the generic kernel, csrc/bm25_gallop.hip); counts are unaffected.  Several indexed
 * fields (decode_positions_multiterm_multifield, add_result.rs:1595-1607: an embedded pointer, or a record whose FIRST field has
 * fewer than 10 positions -> counted, not ranked): "ranked only if every term has >= 10 positions in the lowest field that holds
 * the doc", over the image's merged lists (ss_bm25_fields_info; SS_ENOTSUP without them); under a field filter the reference
 * switches the shortcut off (add_result.rs:3116) and the bit is ignored.  For ss_bm25_search_dev: ops_mask bit 3 = some query
 * carries the bit. */
#define SS_OP_ALL_TERMS_FREQUENT 0x80000000u
typedef struct {
  uint32_t n_terms;                  /* 1..SS_MAX_QUERY_TERMS unique terms (scored; all required for an intersection) */
  uint32_t op;                       /* SS_OP_* | SS_OP_NOT_TERMS(number of NOT terms) */
  uint32_t term[SS_MAX_QUERY_TERMS]; /* term index into the uploaded vocabulary; query terms first, then the NOT terms */
  float idf[SS_MAX_QUERY_TERMS];     /* host-computed, search.rs:3225-3230; entries of NOT terms are ignored */
  uint32_t phrase_len;               /* SS_OP_PHRASE: words of the phrase, 2 .. SS_MAX_PHRASE (non_unique_query_list); else 0 */
  uint8_t phrase_seq[SS_MAX_PHRASE]; /* word i of the phrase is term[phrase_seq[i]]: a repeated word names its unique term again;
                                        SS_PHRASE_SKIP = a place inside an n-gram key (never place 0) */
} ss_bm25_query;
/* SS_OP_PHRASE ("..." queries, QueryType::Phrase): term[] holds the phrase's UNIQUE terms (query_list), phrase_seq its words in
 * order.  A doc matches when it contains every unique term and some position p carries word i at p + i for every i
 * (add_result.rs:3596-3684: the merge over the entries' position lists, fewest positions first; an n-gram key is one entry
 * at its first place -- the positions of its first component term --, its other places are SS_PHRASE_SKIP, its other component
 * terms are unique terms that no place names); it is scored like the
 * intersection of the unique terms (get_bm25f_multiterm_singlefield) and counted only when the phrase matches.  Needs the
 * positions in the image; up to 6 unique terms and k <= 128 run over the probe index (every list with a probe row: on a rationed
 * vocabulary the rows built on demand go to a batch's phrase queries first), 7 .. SS_MAX_PHRASE unique terms or a larger k on the
 * generic kernel (csrc/bm25_gallop.hip; no probe rows needed).  One indexed field: ss_bm25_upload_positions.  Several indexed fields
 * (ss_bm25_upload_fields_positions; add_result.rs:3248-3386): the phrase must stand inside ONE field, fields tried in ascending
 * order, only listed ones under SS_OP_FIELD_FILTER; the score sums all fields of the unique terms.  Host-pointer batches may mix
 * phrase queries with others (run as two sub-batches inside the library, answers back in the callers' order); a DEVICE-resident
 * batch (ss_bm25_search_dev) holds phrase queries only (ops_mask bit 4).  NOT terms work with phrases as with every query type
 * (add_result.rs:3440-3497; ABI v4): a doc found in a NOT list is no match. */

/* Batched BM25 search.  Outputs: out_doc/out_score [n_queries*k], out_count [n_queries] (= results.len()),
 * out_total [n_queries] (= result_count_total: exact match count for Count/TopkCount). */
int ss_bm25_search(ss_shard* s, uint32_t n_queries, const ss_bm25_query* queries, uint32_t k,
                   uint32_t result_type, uint32_t* out_doc, float* out_score, uint32_t* out_count,
                   uint64_t* out_total);
/* Same, everything device-resident and asynchronous on `stream` (a hipStream_t passed as void*;
 * NULL = the shard's own stream).  d_queries is a device array of ss_bm25_query.  ops_mask tells the host, which cannot read
 * the queries, what the batch contains (it picks the kernel variants by it): bit 0 set if any query is an intersection of
 * > 1 terms (variant with match counters) or carries a field filter, bit 1 set if any query is a union of > 1 terms;
 * bits 8..15 = the largest n_terms + NOT terms in the batch (0 = unknown: the generic 10-term kernel is used); bits 16..23 =
 * the largest n_terms alone (0 = same as bits 8..15, i.e. NO query of the batch has NOT terms -- a batch with NOT terms
 * declares bits 8..15 > bits 16..23 even when its longest query has none); bit 2 set if every term of the batch has probe rows
 * (ss_bm25_term_probed; irrelevant when the probe budget covered all lists); bit 3 set if some query carries
 * SS_OP_ALL_TERMS_FREQUENT; bit 4 set if the batch consists of SS_OP_PHRASE queries (then all of them must be); bit 5 set if some
 * query carries a field filter (several indexed fields: without it every query reads its terms' merged lists, one per term);
 * bit 6 set if EVERY query has exactly bits 16..23 terms (optional: a batch of nothing but 2- or 3-term intersections is then
 * answered, under the exhaustive strategy or without probe rows, by the 16-bit scan instead of the f32 scan -- 4-10x faster);
 * bit 7 set if some query is a union of several terms under a field filter; bits 24..27 = the most NOT terms any ONE query of the
 * batch has (optional, 0 = not stated: the host then assumes whatever bits 8..15 leave room for beside one scored term, and a
 * TopkCount / Count request with NOT terms under the exhaustive strategy takes the slower f32-tile kernel unless 1 is stated);
 * bit 28 set if some query may name a term of the SPARSE tier (ss_bm25_append_sparse): the batch is then split into its dense and
 * sparse parts on the host -- the queries are copied back once (a stream synchronisation), errors are reported as the host-pointer
 * entry points report them (return code, not per-query flags), the answers are in the device arrays when the call returns.
 * The assertion is CHECKED ON THE DEVICE, query by query, before the search kernels run: a query that contradicts ops_mask
 * (an intersection in a batch declared union-only, more terms than declared, an unprobed term under bit 2, ...) or is
 * malformed (no terms, a term id outside the vocabulary) is answered as an empty query and flagged
 * d_out_count[q] = UINT32_MAX -- never a silently wrong list.  (idf > 0 and unique terms remain the caller's duty.) */
int ss_bm25_search_dev(ss_shard* s, uint32_t n_queries, const ss_bm25_query* d_queries, uint32_t k,
                       uint32_t result_type, uint32_t ops_mask, uint32_t* d_out_doc, float* d_out_score,
                       uint32_t* d_out_count, uint64_t* d_out_total, void* stream);

/* ------------------------------------------------------------------ facet filter (search.rs FacetFilter, add_result.rs:341-482)
 * facet.bin holds one fixed-size record per doc (facets_size_sum bytes, every facet at its offset: facet.json).  A search
 * with a facet filter drops a doc unless EVERY filter passes -- a numeric value inside the half-open range [lo, hi) as
 * Rust's Range::contains, a String16 / String32 id inside the wanted set -- before it is counted (add_result.rs:3499-
 * 3501): a filtered doc neither counts nor ranks.  The filter is evaluated once per call over all docs into an exclusion
 * bitmap that stands in for the tombstone bitmap; all queries of the call share it (the reference has one filter per
 * search call).  lo / hi: the value's bits (two's complement for I*, IEEE bits for F32 in the low word / F64); Timestamp =
 * I64.  Calls that share a shard must be stream-ordered (one bitmap per shard).
 * SS_FACET_POINT (FieldType::Point: the u64 Morton code of (lat, lon) x 1e7, geo_search.rs:27-41) filters by DISTANCE to a
 * base point (FacetFilter::Point -> FilterSparse::Point, search.rs:2712-2722, add_result.rs:462-478): lo / hi = the f64 bits
 * of the distance range, values[0..1] / values[2..3] = the f64 bits (low word first) of the base's latitude / longitude,
 * n_values = SS_POINT_KM | SS_POINT_MILES.  A doc passes iff its code lies inside the reference's Morton range
 * (point_distance_to_morton_range(base, hi, unit), geo_search.rs:128-144, computed by the library into values[4..7]) and
 * lo <= euclidian_distance(base, doc) < hi (geo_search.rs:115-124, f64).  n_values = SS_POINT_SORTKEY compares
 * simplified_distance (geo_search.rs:82-87, the key of a sort by distance) without a Morton range: the pivots of a result sort. */
enum { SS_FACET_U8 = 0, SS_FACET_U16, SS_FACET_U32, SS_FACET_U64, SS_FACET_I8, SS_FACET_I16, SS_FACET_I32, SS_FACET_I64,
       SS_FACET_F32, SS_FACET_F64, SS_FACET_STRING16, SS_FACET_STRING32, SS_FACET_POINT };
enum { SS_POINT_SORTKEY = 0, SS_POINT_KM = 1, SS_POINT_MILES = 2 };
typedef struct ss_facet_point {   /* the base of a Point facet's distances (QueryFacet::Point / ResultSort.base) */
  double lat, lon;
  uint32_t unit;                  /* SS_POINT_KM / _MILES: euclidian_distance; SS_POINT_SORTKEY: simplified_distance */
  uint32_t reserved;
} ss_facet_point;
#define SS_MAX_FACET_FILTERS 8
typedef struct ss_facet_filter {
  uint32_t offset;      /* of the facet inside a record */
  uint32_t type;        /* SS_FACET_* */
  uint64_t lo, hi;      /* numeric types: passes iff lo <= value < hi */
  uint32_t n_values;    /* string types: passes iff the id is one of values[0 .. n_values) (<= 8); SS_FACET_IDS_EXTERN: one of the
                           `hi` ids of the HOST array at (uintptr_t)lo -- any number (a StringSet filter resolves to every set id
                           that holds the value, search.rs:2643-2710) */
  uint32_t values[8];
  uint32_t reserved;    /* numeric types: SS_FACET_LO_EXCLUSIVE | SS_FACET_HI_INCLUSIVE turn the ends around (0 = [lo, hi)) */
} ss_facet_filter;
#define SS_FACET_IDS_EXTERN 0xFFFFFFFFu
#define SS_FACET_HI_INCLUSIVE 1u
#define SS_FACET_LO_EXCLUSIVE 2u
int ss_facet_upload(ss_shard* s, uint64_t n_docs, uint32_t record_size, const uint8_t* records);
int ss_bm25_search_filtered(ss_shard* s, uint32_t n_queries, const ss_bm25_query* queries, uint32_t k, uint32_t result_type,
                            uint32_t n_filters, const ss_facet_filter* filters /* host */, uint32_t* out_doc, float* out_score,
                            uint32_t* out_count, uint64_t* out_total);
int ss_bm25_search_filtered_dev(ss_shard* s, uint32_t n_queries, const ss_bm25_query* d_queries, uint32_t k,
                                uint32_t result_type, uint32_t ops_mask, uint32_t n_filters,
                                const ss_facet_filter* filters /* host */, uint32_t* d_out_doc, float* d_out_score,
                                uint32_t* d_out_count, uint64_t* d_out_total, void* stream);

/* Facet counting of ONE query (query_facets -> facet_count, add_result.rs:484-640): every counted doc -- the query's match set
 * after NOT terms, tombstones and the facet filter -- adds one to the bucket of its value of the facet at facet_offset:
 * a String16 / String32 facet's id (buckets = ids 0 .. n_buckets-1), or for a numeric facet the range whose lower bound is
 * the last one <= the value (range_lower_bounds[n_buckets], ascending, the value's bits as in ss_facet_filter).
 * out_counts [n_buckets + 1]: the last slot = docs outside the buckets (an id >= n_buckets, a value below the first
 * bound).  *out_total (may be NULL) = the match count.  The match set is read from the probe index's bit records:
 * SS_ENOTSUP if a list of the query has no probe row (ss_bm25_term_probed). */
int ss_bm25_facet_count(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                        uint32_t facet_offset, uint32_t facet_type, uint32_t n_buckets, const uint64_t* range_lower_bounds,
                        uint64_t* out_counts, uint64_t* out_total);

/* Result sort (search.rs ResultSort; ordering min_heap.rs:574-1050: the sort fields in order, each ascending or descending, the
 * score last).  ss_bm25_facet_kth finds the PIVOT of such a sort for ONE query: the k-th best value of a numeric facet among
 * the query's matches (after NOT terms, tombstones and the facet filters) -- *out_value = its stored bits, *out_n_better =
 * matches strictly better, *out_n_equal = matches with exactly that value, *out_total = all matches (fewer than k matches:
 * the pivot is the worst match).  The top-k under the sort is then: the n_better docs of a search filtered to "better than
 * the pivot" (SS_FACET_LO_EXCLUSIVE / _HI_INCLUSIVE) ordered by their values (ss_facet_values), followed by the best
 * k - n_better docs of a search filtered to "equal to the pivot" -- by the next sort field the same way, by score when none
 * is left.  String facets sort by their strings: SS_ENOTSUP on the id column itself -- the host appends a u32 rank column
 * (rank of each id's string) to the records it uploads and sorts by that.
 * ss_facet_values: the stored bits of a facet for a list of docs (host arrays).
 * Point facets: the *_point entry points take the base point; the value that is counted into ranges (Ranges::Point,
 * add_result.rs:605-618; bounds = f64 bits of the distances), selected (morton_ordering, min_heap.rs:510-528 /
 * 1017-1036: unit = SS_POINT_SORTKEY) or returned is the f64 distance of each doc's point to it. */
int ss_bm25_facet_kth(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                      uint32_t facet_offset, uint32_t facet_type, uint32_t descending, uint64_t k, uint64_t* out_value,
                      uint64_t* out_n_better, uint64_t* out_n_equal, uint64_t* out_total);
int ss_facet_values(ss_shard* s, uint32_t n, const uint32_t* doc_ids, uint32_t facet_offset, uint32_t facet_type, uint64_t* out_values);
/* The same sort as ONE call for a batch of queries, pivots on the device: per query the radix selects of all sort fields run
 * back to back on the device (the prefix never visits the host), leaving two doc sets -- the docs that are in the answer for sure
 * (strictly better than a pivot at some field) and the tie group of the last pivot --, which two ordinary searches under exclusion
 * bitmaps turn into scored lists and a compose kernel orders by (field 1, ..., field n, score desc, doc asc).  The host only
 * launches: one synchronisation per call.  n_sorts <= SS_MAX_SORT_FIELDS numeric or Point fields (n_sorts = 0: by score alone);
 * every list of every query needs a probe row (SS_ENOTSUP otherwise, as for ss_bm25_facet_kth); no phrase queries.
 * out_doc / out_score [n_queries][k], out_count [n_queries], out_total [n_queries] = all matches of the query. */
#define SS_MAX_SORT_FIELDS 4
typedef struct ss_result_sort {   /* search.rs ResultSort */
  uint32_t facet_offset;
  uint32_t facet_type;            /* SS_FACET_U8 .. SS_FACET_F64, or SS_FACET_POINT: by simplified_distance to the base */
  uint32_t descending;            /* SortOrder */
  uint32_t reserved;
  double base_lat, base_lon;      /* Point facets */
} ss_result_sort;
int ss_bm25_search_sorted(ss_shard* s, uint32_t n_queries, const ss_bm25_query* queries, uint32_t n_sorts, const ss_result_sort* sorts,
                          uint32_t k, uint32_t n_filters, const ss_facet_filter* filters, uint32_t* out_doc, float* out_score,
                          uint32_t* out_count, uint64_t* out_total);
int ss_bm25_facet_count_point(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                              uint32_t facet_offset, const ss_facet_point* base, uint32_t n_buckets,
                              const uint64_t* range_lower_bounds, uint64_t* out_counts, uint64_t* out_total);
int ss_bm25_facet_kth_point(ss_shard* s, const ss_bm25_query* query, uint32_t n_filters, const ss_facet_filter* filters,
                            uint32_t facet_offset, const ss_facet_point* base, uint32_t descending, uint64_t k, uint64_t* out_value,
                            uint64_t* out_n_better, uint64_t* out_n_equal, uint64_t* out_total);
int ss_facet_point_distances(ss_shard* s, uint32_t n, const uint32_t* doc_ids, uint32_t facet_offset, const ss_facet_point* base,
                             uint64_t* out_values);

/* ------------------------------------------------------------------ vector image
 * rows: row-major [n_rows x dim] f32, already L2-normalised for cosine (vector.rs:585-596); the uploader of
 * a real vector.bin strips the 24-byte VectorHeader (vector.rs:62-73).  row_doc_ids may be NULL (= row index).
 * Several records may share a doc id (one per indexed field x chunk, vector.rs:561-576): the search then returns each
 * doc once with its best record's score, as TopK::push does (vector.rs:441-452, 462-473). */
int ss_vec_upload(ss_shard* s, uint64_t n_rows, uint32_t dim, const float* rows, const uint32_t* row_doc_ids);
/* VectorSimilarity of the shard's vector image (the index-wide setting the reference reads as shard.vector_similarity;
 * similarity arms vector_similarity.rs:118-345, 880-908).  SS_SIM_DOT (default) = Dot and Cosine: a record's similarity is
 * the dot product.  SS_SIM_EUCLIDEAN: MINUS the squared distance -- -euclidean_f32 (912) for f32 records, -euclidean_i8
 * (921) for i8 records, -euclidean_i8_quantized = -max(0, norm1 + norm2 - 2 dot_i32 scale1 scale2) (1721-1735) for i8 records
 * with quantisation scales -- so that larger is still closer and TopK / ANN selection work unchanged; threshold_raw of a
 * search is then -similarity_threshold (vector.rs:398), scores come back as -distance^2 (vector_score = -score, vector.rs:
 * 1495).  Returned f32 scores are recomputed in the reference's own summation order (euclidean_f32_avx2 when dim % 8 == 0).
 * Must be set BEFORE the image is uploaded / generated (the f32 image is laid out with two extra columns); SS_ESTATE
 * otherwise.  Applies to AnnMode::All and the ANN modes (medoids are scored with the same similarity). */
enum { SS_SIM_DOT = 0, SS_SIM_EUCLIDEAN = 1 };
int ss_vec_set_similarity(ss_shard* s, int similarity);
/* i8 records under Euclidean + ScalarQuantizationI8: VectorHeader.norm of every record (vector.bin uploads with
 * use_record_scale keep it themselves).  The query's norm goes into ss_vec_search_i8_ann[_dev]'s query_norm. */
int ss_vec_set_row_norms(ss_shard* s, uint64_t n_rows, const float* row_norm);

/* Append of ONE committed level of vector records (commit.rs:142-148 -> the writer of vector.rs:1066-1094 puts a level's clusters
 * and records behind the earlier levels'): the commit seam of the vector image, beside ss_bm25_append_level.  The records are written
 * behind the image's rows in place -- O(level), not O(shard); the image and its per-row arrays grow by half when their room is used up.
 * The level brings exactly what the image carries per row: doc ids iff the image was uploaded with them, scales / norms iff the i8
 * image has them (ss_vec_upload_i8 / ss_vec_set_row_norms), field ids iff it has them (ss_vec_set_fields), its clusters (child
 * counts summing to n_rows) iff the image has a cluster structure (ss_vec_set_clusters / a vector.bin upload) -- SS_EINVAL otherwise.
 * Searches on the shard wait for the append (the call synchronises the device); answers afterwards are those of a one-shot upload of
 * all rows. */
typedef struct ss_vec_level {
  uint64_t n_rows;
  const void* rows;             /* [n_rows][dim]: f32, or i8 when elem_i8 != 0 -- the image's own precision */
  uint32_t elem_i8;
  uint32_t n_clusters;          /* clusters of the level (0 = the image has no cluster structure) */
  const uint32_t* row_doc_ids;  /* [n_rows] or NULL */
  const float* row_scale;       /* [n_rows] or NULL (i8 image with per-record scales) */
  const float* row_norm;        /* [n_rows] or NULL (i8, Euclidean, quantised) */
  const uint16_t* row_field;    /* [n_rows] or NULL */
  const uint32_t* child_count;  /* [n_clusters] or NULL */
} ss_vec_level;
int ss_vec_append_rows(ss_shard* s, const ss_vec_level* level);
/* Room for n_rows_cap rows in the image and its per-row arrays, taken once (at open time) instead of by the first append that finds
 * none: growing a 30 GB image means a 45 GB allocation and a device-to-device copy -- 1.0 s measured at 10 M x 768 f32 --, an append
 * into reserved room 3.6 ms per 65 536-row level (profiles/r4m_vec_append.log). */
int ss_vec_reserve_rows(ss_shard* s, uint64_t n_rows_cap);
/* Device-side synthetic matrix (generator = oracle so_vec_gen, uniform(-1,1) then normalize_f32). */
/* Rows straight from a shard's vector.bin (writer vector.rs:1066-1094): per level u32 cluster_count + child counts,
 * then 24-byte VectorHeader + dim x f32 records; doc id = (level << 16) | header.doc_id (vector.rs:1448).  f32 only. */
int ss_vec_upload_vector_bin(ss_shard* s, const uint8_t* bytes, uint64_t len, uint32_t dim);
int ss_vec_synth(ss_shard* s, uint64_t seed, uint64_t n_rows, uint32_t dim);
int ss_vec_info(ss_shard* s, uint64_t* n_rows, uint32_t* dim);
/* copy rows [r0, r0+n) back to the host (test accessor) */
int ss_vec_read_rows(ss_shard* s, uint64_t r0, uint64_t n, float* out);

/* Batched brute-force scan (AnnMode::All).  queries: [n_queries x dim] f32 normalised.  threshold_raw is
 * compared as `score < threshold_raw -> reject` (vector.rs:423; pass -FLT_MAX for none).
 * out_total = number of rows that passed the running top-k filter (vector.rs:429 counts accepted pushes;
 * order-dependent in the reference, an upper bound here; exact when n_rows <= k). */
int ss_vec_search(ss_shard* s, uint32_t n_queries, const float* queries, uint32_t k, float threshold_raw,
                  uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total);
/* _dev variants: device pointers, asynchronous on `stream`.  The scan keeps the rows that beat the running threshold in a
 * fixed candidate buffer per query, sized for rows in no particular order; rows sorted by similarity to a query can
 * overflow it.  The host-pointer functions notice and re-run the batch with a schedule that cannot overflow; the _dev
 * functions cannot look at the result, so they flag it: d_out_count[q] = UINT32_MAX for the queries of that batch --
 * re-issue such a batch through the host-pointer function.  (ss_rrf_merge_dev / ss_topk_merge_dev treat it as empty.) */
int ss_vec_search_dev(ss_shard* s, uint32_t n_queries, const float* d_queries, uint32_t k, float threshold_raw,
                      uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total,
                      void* stream);

/* ---- i8 (quantised) vectors: the reference's Precision::I8 embeddings (quantize_f32_to_i8 = round(v * 127) clamped to
 * +-127, vector_similarity.rs:1226-1232; or a per-record scale with ScalarQuantizationI8).  Score of a record =
 * dot_i8 as f32 (vector_similarity.rs:1011-1016; Cosine, and Dot without quantisation scales), or with scales
 * dot_i8_quantized = dot as f32 * query_scale * embedding_scale (1754-1758).  The integer dot product is exact, so
 * scores are bit-identical to the reference's.  A shard holds ONE vector image: uploading i8 rows replaces f32 rows and
 * vice versa.  row_scale / query_scale may be NULL (= no scaling).  dim <= 2560.  Semantics of k, threshold_raw,
 * row_doc_ids, outputs and tombstones as for the f32 functions. */
int ss_vec_upload_i8(ss_shard* s, uint64_t n_rows, uint32_t dim, const int8_t* rows, const float* row_scale,
                     const uint32_t* row_doc_ids);
/* vector.bin with Precision::I8 records (24-byte VectorHeader + dim x i8); use_record_scale keeps VectorHeader.scale
 * for dot_i8_quantized (ScalarQuantizationI8 with Dot), 0 for Cosine / unscaled Dot whose score is the raw integer dot */
int ss_vec_upload_vector_bin_i8(ss_shard* s, const uint8_t* bytes, uint64_t len, uint32_t dim, int use_record_scale);
int ss_vec_synth_i8(ss_shard* s, uint64_t seed, uint64_t n_rows, uint32_t dim); /* ss_vec_synth rows, quantised on the device */
int ss_vec_read_rows_i8(ss_shard* s, uint64_t first_row, uint64_t n, int8_t* out);
int ss_vec_search_i8(ss_shard* s, uint32_t n_queries, const int8_t* queries, const float* query_scale, uint32_t k,
                     float threshold_raw, uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total);
int ss_vec_search_i8_dev(ss_shard* s, uint32_t n_queries, const int8_t* d_queries, const float* d_query_scale, uint32_t k,
                         float threshold_raw, uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count,
                         uint64_t* d_out_total, void* stream);

/* ---- ANN modes (AnnMode::Nprobe / Similaritythreshold / NprobeSimilaritythreshold, vector.rs:1300-1392).
 * The reference clusters the records of every level (65 536 docs) at commit time and stores them cluster after cluster;
 * the first record of a cluster is its medoid.  A query scores the medoids of a level, keeps the best n_probe of them
 * (TopK::new(n_probe, cluster threshold), so ties keep the earlier cluster) and visits only those clusters' records.
 * ss_vec_upload_vector_bin[_i8] keeps the cluster structure of the file; for rows uploaded with ss_vec_upload[_i8] /
 * ss_vec_synth[_i8] in cluster order, ss_vec_set_clusters declares it: level_clusters[n_levels] = clusters per level,
 * child_count[sum of level_clusters] = records per cluster (> 0), sum = n_rows.
 * Medoid similarity uses the reference's own summation order (dot_f32_avx2's 8 fmadd lanes when dim % 8 == 0, else the
 * sequential dot_f32; the i8 dot is exact), so the selected clusters are the reference's, not merely close to them.
 * A batch scans the union of its queries' clusters once; a row is a candidate only for the queries that selected its
 * cluster.  mode == NULL is AnnMode::All.  out_clusters (may be NULL) = observed_cluster_count per query.
 * The selection state of a search with a mode (medoid scores, cluster bitmaps, tile list) is one set of buffers per shard: such
 * searches queued on different streams are ordered one after the other by the library (an event wait), searches without a mode
 * stay concurrent. */
typedef struct ss_ann_mode {
  uint32_t n_probe;              /* clusters visited per level; 0 = no limit (AnnMode::Similaritythreshold) */
  float cluster_threshold_raw;   /* clusters whose medoid scores below it are skipped; -FLT_MAX = none (AnnMode::Nprobe) */
  uint64_t field_mask;           /* field_filter of search_vector_shard (vector.rs:1225-1237, 1397-1400): bit f set = records
                                  * of indexed field f are searched, the others are skipped before they are scored; 0 = every
                                  * field.  Needs the records' field ids (vector.bin upload, or ss_vec_set_fields); fields
                                  * >= 64 cannot be selected.  n_probe = 0 and no cluster threshold = AnnMode::All + filter. */
  uint32_t flags;                /* SS_ANN_* */
  uint32_t reserved;             /* 0 */
} ss_ann_mode;
/* flags bit 0: also report observed_vector_count (TopK::push calls, vector.rs:421, 1510): the records of the visited clusters
 * that pass the field filter and are not tombstoned (vector.rs:1397-1400, 1450-1452).  out_clusters then holds THREE words per
 * query: [3 q] observed_cluster_count, [3 q + 1] / [3 q + 2] the low / high word of observed_vector_count.  A mode that skips
 * no cluster (n_probe 0, no threshold) reports observed_cluster_count 0 -- every cluster, the host knows their number
 * (ss_vec_cluster_info) -- and the count over the whole image. */
#define SS_ANN_REPORT_OBSERVED 1u
/* VectorHeader.field_id of every record, for rows uploaded with ss_vec_upload[_i8] / ss_vec_synth[_i8] */
int ss_vec_set_fields(ss_shard* s, uint64_t n_rows, const uint16_t* row_field);
int ss_vec_set_clusters(ss_shard* s, uint32_t n_levels, const uint32_t* level_clusters, const uint32_t* child_count);
int ss_vec_cluster_info(ss_shard* s, uint32_t* n_levels, uint32_t* n_clusters);
int ss_vec_search_ann(ss_shard* s, uint32_t n_queries, const float* queries, uint32_t k, float threshold_raw,
                      const ss_ann_mode* mode, uint32_t* out_doc, float* out_score, uint32_t* out_count,
                      uint64_t* out_total, uint32_t* out_clusters);
int ss_vec_search_ann_dev(ss_shard* s, uint32_t n_queries, const float* d_queries, uint32_t k, float threshold_raw,
                          const ss_ann_mode* mode, uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count,
                          uint64_t* d_out_total, uint32_t* d_out_clusters, void* stream);
int ss_vec_search_i8_ann(ss_shard* s, uint32_t n_queries, const int8_t* queries, const float* query_scale, uint32_t k,
                         float threshold_raw, const ss_ann_mode* mode, uint32_t* out_doc, float* out_score,
                         uint32_t* out_count, uint64_t* out_total, uint32_t* out_clusters);
int ss_vec_search_i8_ann_dev(ss_shard* s, uint32_t n_queries, const int8_t* d_queries, const float* d_query_scale,
                             uint32_t k, float threshold_raw, const ss_ann_mode* mode, uint32_t* d_out_doc,
                             float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, uint32_t* d_out_clusters,
                             void* stream);
/* the same with the per-query norm euclidean_i8_quantized needs (QuantizedVector.norm of the quantised query; NULL = 0) */
int ss_vec_search_i8_euclid(ss_shard* s, uint32_t n_queries, const int8_t* queries, const float* query_scale,
                            const float* query_norm, uint32_t k, float threshold_raw, const ss_ann_mode* mode, uint32_t* out_doc,
                            float* out_score, uint32_t* out_count, uint64_t* out_total, uint32_t* out_clusters);
int ss_vec_search_i8_euclid_dev(ss_shard* s, uint32_t n_queries, const int8_t* d_queries, const float* d_query_scale,
                                const float* d_query_norm, uint32_t k, float threshold_raw, const ss_ann_mode* mode,
                                uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total,
                                uint32_t* d_out_clusters, void* stream);

/* ------------------------------------------------------------------ cross-shard merge + RRF (host side)
 * Inputs are the concatenation over shards of per-shard top-(offset+length) lists with GLOBAL ids
 * (global = local*S + shard, search.rs:1671).  Hybrid = RRF k=0.6, 0-based ranks (search.rs:1962-2035).
 * Returns the number of results written (<= length) or a negative code. */
int ss_merge_results(int mode, const uint64_t* lex_doc, const float* lex_score, uint32_t n_lex,
                     const uint64_t* vec_doc, const float* vec_score, uint32_t n_vec, uint32_t offset,
                     uint32_t length, uint64_t* out_doc, float* out_score, uint8_t* out_source);

/* Device-side variant for the multi-GPU path: per-shard top-k lists of a whole query batch, laid out
 * [n_shards][n_queries][k] exactly as an RCCL all-gather of the ss_*_search_dev outputs leaves them, are merged
 * per query into global ids (local*S + shard), sorted by score desc (ties: shard order, as the reference's stable
 * sort of the concatenation).  Unused slots: doc = UINT64_MAX.  Any n_shards * k: up to 8192 keys per query are sorted in LDS; beyond
 * that (more than 8 shards at k = 1024, a deep page) every entry finds its slot by binary searches in the other shards' lists -- the
 * lists must then arrive sorted by score descending, as the searches leave them. */
int ss_topk_merge_dev(int device, uint32_t n_queries, uint32_t n_shards, uint32_t k, const uint32_t* d_doc,
                      const float* d_score, const uint32_t* d_count, uint64_t* d_out_doc, float* d_out_score,
                      uint32_t* d_out_count, void* stream);
/* the same over ONE gathered buffer: per shard [n_queries * k doc ids | n_queries * k score bits | n_queries counts] as
 * 32-bit words -- a single all-gather per batch (latency bound: three collectives cost three latencies) */
int ss_topk_merge_dev_packed(int device, uint32_t n_queries, uint32_t n_shards, uint32_t k, const uint32_t* d_packed,
                             uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, void* stream);
/* the WHOLE concatenation of the gathered lists, sorted: outputs [n_queries][n_shards * k] -- what the RRF ranks of a hybrid
 * search over several shards run over (search.rs:1962-2035 sorts the appended lists untruncated); feeds ss_rrf_merge_dev with
 * k_lex / k_vec = n_shards * k (ss_rrf_merge_dev takes k_lex + k_vec <= 8192). */
int ss_topk_concat_dev_packed(int device, uint32_t n_queries, uint32_t n_shards, uint32_t k, const uint32_t* d_packed,
                              uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, void* stream);
/* ---- the exchange itself, behind the ABI (RCCL over xGMI; replaces search.rs:1875-1940 + 2098-2119 for shards on different
 * GPUs).  A communicator is one rank of a group of shards, rank = shard id (global id = local * n_ranks + rank, search.rs:1671).
 *   one process per GPU : rank 0 calls ss_comm_unique_id and hands the 128 bytes to the other ranks (any channel: MPI,
 *                         a TCP store, a file); every rank calls ss_comm_create with the same id;
 *   one process, S GPUs : ss_comm_create_all(S, devices, out[S]) (the reference's shape: one process, one task per shard).
 * ss_topk_allgather_merge: packs this shard's top-k lists of the batch (outputs of ss_*_search_dev, [n_queries][k]), ONE
 * all-gather, then ss_topk_merge_dev_packed -- every rank ends with the same merged lists.  Asynchronous on `stream`; all
 * ranks must call it with the same n_queries and k (in a single process: from one thread per rank, as Index::search runs its
 * shard tasks).  Any n_ranks * k (see ss_topk_merge_dev). */
#define SS_COMM_ID_BYTES 128
typedef struct ss_comm ss_comm;
int ss_comm_unique_id(uint8_t id_out[SS_COMM_ID_BYTES]);
int ss_comm_create(int device, int rank, int n_ranks, const uint8_t id[SS_COMM_ID_BYTES], ss_comm** out);
int ss_comm_create_all(int n_devices, const int* devices, ss_comm** out /*[n_devices]*/);
int ss_comm_destroy(ss_comm* c);
int ss_comm_info(const ss_comm* c, int* rank, int* n_ranks, int* device);
int ss_topk_allgather_merge(ss_comm* c, uint32_t n_queries, uint32_t k, const uint32_t* d_doc, const float* d_score,
                            const uint32_t* d_count, uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, void* stream);
/* One shard's part of <IndexArc as Search>::search when the shards live on different GPUs (search.rs:1637-1743 per-shard
 * task, 1669-1673 global ids, 1875-1940 gather, 1884-1921 totals summed, 2098-2119 sort / truncate): searches shard `s`
 * (host queries, as ss_bm25_search), exchanges through `c` -- ONE all-gather: lists, totals and a status word --
 * (the totals ride in the same gather) and hands EVERY rank the merged answer: out_doc [n_queries][k] GLOBAL ids (local * n_ranks + rank; UINT64_MAX = unused),
 * out_score, out_count, out_total (sum over the shards).  Collective over the communicator: every rank calls it with the
 * same n_queries / k / result_type (each with its own shard's idf in the queries).  c's device must be s's. */
int ss_bm25_search_sharded(ss_shard* s, ss_comm* c, uint32_t n_queries, const ss_bm25_query* queries, uint32_t k,
                           uint32_t result_type, uint64_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total);

/* The vector and the hybrid shard task of the same search (search.rs:1680-1689, 1723-1732), same conventions and the same
 * single all-gather: ss_vec_search_sharded = this shard's AnnMode::All f32 scan (host queries [n_queries][dim]; any number of
 * them: SS_VEC_BATCH per pass over the matrix, one all-gather for the call) + merged top-k on every rank.  ss_hybrid_search_sharded = SearchMode::Hybrid: both shard tasks at
 * k = offset + length, both lists in the one all-gather, the two cross-shard concatenations sorted, RRF over them (ranks run
 * over the whole concatenation, search.rs:1962-2035), sort / offset / length (2098-2119); out_* [n_queries][length] with
 * out_source SS_SRC_* (may be NULL); out_total = sum over the shards of max(lexical, vector) totals (1919-1921).
 * result_type: SS_RT_TOPK or SS_RT_TOPKCOUNT (the lexical task's).  Any k and any n_ranks * k on all three (k > SS_MAX_K: this
 * shard's lists in passes, see SS_MAX_K; the hybrid fusion runs on the device while n_ranks * k * 2 <= 8192 and k <= SS_MAX_K, on the
 * host -- the same f32 operations, ss_merge_results -- beyond).
 * Every ss_*_search_sharded call is a collective in which a rank takes part EVEN IF its own search failed (with empty lists and
 * a status word): that rank returns its own error, every other rank SS_EPEER -- no rank is left blocking in the exchange. */
int ss_vec_search_sharded(ss_shard* s, ss_comm* c, uint32_t n_queries, const float* queries, uint32_t k, float threshold_raw,
                          uint64_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total);
int ss_hybrid_search_sharded(ss_shard* s, ss_comm* c, uint32_t n_queries, const ss_bm25_query* queries, uint32_t result_type,
                             const float* query_vectors, float threshold_raw, uint32_t k, uint32_t offset, uint32_t length,
                             uint64_t* out_doc, float* out_score, uint8_t* out_source, uint32_t* out_count, uint64_t* out_total);
/* time of the collective itself: when on, every all-gather of the ss_*_search_sharded calls is bracketed with HIP events on
 * the shard's stream; read = (collectives, their summed microseconds) */
int ss_comm_profile(ss_comm* c, int on);
int ss_comm_profile_read(ss_comm* c, uint64_t* collectives, double* total_us, int reset);

/* Hybrid fusion of a query batch on the device: RRF (k = 0.6, 0-based ranks, search.rs:1962-2035) of the lexical and the
 * vector list of every query, then sort / offset / length (2098-2119) -- ss_merge_results(SS_MODE_HYBRID) for n_queries
 * queries without a host round trip.  d_lex_doc [n_queries][k_lex] and d_vec_doc [n_queries][k_vec] are the doc-id
 * outputs of ss_bm25_search_dev / ss_vec_search*_dev (u32 shard-local ids, doc_ids_are_u64 = 0) or of ss_topk_merge_dev*
 * (u64 global ids, = 1), with their counts; both sorted by score descending with unique ids (the scores themselves are not
 * needed: only ranks enter the fusion).  Outputs [n_queries][length]: doc (UINT64_MAX = unused), fused score,
 * source (SS_SRC_*, may be NULL), count.  Equal fused scores: doc id ascending.  k_lex + k_vec <= 8192 (SS_ENOTSUP beyond); k_lex = 0 or
 * k_vec = 0 leaves the other list's ranks as scores. */
int ss_rrf_merge_dev(int device, uint32_t n_queries, uint32_t k_lex, const void* d_lex_doc, const uint32_t* d_lex_count,
                     uint32_t k_vec, const void* d_vec_doc, const uint32_t* d_vec_count, int doc_ids_are_u64, uint32_t offset,
                     uint32_t length, uint64_t* d_out_doc, float* d_out_score, uint8_t* d_out_source, uint32_t* d_out_count,
                     void* stream);

/* ------------------------------------------------------------------ measurement hooks
 * When enabled the library brackets every launch of the dominant kernels with HIP events on the stream
 * the kernel is launched on and accumulates (launches, milliseconds).  kernel: 0 = bm25 scan, 1 = vector scan. */
int ss_profile_enable(ss_shard* s, int on);
int ss_profile_read(ss_shard* s, int kernel, uint64_t* launches, double* total_ms, int reset);

#ifdef __cplusplus
}
#endif
#endif
