// Internal declarations shared by the HIP translation units.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/seekstorm_hip.h"

// ---------------------------------------------------------------- vector scan geometry
constexpr int VS_WAVES = 4;                 // waves per workgroup
constexpr int VS_TR = VS_WAVES * 32;        // X rows per tile (one 32-row MFMA block per wave)
constexpr int VS_KC = 32;                   // K chunk (floats) = one 128-byte line per row
constexpr int VS_STAGES = 3;                // LDS ring depth
constexpr int VS_XS = VS_TR * VS_KC * 4;    // bytes of X per stage
constexpr int VS_QS = SS_VEC_BATCH * VS_KC * 4;  // bytes of Q per stage (fragment order)
constexpr int VS_STAGE = VS_XS + VS_QS;
constexpr int VS_LDS = VS_STAGES * VS_STAGE;
constexpr uint32_t VS_CAP = 8192;           // candidate slots per query
#ifndef VS_GRID_MULT
#define VS_GRID_MULT 32u                    // scan workgroups per launch = 512 x this (vec_scan.hip: how long a workgroup holds its CU)
#endif
constexpr int VS_FIRST_TILES = 16;          // first chunk: 2048 rows, everything is a candidate (64 tiles: the 8192-entry
                                            // refine after it costs more than the launch it saves)

// f32 operations rounded ON THEIR OWN, as the reference (Rust never contracts a * b + c) computes them.  HIP's __fmul_rn /
// __fadd_rn are plain `*` / `+` and the default -ffp-contract=fast fuses them into an fma; an operation emitted under
// contract(off) carries no `contract` flag and stays unfused after inlining.
__device__ __forceinline__ float ss_fmul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float ss_fadd(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float ss_fsub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// SmallFloat decode (DOCUMENT_LENGTH_COMPRESSION[b]), index.rs:4255-4268
__host__ __device__ inline uint32_t ss_byte4_to_int(uint32_t b) {
  if (b < 24u) return b;
  uint32_t i = b - 24u, bits = i & 7u, shift = i >> 3;
  return shift == 0 ? 24u + bits : 24u + ((bits | 8u) << (shift - 1u));
}

// ---------------------------------------------------------------- BM25 image geometry
constexpr int BM_SUB_LOG2 = 12;             // docs per sub-block = 4096 = one wave's 16 KB LDS accumulator tile
                                            // (2048 with 16 waves/CU measured 17% slower: per-item overhead dominates)
constexpr int BM_SUB = 1 << BM_SUB_LOG2;
#ifndef SS_BM_WAVES_OR
#define SS_BM_WAVES_OR 8
#endif
#ifndef SS_BM_WAVES_AND
#define SS_BM_WAVES_AND 6
#endif
constexpr int BM_WAVES_OR = SS_BM_WAVES_OR;    // waves per workgroup, union-only kernels (16.4 KB of LDS per wave)
constexpr int BM_WAVES_AND = SS_BM_WAVES_AND;  // kernels that also carry match counters (20.5 KB per wave)
constexpr uint32_t BM_NO_PROBE_ROW = 0xFFFFFFFFu;
// Packed posting (one dword): the doc AND its finished BM25 weight, so that scoring a posting is one multiply-add with
// idf and never a table or doc-length lookup:
//   bits  0..12  doc field = doc-in-sub-block + 1 (1..4096); 0 = the NULL posting's dump slot
//   bits 13..31  W19 = the posting's weight  tf (K + 1) / (tf + bm25_component_cache[len])  (add_result.rs:1445-1447 without
//                idf), computed in f32 exactly as the reference computes it and then rounded to 15 mantissa bits:
//                weight = as_float((W19 << 8) + BM_W_BASE), binades 2^-14 .. 2^2, relative rounding error <= 2^-16 = 1.5e-5
//                (north_star's tolerance is 1e-4).  Any tf fits: there is no tf field, no table and no exception list.
//                Lists that can take part in the all_terms_frequent shortcut (df >= N / 2, intersection.rs:198-209) give
//                up the last mantissa bit for the one thing that rule asks of a posting: W19 bit 0 = (tf < 10).
// The all-zero dword is the NULL posting (segment padding, and what an out-of-range buffer load returns): its doc field
// addresses the dump slot in front of a wave's accumulator tile; whatever weight it decodes to lands there.
constexpr uint32_t BM_W_BASE = 0x38800000u;  // f32 bits of 2^-14
__host__ __device__ inline uint32_t bm_f2u(float f) { union { float f; uint32_t u; } x; x.f = f; return x.u; }
__host__ __device__ inline float bm_u2f(uint32_t u) { union { float f; uint32_t u; } x; x.u = u; return x.f; }
__host__ __device__ inline uint32_t bm_wcode(float w) {  // round to nearest; weights below 2^-14 clamp to the smallest code
  const uint32_t b = bm_f2u(w);
  if (b < BM_W_BASE + 128u) return 1u;
  const uint32_t c = (b - BM_W_BASE + 128u) >> 8;
  return c > 0x7FFFFu ? 0x7FFFFu : c;
}
__host__ __device__ inline float bm_wdecode(uint32_t c) { return bm_u2f((c << 8) + BM_W_BASE); }
__host__ __device__ inline float bm_weight(uint32_t p) { return bm_u2f(((p >> 13) << 8) + BM_W_BASE); }  // weight of a posting
__host__ __device__ inline uint32_t bm_pack(uint32_t doc_in_sub, uint32_t wcode) { return (wcode << 13) | ((doc_in_sub + 1u) & 0x1FFFu); }
__host__ __device__ inline uint32_t bm_doc_field(uint32_t p) { return p & 0x1FFFu; }                  // doc-in-sub-block + 1
__host__ __device__ inline uint32_t bm_tf_lt10(uint32_t p) { return (p >> 13) & 1u; }                 // flagged lists only
// one term of get_bm25f_multiterm_singlefield without idf (add_result.rs:1445-1447; SIGMA = 0): the operations and their
// order are the reference's (tf * (K + 1.0) / (tf + bm25_component)), each rounded to f32
inline float bm_weight_exact(uint32_t tf, float comp_len) {
  const float t = (float)tf;
  const volatile float num = t * (1.2f + 1.0f);  // volatile: no contraction / reassociation whatever the host flags
  const volatile float den = t + comp_len;
  return num / den;
}
// per-wave LDS: [12 B pad][dump f32][tile BM_SUB f32] (+ [3 B pad][dump u8][BM_SUB u8 match counters])
constexpr int BM_WAVE_ACC = 16 + BM_SUB * 4;
constexpr int BM_WAVE_CNT = 16 + BM_SUB;  // counters of doc d at byte 4 + d; the dump counter at byte 3

struct ss_prof {
  bool on = false;
  uint64_t launches[2] = {0, 0};
  double ms[2] = {0.0, 0.0};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[2];
};

// BM25 workspace of ONE stream: expanded queries + partition lists / totals / thresholds / bounds.  Searches queued on
// different streams of one shard run concurrently on the device, so each stream owns its buffers (grow-only; a buffer is
// only replaced after its stream has drained).  The host-pointer entry points use the shard's own stream.
struct ss_bm_ws {
  void* d_vq = nullptr;
  size_t vq_cap = 0;
  uint64_t* d_part = nullptr;
  size_t part_cap = 0;
};

// The vector scans' per-batch buffers (queries in fragment order, thresholds / counters, candidate keys, i8 Euclidean side
// values): the shard's own set serves its own stream; a _dev search on another stream gets a set of its own, so that scans
// queued on different streams of one shard overlap safely.  (The ANN preparation buffers stay per shard: ANN searches of
// one shard must be stream-ordered.)
struct ss_vec_ws {
  float* d_Qf = nullptr;
  uint32_t* d_vstate = nullptr;
  uint64_t* d_cand = nullptr;
  float* d_qaux = nullptr;
};

// ---- coalescing of concurrent small host-pointer searches (ss_api.hip; seekstorm_hip.h ss_shard_set_coalescing).  Group commit:
// requests that arrive while a batch runs are queued; the previous leader hands leadership to the thread of the queue's front,
// which runs every compatible queued request as ONE device batch.  A request's thread waits on its own state word (spin, then
// futex), so completing a batch wakes exactly its members.
struct ss_co_req {
  const void* q = nullptr;          // lexical: ss_bm25_query[nq]; vector: nq rows of dim elements (f32 or i8)
  const float* qscale = nullptr;    // i8 vectors: per-query scales or null
  uint32_t nq = 0, k = 0, rt = 0, elem = 0;
  float thr = 0.f;
  uint32_t* out_doc = nullptr;
  float* out_score = nullptr;
  uint32_t* out_count = nullptr;
  uint64_t* out_total = nullptr;
  int rc = 0;
  std::atomic<uint32_t> state{0};   // 0 = pending, 1 = done, 3 = "lead the next batch"; bit 2 (value 4) = its thread sleeps on the futex
  uint32_t lane = 0xFFFFFFFFu;      // the lane a leader works on (its own on arrival, its predecessor's when told to lead); none: a follower
};
struct ss_coalescer {
  std::mutex mu;
  std::deque<ss_co_req*> queue;
  std::atomic<uint32_t> waking{0};     // members of finished batches whose leader has not signalled them yet (see the linger in co_submit)
  std::atomic<uint32_t> queued_nq{0};  // queries of the queued requests: what a lingering leader polls WITHOUT taking mu (it used to lock mu
                                       // every ~0.3 us of its wait, against every caller that was trying to enqueue)
  // LANES (round 4): up to n_lanes batches in flight at once.  Every lane has its own pinned staging and completion event; the device
  // work of all lanes goes to the shard's ONE stream in the order it was enqueued (so every stream synchronisation elsewhere in the
  // library still covers it), but a leader holds the shard mutex only while it enqueues and waits for its batch's event outside --
  // the next leader stages, checks and enqueues its batch while this one runs, and distributes results while the next one runs.
  uint32_t leaders = 0;             // leaders at work (<= n_lanes); invariant: a non-empty queue has a leader or a successor told to lead
  uint32_t n_lanes = 1;
  bool lanes_forced = false;        // SS_COALESCE_LANES=2: the second lane whatever the number of callers
  uint32_t lanes_from = 32;         // adaptive: the second lane opens while this many callers seem to be around (SS_COALESCE_LANES=autoN: N;
                                    // 96 while both lanes fed one stream, 32 since each has its own: ss_api.hip ss_shard_create)
  struct Lane { char* h_pin = nullptr; size_t h_pin_cap = 0; hipEvent_t ev = nullptr; bool busy = false; } lane[2];
  uint32_t max_batch = 0, max_wait_us = 0;
  uint64_t batches = 0, queries = 0;
  // linger: how many callers seem to be around (members of the last batch + what was queued when it finished) and how long that
  // batch took -- the next leader gives the callers the last batch has just released a moment to come back (co_submit)
  uint32_t callers_est = 0, last_batch_us = 0;
  uint32_t seen_ring[4] = {0, 0, 0, 0}, seen_at = 0;  // callers seen at the end of the last four batches
  // (host staging of a merged batch: PINNED, hipHostMalloc, grow-only, per lane -- the copies to and from the device are then real
  // asynchronous DMA instead of staged pageable copies; only the lane's leader of the moment touches it)
};

// Incremental images (ss_bm25_append_level): the decoded postings of every committed level stay in HBM as they arrived -- (doc, tf)
// per term -- and the image is REBUILT from them on the device after each commit (bm25 weights depend on avgdl, which every commit
// moves: there is nothing to patch).  A level = 65 536 docs = 16 sub-blocks: every sub-block belongs to exactly one level's arrays.
struct ss_raw_level {
  uint32_t n_docs = 0, n_terms = 0;  // docs of the level (65 536, the last one may hold fewer), terms it knows (<= the image's)
  uint64_t n_post = 0, psum = 0;     // postings; sum of the SmallFloat-decoded length bytes of its docs
  uint64_t* d_off = nullptr;         // [n_terms + 1]
  uint32_t* d_doc = nullptr;         // shard-local doc ids, ascending inside a term
  uint16_t* d_tf = nullptr;
  // positions (ss_bm25_append_level_positions; all levels of an image carry them, or none)
  uint16_t* d_npos = nullptr;        // [n_post] positions of every posting (tf, or what the caller said: n-gram component terms)
  uint32_t* d_prel = nullptr;        // [n_post] positions of the term's earlier postings IN THIS LEVEL
  uint64_t* d_tpos = nullptr;        // [n_terms + 1] first position of every term in d_pos
  uint16_t* d_pos = nullptr;         // the level's positions, term after term, posting after posting, ascending
  uint64_t n_pos = 0;
};

// ... of an image with SEVERAL indexed fields (ss_bm25_append_level_fields, round 6): kept on the HOST -- the multi-field image is built by
// the host builder (per-field lists + merged lists), so a commit re-assembles the shard's postings term by term and rebuilds the image
struct ss_raw_level_f {
  uint32_t n_docs = 0, n_terms = 0;
  std::vector<uint64_t> off;      // [n_terms + 1], from 0
  std::vector<uint32_t> doc;      // shard-local doc ids; a term's entries sorted by (doc, field)
  std::vector<uint8_t> field;
  std::vector<uint16_t> tf;
  std::vector<uint8_t> doclen;    // [n_fields][n_docs]
};

// Device blocks of an incremental image, recycled from commit to commit.  Every commit builds a slightly larger image beside the old
// one; a fresh multi-GB hipMalloc now and then stalls for hundreds of milliseconds (measured: 80 ms .. 1.2 s, against 35 ms for the
// whole rebuild), so the arrays are handed out with a quarter of headroom and the previous image's blocks serve the next commit.
struct ss_block_pool {
  struct Idle { void* p; size_t cap; uint64_t gen; };
  std::map<void*, size_t> live;   // handed out: capacity
  std::vector<Idle> idle;         // returned by a swap, reusable
  uint64_t gen = 0;               // commits seen
  int alloc(void** out, size_t bytes) {
    size_t best = (size_t)-1;
    for (size_t i = 0; i < idle.size(); i++)
      if (idle[i].cap >= bytes && idle[i].cap / 2 <= bytes && (best == (size_t)-1 || idle[i].cap < idle[best].cap)) best = i;
    if (best != (size_t)-1) {
      *out = idle[best].p;
      live[*out] = idle[best].cap;
      idle.erase(idle.begin() + (long)best);
      return 0;
    }
    const size_t cap = bytes + bytes / 4 + 4096;
    const hipError_t e = hipMalloc(out, cap);
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? -2 : -3;  // SS_ENOMEM / SS_EDEVICE
    live[*out] = cap;
    return 0;
  }
  bool release(void* p) {  // live -> idle; false: not one of ours
    auto it = live.find(p);
    if (it == live.end()) return false;
    idle.push_back(Idle{p, it->second, gen});
    live.erase(it);
    return true;
  }
  void drop(void* p) { live.erase(p); }  // its owner frees it
  void trim(uint64_t keep_gens) {        // idle blocks no commit has picked for a while go back to the driver
    for (size_t i = 0; i < idle.size();)
      if (idle[i].gen + keep_gens <= gen) { (void)hipFree(idle[i].p); idle.erase(idle.begin() + (long)i); } else i++;
  }
  void clear_idle() { for (auto& b : idle) (void)hipFree(b.p); idle.clear(); }
};

struct ss_shard {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  ss_coalescer co_lex, co_vec;
  uint64_t synth_stride = 1, synth_offset = 0;  // ss_synth_set_partition: which slice of the generator stream ss_*_synth build
  // ---- vector image
  float* d_X = nullptr;          // [n_rows_pad][dim_pad]   (f32 image)
  int8_t* d_X8 = nullptr;        // i8 image (quantised embeddings) in MFMA fragment order, vec8_scan.hip v8_index; one of the two is set
  float* d_row_scale = nullptr;  // i8 image: per-record scale (VectorHeader.scale) for dot_i8_quantized, null = raw integer dot
  float* d_row_norm = nullptr;   // i8 image, Euclidean + ScalarQuantizationI8: per-record norm (VectorHeader.norm), euclidean_i8_quantized
  int32_t* d_row_sq = nullptr;   // i8 image, Euclidean without scales: sum of squares of every record (euclidean_i8 as dot products)
  float* d_qaux = nullptr;       // [2][64] per-query side values of the batch in flight (i8 Euclidean: norm | sum of squares)
  int vec_similarity = SS_SIM_DOT;  // ss_vec_set_similarity: Dot / Cosine (dot product) or Euclidean (minus the squared distance)
  uint32_t dim_pad8 = 0;         // row stride of the i8 image in bytes (multiple of 128)
  uint32_t* d_row_doc = nullptr; // optional row -> doc id
  uint16_t* d_row_field = nullptr; // optional row -> indexed field id (VectorHeader.field_id): field_filter
  bool vec_multi_record = false; // several records per doc: TopK::push dedup (vector.rs:441-452) in the refine kernel
  uint64_t n_rows = 0, n_rows_pad = 0;
  uint64_t vec_rows_cap = 0;     // rows the image and its per-row side arrays have room for (0: exactly n_rows_pad / n_rows -- ss_vec_append_rows grows them)
  uint32_t dim = 0, dim_pad = 0;
  // cluster structure (ANN modes, vec_ann.hip); null = none declared
  uint32_t* d_row_cluster = nullptr;    // [n_rows] shard-wide cluster index of each row
  uint32_t* d_cluster_first = nullptr;  // [vec_n_clusters] first row (= medoid record) of each cluster
  uint32_t* d_level_off = nullptr;      // [vec_n_levels + 1] cluster index range of each level
  void* d_medoids = nullptr;            // the medoid records, transposed: [k / 8][cluster][8] f32 or [k / 16][cluster][16] i8
  uint32_t vec_n_clusters = 0, vec_n_levels = 0, vec_max_level_clusters = 0;
  std::vector<uint32_t> h_level_clusters, h_child_count;  // the declared structure (ss_vec_append_rows extends it by a level)
  float* d_ann_score = nullptr;         // [64][vec_n_clusters] medoid similarity per query
  float* d_ann_its = nullptr;           // [64][vec_n_clusters] TopK arrays of the per-level selection (scores)
  uint32_t* d_ann_itc = nullptr;        //                      (cluster ids)
  uint32_t* d_ann_sel = nullptr;        // [65][words] selected-cluster bits per query; row 64 = union over the batch
  uint32_t* d_ann_tiles = nullptr;      // [tiles + 1] ascending tile list of the batch, count in the last slot
  uint32_t* d_ann_ncl = nullptr;        // [64] observed_cluster_count
  uint32_t* d_ann_live = nullptr;       // [2 + n_clusters]: u64 live records of the image, then per cluster (ssi_vec_observed_prepare)
  size_t ann_live_cap = 0;
  hipEvent_t ann_ev = nullptr;          // end of the last search that used the (per-shard) ANN selection state, and its stream:
  hipStream_t ann_ev_stream = nullptr;  // a search with an ss_ann_mode on ANOTHER stream waits for it before it overwrites that state
  bool ann_ev_set = false;
  // vector workspace (one 64-query batch in flight per shard)
  float* d_Qf = nullptr;
  uint32_t* d_vstate = nullptr;  // tau[64] | cnt[64] | kept[64] | flags[64] | total_lo/hi ...
  uint64_t* d_cand = nullptr;    // [64][VS_CAP]
  std::map<hipStream_t, ss_vec_ws> vec_ws;  // the same four buffers for searches on other streams than the shard's own
  void* d_qstage = nullptr;      // host-variant staging of queries (grow-only)
  size_t qstage_cap = 0;
  uint32_t* d_out_doc = nullptr; // host-variant staging of outputs
  float* d_out_score = nullptr;
  uint32_t* d_out_count = nullptr;
  uint64_t* d_out_total = nullptr;
  size_t out_cap = 0, q_cap = 0;
  // Coalesced vector batches (ss_api.hip vec_search_host_lane) run on a stream and staging of their OWN: a batch-64 scan of a 10 M x 768
  // image is 9 ms of device time, during which the shard's other work -- a hybrid caller's lexical half above all -- must neither wait for
  // the shard mutex nor queue behind the scan on the shard's stream (p99 2.3-2.5 x p50 on a deterministic pass).
  // vmu serialises those batches among themselves (lock order: vmu, then mu); mu is held only while a batch is ENQUEUED.
  std::mutex vmu;
  hipStream_t vstream = nullptr;
  hipEvent_t vev = nullptr;          // behind a coalesced pass (blocking-sync: the batch's leader SLEEPS through the 9 ms instead of spinning)
  void* d_vq = nullptr; size_t vq_cap = 0;                 // staged queries (+ scales)
  uint32_t* d_vdoc = nullptr; float* d_vscore = nullptr; uint32_t* d_vcount = nullptr; uint64_t* d_vtotal = nullptr;
  size_t vout_cap = 0, vq_rows_cap = 0;
  // ---- bm25 image
  uint64_t bm_n_docs = 0;
  uint32_t bm_n_terms = 0, bm_n_sub = 0;  // bm_n_terms: VIRTUAL terms (posting lists) = query-able terms x fields
  uint32_t bm_n_fields = 1;               // posting lists per query-able term: the indexed fields (BM25F), + 1 when bm_merged;
                                          // the public API speaks of bm_n_terms / bm_n_fields terms
  // Several indexed fields: BM25F is additive per (term, field), so beside the (term, field) lists the image carries one
  // MERGED list per term -- the last of the term's lists: every doc that holds the term in any field, with the weight
  // sum_f boost_f * w_f / S (S = the scale that keeps it inside the weight code's range; boost[last] = S gives it back
  // through idf).  A query WITHOUT a field filter reads only these: it is a single-field query to every kernel (pruned
  // strategy, 16-bit scan, plain intersections); a query with a field filter reads the (term, field) lists as before.
  bool bm_merged = false;
  float* d_boost = nullptr;               // [bm_n_fields] schema boost per field (add_result.rs:1253); merged list: S
  std::vector<uint64_t> h_df_real;        // multi-field: docs containing the term in any field (the df idf needs)
  std::vector<float> h_boost;             // host copy of d_boost (several indexed fields)
  std::map<hipStream_t, ss_bm_ws> bm_ws;   // per-stream search workspaces (guarded by mu)
  uint64_t bm_n_post = 0;
  float bm_avgdl = 0.f;
  uint64_t bm_n_post_pad = 0;     // dwords in d_post (segments padded to 16 bytes)
  uint32_t* d_post = nullptr;     // packed postings, ordered (term, sub-block, doc); every (term, sub-block) segment
                                  // starts 16-byte aligned and is zero-padded (NULL postings) to a multiple of 16 bytes
  uint64_t* d_term_base = nullptr; // [n_terms+1] first 16-byte unit of each term
  uint32_t* d_sub_off = nullptr;   // [n_terms][n_sub+1] segment boundaries in 16-byte units relative to the term base
  float* d_comp = nullptr;         // bm25_component_cache[256] (kept for inspection; the kernels read weights from the postings)
  std::vector<uint64_t> h_df;      // posting_count per term (the df the host needs for idf)
  // probe index: membership (64-doc bit records) and rank (index of each group's first posting) of a doc in a term's
  // segment without reading the segment; the bit records are also the bitmaps exact union counts are popcounted from
  uint8_t* d_facets = nullptr;     // facet.bin: one record of facet_record_size bytes per doc (ss_facet_upload)
  uint64_t facet_docs = 0;
  uint32_t facet_record_size = 0;
  uint32_t* d_filter_bits = nullptr;  // exclusion bitmap of the facet-filtered search in flight (facet.hip), grow-only
  uint64_t filter_words_cap = 0;
  void* d_facet_ws = nullptr;      // ss_bm25_facet_count workspace (query, match words, histogram, bounds), grow-only
  size_t facet_ws_cap = 0;
  uint32_t* d_deleted = nullptr;   // tombstone bitmap by shard-local doc id (delete.bin / delete_hashset), null = none
  uint64_t deleted_words = 0, n_deleted = 0;
  uint2* d_probe = nullptr;        // [probe_rows + 1][n_sub][BM_SUB / 64] 64 doc bits; the last row is all zero (absent terms)
  uint32_t* d_probe_z = nullptr;   // same shape: index inside the term of the group's first posting (read on hits only)
  uint32_t* d_probe_row = nullptr; // [n_terms + 1] row of each (virtual) term, BM_NO_PROBE_ROW for the lists left without one:
                                   // rows go to the longest lists until the budget is spent (1.9 MB per list at 10 M docs is
                                   // the right price for the lists that cost query time, not for a vocabulary's long tail)
  std::vector<uint32_t> h_probe_row;
  uint32_t bm_probe_rows = 0;
  // When the rows are rationed, the last probe_pool_rows of them are a POOL: rows built on demand for the row-less lists a
  // batch touches (ssi_bm25_ensure_probe_rows, ss_api.hip), least recently used first.  pool_list[i] = the list that holds
  // pool row i (BM_NO_PROBE_ROW = free), pool_tick[i] = the last batch that needed it.
  uint32_t probe_pool_begin = 0, probe_pool_rows = 0;
  std::vector<uint32_t> pool_list;
  std::vector<uint64_t> pool_tick;
  uint64_t pool_clock = 0;
  uint32_t* d_pool_stage = nullptr;  // (list, row) pairs of the build in flight, grow-only
  size_t pool_stage_cap = 0;
  uint64_t probe_budget = 0;       // ss_bm25_set_probe_budget: bytes, 0 = half of the free device memory
  int bm_strategy = SS_BM25_AUTO;  // ss_bm25_set_strategy
  float* d_umax = nullptr;         // [n_terms + 1] largest weight tf*(K+1)/(tf+comp[len]) of the term (max_list_score / idf)
  // [n_terms + 1][4] the 10th / 100th / 1000th largest weight of every list (0 = the list is shorter), built from the finished image
  // the first time a search wants it (ssi_bm25_ensure_kth, bm25.hip): idf * that weight is a score k docs of a UNION reach for sure,
  // so the shared threshold of such a query starts there instead of at zero
  float* d_kthw = nullptr;
  std::vector<float> h_kthw;
  bool bm_partmax = false;         // the pruned kernel bounds every partition by its own block maxima (set at image build when the
                                   // maxima vary over the doc ids; SS_BM25_SUBMAX=1 / 0 forces it on / off)
  // positions of every posting (ss_bm25_upload_positions): only phrase queries read them
  uint16_t* d_pos = nullptr;       // the positions, posting after posting in image order
  uint32_t* d_pos32 = nullptr;     // several indexed fields (ss_bm25_upload_fields_positions): the positions of the MERGED lists' postings,
                                   // field << BM_POS_FIELD_SHIFT | position inside the field, ascending (fields ascending inside a posting)
  uint32_t* d_pos_off = nullptr;   // [bm_n_post_pad + 1] first position of the posting at that (padded) image index, relative to its term's
                                   // base; the posting's count is the distance to the next entry (padding slots repeat their successor)
  uint64_t* d_pos_base = nullptr;  // [n_terms + 1] first position of every term in d_pos
  float* d_submax = nullptr;       // [n_terms + 1][n_sub] largest weight of every (term, 4096-doc sub-block) segment, 0 = empty: the
                                   // reference's per-block max_block_score / idf (get_max_score, index.rs:2938-3200) at this image's
                                   // block size; the last row (absent terms) is all zero
  // ---- sparse tier (bm25_sparse.hip): rare terms as plain sorted lists, no directory row, no probe row.  Sparse list i is public
  // term bm_n_terms / bm_n_fields + i; a posting = weight code << 32 | doc (several indexed fields: the code of the term's MERGED weight)
  uint8_t* d_doclen = nullptr;       // the length bytes of the docs ([indexed fields][n_docs]; several fields: images with merged lists only),
                                     // kept for ss_bm25_append_sparse[_fields]
  uint64_t* d_sp_base = nullptr;     // [sp_n + 1] first posting of every sparse list
  uint64_t* d_sp_post = nullptr;     // the postings, list after list, ascending docs inside a list
  uint32_t sp_n = 0;
  void* d_sp_pos = nullptr;          // positions of the sparse postings (phrase queries): u16 (one indexed field) or u32 field << 20 | position
  uint64_t* d_sp_pos_end = nullptr;  // [sparse postings] END of every posting's positions in d_sp_pos (its start = the posting before's end)
  uint64_t sp_pos_n = 0;
  uint32_t sp_pos_elem = 0;          // 2 | 4
  std::vector<uint64_t> h_sp_base;   // host copy of d_sp_base (posting counts = the df the host needs for idf)
  void* d_tier_ws = nullptr;         // workspace of a tiered search (sub-queries, row maps, sparse lists, merged answers), grow-only
  size_t tier_ws_cap = 0;
  uint64_t bm_batch_postings = 0;    // mean postings per query of the host-pointer batch about to run (0 = unknown: a device-resident batch)
  uint32_t del_per_query = 0;        // d_deleted holds one bitmap of deleted_words words PER QUERY of the batch in flight (ss_bm25_search_sorted)
  void* d_sort_ws = nullptr;         // workspace of ss_bm25_search_sorted, grow-only
  size_t sort_ws_cap = 0;
  void* d_tier_hold = nullptr;       // answers of the queries a tiered batch runs one by one (unions with a sparse NOT term), grow-only
  size_t tier_hold_cap = 0;
  uint32_t* d_excl_bits = nullptr;   // per-query exclusion bitmap of such a query: tombstones | docs of its sparse NOT lists
  size_t excl_words_cap = 0;
  // pages deeper than SS_MAX_K results (ss_api.hip "deep pages"): the exclusion bitmap(s) of the passes -- whatever excluded docs before
  // (tombstones, a facet filter) | the docs the earlier passes returned; one row of peel_words words per query of a vector group
  uint32_t* d_peel_bits = nullptr;
  size_t peel_words_cap = 0;         // dwords allocated
  void* d_gate_ws = nullptr;         // match sets of a gated union's single-term queries (ss_api.hip bm25_search_gated_scan_rule), grow-only
  size_t gate_ws_cap = 0;
  uint32_t vec_del_stride = 0;       // != 0: d_deleted holds one bitmap of that many words per query of the vector batch in flight
  // threshold seeds from OUTSIDE a batch's own lists (bm25_search_tiered: the k-th FULL score the sparse kernel found for a union whose dense
  // terms this batch carries): [ext_seed_n] floats, row i for query i of the NEXT ssi_bm25_search call of exactly ext_seed_n queries
  const float* d_ext_seed = nullptr;
  uint32_t ext_seed_n = 0;
  void* d_route_ws = nullptr;        // a batch that holds shapes of several kernel families (ss_api.hip bm25_route_shapes): the sub-batches'
  size_t route_ws_cap = 0;           // queries + row maps, and the answers until every sub-batch has run; grow-only
  uint64_t gallop_batches = 0;       // sub-batches the generic kernels (bm25_gallop.hip) answered
  // incremental image (ss_bm25_append_level)
  ss_block_pool blocks;              // the image arrays of incremental images come from here
  ss_block_pool* pool = nullptr;     // set on the scratch shard a rebuild fills: its image arrays are taken from the owner's pool
  std::vector<ss_raw_level> raw;
  std::vector<ss_raw_level_f> raw_f;  // several indexed fields: the committed levels' entries, on the host (ss_bm25_append_level_fields)
  uint32_t raw_f_fields = 0;
  std::vector<uint8_t> h_doclen;     // the length bytes of every doc committed so far
  double raw_last_append_ms = 0.0, raw_last_rebuild_ms = 0.0;
  // bm25 workspace
  void* d_bq = nullptr; size_t bq_cap = 0;       // staged queries
  // ... and their way there: PINNED host staging the asynchronous copy reads from (the callers' own arrays -- pageable, often a local
  // of a frame that returns before the stream is synchronised -- are never handed to hipMemcpyAsync), with the event behind the last
  // copy out of it: the next batch waits for it before it overwrites the staging
  void* h_bq = nullptr; size_t h_bq_cap = 0; hipEvent_t bq_ev = nullptr; bool bq_ev_set = false;
  uint64_t* d_ptotal = nullptr;                   // per (query, partition) match counts
  // one-launch path of small host-pointer batches (bm25_small.hip): device workspace (zero between launches), pinned answer
  // staging + completion flags (slot 0: direct calls, 1 / 2: the coalescer's lanes), launch counter
  void* d_small_ws = nullptr;
  char* h_small = nullptr;
  uint32_t small_seq = 0;
  // The one-launch kernels of the coalescer's two LANES run on a stream (and a workspace) of their own each, so that two batches of
  // ~ 20 - 30 concurrent callers overlap on the device instead of queueing on the shard's one stream (a launch of that size is a chain of
  // dependent round trips, not a full chip).  Ordering: a lane launch waits for the tail of s->stream as it stood when it was enqueued
  // (ev_main: uploads, commits, tombstones, probe rows built on demand -- everything that writes what the kernel reads goes through
  // s->stream); everything ELSE that takes the shard mutex drains the lane streams first (ShardLock in ss_api.hip), so a writer never
  // meets a lane kernel in flight.  Lane kernels only read the image; their answers go to the lanes' pinned staging.
  hipStream_t lstream[2] = {nullptr, nullptr};
  hipStream_t lane_last[2] = {nullptr, nullptr};  // the stream a lane's latest launch went to (big batches share lane 0's)
  void* d_small_ws_l[2] = {nullptr, nullptr};
  hipEvent_t ev_main = nullptr;
  bool lanes_inflight = false;  // a lane launch since the lane streams were last drained (under mu)
  bool main_dirty = true;       // s->stream may hold work the lane streams have not been ordered behind yet (set by every ShardLock section;
                                // a cross-stream wait costs a launch ~ 15 us on this runtime, so it is only enqueued when there is something to wait for)
  // the STAGED pipeline's answers of a direct host-pointer call take the same road home (ss_api.hip bm25_answers_home): one kernel writes
  // them into this pinned block and raises flag slot 0 behind them -- instead of four copies into the caller's pageable arrays and a
  // stream synchronisation
  char* h_ans = nullptr; size_t h_ans_cap = 0;
  uint32_t* d_ans_done = nullptr;  // blocks of that kernel that have finished (zero between launches)
  uint64_t small_launches = 0;
  ss_prof prof;
};
int ssi_bm25_ensure_kth(ss_shard* s, hipStream_t st);   // bm25.hip
void ssi_bm25_drop_kth(ss_shard* s);
inline uint32_t bm_kth_sel(uint32_t k) { return k == 0u ? 3u : k <= 10u ? 0u : k <= 100u ? 1u : k <= 1000u ? 2u : 3u; }  // column of d_kthw (3 = none)
// bm25_small.hip
size_t ssi_bm25_small_ws_bytes();
bool ssi_bm25_small_serves(const ss_shard* s, uint32_t nq, uint32_t k, uint32_t np_max, uint32_t nn_max);
// what a batch of the one-launch path holds: counts wanted; some query is an intersection of several DENSE terms / a union of two or more
// dense lists (the counting workgroups); NOT terms; a term of the sparse tier; a phrase (naming a sparse term); the most DENSE scored terms
struct ss_small_shape { bool want_counts, has_and, has_or, any_not, any_sparse, any_phrase; uint32_t np_max; };
int ssi_bm25_small_launch(ss_shard* s, void* ws, uint32_t nq, const ss_bm25_query* hq, uint32_t k, const ss_small_shape& sh, uint32_t* out_doc,
                          float* out_score, uint32_t* out_count, uint64_t* out_total, uint32_t* flag, uint32_t seq, hipStream_t st);

// Opt-in to more than 64 KB of dynamic LDS is a per-device function attribute: set it once per (kernel, device) --
// one process may hold shards on several GPUs (C++ host Index), and concurrent searches may race to be first.
#define SS_SET_MAX_LDS(kernel_fn, bytes)                                                            \
  do {                                                                                              \
    static std::atomic<uint32_t> _done_mask{0};                                                     \
    int _dev = 0;                                                                                   \
    SS_HIP(hipGetDevice(&_dev));                                                                    \
    const uint32_t _bit = 1u << (_dev & 31);                                                        \
    if (!(_done_mask.load(std::memory_order_acquire) & _bit)) {                                     \
      SS_HIP(hipFuncSetAttribute((const void*)(kernel_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
      _done_mask.fetch_or(_bit, std::memory_order_release);                                         \
    }                                                                                               \
  } while (0)

// error helper
#define SS_HIP(x)                                   \
  do {                                              \
    hipError_t _e = (x);                            \
    if (_e != hipSuccess) return (_e == hipErrorOutOfMemory) ? SS_ENOMEM : SS_EDEVICE; \
  } while (0)

// ---- implemented in vec_scan.hip
// d_queries: f32 [nq][dim] for the f32 image, i8 [nq][dim] for the i8 image (d_qscale: per-query scale or null)
// ann_mode: null = AnnMode::All; d_out_clusters: null or [nq] observed_cluster_count
// d_qnorm: i8 Euclidean with quantisation scales: per-query norm (QuantizedVector.norm), else null
int ssi_vec_search(ss_shard* s, uint32_t nq, const void* d_queries, const float* d_qscale, uint32_t k, float thr,
                   uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, hipStream_t st,
                   bool safe_mode, const ss_ann_mode* ann_mode = nullptr, uint32_t* d_out_clusters = nullptr,
                   const float* d_qnorm = nullptr);
int ssi_vec_augment(ss_shard* s, hipStream_t st, uint64_t r0 = 0);   // f32 Euclidean image: columns dim, dim + 1 = |x|^2, 1 (rows r0 .. n_rows)
int ssi_vec8_row_sq(ss_shard* s, hipStream_t st, uint64_t r0 = 0);   // i8 Euclidean image without scales: d_row_sq
int ssi_vec8_permute_range(ss_shard* s, const int8_t* d_rows_row_major, uint64_t r0, uint64_t n, hipStream_t st);  // appended rows -> fragment order
struct VAnn;
int ssi_vec8_qprep(ss_shard* s, const int8_t* d_queries, uint32_t nb, hipStream_t st);
int ssi_vec8_launch_scan(ss_shard* s, uint32_t tile0, uint32_t ntiles, const float* d_qscale, const VAnn* ann, hipStream_t st);
int ssi_vec8_qaux(ss_shard* s, const int8_t* d_queries, uint32_t nb, const float* d_qnorm, hipStream_t st);
int ssi_bm25_match_bits(ss_shard* s, const ss_bm25_query* d_q, unsigned long long* d_bits, unsigned long long* d_total, hipStream_t st, uint32_t nq = 1);
// facet histogram over a match bitmap: d_counts [n_buckets + 1] (last = values outside the buckets)
int ssi_facet_kth(ss_shard* s, const unsigned long long* d_bits, uint64_t n_docs, uint64_t n_matches, uint32_t offset, uint32_t type,
                  bool descending, uint64_t k, unsigned long long* d_hist, uint64_t* value_bits, uint64_t* n_better, uint64_t* n_equal,
                  hipStream_t st);
int ssi_facet_values(ss_shard* s, const uint32_t* d_docs, uint32_t n, uint32_t offset, uint32_t type, unsigned long long* d_out, hipStream_t st);
int ssi_facet_count(ss_shard* s, const unsigned long long* d_bits, uint64_t n_docs, uint32_t offset, uint32_t type, uint32_t n_buckets,
                    const uint64_t* d_bounds, unsigned long long* d_counts, hipStream_t st);
// ---- implemented in facet.hip: exclusion bitmap (failed facet filters | tombstones) into s->d_filter_bits
int ssi_facet_build(ss_shard* s, uint32_t n_filters, const ss_facet_filter* filters, hipStream_t st);
int ssi_sort_select(ss_shard* s, uint32_t nq, unsigned long long* d_E, unsigned long long* d_B, unsigned long long* d_ex_b, unsigned long long* d_ex_e,
                    const unsigned long long* d_total, unsigned long long* d_hist, void* d_state, uint32_t n_sorts, const ss_result_sort* sorts,
                    uint32_t k, hipStream_t st);
int ssi_sort_compose(ss_shard* s, uint32_t nq, const uint32_t* a_doc, const float* a_score, const uint32_t* a_cnt, const uint32_t* c_doc,
                     const float* c_score, const uint32_t* c_cnt, const unsigned long long* d_total, uint32_t n_sorts, const ss_result_sort* sorts,
                     uint32_t k, uint32_t* out_doc, float* out_score, uint32_t* out_count, unsigned long long* out_total, hipStream_t st);
// ---- implemented in vec_ann.hip
// observed_vector_count (SS_ANN_REPORT_OBSERVED): live records per cluster once per call, then the triples of every batch
int ssi_vec_observed_prepare(ss_shard* s, unsigned long long field_mask, hipStream_t st);
int ssi_vec_observed_report(ss_shard* s, uint32_t nb, bool clusters_selected, uint32_t* d_out3, hipStream_t st);
// after the batch's queries are in s->d_Qf (qprep): medoid scores, per-query selection, tile list -> *out
int ssi_vec_ann_prepare(ss_shard* s, uint32_t nb, const float* d_qscale, const ss_ann_mode* mode, VAnn* out,
                        uint32_t* d_out_clusters, hipStream_t st, const float* d_qnorm = nullptr);
int ssi_vec_set_clusters(ss_shard* s, uint32_t n_levels, const uint32_t* level_clusters, const uint32_t* child_count);
void ssi_vec_free_clusters(ss_shard* s);
int ssi_vec8_quantize(ss_shard* s, hipStream_t st);
int ssi_vec8_permute(ss_shard* s, const int8_t* d_rows_row_major, hipStream_t st);
int ssi_vec8_gather_rows(ss_shard* s, uint64_t r0, uint64_t n, int8_t* d_out, hipStream_t st);
int ssi_vec_alloc_ws(ss_shard* s);
// ---- implemented in bm25.hip
// ss_bm25_query::op = operator (bits 0-7) | number of NOT terms (bits 8-15); the NOT terms follow the n_terms query
// terms in term[] (their idf entries are ignored).  A doc found in a NOT list neither counts nor ranks
// (add_result.rs:3440-3497).  On the device a NOT term is a term with idf = BM_NOT_IDF: its docs' scores become hugely
// negative, and only positive scores are candidates or counted.
#define BM_NOT_IDF (-1.0e30f)
// What the kernels read: the public query expanded over the shard's indexed fields.  A posting list of the image is a
// VIRTUAL term = (term, field): v = term * n_fields + field (one field: v = term).  A query term becomes one virtual
// term per field it occurs in, each scored with idf * boost(field) like get_bm25f_multiterm_multifield
// (add_result.rs:1171-1426: bm25f += boost * idf * (tf (K+1) / (tf + comp[len_field]))); an intersection needs one
// virtual term of every GROUP (= query term): a doc's match byte collects and_val of each posting and is compared with
// and_target -- bits of a mask (<= 8 query terms), or, for 9-10 single-field terms, 0xFF = "count one more".
constexpr int BM_MAX_VTERMS = 32;
struct bm_vquery {
  uint32_t n_terms;     // scored virtual terms
  uint32_t op;          // SS_OP_* | number of virtual NOT terms << 8; a query of ONE term is always a union (of its fields)
  uint32_t n_groups;    // query terms
  uint32_t and_target;  // 0 unless the query is an intersection of > 1 terms
  uint32_t term[BM_MAX_VTERMS];
  float idf[BM_MAX_VTERMS];
  uint8_t and_val[BM_MAX_VTERMS];
  uint8_t group[BM_MAX_VTERMS];  // query term of each virtual term (the virtual terms of a group are contiguous)
  uint32_t phrase_len;           // SS_OP_PHRASE: words of the phrase (virtual term = the term's only / merged list), else 0
  uint8_t phrase_seq[SS_MAX_PHRASE];
  uint32_t phrase_fields;        // several indexed fields: the fields the phrase may stand in (bit f; the query's field filter or all ones)
};
__host__ __device__ inline uint32_t bm_q_op(uint32_t op) { return op & 0xFFu; }
__host__ __device__ inline uint32_t bm_q_nnot(uint32_t op) { return (op >> 8) & 0xFFu; }
__host__ __device__ inline uint32_t bm_q_field_filter(uint32_t op) { return (op >> 16) & 0x7FFFu; }  // SS_OP_FIELD_FILTER: bit f = field f
__host__ __device__ inline bool bm_q_all_frequent(uint32_t op) { return (op >> 31) != 0u; }         // SS_OP_ALL_TERMS_FREQUENT
// bm_vquery::and_target bit 8 / a term's and_val bit 8 inside the scan kernels: the intersection runs under the reference's
// all_terms_frequent shortcut -- a posting with tf < 10 sets bit 7 of its doc's match byte, which keeps the doc counted
// but out of the ranking (add_result.rs:2091-2104, 3541-3556)
constexpr uint32_t BM_AND_FREQ = 0x100u;
// positions of a multi-field image: field id above the position inside the field (positions < 65 536, phrases <= SS_MAX_PHRASE words:
// start + word index never reaches bit 20)
constexpr uint32_t BM_POS_FIELD_SHIFT = 20u;
// a sparse posting's upper word: the 19-bit weight code, above it (several indexed fields) bit f = the doc holds the term in field f
constexpr uint32_t BM_SP_FIELD_SHIFT = 19u, BM_SP_CODE_MASK = 0x7FFFFu;
// A UNION under a field filter (add_result.rs:3124-3136 applied inside union_docid_3's sub-queries, union.rs:1330-1425: a doc ends
// with the sum over its terms that occur in a LISTED field, all fields of those terms counted; a doc none of whose terms passes is
// no result).  BM_AND_GATED in and_target / a term's av: the lists of a term come listed fields first (their postings add and set
// the term's bit in the doc's match byte), then the unlisted fields, whose list is marked 0x80 in its and_val and adds only where
// the term's bit is set; a doc is a result iff its score is positive.  BM_AND_TOUCH (more than two terms): every posting also sets
// bit 7 of the match byte and the exact count is that of the UNFILTERED union -- union_scan counts a doc before the filter sees it
// (union.rs:552-553); two terms count |pass(X) u pass(Y)| (union_docid_2).  <= 7 terms (bits 0..6).
constexpr uint32_t BM_AND_GATED = 0x200u, BM_AND_TOUCH = 0x400u;
int ssi_bm25_search(ss_shard* s, uint32_t nq, const ss_bm25_query* d_q, uint32_t k, uint32_t rt, uint32_t* d_out_doc,
                    float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, bool has_and, bool has_or,
                    uint32_t nt_max, uint32_t np_max, bool all_probed, hipStream_t st, bool any_frequent = false, bool phrase = false,
                    bool any_field_filter = true, bool uniform_terms = false, bool any_gated = false, uint32_t nn_max = 0xFFFFFFFFu);
// indexed fields of the image (bm_n_fields counts the merged list as well)
inline uint32_t bm_real_fields(const ss_shard* s) { return s->bm_n_fields - (s->bm_merged ? 1u : 0u); }
// ---- implemented in synth.hip
int ssi_vec_synth(ss_shard* s, uint64_t seed, hipStream_t st);
int ssi_bm25_synth(ss_shard* s, uint64_t seed, const uint32_t* d_thresh, const uint8_t* d_lentab, hipStream_t st);
// positions_sum = 0: avgdl from the decoded length bytes; else the reference's stored positions_sum_normalized (index.rs:3480)
// merged_boost (s->bm_merged images): the real fields' boosts; the last list of every term is then built here as the merged
// list -- its offs / docs give the docs, its weights come from the term's field lists
int ssi_bm25_build_from_host(ss_shard* s, const uint8_t* doclen, const uint64_t* offs, const uint32_t* docs,
                             const uint16_t* tfs, uint64_t positions_sum, const float* merged_boost = nullptr, float merged_scale = 1.0f);
int ssi_bm25_upload(ss_shard* s, uint64_t n_docs, const uint8_t* doclen, uint32_t n_terms, const uint64_t* offs,
                    const uint32_t* docs, const uint16_t* tfs, uint64_t positions_sum);
int ssi_bm25_attach_positions(ss_shard* s, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs, const uint16_t* positions,
                              uint64_t n_positions, const uint16_t* npos = nullptr);
int ssi_bm25_upload_positions_fields(ss_shard* s, uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                     const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions, const uint16_t* npos = nullptr);
int ssi_bm25_upload_positions(ss_shard* s, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs, const uint16_t* positions,
                              uint64_t n_positions, const uint16_t* npos = nullptr);
int ssi_bm25_upload_fields_positions(ss_shard* s, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                                     uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                     const uint16_t* tfs, uint64_t positions_sum, const uint16_t* positions, uint64_t n_positions, const uint16_t* npos = nullptr);
int ssi_bm25_upload_fields(ss_shard* s, uint64_t n_docs, uint32_t n_fields, const uint8_t* doclen, const float* boost,
                           uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                           const uint16_t* tfs, uint64_t positions_sum);
void ssi_prof_begin(ss_shard* s, int kernel, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1);
void ssi_prof_end(ss_shard* s, int kernel, hipStream_t st, hipEvent_t e0, hipEvent_t e1);

// ---- incremental image (synth.hip): `img` = a scratch ss_shard that receives the new image (its bm_* / d_* image fields); the raw
// levels are read from `s`
int ssi_bm25_rebuild_from_raw(const ss_shard* s, const std::vector<ss_raw_level>& levels, uint32_t n_terms, const uint8_t* doclen, uint64_t n_doclen,
                              ss_shard* img, hipStream_t st, bool one_shot = false);
// ---- sparse tier (synth.hip: append; bm25_sparse.hip: kernels)
int ssi_bm25_fill_fixed_probe_rows(ss_shard* s, hipStream_t st);  // the fixed probe rows from the uploaded postings (ss_api.hip)
int ssi_bm25_append_sparse(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                           const uint16_t* positions = nullptr, uint64_t n_positions = 0, const uint16_t* npos = nullptr);
int ssi_bm25_append_sparse_fields(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs,
                                  const uint16_t* positions = nullptr, uint64_t n_positions = 0, const uint16_t* npos = nullptr);
// keys per lane of the sparse tier's top-k lists (rows of 64 * KPL keys), as the dense kernels choose theirs
inline int ssi_bm25_sparse_kpl(uint32_t kk) { return kk <= 64 ? 1 : kk <= 128 ? 2 : kk <= 256 ? 4 : 16; }
int ssi_bm25_launch_sparse(const ss_shard* s, const ss_bm25_query* d_q, uint32_t nq, uint32_t k, unsigned long long* d_keys,
                           unsigned long long* d_extra, hipStream_t st);
int ssi_bm25_launch_sparse_phrase(const ss_shard* s, const ss_bm25_query* d_q, uint32_t nq, uint32_t k, unsigned long long* d_keys,
                                  unsigned long long* d_extra, hipStream_t st);
int ssi_bm25_launch_sparse_seeds(uint32_t nq, uint32_t k, const uint32_t* d_dense_row, const uint32_t* d_sparse_row, const ss_bm25_query* d_spq,
                                 const unsigned long long* d_keys, float* d_seed, hipStream_t st);
int ssi_bm25_sparse_excl_bits(const ss_shard* s, const uint32_t* d_base_bits, uint32_t base_words, const uint32_t* lists, uint32_t n_lists,
                              uint32_t* d_out, uint32_t words, hipStream_t st);
int ssi_bm25_sparse_mark_unlisted(const ss_shard* s, const uint32_t* lists, uint32_t n_lists, uint32_t filter, uint32_t* d_out, uint32_t words, hipStream_t st);
int ssi_bm25_launch_tier_merge(uint32_t nq, uint32_t k, const uint32_t* d_dense_row, const uint32_t* d_sparse_row, const uint32_t* d_doc,
                               const float* d_score, const uint32_t* d_count, const unsigned long long* d_total, const unsigned long long* d_keys,
                               const unsigned long long* d_extra, uint32_t* o_doc, float* o_score, uint32_t* o_count, unsigned long long* o_total,
                               hipStream_t st);
// generic intersections / phrases by galloping lookups (bm25_gallop.hip): any number of terms, tiers, fields, k -- the shapes the
// specialised kernels leave out.  d_q: PUBLIC queries on the device, validated by the host (ss_api.hip bm25_route_shapes)
int ssi_bm25_gallop_search(ss_shard* s, uint32_t nq, const ss_bm25_query* d_q, bool phrase, uint64_t longest_driver, uint32_t k, uint32_t rt,
                           uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, hipStream_t st);
struct ss_comm;
// one per-shard result list of a batch on the device: [nq][k] doc ids / scores, [nq] counts
struct ss_dev_list { const uint32_t* doc; const float* score; const uint32_t* count; uint32_t k; };
// The exchange of the ss_*_search_sharded entry points (comm.hip): one all-gather of the rank's lists + totals + status, merge,
// (hybrid: RRF over the merged lists), answers to the host of every rank.  local_rc != 0: this rank's search failed -- it still
// takes part (empty lists) so that no peer blocks; every rank then returns an error (the failing one its own, the others SS_EPEER).
int ssi_comm_exchange(ss_comm* c, uint32_t nq, int n_lists, const ss_dev_list* L, const uint64_t* d_tot_a, const uint64_t* d_tot_b,
                      int local_rc, bool hybrid, uint32_t offset, uint32_t length, uint64_t* out_doc, float* out_score,
                      uint8_t* out_source, uint32_t* out_count, uint64_t* out_total, hipStream_t st);
int ssi_topk_merge_launch(int device, uint32_t nq, uint32_t S, uint32_t k, const uint32_t* d_doc, const float* d_score, const uint32_t* d_count,
                          size_t stride_ds, size_t stride_c, uint32_t out_len, uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count,
                          hipStream_t st);
