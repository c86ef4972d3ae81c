// Internal declarations shared by the HIP translation units.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
constexpr int VS_FIRST_TILES = 16;          // first chunk: 2048 rows, everything is a candidate

// ---------------------------------------------------------------- BM25 image geometry
constexpr int BM_SUB_LOG2 = 12;             // docs per sub-block = 4096 = one wave's 16 KB LDS accumulator tile
                                            // (2048 with 16 waves/CU measured 17% slower: per-item overhead dominates)
constexpr int BM_SUB = 1 << BM_SUB_LOG2;
constexpr int BM_WAVES_OR = 8 << (12 - BM_SUB_LOG2);   // waves per workgroup (one workgroup per CU), union-only kernels
constexpr int BM_WAVES_AND = 6 << (12 - BM_SUB_LOG2);  // kernels that also carry match counters
constexpr uint32_t BM_TF_MAX = 2046;        // 11-bit tf field, 2047 reserved
// packed posting: bits 0..12 doc-in-sub-block (13 bits), 13..20 SmallFloat length byte, 21..31 tf
__host__ __device__ inline uint32_t bm_pack(uint32_t doc_in_sub, uint32_t len_byte, uint32_t tf) {
  return (doc_in_sub & 0x1FFFu) | ((len_byte & 0xFFu) << 13) | (tf << 21);
}

struct ss_prof {
  bool on = false;
  uint64_t launches[2] = {0, 0};
  double ms[2] = {0.0, 0.0};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[2];
};

struct ss_shard {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  // ---- vector image
  float* d_X = nullptr;          // [n_rows_pad][dim_pad]
  uint32_t* d_row_doc = nullptr; // optional row -> doc id
  uint64_t n_rows = 0, n_rows_pad = 0;
  uint32_t dim = 0, dim_pad = 0;
  // vector workspace (one 64-query batch in flight per shard)
  float* d_Qf = nullptr;
  uint32_t* d_vstate = nullptr;  // tau[64] | cnt[64] | kept[64] | flags[64] | total_lo/hi ...
  uint64_t* d_cand = nullptr;    // [64][VS_CAP]
  float* d_qstage = nullptr;     // host-variant staging of queries
  uint32_t* d_out_doc = nullptr; // host-variant staging of outputs
  float* d_out_score = nullptr;
  uint32_t* d_out_count = nullptr;
  uint64_t* d_out_total = nullptr;
  size_t out_cap = 0, q_cap = 0;
  // ---- bm25 image
  uint64_t bm_n_docs = 0;
  uint32_t bm_n_terms = 0, bm_n_sub = 0;
  uint64_t bm_n_post = 0;
  float bm_avgdl = 0.f;
  uint32_t* d_post = nullptr;     // packed postings, ordered (term, doc)
  uint64_t* d_term_base = nullptr; // [n_terms+1] first posting of each term
  uint32_t* d_sub_off = nullptr;   // [n_terms][n_sub+1] offsets relative to term base
  float* d_comp = nullptr;         // bm25_component_cache[256]
  std::vector<uint64_t> h_term_base;
  // bm25 workspace
  void* d_bq = nullptr; size_t bq_cap = 0;       // staged queries
  uint64_t* d_part = nullptr; size_t part_cap = 0; // partition-local top-k keys
  uint64_t* d_ptotal = nullptr;                   // per (query, partition) match counts
  ss_prof prof;
};

// error helper
#define SS_HIP(x)                                   \
  do {                                              \
    hipError_t _e = (x);                            \
    if (_e != hipSuccess) return (_e == hipErrorOutOfMemory) ? SS_ENOMEM : SS_EDEVICE; \
  } while (0)

// ---- implemented in vec_scan.hip
int ssi_vec_search(ss_shard* s, uint32_t nq, const float* d_queries, uint32_t k, float thr, uint32_t* d_out_doc,
                   float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, hipStream_t st, bool safe_mode);
int ssi_vec_alloc_ws(ss_shard* s);
// ---- implemented in bm25.hip
int ssi_bm25_search(ss_shard* s, uint32_t nq, const ss_bm25_query* d_q, uint32_t k, uint32_t rt, uint32_t* d_out_doc,
                    float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, bool has_and, uint32_t nt_max, hipStream_t st);
// ---- implemented in synth.hip
int ssi_vec_synth(ss_shard* s, uint64_t seed, hipStream_t st);
int ssi_bm25_synth(ss_shard* s, uint64_t seed, const uint32_t* d_thresh, const uint8_t* d_lentab, hipStream_t st);
int ssi_bm25_build_from_host(ss_shard* s, const uint8_t* doclen, const uint64_t* offs, const uint32_t* docs,
                             const uint16_t* tfs);
void ssi_prof_begin(ss_shard* s, int kernel, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1);
void ssi_prof_end(ss_shard* s, int kernel, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
