// Small host-pointer batches (<= 64 queries) of the pruned strategy as ONE launch: the reference's real call shape is one query per
// call (search.rs:1637-1743, one task per shard and query), where the staged pipeline of bm25.hip -- copy the queries in, expand them,
// probe, merge the partition lists, four copies out, synchronise -- is all latency and no throughput (139 us for a
// single query of which the kernels themselves take about 50).  Here:
//
//   * the queries travel IN THE KERNEL ARGUMENTS (60 bytes each: <= 4 scored terms, <= 4 NOT terms; the host has validated and
//     "expanded" them -- one indexed field: there is nothing to expand) -- no copy in, no expansion launch;
//   * a workgroup = 8 partitions of ONE query (pb_wave, the body of bm25_probe_kernel); its 8 lists are merged in LDS;
//   * the LAST workgroup of a query to arrive (one atomic per workgroup) merges the per-workgroup lists and writes the answers
//     straight into the caller's PINNED host buffers; the last query to finish raises a flag word there, which the host thread
//     polls -- no merge launch, no copies out, no completion interrupt;
//   * exact union counts (TopkCount) are further workgroups of the SAME launch popcounting the probe index's bit records
//     (the job of bm25_union_count_kernel), arriving on the same per-query counter;
//   * the per-query state (shared threshold, match count, arrival counter) is left ZERO by whoever consumed it: the next launch
//     starts clean without a memset.
//
// Answers are bit-identical to the staged pipeline's: same body, same fma chain, same total order of the keys.
#include <cstring>

#include "bm25_probe_body.h"

constexpr uint32_t SM_MAX_Q = 64;   // queries per launch (their 60-byte forms must fit the 4 KB of kernel arguments)
constexpr uint32_t SM_MAX_PB = 64;  // workgroups (of 8 partitions) per query: the final tournament plays one list per lane
constexpr uint32_t SM_MAX_CB = 20;  // counting workgroups per query
#ifndef SM_G
#define SM_G 8  // chunks of 64 driver postings per group (bm25_probe_body.h)
#endif
#ifndef SM_ARRIVE_ACQREL
#define SM_ARRIVE_ACQREL 0  // 1: acquire / release at agent scope on the arrival counter (see the arrival below)
#endif

struct pb_squery {
  uint32_t n_terms, op;  // op = SS_OP_* | NOT terms << 8 (bm_q_op / bm_q_nnot)
  uint32_t term[8];      // scored terms, then the NOT terms
  float idf[4];
  float thr0;            // threshold seed: a score k docs of the query reach for sure (0 = none; bm_kth_kernel)
};
static_assert(sizeof(pb_squery) == 60, "layout");

struct PbSmall {
  pb_squery q[SM_MAX_Q];
  const uint32_t* post;
  const unsigned long long* term_base;
  const uint32_t* sub_off;
  const uint2* probe;
  const uint32_t* probe_z;
  const uint32_t* probe_row;
  const float* umax;
  const uint32_t* del;
  unsigned long long* part_keys;  // [nq][PB][64 * KPL]
  unsigned long long* total;      // [SM_MAX_Q]      zero between launches
  uint32_t* tau;                  // [SM_MAX_Q][BM_TAU_STRIDE]  zero between launches
  uint32_t* arrive;               // [SM_MAX_Q + 1]  zero between launches; the last word counts finished queries
  uint32_t* out_doc;              // caller's buffers (pinned host memory or device memory)
  float* out_score;
  uint32_t* out_count;
  unsigned long long* out_total;
  unsigned long long* bests;      // [SM_MAX_Q][SM_MAX_PB * 8] best key of every partition (zero between launches): pb_publish_kth_best
  uint32_t* flag;                 // pinned host word: = seq when every answer is in place
  uint32_t del_words, n_sub, n_terms, nq, PB, CB, k, count, seq;
};
static_assert(sizeof(PbSmall) <= 4096, "kernel arguments are limited to 4 KB");

typedef __attribute__((address_space(3))) unsigned long long bm_lds_u64;
__device__ __forceinline__ u64 lds_ld64(uint32_t off) { return *(bm_lds_u64*)(uintptr_t)off; }
__device__ __forceinline__ void lds_st64(uint32_t off, u64 v) { *(bm_lds_u64*)(uintptr_t)off = v; }

// Tournament over <= 64 sorted lists (descending, 0 = exhausted; keys are unique): lane p plays list p through ld(rank); k <= 32
// rounds of a wave-wide maximum, the winning lane advances.  Four entries of every list are fetched ahead, so that the rounds do
// not wait on memory (a list that places more than four keys fetches the next four).  Returns rank r of the merged list in lane r.
template <typename LD>
__device__ __forceinline__ u64 pb_tournament(bool have, uint32_t k, LD ld, int lane) {
  u64 h0 = have ? ld(0u) : 0ull, h1 = have ? ld(1u) : 0ull, h2 = have ? ld(2u) : 0ull, h3 = have ? ld(3u) : 0ull;
  uint32_t cur = 0;
  u64 mine = 0ull;
  for (uint32_t r = 0; r < k; r++) {
    u64 m = h0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 x = shflx64(m, o); m = x > m ? x : m; }
    if (m == 0ull) break;  // wave-uniform: every list is exhausted
    if ((uint32_t)lane == r) mine = m;
    if (h0 == m) {
      h0 = h1; h1 = h2; h2 = h3; h3 = 0ull;
      cur++;
      if ((cur & 3u) == 0u) { h0 = ld(cur); h1 = ld(cur + 1u); h2 = ld(cur + 2u); h3 = ld(cur + 3u); }
    }
  }
  return mine;
}

template <int NT, int KPL, bool FILT>
__global__ void __launch_bounds__(PB_WAVES * 64, 4) bm25_small_kernel(const PbSmall fz_) {
  // the arguments are read where they lie (the kernel argument segment: constant address space, scalar loads with a run-time
  // index for the query) -- indexing the by-value copy would put it into scratch
  typedef __attribute__((address_space(4))) const PbSmall KArgs;
  KArgs* fz = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t nq = fz->nq, PB = fz->PB, k = fz->k, b = blockIdx.x;
  constexpr uint32_t KS = 64u * KPL;
  constexpr uint32_t WREG = pb_qcap(SM_G) * 12u;  // a wave's LDS region (its survivor queue while it probes, its list afterwards)
  // What one workgroup hands to another (partition lists, counts) travels in device-scope atomic accesses -- they meet at the
  // coherence point of the 8 XCDs' L2s by themselves.  No __threadfence(): on this part an agent-scope fence writes back and
  // invalidates the XCD's L2, and one per wave made a batch of 64 TopkCount queries take 595 us instead of 230.
  auto ldk = [&](const u64* p) -> u64 { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto stk = [&](u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  uint32_t qi;
  if (b < nq * PB) {
    // ---- role 1: eight partitions of query qi
    qi = b % nq;
    const uint32_t pb = b / nq, part = pb * PB_WAVES + (uint32_t)w;
    PbQueryRegs<NT> Q;
    Q.nt_ = fz->q[qi].n_terms;
    Q.op_ = fz->q[qi].op;
#pragma unroll
    for (int t = 0; t < NT; t++) { Q.term_[t] = fz->q[qi].term[t]; Q.idf_[t] = fz->q[qi].idf[t]; }
#pragma unroll
    for (int j = 0; j < 4; j++) Q.not_[j] = FILT ? fz->q[qi].term[min(Q.nt_ + (uint32_t)j, 7u)] : 0u;
    BmTop<KPL> T = pb_wave<NT, KPL, FILT, false, true, SM_G, true>(fz->post, fz->term_base, fz->sub_off, fz->probe, fz->probe_z, fz->probe_row, fz->umax, nullptr, nullptr,
                                                 Q, fz->tau, fz->del, fz->del_words, fz->n_sub, fz->n_terms, PB * PB_WAVES, k, fz->count & 1u, qi, part, w, lane, fz->q[qi].thr0,
                                                 (fz->count & 4u) && k <= 64u && PB * PB_WAVES >= k ? fz->bests + (size_t)qi * (SM_MAX_PB * PB_WAVES) : nullptr);
    // the workgroup's eight lists -> one (LDS: every wave's queue is empty by now and its region its own)
    const uint32_t lb = (uint32_t)w * WREG;
#pragma unroll
    for (int r = 0; r < KPL; r++) lds_st64(lb + ((uint32_t)r * 64u + (uint32_t)lane) * 8u, T.keys[r]);
    if (lane == 0) lds_st64(lb + KS * 8u, T.matched);
    // every wave's device-scope traffic -- the no-return atomicMax on the query's threshold, its best-key slots -- is PERFORMED before
    // the barrier: wave 0 arrives behind it, so whoever arrives last may zero that state for the next launch without a late atomic
    // landing on top (the workgroup barrier alone does not wait for vmcnt)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (w != 0) return;
    u64 matched = 0ull;
#pragma unroll
    for (int ww = 0; ww < PB_WAVES; ww++) matched += lds_ld64((uint32_t)ww * WREG + KS * 8u);
    u64* mine = fz->part_keys + ((size_t)qi * PB + pb) * KS;
    if (k <= 32u) {
      const uint32_t ob = (uint32_t)(lane & 7) * WREG;
      const u64 m = pb_tournament(lane < PB_WAVES, k, [&](uint32_t rank) -> u64 { return rank < KS ? lds_ld64(ob + rank * 8u) : 0ull; }, lane);
      stk(mine + lane, m);  // ranks >= k: 0 (KPL = 1: k <= 32 < 64)
      if (KPL == 2) stk(mine + 64 + lane, 0ull);
    } else {
#pragma unroll 1
      for (int ww = 1; ww < PB_WAVES; ww++) {
        const uint32_t ob = (uint32_t)ww * WREG;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
          const u64 key = lds_ld64(ob + ((uint32_t)r * 64u + (uint32_t)lane) * 8u);
          if (__ballot(key > T.worst)) T = bm_offer_lane_keys<KPL>(T, key > T.worst ? key : 0ull, k, nullptr);
        }
      }
#pragma unroll
      for (int r = 0; r < KPL; r++) stk(mine + r * 64 + lane, T.keys[r]);
    }
    if (lane == 0 && matched) atomicAdd(&fz->total[qi], matched);
  } else {
    // ---- role 2: exact count of a union of >= 2 lists, popcounted from the bit records (bm25_union_count_kernel's job; the body
    // above counts intersections and single lists itself)
    const uint32_t c = b - nq * PB;
    qi = c % nq;
    const uint32_t cpart = (c / nq) * PB_WAVES + (uint32_t)w, CP = fz->CB * PB_WAVES;
    const uint32_t np = fz->q[qi].n_terms, op = fz->q[qi].op, n_not = bm_q_nnot(op);
    if (bm_q_op(op) == SS_OP_UNION && np >= 2u) {
      const uint32_t n_groups = fz->n_sub * (uint32_t)(BM_SUB / 64);
      const uint32_t g_begin = (uint32_t)(((u64)n_groups * cpart) / CP), g_end = (uint32_t)(((u64)n_groups * (cpart + 1u)) / CP);
      const uint2* rows[8];
#pragma unroll
      for (int t = 0; t < 8; t++) rows[t] = fz->probe + (size_t)fz->probe_row[fz->q[qi].term[(uint32_t)t < np + n_not ? t : 0]] * n_groups;
      const uint32_t* __restrict__ del = fz->del;
      const uint32_t del_words = fz->del_words;
      uint32_t cnt = 0;
      constexpr int U = 4;
      for (uint32_t g0 = g_begin; g0 < g_end; g0 += 64u * U) {
        u64 acc[U], neg[U];
#pragma unroll
        for (int u = 0; u < U; u++) { acc[u] = 0ull; neg[u] = 0ull; }
#pragma unroll
        for (int t = 0; t < 8; t++) {
          if ((uint32_t)t >= np + n_not) break;
#pragma unroll
          for (int u = 0; u < U; u++) {
            const uint32_t g = g0 + 64u * u + (uint32_t)lane;
            uint2 r = make_uint2(0u, 0u);
            if (g < g_end) r = rows[t][g];
            const u64 bts = ((u64)r.y << 32) | r.x;
            if ((uint32_t)t < np) acc[u] |= bts; else neg[u] |= bts;
          }
        }
        if (del) {
#pragma unroll
          for (int u = 0; u < U; u++) {
            const uint32_t g = g0 + 64u * u + (uint32_t)lane;  // group g = bitmap words 2g, 2g + 1
            if (g < g_end && 2u * g + 1u < del_words) {
              const uint2 r = ((const uint2*)del)[g];
              neg[u] |= ((u64)r.y << 32) | r.x;
            } else if (g < g_end && 2u * g < del_words) {
              neg[u] |= (u64)del[2u * g];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++) cnt += (uint32_t)__popcll(acc[u] & ~neg[u]);
      }
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
      if (lane == 0) lds_st32((uint32_t)w * WREG, cnt);
      __syncthreads();
      if (w != 0) return;
      uint32_t sum = 0u;
#pragma unroll
      for (int ww = 0; ww < PB_WAVES; ww++) sum += lds_ld32((uint32_t)ww * WREG);
      if (lane == 0 && sum) atomicAdd(&fz->total[qi], (unsigned long long)sum);
    } else if (w != 0) {
      return;
    }
  }

  // ---- arrival: one atomic per workgroup, behind its list and count (s_waitcnt: stores and atomics without return are counted
  // until they are performed); the last workgroup of the query merges and answers
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the compiler keeps the order; the hardware's part is the s_waitcnt)
  __builtin_amdgcn_s_waitcnt(0);
  uint32_t prev = 0u;
#if SM_ARRIVE_ACQREL
  // the memory model's own form: release what this workgroup wrote, acquire what the others released (one fence pair per WORKGROUP).
  // Measured against the relaxed form below (profiles/r6_small_arrive.log); the litmus probe tools/probes/small_litmus.hip is the
  // evidence the relaxed form rests on.
  if (lane == 0) prev = __hip_atomic_fetch_add(&fz->arrive[qi], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
  if (lane == 0) prev = atomicAdd(&fz->arrive[qi], 1u);
#endif
  prev = __builtin_amdgcn_readfirstlane(prev);
  if (prev + 1u != PB + fz->CB) return;
  const u64* lists = fz->part_keys + (size_t)qi * PB * KS;
  uint32_t* o_doc = fz->out_doc + (size_t)qi * k;
  float* o_score = fz->out_score + (size_t)qi * k;
  uint32_t n_res = 0u;
  if (k <= 32u) {
    const u64* lst = lists + (size_t)min((uint32_t)lane, PB - 1u) * KS;
    const u64 m = pb_tournament((uint32_t)lane < PB, k, [&](uint32_t rank) -> u64 { return rank < KS ? ldk(lst + rank) : 0ull; }, lane);
    if ((uint32_t)lane < k) {
      o_doc[lane] = m ? 0xFFFFFFFFu - (uint32_t)m : SS_NO_DOC;
      o_score[lane] = m ? __uint_as_float((uint32_t)(m >> 32)) : 0.f;
    }
    n_res = (uint32_t)__popcll(__ballot(m != 0ull));
  } else {
    BmTop<KPL> F;
#pragma unroll
    for (int r = 0; r < KPL; r++) F.keys[r] = 0ull;
    F.worst = 0ull; F.wsc = -1.0f; F.matched = 0;
#pragma unroll 1
    for (uint32_t p = 0; p < PB; p++) {
#pragma unroll
      for (int r = 0; r < KPL; r++) {
        const u64 key = ldk(lists + (size_t)p * KS + (uint32_t)r * 64u + (uint32_t)lane);
        if (__ballot(key > F.worst)) F = bm_offer_lane_keys<KPL>(F, key > F.worst ? key : 0ull, k, nullptr);
      }
    }
#pragma unroll
    for (int r = 0; r < KPL; r++) {
      const uint32_t rank = (uint32_t)r * 64u + (uint32_t)lane;
      const u64 m = F.keys[r];
      if (rank < k) {
        o_doc[rank] = m ? 0xFFFFFFFFu - (uint32_t)m : SS_NO_DOC;
        o_score[rank] = m ? __uint_as_float((uint32_t)(m >> 32)) : 0.f;
      }
      n_res += (uint32_t)__popcll(__ballot(m != 0ull && rank < k));
    }
  }
  if (lane == 0) {
    fz->out_count[qi] = n_res;
    fz->out_total[qi] = __hip_atomic_load(&fz->total[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // leave the query's state as the next launch expects it
    fz->total[qi] = 0ull;
    if (!(fz->count & 2u)) fz->tau[(size_t)qi * BM_TAU_STRIDE] = 0u;  // (bit 1: an experiment -- the next launch starts from these thresholds)
    fz->arrive[qi] = 0u;
  }
  for (uint32_t p_ = (uint32_t)lane; p_ < PB * PB_WAVES; p_ += 64u) fz->bests[(size_t)qi * (SM_MAX_PB * PB_WAVES) + p_] = 0ull;
  __threadfence_system();  // the answers (host memory) before the flag
  if (lane == 0) {
    const uint32_t done = atomicAdd(&fz->arrive[SM_MAX_Q], 1u);
    if (done + 1u == nq) {
      fz->arrive[SM_MAX_Q] = 0u;
      __hip_atomic_store(fz->flag, fz->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------- host side
size_t ssi_bm25_small_ws_bytes() {
  return (size_t)SM_MAX_Q * SM_MAX_PB * 128u * sizeof(u64) + SM_MAX_Q * sizeof(u64) + (size_t)SM_MAX_Q * BM_TAU_STRIDE * 4u + (SM_MAX_Q + 1u) * 4u + 64u +
         (size_t)SM_MAX_Q * SM_MAX_PB * PB_WAVES * sizeof(u64) + 8u;
}

// can this batch take the one-launch path?  (the caller has run check_queries: no phrase, no all_terms_frequent, every list with a probe row)
bool ssi_bm25_small_serves(const ss_shard* s, uint32_t nq, uint32_t k, uint32_t np_max, uint32_t nn_max) {
  static const int on = [] { const char* e = getenv("SS_BM25_SMALL"); return e ? atoi(e) : 1; }();
  // (several indexed fields: images with MERGED lists -- a query without a field filter reads one list per term, ss_common.h bm_merged)
  return on && nq >= 1 && nq <= SM_MAX_Q && (s->bm_n_fields == 1 || (s->bm_merged && s->h_boost.size() == s->bm_n_fields)) && s->d_probe && s->d_probe_z && s->d_probe_row && s->d_umax && k >= 1 && k <= 128 &&
         np_max >= 1 && np_max <= 4 && nn_max <= 4 && !s->del_per_query && s->bm_strategy != SS_BM25_EXHAUSTIVE && s->bm_strategy != SS_BM25_EXHAUSTIVE_F32;
}

// ws: ssi_bm25_small_ws_bytes() of device memory, zero when first used.  Outputs / flag: device-visible addresses (pinned host memory).
int ssi_bm25_small_launch(ss_shard* s, void* ws, uint32_t nq, const ss_bm25_query* hq, uint32_t k, bool want_counts, bool has_and, bool has_or,
                          uint32_t np_max, bool any_not, uint32_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total,
                          uint32_t* flag, uint32_t seq, hipStream_t st) {
  if (nq == 0 || nq > SM_MAX_Q || !ws) return SS_EINVAL;
  PbSmall a;
  static const int keep_tau = [] { const char* e = getenv("SS_BM25_KEEP_TAU"); return e ? atoi(e) : 0; }();
  const uint32_t ksel = keep_tau ? 3u : bm_kth_sel(k);
  for (uint32_t i = 0; i < nq; i++) {
    const ss_bm25_query& Q = hq[i];
    const uint32_t np = Q.n_terms, n_not = bm_q_nnot(Q.op);
    if (np == 0 || np > 4 || n_not > 4) return SS_EINVAL;
    pb_squery& o = a.q[i];
    o.n_terms = np;
    // a query of ONE term is always a union (bm_expand_kernel); nothing else of `op` reaches the kernel
    o.op = (np > 1 ? bm_q_op(Q.op) : (uint32_t)SS_OP_UNION) | (n_not << 8);
    // the list a term reads: its only one, or -- several indexed fields -- its MERGED list (the last of the term's L lists), whose
    // weights carry sum_f boost_f w_f / S and whose "boost" S gives the scale back through idf (bm_expand_kernel does the same)
    const uint32_t L = s->bm_n_fields;
    const float scale = L > 1 ? s->h_boost[L - 1] : 1.0f;
    for (uint32_t t = 0; t < 8; t++) o.term[t] = t < np + n_not ? Q.term[t] * L + (L - 1u) : s->bm_n_terms;  // (absent: the all-zero row)
    for (uint32_t t = 0; t < 4; t++) o.idf[t] = t < np ? (L > 1 ? scale * Q.idf[t] : Q.idf[t]) : 0.f;
    // threshold seed: a union with nothing that takes a doc away again (no NOT terms, no tombstones / filter bitmap)
    o.thr0 = 0.f;
    if (ksel < 3u && n_not == 0 && !s->n_deleted && !s->h_kthw.empty() && (np == 1 || bm_q_op(Q.op) == SS_OP_UNION))
      for (uint32_t t = 0; t < np; t++) o.thr0 = std::max(o.thr0, o.idf[t] * s->h_kthw[(size_t)o.term[t] * 4u + ksel]);
  }
  for (uint32_t i = nq; i < SM_MAX_Q; i++) memset(&a.q[i], 0, sizeof(pb_squery));
  const int KPL = k <= 64 ? 1 : 2;
  const uint32_t KS = 64u * KPL;
  char* w = (char*)ws;
  a.part_keys = (unsigned long long*)w;
  w += (size_t)SM_MAX_Q * SM_MAX_PB * 128u * sizeof(u64);
  a.total = (unsigned long long*)w;
  w += SM_MAX_Q * sizeof(u64);
  a.tau = (uint32_t*)w;
  w += (size_t)SM_MAX_Q * BM_TAU_STRIDE * 4u;
  a.arrive = (uint32_t*)w;
  w += ((SM_MAX_Q + 1u) * 4u + 64u + 7u) & ~(size_t)7u;
  a.bests = (unsigned long long*)w;
  (void)KS;
  a.post = s->d_post;
  a.term_base = (const unsigned long long*)s->d_term_base;
  a.sub_off = s->d_sub_off;
  a.probe = s->d_probe;
  a.probe_z = s->d_probe_z;
  a.probe_row = s->d_probe_row;
  a.umax = s->d_umax;
  a.del = s->n_deleted ? s->d_deleted : nullptr;
  a.del_words = (uint32_t)s->deleted_words;
  a.n_sub = s->bm_n_sub;
  a.n_terms = s->bm_n_terms;
  a.nq = nq;
  a.k = k;
  // bit 2: the query's threshold from the partitions' best keys (pb_publish_kth_best).  From 16 queries per call on: 64 queries 164 -> 105 us,
  // 32 queries 112 -> 87 us; a call of 1 / 8 queries -- 512 partitions per query, a handful of groups each -- pays 6 / 13 us for the
  // re-computations and gains nothing (tools/probes/small_fused.py, profiles/r5_small_kth_best.log)
  static const int bests_min = [] { const char* e = getenv("SS_BM25_SMALL_BESTS_MIN"); return e ? atoi(e) : 16; }();
  const bool use_bests = nq >= (uint32_t)bests_min;
  a.count = (want_counts ? 1u : 0u) | (keep_tau ? 2u : 0u) | (use_bests ? 4u : 0u);
  a.seq = seq;
  a.out_doc = out_doc; a.out_score = out_score; a.out_count = out_count; a.out_total = (unsigned long long*)out_total; a.flag = flag;
  // partitions: about 4096 waves in all (the staged path's rule), 16 .. 256 per query; intersections at least 48 (the shortest list
  // drives, shorter assignments balance better); never more than the sub-blocks can feed
  static const int pb_env = [] { const char* e = getenv("SS_BM25_SMALL_PB"); return e ? atoi(e) : 0; }();
  // (measured on C2, tools/probes/small_fused.py: one query 54.9 us at 32 workgroups, 50.6 at 64; 8 queries 71.5 / 68.4; 32 queries best at 16)
  // (with the best-keys threshold 16 .. 31 queries are best at 2048 waves in all: 16 queries 84.5 us at 32 workgroups each, 77.7 at 16)
  uint32_t PB = std::max<uint32_t>(has_and ? 6u : 2u, std::min<uint32_t>(SM_MAX_PB, ((use_bests && nq < 32u && !has_and) ? 256u : 512u) / nq));
  if (pb_env > 0) PB = (uint32_t)pb_env;
  PB = std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(PB, SM_MAX_PB), (s->bm_n_sub + PB_WAVES - 1) / PB_WAVES));
  a.PB = PB;
  // counting workgroups: a wave per >= 1024 groups of 64 docs, at most 2048 workgroups more in all
  uint32_t CB = 0;
  if (want_counts && has_or) {
    const uint32_t n_groups = s->bm_n_sub * (uint32_t)(BM_SUB / 64);
    CB = std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(SM_MAX_CB, (n_groups / 1024u + PB_WAVES - 1) / PB_WAVES), std::max<uint32_t>(1u, 2048u / nq)));
  }
  a.CB = CB;
  const uint32_t NT = np_max <= 2 ? 2u : np_max;
  const bool filt = any_not || a.del != nullptr;
  const dim3 grid(nq * (PB + CB)), block(PB_WAVES * 64);
  const size_t lds = (size_t)PB_WAVES * pb_qcap(SM_G) * 12;
#define SS_S(NT_, KPL_)                                                                    \
  if (NT == NT_ && KPL == KPL_) {                                                          \
    if (filt) bm25_small_kernel<NT_, KPL_, true><<<grid, block, lds, st>>>(a);             \
    else bm25_small_kernel<NT_, KPL_, false><<<grid, block, lds, st>>>(a);                 \
  }
  SS_S(2, 1) SS_S(3, 1) SS_S(4, 1) SS_S(2, 2) SS_S(3, 2) SS_S(4, 2)
#undef SS_S
  SS_HIP(hipGetLastError());
  return SS_OK;
}
