// The wave-level body of the pruned strategy (bm25_probe.hip has the story): ONE (query, partition) assignment walked by one wave,
// shared by bm25_probe_kernel (batches: one launch per stage) and bm25_small_kernel (bm25_small.hip: small host-pointer batches, the
// whole search in one launch).  Device code only; included by both translation units.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "bm25_dev.h"

constexpr int PB_WAVES = 8;
// survivor queue entries per wave: < 64 left over + one group of G x 64 pushed (G = chunks of 64 driver postings evaluated together)
constexpr int pb_qcap(int G) { return 64 * G + 64; }
constexpr int PB_QCAP = pb_qcap(4);

// weight of one posting: it is IN the posting (ss_common.h) -- the same decode as the scan kernels'
__device__ __forceinline__ float pb_weight(uint32_t p) { return bm_weight(p); }

// How the body reads its query: the expanded query in device memory (bm_expand_kernel's bm_vquery) ...
struct PbQueryMem {
  const bm_vquery* __restrict__ q;
  __device__ __forceinline__ uint32_t n_terms() const { return q->n_terms; }
  __device__ __forceinline__ uint32_t op() const { return q->op; }
  __device__ __forceinline__ uint32_t term(int t) const { return q->term[t]; }
  __device__ __forceinline__ float idf(int t) const { return q->idf[t]; }
  __device__ __forceinline__ uint32_t not_term(uint32_t nt, uint32_t j) const { return q->term[nt + j]; }
};
// ... or a query held in registers (bm25_small_kernel reads it from its kernel arguments): <= NT scored terms, <= 4 NOT terms
template <int NT>
struct PbQueryRegs {
  uint32_t nt_, op_, term_[NT], not_[4];
  float idf_[NT];
  __device__ __forceinline__ uint32_t n_terms() const { return nt_; }
  __device__ __forceinline__ uint32_t op() const { return op_; }
  __device__ __forceinline__ uint32_t term(int t) const { return term_[t]; }
  __device__ __forceinline__ float idf(int t) const { return idf_[t]; }
  __device__ __forceinline__ uint32_t not_term(uint32_t, uint32_t j) const { return j == 0u ? not_[0] : j == 1u ? not_[1] : j == 2u ? not_[2] : not_[3]; }
};

// The query's threshold from the partitions' BEST keys (bm25_scan16.hip has the measurement that led here): `bests` holds one key per
// partition of the query (0 = none yet); a partition that has just raised its own stores it, reads all of them and publishes the k-th
// largest -- k distinct docs reach it, so the query's k-th best score does -- into the shared threshold.  Out of line: rare, and its
// sort would cost the probe loop registers.
__device__ __attribute__((noinline)) void pb_publish_kth_best(unsigned long long* bests, uint32_t stride, uint32_t P, uint32_t part, uint32_t k, u64 best, uint32_t* tau_q) {
  const int lane = __lane_id();
  if (lane == 0) __hip_atomic_store(bests + (size_t)part * stride, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (at most 256 of the partitions are read -- every (P / 256)-th: the k-th largest over a SUBSET of the docs is a lower bound all the same,
  // and a single query runs as 512 partitions whose every wave would otherwise read 512 slots each time its best key moves)
  const uint32_t step = (P + 255u) / 256u;
  u64 m = 0ull;
  for (uint32_t p_ = (uint32_t)lane * step; p_ < P; p_ += 64u * step) {
    const u64 x = p_ == part ? best : __hip_atomic_load(bests + (size_t)p_ * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    m = x > m ? x : m;
  }
  m = wave_sort_desc(m, lane);
  const u64 kth = rdlane64(m, (int)k - 1);
  if (kth && lane == 0) bm_publish_tau(tau_q, __uint_as_float((uint32_t)(kth >> 32)));
}

// The same for 64 < k <= 128 (KPL = 2): every partition publishes its TWO best keys (`bests` holds [P][2]) -- 64 partitions of a query in a
// batch of 64 then show 128 distinct docs, enough for a top-100 -- and the k-th largest of the lanes' best two is the bound (any k distinct
// docs give one).  A reader may catch a partition between its two stores: (new best, old second) are two real docs; (old best, new second
// = the old best) is ONE doc twice -- equal keys are the same doc, the copy is dropped.
__device__ __attribute__((noinline)) void pb_publish_kth_best2(unsigned long long* bests, uint32_t P, uint32_t part, uint32_t k, u64 best, u64 second, uint32_t* tau_q) {
  const int lane = __lane_id();
  if (lane == 0) {
    __hip_atomic_store(bests + (size_t)part * 2u, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(bests + (size_t)part * 2u + 1u, second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const uint32_t step = (P + 255u) / 256u;
  u64 m0 = 0ull, m1 = 0ull;  // this lane's best two
  for (uint32_t p_ = (uint32_t)lane * step; p_ < P; p_ += 64u * step) {
    const u64 x0 = p_ == part ? best : __hip_atomic_load(bests + (size_t)p_ * 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 x1 = p_ == part ? second : __hip_atomic_load(bests + (size_t)p_ * 2u + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (x1 >= x0) x1 = 0ull;  // (the same doc read twice, or a second that overtook the best it was read after)
    if (x0 > m0) { m1 = m0 > x1 ? m0 : x1; m0 = x0; }
    else if (x0 > m1) m1 = x0;
  }
  u64 kk[2] = {0ull, 0ull};
  (void)topk_merge64<2>(kk, m0, 128u, lane);
  (void)topk_merge64<2>(kk, m1, 128u, lane);
  const u64 kth = k <= 64u ? rdlane64(kk[0], (int)k - 1) : rdlane64(kk[1], (int)k - 65);
  if (kth && lane == 0) bm_publish_tau(tau_q, __uint_as_float((uint32_t)(kth >> 32)));
}

// FILT: tombstones and / or NOT terms are present (a separate instantiation: the unfiltered kernel pays nothing for them)
// SKIP: the driver streams jump over sub-blocks whose block-max bound (qbound) lies below the threshold.  A separate
// instantiation: the few registers the skip needs pushed the common kernel into scratch (C2: 0.55 -> 0.80 ms per 1000 queries).
// SEEDED: the caller knows a score k docs of the query reach for sure (thr0, bm_kth_kernel): the threshold never lies below it
// G: chunks of 64 driver postings per group -- their gathers are in flight together (4 in the staged kernel, whose registers are capped
// for occupancy; 8 in the one-launch kernel of small batches, which is bound by the number of dependent round trips per wave)
// KTHB: the partitions publish their best keys and derive the query's threshold from them (pb_publish_kth_best; `bests` + `bests_stride`)
// EXT: the query's shared threshold may rise from OUTSIDE these lists while they are read (bm25_small.hip, the tiered instances)
template <int NT, int KPL, bool FILT, bool SKIP, bool SEEDED, int G, bool KTHB, bool EXT = false, typename QV>
__device__ __forceinline__ BmTop<KPL> pb_wave(
    const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base, const uint32_t* __restrict__ sub_off,
    const uint2* __restrict__ probe, const uint32_t* __restrict__ probe_z,
    const uint32_t* __restrict__ probe_row, const float* __restrict__ umax, const float* __restrict__ pmax, const float* __restrict__ qbound,
    const QV Q, uint32_t* tau, const uint32_t* __restrict__ del, uint32_t del_words, uint32_t n_sub, uint32_t n_terms,
    uint32_t P, uint32_t k, uint32_t count, const uint32_t qi, const uint32_t part, const int w, const int lane, const float thr0 = 0.f,
    unsigned long long* bests = nullptr /* KTHB: best key of partition p of the query at bests[p * bests_stride] */, const uint32_t bests_stride = 1u) {
  // del_words bit 31: ONE exclusion bitmap PER QUERY, del_words words each, back to back (ss_bm25_search_sorted: every query of the
  // batch is searched inside its own doc set)
  if (FILT && (del_words >> 31)) { del_words &= 0x7FFFFFFFu; del += (size_t)qi * del_words; }
    const uint32_t nt = Q.n_terms(), n_not = FILT ? bm_q_nnot(Q.op()) : 0u;  // NT covers the query terms; NOT terms are probed at the end
  const bool is_and = (bm_q_op(Q.op()) == SS_OP_INTERSECTION) && nt > 1;
  const uint32_t row_len = n_sub + 1;

  // per-term state in PROCESSING order (sorted below); qpos = position in the query (order of the score sum)
  const uint32_t* tptr[NT];
  const uint32_t* rowp[NT];
  const uint2* prow[NT];   // 64 doc bits per group
  const uint32_t* zrow[NT];  // index of the group's first posting (fetched on hits only)
  float idf[NT], U[NT];
  uint32_t qpos[NT];
  unsigned long long size[NT];
  const uint32_t s_begin = (uint32_t)(((u64)n_sub * part) / P);
  const uint32_t s_end = (uint32_t)(((u64)n_sub * (part + 1)) / P);
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const bool have = (uint32_t)t < nt;
    const uint32_t term = have ? Q.term(t) : n_terms;
    idf[t] = have ? Q.idf(t) : 0.f;
    // Upper bound of idf * w over the docs THIS wave can see: the largest weight of the term inside the partition's
    // sub-blocks (bm_partmax_kernel below, from the per-(term, block) maxima -- the reference's max_block_score,
    // index.rs:2938-3200, used like intersection.rs:2090-2097 / single.rs:373-386): tighter than the list-level maximum
    // wherever weights are not spread evenly over the doc ids, never looser.  (umax = the largest DECODED weight of the list.)
    U[t] = idf[t] * (pmax ? pmax[((size_t)qi * P + part) * 4u + t] : umax[term]);
    qpos[t] = t;
    tptr[t] = post + term_base[term] * 4ull;
    rowp[t] = sub_off + (size_t)term * row_len;
    prow[t] = probe + (size_t)probe_row[term] * n_sub * (BM_SUB / 64);  // the host sends only queries whose lists all have a row
    zrow[t] = probe_z + (size_t)probe_row[term] * n_sub * (BM_SUB / 64);
    size[t] = have ? term_base[term + 1] - term_base[term] : ~0ull;
  }
  // sort: unions by U descending (absent terms have U = 0: last), intersections by list size ascending (absent: last)
  auto cswap = [&](int x, int y) {
    const bool sw = is_and ? (size[y] < size[x]) : (U[y] > U[x]);
    if (sw) {
      { auto t_ = tptr[x]; tptr[x] = tptr[y]; tptr[y] = t_; }
      { auto t_ = rowp[x]; rowp[x] = rowp[y]; rowp[y] = t_; }
      { auto t_ = prow[x]; prow[x] = prow[y]; prow[y] = t_; }
      { auto t_ = zrow[x]; zrow[x] = zrow[y]; zrow[y] = t_; }
      { float t_ = idf[x]; idf[x] = idf[y]; idf[y] = t_; }
      { float t_ = U[x]; U[x] = U[y]; U[y] = t_; }
      { uint32_t t_ = qpos[x]; qpos[x] = qpos[y]; qpos[y] = t_; }
      { auto t_ = size[x]; size[x] = size[y]; size[y] = t_; }
    }
  };
  if (NT == 2) { cswap(0, 1); }
  if (NT == 3) { cswap(0, 1); cswap(1, 2); cswap(0, 1); }
  if (NT == 4) { cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2); }
  float SU[NT + 1];  // SU[j] = sum of U[j..]
  SU[NT] = 0.f;
#pragma unroll
  for (int j = NT - 1; j >= 0; j--) SU[j] = SU[j + 1] + U[j];

  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull;
  T.wsc = -1.0f;
  T.matched = 0;
  uint32_t* tau_q = tau + (size_t)qi * BM_TAU_STRIDE;
  u64 last_best = 0ull, last_second = 0ull;  // (KTHB: the best key(s) this partition has published)
  const int lane4 = lane * 4;

  // score in QUERY order with the exhaustive kernels' fma chain (bit-identical results)
  auto combine = [&](const float (&wv)[NT], uint32_t pres) -> float {
    float score = 0.f;
#pragma unroll
    for (uint32_t qp = 0; qp < (uint32_t)NT; qp++) {
#pragma unroll
      for (int t = 0; t < NT; t++)
        if (qpos[t] == qp && (pres >> t) & 1u) score = fmaf(idf[t], wv[t], score);
    }
    return score;
  };

  auto offer = [&](bool cand, float score, uint32_t doc) {
    if (__ballot(cand)) {
      const u64 key = cand ? (((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - doc)) : 0ull;
      const u64 key2 = key > T.worst ? key : 0ull;
      if (__ballot(key2 != 0ull)) {
        T = bm_offer_lane_keys<KPL>(T, key2, k, tau_q);
        if (KTHB && KPL == 1) {  // (k <= 64: one key per lane)
          const u64 best = rdlane64(T.keys[0], 0);
          // (intersections: the driver list is read whatever the threshold -- nothing to gain there)
          if (bests && !is_and && best != last_best) { last_best = best; pb_publish_kth_best(bests, bests_stride, P, part, k, best, tau_q); }
        }
        if (KTHB && KPL == 2) {  // (64 < k <= 128: the partition's best TWO keys, bests = [P][2])
          const u64 best = rdlane64(T.keys[0], 0), second = rdlane64(T.keys[0], 1);
          if (bests && !is_and && (best != last_best || second != last_second)) {
            last_best = best; last_second = second;
            pb_publish_kth_best2(bests, P, part, k, best, second, tau_q);
          }
        }
      }
    }
  };
  // tombstone test (delete_hashset, add_result.rs:3435): only ever evaluated for the few lanes that still hold a candidate
  auto is_deleted = [&](bool lanes, uint32_t doc) -> bool {
    uint32_t wd = 0u;
    if (lanes && (doc >> 5) < del_words) wd = del[doc >> 5];
    return (wd >> (doc & 31u)) & 1u;
  };
  // NOT terms (add_result.rs:3440-3497): a candidate found in one of their lists is dropped; evaluated like the
  // tombstones, for the few lanes that still hold a candidate
  auto in_not_list = [&](bool lanes, uint32_t doc) -> bool {
    bool found = false;
    for (uint32_t j = 0; j < n_not; j++) {
      const uint32_t term = Q.not_term(nt, j);
      uint2 r = make_uint2(0u, 0u);
      if (lanes) r = probe[((size_t)probe_row[term] * n_sub + (doc >> BM_SUB_LOG2)) * (BM_SUB / 64) + ((doc & (BM_SUB - 1)) >> 6)];
      const u64 bits = ((u64)r.y << 32) | r.x;
      found = found || ((bits >> (doc & 63u)) & 1ull);
    }
    return found;
  };
  auto cur_thr = [&]() -> float {
    const float own = SEEDED ? fmaxf(T.wsc, thr0) : T.wsc;
    return fmaxf(own, __uint_as_float(__hip_atomic_load(tau_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
  };

  // Survivor queue (LDS, private to the wave): candidates that passed the DENSE stage -- driver posting weighed, first
  // other term probed -- wait here until 64 of them can take the SPARSE stage together (posting fetch of the probed
  // term, remaining probes, score).  After the first probe only a few percent of the lanes are still alive; without
  // the queue every later gather round trip would be paid for a handful of lanes.
  constexpr uint32_t QCAP = pb_qcap(G);
  const uint32_t q_doc = (uint32_t)w * (QCAP * 12u), q_w0 = q_doc + QCAP * 4u, q_pos = q_w0 + QCAP * 4u;
  uint32_t qn = 0;

  // one driver stream: the postings of processing term J inside this partition's sub-block range
  auto stream = [&](auto Jc) {
    constexpr int J = decltype(Jc)::value;
    constexpr int A = J == 0 ? 1 : 0;  // the first other term in probe order
    const uint32_t x_begin = rowp[J][s_begin] * 4u, x_end = rowp[J][s_end] * 4u;  // dword range of the stream
    if (x_begin == x_end) return;
    // J = 0 and a threshold above SU[0]: NO list is essential -- the threshold came from outside these lists (a tiered query's sparse lists,
    // whose docs carry the rare terms' idf: bm25_small.hip role 3 while this runs, or the staged tiered pipeline's seeds before it,
    // bm_ext_seed_kernel) and nothing the dense lists hold alone can reach it.  (A single list under exact counts is read to its end: it
    // counts its own matches.)
    if (J == 0 && !is_and && k && !(count && nt == 1) && SU[0] < cur_thr() * 0.99999f) return;

    // ---- sparse stage: the LAST n queue entries (n <= 64), one per lane
    auto drain = [&](uint32_t n) {
      const float thr = cur_thr();
      bool alive = (uint32_t)lane < n;
      const uint32_t qb = (qn - n + (uint32_t)lane) * 4u;
      const uint32_t doc = alive ? lds_ld32(q_doc + qb) : 0u;
      const uint32_t pos = alive ? lds_ld32(q_pos + qb) : 0xFFFFFFFFu;
      const uint32_t tile = doc >> BM_SUB_LOG2, d = doc & (BM_SUB - 1);
      float wv[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) wv[t] = 0.f;
      wv[J] = alive ? lds_ldf(q_w0 + qb) : 0.f;
      uint32_t pres = 1u << J;
      float known = idf[J] * wv[J];
      // round 1: z of the A hit, and (speculatively: these are the few survivors) the bits of every remaining term
      const uint32_t gidx = tile * (uint32_t)(BM_SUB / 64) + (d >> 6);
      const bool hit_a = alive && pos != 0xFFFFFFFFu;  // pos = rank inside the group
      const uint32_t za = zrow[A][hit_a ? gidx : 0u];  // unconditional loads (dead lanes: element 0), no exec-masked branches
      uint2 rb[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        rb[t] = make_uint2(0u, 0u);
        if (t == J || t == A || (uint32_t)t >= nt) continue;
        rb[t] = prow[t][alive ? gidx : 0u];
      }
      // round 2: the A posting, z of the other hits
      uint32_t pa = 0u;
      if (hit_a) pa = tptr[A][za + pos];
      bool hit[NT];
      uint32_t zt[NT], rk[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        hit[t] = false; zt[t] = 0u; rk[t] = 0u;
        if (t == J || t == A || (uint32_t)t >= nt) continue;
        const u64 bits = ((u64)rb[t].y << 32) | rb[t].x;
        hit[t] = alive && ((bits >> (d & 63u)) & 1ull);
        if (is_and) alive = hit[t];
        else if (t < J && hit[t]) { alive = false; }  // evaluated in the earlier term's stream
        rk[t] = (uint32_t)__popcll(bits & ((1ull << (d & 63u)) - 1ull));
      }
#pragma unroll
      for (int t = 0; t < NT; t++) {
        hit[t] = hit[t] && alive;
        if (t == J || t == A || (uint32_t)t >= nt) continue;
        zt[t] = zrow[t][hit[t] ? gidx : 0u];
      }
      if (hit_a) {
        wv[A] = pb_weight(pa);
        pres |= 1u << A;
        known += idf[A] * wv[A];
      }
      // round 3: postings of the other hits (skipped when the bounds already rule the doc out)
      float rest = SU[0] - U[J] - ((uint32_t)A < nt ? U[A] : 0.f);
      if (!is_and && k) alive = alive && (known + rest) >= thr * 0.99999f;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        if (t == J || t == A || (uint32_t)t >= nt) continue;
        if (hit[t] && alive) {
          const uint32_t pt = tptr[t][zt[t] + rk[t]];
          wv[t] = pb_weight(pt);
          pres |= 1u << t;
        }
      }
      if (FILT && del && __ballot(alive)) alive = alive && !is_deleted(alive, doc);
      if (FILT && n_not && !(count && is_and) && k) alive = alive && combine(wv, pres) >= thr;  // probe the NOT lists for real candidates only
      if (FILT && n_not && __ballot(alive)) alive = alive && !in_not_list(alive, doc);
      if (__ballot(alive)) {
        if (count && is_and) T.matched += __popcll(__ballot(alive));
        if (k) {
          const float score = combine(wv, pres);
          offer(alive && score >= thr && score > 0.f, score, doc);
        }
      }
      qn -= n;
    };

    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[J], 0, (int)(x_end * 4u), BM_RSRC_FLAGS);
    // sub-block boundaries of the driver (dword offsets), 64 per block load: lane i = sub-block blk0 + i
    uint32_t blk0 = s_begin;
    uint32_t vb = rowp[J][min(blk0 + (uint32_t)lane, s_end)] * 4u;
    const bool skip_blocks = SKIP && k != 0 && !count;  // exact counts need every match
    const float* __restrict__ qbrow = SKIP ? qbound + (size_t)qi * n_sub : nullptr;
    uint32_t vqb = 0u;  // bound of sub-block blk0 + lane
    if constexpr (SKIP) { if (skip_blocks) vqb = __float_as_uint(qbrow[min(blk0 + (uint32_t)lane, s_end - 1u)]); }
    auto bnd = [&](uint32_t s) -> uint32_t {  // s in [blk0, blk0 + 64), uniform
      return __builtin_amdgcn_readlane(vb, s - blk0);
    };
    uint32_t s_cur = s_begin;  // sub-block containing the stream position x (bnd(s_cur) <= x)
    uint32_t pn[G];            // driver postings of the NEXT group (prefetched)
#pragma unroll
    for (int g = 0; g < G; g++) pn[g] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4 + g * 256, (int)(x_begin * 4u), 0);
    for (uint32_t x = x_begin; x < x_end; x += 64u * G) {
      const float thr = cur_thr();
      // this and all later terms are non-essential now
      // (J = 0 is tested inside the loop by the EXT instances only -- the test in the first, longest stream of every query cost the C2
      // headline 11 %, 0.567 -> 0.631 ms per 1000 queries, profiles/r6_break_ab.log; the others test it once, before the stream: above)
      if (!is_and && k && (J > 0 || (EXT && !(count && nt == 1))) && SU[J] < thr * 0.99999f) break;
      if constexpr (SKIP) if (skip_blocks) {
        // block-max skip: while the sub-block holding position x cannot contain a doc that reaches thr, jump to the next one
        bool moved = false;
        for (;;) {
          if (s_cur + 1u - blk0 >= 64u) {
            blk0 = s_cur;
            vb = rowp[J][min(blk0 + (uint32_t)lane, s_end)] * 4u;
            vqb = __float_as_uint(qbrow[min(blk0 + (uint32_t)lane, s_end - 1u)]);
          }
          if (s_cur + 1u > s_end) break;
          const uint32_t b1 = bnd(s_cur + 1u);
          if (b1 <= x) { s_cur++; continue; }  // x lies beyond this sub-block
          if (__uint_as_float(__builtin_amdgcn_readlane(vqb, s_cur - blk0)) >= thr * 0.99999f) break;
          x = b1;  // nothing in the rest of this sub-block can enter the top-k
          s_cur++;
          moved = true;
          if (x >= x_end) break;
        }
        if (moved) {
          if (x >= x_end) break;
#pragma unroll
          for (int g = 0; g < G; g++) pn[g] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4 + g * 256, (int)(x * 4u), 0);
        }
      }
      uint32_t pg[G], tile[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        pg[g] = pn[g];
        tile[g] = s_cur;
      }
      if (x + 64u * G < x_end) {
#pragma unroll
        for (int g = 0; g < G; g++) pn[g] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4 + g * 256, (int)((x + 64u * G) * 4u), 0);
      }
      // each lane's sub-block: count the boundaries at or before its stream position
      const uint32_t x_hi = min(x + 64u * G, x_end);
      for (;;) {
        if (s_cur + 1u - blk0 >= 64u) {  // next boundary lies outside the loaded block
          blk0 = s_cur;
          vb = rowp[J][min(blk0 + (uint32_t)lane, s_end)] * 4u;
          if constexpr (SKIP) { if (skip_blocks) vqb = __float_as_uint(qbrow[min(blk0 + (uint32_t)lane, s_end - 1u)]); }
        }
        if (s_cur + 1u > s_end) break;
        const uint32_t b = bnd(s_cur + 1u);
        if (b >= x_hi) break;
        s_cur++;
#pragma unroll
        for (int g = 0; g < G; g++) tile[g] += (x + 64u * g + (uint32_t)lane) >= b ? 1u : 0u;
      }
      // ---- dense stage: weigh the driver posting, probe the first other term
      uint32_t dg[G];
      bool alive[G];
      float w0[G];
      uint2 rec[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        alive[g] = pg[g] != 0u;
        dg[g] = bm_doc_field(pg[g]) - 1u;  // doc inside its sub-block
        // The probe of the first other term is issued NOW, for every real posting and before its weight is known: the
        // weight lookups and bound tests below then run under the gather's latency instead of in front of it (postings the
        // bound test would have spared cost a cached read).  Unconditional load: NULL lanes read record 0.
        if (NT > 1) rec[g] = prow[A][alive[g] ? tile[g] * (uint32_t)(BM_SUB / 64) + (dg[g] >> 6) : 0u];
      }
#pragma unroll
      for (int g = 0; g < G; g++) w0[g] = alive[g] ? pb_weight(pg[g]) : 0.f;
      if (nt == 1) {  // single-term query: the driver posting is the whole score
#pragma unroll
        for (int g = 0; g < G; g++) {
          if (__ballot(alive[g]) == 0ull) continue;
          const uint32_t doc1 = (tile[g] << BM_SUB_LOG2) + dg[g];
          if (FILT && del && count) alive[g] = alive[g] && !is_deleted(alive[g], doc1);
          if (FILT && n_not && count) alive[g] = alive[g] && !in_not_list(alive[g], doc1);
          if (count) T.matched += __popcll(__ballot(alive[g]));
          if (k) {
            const float score = fmaf(idf[J], w0[g], 0.f);
            bool cand = alive[g] && score >= thr && score > 0.f;
            if (FILT && del && !count && __ballot(cand)) cand = cand && !is_deleted(cand, doc1);
            if (FILT && n_not && !count && __ballot(cand)) cand = cand && !in_not_list(cand, doc1);
            offer(cand, score, doc1);
          }
        }
        continue;
      }
      const float rest0 = SU[0] - U[J];
#pragma unroll
      for (int g = 0; g < G; g++)
        if (!is_and && k) alive[g] = alive[g] && (idf[J] * w0[g] + rest0) >= thr * 0.99999f;
#pragma unroll
      for (int g = 0; g < G; g++) {
        const u64 bits = ((u64)rec[g].y << 32) | rec[g].x;
        bool hit = alive[g] && ((bits >> (dg[g] & 63u)) & 1ull);
        uint32_t pos = 0xFFFFFFFFu;
        if (is_and) alive[g] = hit;
        else if (A < J && hit) { alive[g] = false; hit = false; }  // evaluated in the earlier term's stream
        else if (!hit && k) alive[g] = alive[g] && (idf[J] * w0[g] + rest0 - U[A]) >= thr * 0.99999f;
        if (hit) pos = (uint32_t)__popcll(bits & ((1ull << (dg[g] & 63u)) - 1ull));  // rank inside the group
        // push the survivors of this chunk
        const u64 m = __ballot(alive[g]);
        if (m) {
          if (alive[g]) {
            const uint32_t at = qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            lds_st32(q_doc + at * 4u, (tile[g] << BM_SUB_LOG2) + dg[g]);
            lds_stf(q_w0 + at * 4u, w0[g]);
            lds_st32(q_pos + at * 4u, pos);
          }
          qn += (uint32_t)__popcll(m);
        }
      }
      while (qn >= 64u) drain(64u);
    }
    if (qn) drain(qn);
  };
  // drivers: unions -> every term in upper-bound order (a stream ends as soon as its term is non-essential);
  // intersections -> the shortest list only
  // (a union whose top-k is not wanted has nothing to do here: its count comes from bm25_union_count_kernel)
  if (k || is_and || nt == 1) stream(std::integral_constant<int, 0>{});
  if (!is_and && k) {
    if (NT > 1 && nt > 1) stream(std::integral_constant<int, (NT > 1 ? 1 : 0)>{});
    if (NT > 2 && nt > 2) stream(std::integral_constant<int, (NT > 2 ? 2 : 0)>{});
    if (NT > 3 && nt > 3) stream(std::integral_constant<int, (NT > 3 ? 3 : 0)>{});
  }

  return T;
}
