// BM25 scan kernels: exhaustive union / intersection scoring of a batch of queries with an exact top-k per query.
// Reference path replaced: union_docid_2/3 (union.rs:1168/1308), union_blockid -> union_scan (union.rs:265/403),
// intersection_blockid/docid (intersection.rs:2023/112), single_blockid (single.rs:292) together with
// add_result_multiterm_singlefield (add_result.rs:3418), get_bm25f_multiterm_singlefield (add_result.rs:1429) and
// MinHeap::add_topk (min_heap.rs:1193).
//
// One wave owns a (query, partition-of-sub-blocks) assignment.  Per 4096-doc sub-block ("item"):
//   phase 1  every term's segment is streamed with raw buffer loads (16 bytes per lane, CPT x 1 KB per term in
//            flight, prefetched one item ahead).  Segments are 16-byte aligned and NULL padded, and the buffer
//            descriptor's num_records is the segment end: lanes past it read zeros (= NULL postings) without touching
//            memory, so nothing in the loop is predicated.  A posting carries its finished weight (ss_common.h):
//            acc[doc] += idf * weight is a plain LDS gather / scatter on a tile private to the wave (bm_chunk,
//            bm25_dev.h), with no table and no doc-length lookup.
//   trigger  the running maximum of the updated scores is compared once per item with the current k-th best; only if
//            some doc could enter the list (or exact counts are wanted) is the tile scanned, otherwise just cleared.
// bm25_scan_fast_kernel<NT>: <= 4 terms, per-term state in scalar registers.  bm25_scan_group_kernel: up to 10 terms,
// processed four at a time per item with the term state re-read from a per-wave LDS table (no cross-item prefetch).
#include "bm25_dev.h"

template <int NT> struct FastCfg { static constexpr int CPT = NT <= 2 ? 4 : NT <= 4 ? 3 : 2; static constexpr int RC = NT * CPT; };

#define BM_KERNEL_ARGS                                                                                              \
  const uint32_t *__restrict__ post, const unsigned long long *__restrict__ term_base,                             \
      const uint32_t *__restrict__ sub_off, const bm_vquery *__restrict__ qs,                                      \
      unsigned long long *__restrict__ part_keys, unsigned long long *__restrict__ total, uint32_t *tau,            \
      const uint32_t *__restrict__ del, uint32_t del_words, uint32_t n_sub,                                         \
      uint32_t n_terms, uint32_t nq, uint32_t P, uint32_t k, uint32_t count

template <bool HAS_AND>
__device__ __forceinline__ BmLds bm_lds_setup(int lane, int w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WAVE_LDS = BM_WAVE_ACC + (HAS_AND ? BM_WAVE_CNT : 0);
  // the kernels have no static LDS, so the dynamic segment starts at LDS address 0: offsets below are absolute
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  const uint32_t wb = (uint32_t)w * WAVE_LDS;
  for (int i = lane; i < WAVE_LDS / 4; i += 64) lds_st32(wb + i * 4, 0u);
  BmLds L;
  L.accb = wb + 12;
  L.tile = wb + 16;
  L.cnt = wb + BM_WAVE_ACC + 3;
  L.cntw = wb + BM_WAVE_ACC + 4;
  return L;  // every wave initialises and uses only its own slice: no barrier
}

template <int NT, bool HAS_AND, int KPL>
__global__ void __launch_bounds__((HAS_AND ? BM_WAVES_AND : BM_WAVES_OR) * 64) bm25_scan_fast_kernel(BM_KERNEL_ARGS) {
  constexpr int WAVES = HAS_AND ? BM_WAVES_AND : BM_WAVES_OR;
  constexpr int CPT = FastCfg<NT>::CPT;
  constexpr int RC = FastCfg<NT>::RC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const BmLds L = bm_lds_setup<HAS_AND>(lane, w);

  const uint32_t row_len = n_sub + 1;
  const bool count_mode = count != 0;
  const int lane16 = lane * 16;

  // one (query, partition) assignment per wave; the grid covers all nq * P of them (workgroups are short-lived, the
  // dispatcher balances them over the CUs)
  const uint32_t a = blockIdx.x * WAVES + w;
  if (a < nq * P) {
    const uint32_t qi = a % nq, part = a / nq;
    const bm_vquery* __restrict__ Q = qs + qi;
    const uint32_t np = Q->n_terms, nt = np + bm_q_nnot(Q->op);  // scored (virtual) terms, then the NOT terms
    const uint32_t nt_and = HAS_AND ? Q->and_target : 0u;        // 0: union; else what a doc's match byte must reach
    const bool is_and = nt_and != 0u;
    // Per-term state in scalar registers: descriptor base, idf and a rolling window of three segment boundaries.
    // The boundaries themselves are fetched 64 at a time into vector registers (lane i = sub-block s0 + i, one
    // coalesced load per term every BLK items) and picked with v_readlane, so the item loop carries no scalar loads,
    // no 64-bit address arithmetic and no pending boundary load across its back edge.  Terms >= nt use the all-zero
    // row n_terms of sub_off: their segments have zero length.
    constexpr uint32_t BLK = 62;  // items per boundary block: item i needs boundaries i, i+1 and (for the prefetch) i+2
    const uint32_t* tptr[NT];
    const uint32_t* rowp[NT];
    float idf[NT];
    uint32_t av[NT];  // av: what a posting of the term does to its doc's match byte (0: nothing)
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const bool have = (uint32_t)t < nt;
      const uint32_t term = have ? Q->term[t] : n_terms;
      av[t] = (is_and && (uint32_t)t < np && Q->and_val[t]) ? (uint32_t)Q->and_val[t] | (nt_and & (BM_AND_FREQ | BM_AND_GATED | BM_AND_TOUCH)) : 0u;
      idf[t] = have ? ((uint32_t)t < np ? Q->idf[t] : BM_NOT_IDF) : 0.f;
      tptr[t] = post + term_base[term] * 4ull;
      rowp[t] = sub_off + (size_t)term * row_len;
    }
    const uint32_t s_begin = (uint32_t)(((u64)n_sub * part) / P);
    const uint32_t s_end = (uint32_t)(((u64)n_sub * (part + 1)) / P);

    BmTop<KPL> T;
#pragma unroll
    for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
    T.worst = 0ull;
    T.wsc = -1.0f;
    T.matched = 0;

    // RC loads per item; lanes (and whole chunks) past the segment end are out of range: zeros, no memory access
    auto issue_loads = [&](u32x4(&v)[RC], const uint32_t (&b0)[NT], const uint32_t (&b1)[NT]) {
#pragma unroll
      for (int t = 0; t < NT; t++) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[t], 0, (int)(b1[t] << 4), BM_RSRC_FLAGS);
#pragma unroll
        for (int c = 0; c < CPT; c++)
          v[t * CPT + c] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + c * 1024, (int)(b0[t] << 4), 0);
      }
    };

    u32x4 vA[RC], vB[RC];
    uint32_t B0[NT], B1[NT], B2[NT];
    uint32_t vbnd[NT];  // boundaries s0 .. s0+63 of term t, one per lane (indices past s_end clamp: empty items)

    uint32_t* tau_q = tau + (size_t)qi * BM_TAU_STRIDE;
    auto body = [&](u32x4(&cur)[RC], u32x4(&nxt)[RC], uint32_t s, uint32_t i) {
      // the query's shared threshold, refreshed every item (issued ahead of the posting loads: in-order return)
      const uint32_t tau_bits = __hip_atomic_load(tau_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int t = 0; t < NT; t++) B2[t] = __builtin_amdgcn_readlane(vbnd[t], i + 2);
      issue_loads(nxt, B1, B2);
      uint32_t maxn = 0;
#pragma unroll
      for (int t = 0; t < NT; t++) maxn = max(maxn, B1[t] - B0[t]);
      const uint32_t nlast = B1[NT - 1] - B0[NT - 1];
      if (!HAS_AND && nlast != 0 && maxn <= (uint32_t)CPT * 64u) {
        // Fused path (unions, last term present, no oversized segment).  The tile is all zero when an item starts:
        // the first term is a pure scatter, middle terms gather / add / scatter, and the LAST term is only gathered --
        // its new scores stay in registers until the trigger is known.  No trigger (the common case): the tile is never
        // read again, so every touched entry is zeroed through the addresses kept in registers (4-byte stores instead
        // of a 16 KB dense clear).  Trigger: the last term's scores are scattered and the tile is scanned.
        float mx = 0.f;
        uint32_t ao[(NT > 1 ? NT - 1 : 1) * CPT][4];
#pragma unroll
        for (int t = 0; t + 1 < NT; t++) {
          const uint32_t n16 = B1[t] - B0[t];
#pragma unroll
          for (int c = 0; c < CPT; c++)
            if ((uint32_t)c * 64u < n16) {
              if (t == 0) mx = bm_chunk_first(cur[t * CPT + c], idf[t], L, mx, ao[t * CPT + c]);
              else mx = bm_chunk_keep(cur[t * CPT + c], idf[t], L, mx, ao[t * CPT + c]);
            }
        }
        uint32_t aoL[CPT][4];
        float nwL[CPT][4];
#pragma unroll
        for (int c = 0; c < CPT; c++)
          if ((uint32_t)c * 64u < nlast) mx = bm_chunk_read(cur[(NT - 1) * CPT + c], idf[NT - 1], L, mx, aoL[c], nwL[c]);
        const float thr = fmaxf(T.wsc, __uint_as_float(tau_bits));
        if (count_mode || (k && __ballot(mx >= thr))) {
#pragma unroll
          for (int c = 0; c < CPT; c++)
            if ((uint32_t)c * 64u < nlast) {
#pragma unroll
              for (int x = 0; x < 4; x++) lds_stf(aoL[c][x], nwL[c][x]);
            }
          T = bm_scan_tile<HAS_AND, KPL>(T, L.tile, L.cntw, nt_and, s << BM_SUB_LOG2, count_mode, k, thr, tau_q, del, del_words);
        } else {
#pragma unroll
          for (int c = 0; c < CPT; c++)
            if ((uint32_t)c * 64u < nlast) {
#pragma unroll
              for (int x = 0; x < 4; x++) lds_stf(aoL[c][x], 0.f);
            }
#pragma unroll
          for (int t = 0; t + 1 < NT; t++) {
            const uint32_t n16 = B1[t] - B0[t];
#pragma unroll
            for (int c = 0; c < CPT; c++)
              if ((uint32_t)c * 64u < n16) {
#pragma unroll
                for (int x = 0; x < 4; x++) lds_stf(ao[t * CPT + c][x], 0.f);
              }
          }
        }
      } else if (maxn) {
        // general path: every term completely (prefetched chunks, then the rest of an oversized segment loaded
        // synchronously) before the next one, so that a doc's score is always summed in query-term order
        float mx = 0.f;  // running maximum of the scores written in this item
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const uint32_t n16 = B1[t] - B0[t];
#pragma unroll
          for (int c = 0; c < CPT; c++)
            if ((uint32_t)c * 64u < n16)
              mx = bm_chunk<HAS_AND>(cur[t * CPT + c], idf[t], L, av[t], mx);
          if (n16 > (uint32_t)CPT * 64u) {  // df above ~CPT/16 of the docs
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[t], 0, (int)(B1[t] << 4), BM_RSRC_FLAGS);
            for (uint32_t u = B0[t] + CPT * 64u; u < B1[t]; u += 64u) {
              const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (int)(u << 4), 0);
              mx = bm_chunk<HAS_AND>(q, idf[t], L, av[t], mx);
            }
          }
        }
        const float thr = fmaxf(T.wsc, __uint_as_float(tau_bits));
        if (count_mode || (k && __ballot(mx >= thr)))
          T = bm_scan_tile<HAS_AND, KPL>(T, L.tile, L.cntw, nt_and, s << BM_SUB_LOG2, count_mode, k, thr, tau_q, del, del_words);
        else
          bm_clear_tile<HAS_AND>(L, is_and, lane);
      }
#pragma unroll
      for (int t = 0; t < NT; t++) { B0[t] = B1[t]; B1[t] = B2[t]; }
    };

    for (uint32_t s0 = s_begin; s0 < s_end; s0 += BLK) {
      const uint32_t j = s0 + (uint32_t)lane;
#pragma unroll
      for (int t = 0; t < NT; t++) vbnd[t] = rowp[t][j < s_end ? j : s_end];
      // re-derived in EVERY block (same values as the rolled ones): the first use of the freshly loaded boundary
      // registers, and with it their vmcnt wait, sits here and not at the top of the item loop
#pragma unroll
      for (int t = 0; t < NT; t++) {
        B0[t] = __builtin_amdgcn_readlane(vbnd[t], 0);
        B1[t] = __builtin_amdgcn_readlane(vbnd[t], 1);
      }
      if (s0 == s_begin) issue_loads(vA, B0, B1);
      const uint32_t cnt = min(BLK, s_end - s0);  // BLK is even: every block starts on the vA buffer
      for (uint32_t i = 0; i < cnt; i += 2) {
        body(vA, vB, s0 + i, i);
        if (i + 1 < cnt) body(vB, vA, s0 + i + 1, i + 1);
      }
    }

    u64* out = part_keys + ((size_t)qi * P + part) * (64 * KPL);
#pragma unroll
    for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
    if (lane == 0 && T.matched) atomicAdd(&total[qi], T.matched);
  }
}

// Up to SS_MAX_QUERY_TERMS terms and k <= 1024: terms are processed in groups of four per item; the group's
// descriptors are rebuilt from the query each time (scalar loads), loads are not prefetched across items.
template <bool HAS_AND, int KPL>
__global__ void __launch_bounds__((HAS_AND ? BM_WAVES_AND : BM_WAVES_OR) * 64) bm25_scan_group_kernel(BM_KERNEL_ARGS) {
  constexpr int WAVES = HAS_AND ? BM_WAVES_AND : BM_WAVES_OR;
  constexpr int G = 4, CPT = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const BmLds L = bm_lds_setup<HAS_AND>(lane, w);
  const uint32_t total_waves = gridDim.x * WAVES;
  const uint32_t A = nq * P;
  const uint32_t row_len = n_sub + 1;
  const bool count_mode = count != 0;
  const int lane16 = lane * 16;

  for (uint32_t a = blockIdx.x * WAVES + w; a < A; a += total_waves) {
    const uint32_t qi = a % nq, part = a / nq;
    const bm_vquery* __restrict__ Q = qs + qi;
    const uint32_t np = Q->n_terms, nt = np + bm_q_nnot(Q->op);
    const uint32_t nt_and = HAS_AND ? Q->and_target : 0u;
    const bool is_and = nt_and != 0u;
    const uint32_t s_begin = (uint32_t)(((u64)n_sub * part) / P);
    const uint32_t s_end = (uint32_t)(((u64)n_sub * (part + 1)) / P);
    BmTop<KPL> T;
#pragma unroll
    for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
    T.worst = 0ull;
    T.wsc = -1.0f;
    T.matched = 0;

    for (uint32_t s = s_begin; s < s_end; s++) {
      float mx = 0.f;
      bool any = false;
      for (uint32_t g0 = 0; g0 < nt; g0 += G) {
        u32x4 v[G * CPT];
        uint32_t b0[G], b1[G];
        float idf[G];
        const uint32_t* tp[G];
        uint32_t av[G];
#pragma unroll
        for (int t = 0; t < G; t++) {
          const bool have = g0 + t < nt;
          const uint32_t term = have ? Q->term[have ? g0 + t : 0] : n_terms;
          av[t] = (is_and && g0 + t < np && Q->and_val[have ? g0 + t : 0]) ? (uint32_t)Q->and_val[have ? g0 + t : 0] | (nt_and & (BM_AND_FREQ | BM_AND_GATED | BM_AND_TOUCH)) : 0u;
          idf[t] = have ? (g0 + t < np ? Q->idf[have ? g0 + t : 0] : BM_NOT_IDF) : 0.f;
          tp[t] = post + term_base[term] * 4ull;
          b0[t] = sub_off[term * row_len + s];
          b1[t] = sub_off[term * row_len + s + 1];
          __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tp[t], 0, (int)(b1[t] << 4), BM_RSRC_FLAGS);
#pragma unroll
          for (int c = 0; c < CPT; c++)
            v[t * CPT + c] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + c * 1024, (int)(b0[t] << 4), 0);
        }
#pragma unroll
        for (int t = 0; t < G; t++) {
          const uint32_t n16 = b1[t] - b0[t];
          any |= n16 != 0;
#pragma unroll
          for (int c = 0; c < CPT; c++)
            if ((uint32_t)c * 64u < n16)
              mx = bm_chunk<HAS_AND>(v[t * CPT + c], idf[t], L, av[t], mx);
          if (n16 > (uint32_t)CPT * 64u) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tp[t], 0, (int)(b1[t] << 4), BM_RSRC_FLAGS);
            for (uint32_t u = b0[t] + CPT * 64u; u < b1[t]; u += 64u) {
              const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (int)(u << 4), 0);
              mx = bm_chunk<HAS_AND>(q, idf[t], L, av[t], mx);
            }
          }
        }
      }
      if (any) {
        if (count_mode || (k && __ballot(mx >= T.wsc)))
          T = bm_scan_tile<HAS_AND, KPL>(T, L.tile, L.cntw, nt_and, s << BM_SUB_LOG2, count_mode, k, T.wsc, nullptr, del, del_words);
        else
          bm_clear_tile<HAS_AND>(L, is_and, lane);
      }
    }
    u64* out = part_keys + ((size_t)qi * P + part) * (64 * KPL);
#pragma unroll
    for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
    if (lane == 0 && T.matched) atomicAdd(&total[qi], T.matched);
  }
}

#define BM_PASS_ARGS                                                                                                 \
  p.post, p.term_base, p.sub_off, p.q, p.part_keys, p.total, p.tau, p.del, p.del_words, p.n_sub, p.n_terms, p.nq, p.P, p.k, p.count

template <int NT, bool HAS_AND, int KPL>
static int launch_fast(const BmParams& p, hipStream_t st) {
  constexpr int WAVES = HAS_AND ? BM_WAVES_AND : BM_WAVES_OR;
  constexpr int lds = WAVES * (BM_WAVE_ACC + (HAS_AND ? BM_WAVE_CNT : 0));
  SS_SET_MAX_LDS((bm25_scan_fast_kernel<NT, HAS_AND, KPL>), lds);
  const uint32_t A = p.nq * p.P;
  bm25_scan_fast_kernel<NT, HAS_AND, KPL><<<(A + WAVES - 1) / WAVES, WAVES * 64, lds, st>>>(BM_PASS_ARGS);
  return SS_OK;
}

template <bool HAS_AND, int KPL>
static int launch_group(const BmParams& p, hipStream_t st) {
  constexpr int WAVES = HAS_AND ? BM_WAVES_AND : BM_WAVES_OR;
  constexpr int lds = WAVES * (BM_WAVE_ACC + (HAS_AND ? BM_WAVE_CNT : 0));
  SS_SET_MAX_LDS((bm25_scan_group_kernel<HAS_AND, KPL>), lds);
  const uint32_t A = p.nq * p.P;
  const uint32_t grid = std::min<uint32_t>((A + WAVES - 1) / WAVES, 256);
  bm25_scan_group_kernel<HAS_AND, KPL><<<grid, WAVES * 64, lds, st>>>(BM_PASS_ARGS);
  return SS_OK;
}

int ssi_bm25_launch_scan(const BmParams& p, uint32_t nt_max, bool has_and, int KPL, hipStream_t st) {
  if (nt_max >= 1 && nt_max <= 6 && (KPL == 1 || KPL == 2) && !(nt_max > 4 && KPL == 2)) {
    const int NT = nt_max <= 2 ? 2 : nt_max <= 4 ? (int)nt_max : 6;  // 5-6 lists (several fields): one instantiation
#define SS_F(NT_, AND_, KPL_) \
  if (NT == NT_ && has_and == AND_ && KPL == KPL_) return launch_fast<NT_, AND_, KPL_>(p, st);
    SS_F(2, false, 1) SS_F(3, false, 1) SS_F(4, false, 1) SS_F(2, true, 1) SS_F(3, true, 1) SS_F(4, true, 1)
    SS_F(6, false, 1) SS_F(6, true, 1)
    SS_F(2, false, 2) SS_F(3, false, 2) SS_F(4, false, 2) SS_F(2, true, 2) SS_F(3, true, 2) SS_F(4, true, 2)
#undef SS_F
  }
#define SS_G(AND_)                                           \
  switch (KPL) {                                             \
    case 1: return launch_group<AND_, 1>(p, st);             \
    case 2: return launch_group<AND_, 2>(p, st);             \
    case 4: return launch_group<AND_, 4>(p, st);             \
    default: return launch_group<AND_, 16>(p, st);           \
  }
  if (has_and) { SS_G(true) } else { SS_G(false) }
#undef SS_G
}
