// Device-side helpers shared by the BM25 scan kernels (generic and NT-specialised).  Internal.
#pragma once
#include "ss_common.h"

// LDS layout per workgroup: [per wave: BM_WAVE_ACC (+ BM_WAVE_CNT) bytes, ss_common.h] -- nothing else: a posting carries
// its weight, the kernels hold no table
constexpr int BM_RSRC_FLAGS = 0x00020000;  // raw buffer descriptor word 3: 32-bit data format
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BmParams {
  const uint32_t* post;
  const unsigned long long* term_base;
  const uint32_t* sub_off;
  const bm_vquery* q;              // expanded queries (bm25.hip bm_expand_kernel)
  unsigned long long* part_keys;   // [nq][P][KS]
  unsigned long long* total;       // [nq] exact match counts
  uint32_t* tau;                   // [nq] shared admission threshold: bits of the best k-th score any partition holds
  const uint32_t* del;             // tombstone bitmap (bit doc & 31 of word doc >> 5), null = no deleted docs
  uint32_t del_words;
  uint32_t n_sub, n_terms, nq, P, k, count;
};

typedef unsigned long long u64;

// tau[query]: every query's shared threshold on its own 128-byte line (device-scope atomics to one line serialise)
constexpr uint32_t BM_TAU_STRIDE = 32;
// publish a raised k-th best score; skipped when another partition already holds a higher one
__device__ __forceinline__ void bm_publish_tau(uint32_t* tau_q, float wsc) {
  const uint32_t bits = __float_as_uint(wsc);  // scores are positive floats: their bit patterns order like unsigned integers
  if (bits > __hip_atomic_load(tau_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(tau_q, bits);
}

__device__ __forceinline__ u64 shfl64(u64 v, int src) {
  uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shflx64(u64 v, int m) {
  uint32_t lo = __shfl_xor((uint32_t)v, m), hi = __shfl_xor((uint32_t)(v >> 32), m);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 rdlane64(u64 v, int l) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
// full bitonic sort of one key per lane, descending by lane index
__device__ __forceinline__ u64 wave_sort_desc(u64 x, int lane) {
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      u64 p = shflx64(x, j);
      bool up = (lane & k2) == 0;  // this block sorts descending
      bool lower = (lane & j) == 0;
      bool keep_max = (lower == up);
      x = keep_max ? (x > p ? x : p) : (x < p ? x : p);
    }
  }
  return x;
}
// sort a bitonic sequence descending
__device__ __forceinline__ u64 wave_bitonic_merge_desc(u64 x, int lane) {
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    u64 p = shflx64(x, j);
    bool lower = (lane & j) == 0;
    x = lower ? (x > p ? x : p) : (x < p ? x : p);
  }
  return x;
}

// Wave-resident sorted top-k: rank r*64+lane lives in keys[r] of `lane`; 0 = empty.  The list is only touched on
// the (rare) candidate path, which is kept out of line so the unrolled posting loops stay small.
template <int KPL>
__device__ __forceinline__ u64 topk_finish(u64 (&keys)[KPL], uint32_t k, int lane) {
#pragma unroll
  for (int r = 0; r < KPL; r++)
    if ((uint32_t)(r * 64 + lane) >= k) keys[r] = 0ull;
  const uint32_t kr = (k - 1) >> 6, kl = (k - 1) & 63;
  u64 w = 0ull;
#pragma unroll
  for (int r = 0; r < KPL; r++)
    if ((uint32_t)r == kr) w = rdlane64(keys[r], kl);
  return w;  // key at rank k-1 (0 while not full): admission threshold, strict '>'
}
// insert one (wave-uniform) key
template <int KPL>
__device__ __forceinline__ u64 topk_insert1(u64 (&keys)[KPL], u64 key, uint32_t k, int lane) {
  uint32_t pos = 0;
#pragma unroll
  for (int r = 0; r < KPL; r++) pos += __popcll(__ballot(keys[r] > key));
  u64 carry = key;
  bool active = false;
#pragma unroll
  for (int r = 0; r < KPL; r++) {
    if (!active && pos < (uint32_t)(64 * (r + 1))) {
      active = true;
      pos -= 64 * r;
    } else if (active) {
      pos = 0;
    }
    if (active) {
      u64 out = rdlane64(keys[r], 63);
      u64 up = shfl64(keys[r], lane > 0 ? lane - 1 : 0);
      keys[r] = (uint32_t)lane < pos ? keys[r] : ((uint32_t)lane == pos ? carry : up);
      carry = out;
    }
  }
  return topk_finish<KPL>(keys, k, lane);
}
// merge 64 new keys (one per lane, 0 = none)
template <int KPL>
__device__ __forceinline__ u64 topk_merge64(u64 (&keys)[KPL], u64 nk, uint32_t k, int lane) {
  u64 c = wave_sort_desc(nk, lane);
#pragma unroll
  for (int r = 0; r < KPL; r++) {
    u64 crev = shfl64(c, 63 - lane);
    u64 hi = keys[r] > crev ? keys[r] : crev;
    u64 lo = keys[r] > crev ? crev : keys[r];
    keys[r] = wave_bitonic_merge_desc(hi, lane);
    if (r + 1 < KPL) c = wave_bitonic_merge_desc(lo, lane);
  }
  return topk_finish<KPL>(keys, k, lane);
}
// offer up to 4 candidate keys per lane (0 = none); returns the new admission threshold
template <int KPL>
__device__ __forceinline__ u64 topk_offer(u64 (&keys)[KPL], u64 k0, u64 k1, u64 k2, u64 k3, u64 worst,
                                                    uint32_t k) {
  const int lane = __lane_id();
  for (;;) {
    u64 a = k0 > k1 ? k0 : k1, b = k2 > k3 ? k2 : k3;
    u64 mk = a > b ? a : b;  // this lane's best remaining candidate
    bool cand = mk > worst;
    u64 m = __ballot(cand);
    if (m == 0) break;
    if (__popcll(m) > 6) {
      worst = topk_merge64<KPL>(keys, cand ? mk : 0ull, k, lane);
    } else {
      while (m) {
        int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        u64 kk = rdlane64(mk, l);
        if (kk > worst) worst = topk_insert1<KPL>(keys, kk, k, lane);
      }
    }
    if (cand) {  // consumed (inserted or rejected against a threshold that only rises)
      if (k0 == mk) k0 = 0;
      else if (k1 == mk) k1 = 0;
      else if (k2 == mk) k2 = 0;
      else k3 = 0;
    }
  }
  return worst;
}


// ---------------------------------------------------------------- LDS access by byte offset
// (integer offsets instead of pointers: the compiler folds the constant part into the DS instruction's offset field
// and needs no address arithmetic on a generic pointer)
typedef __attribute__((address_space(3))) float bm_lds_f32;
typedef __attribute__((address_space(3))) uint32_t bm_lds_u32;
typedef __attribute__((address_space(3))) uint8_t bm_lds_u8;
typedef __attribute__((address_space(3))) f32x4 bm_lds_f32x4;
__device__ __forceinline__ float lds_ldf(uint32_t off) { return *(bm_lds_f32*)(uintptr_t)off; }
__device__ __forceinline__ void lds_stf(uint32_t off, float v) { *(bm_lds_f32*)(uintptr_t)off = v; }
__device__ __forceinline__ uint32_t lds_ld32(uint32_t off) { return *(bm_lds_u32*)(uintptr_t)off; }
__device__ __forceinline__ void lds_st32(uint32_t off, uint32_t v) { *(bm_lds_u32*)(uintptr_t)off = v; }
__device__ __forceinline__ uint32_t lds_ld8(uint32_t off) { return *(bm_lds_u8*)(uintptr_t)off; }
__device__ __forceinline__ void lds_st8(uint32_t off, uint32_t v) { *(bm_lds_u8*)(uintptr_t)off = (uint8_t)v; }
__device__ __forceinline__ f32x4 lds_ldf4(uint32_t off) { return *(bm_lds_f32x4*)(uintptr_t)off; }
__device__ __forceinline__ void lds_stf4(uint32_t off, f32x4 v) { *(bm_lds_f32x4*)(uintptr_t)off = v; }

// Per-wave LDS addresses (bytes): accb + 4*field (field 0 = dump), tile = accb + 4 (16-byte aligned), match counters
// cnt + field (AND only).
struct BmLds {
  uint32_t accb, tile, cnt, cntw;
};
// accumulator address / weight of a posting (2 VALU operations each)
__device__ __forceinline__ uint32_t bm_acc_addr(uint32_t p, const BmLds& L) { return (bm_doc_field(p) << 2) + L.accb; }

// One 256-posting chunk (4 per lane) of ONE term: acc[doc] += idf * weight   (add_result.rs:1445-1447).
// Plain gather / scatter: the docs of one term are distinct and the tile is private to the wave, whose LDS operations
// execute in order.  NULL postings (padding, out-of-range lanes) touch only the dump slot.  Returns max of the new scores.
template <bool HAS_AND>
__device__ __forceinline__ float bm_chunk(const u32x4 q, float idf, const BmLds& L, uint32_t and_val, float mx) {
  const uint32_t pv[4] = {q.x, q.y, q.z, q.w};
  uint32_t ao[4], co[4], cold[4];
  float old[4];
#pragma unroll
  for (int x = 0; x < 4; x++) {
    ao[x] = bm_acc_addr(pv[x], L);
    old[x] = lds_ldf(ao[x]);
    if (HAS_AND && and_val) {
      co[x] = bm_doc_field(pv[x]) + L.cnt;
      cold[x] = lds_ld8(co[x]);
    }
  }
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const bool gated_list = HAS_AND && (and_val & BM_AND_GATED) && (and_val & 0x80u);  // an unlisted field of a filtered union's term
    float nw = __builtin_fmaf(idf, bm_weight(pv[x]), old[x]);                           // (one fma: the chain every kernel sums a score with)
    if (gated_list && !(cold[x] & (and_val & 0x7Fu))) nw = old[x];                      // ... adds only where the term passed
    lds_stf(ao[x], nw);
    mx = fmaxf(mx, pv[x] ? nw : 0.f);  // a NULL posting's decoded weight is not zero: keep the dump slot out of the maximum
    if (HAS_AND && and_val) {
      // count one more | set the term's bit (+ bit 7 under the all_terms_frequent shortcut when this posting's tf < 10)
      uint32_t nb = and_val == 0xFFu ? cold[x] + 1u : (cold[x] | (and_val & 0xFFu));
      if (and_val & BM_AND_FREQ) nb |= bm_tf_lt10(pv[x]) ? 0x80u : 0u;
      if (and_val & BM_AND_GATED) nb = cold[x] | (gated_list ? 0u : (and_val & 0x7Fu)) | ((and_val & BM_AND_TOUCH) ? 0x80u : 0u);
      lds_st8(co[x], nb);
    }
  }
  return mx;
}

// ---- chunk flavours of the fused item path (unions; bm25_fast.hip):
//   first : the tile is still all zero for this term's docs -> score = idf * w, scattered without a read
//   keep  : gather / add / scatter, addresses handed back
//   read  : gather / add only, the new scores stay in registers (last term: scattered or zeroed once the trigger is known)
// The dump slot (NULL postings) takes part like any accumulator and is zeroed with the rest, so what it holds is bounded by
// NT * idf * 2^-14 per item -- far below any real score; it is never scanned.
__device__ __forceinline__ float bm_chunk_first(const u32x4 q, float idf, const BmLds& L, float mx, uint32_t (&ao)[4]) {
  const uint32_t pv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int x = 0; x < 4; x++) {
    ao[x] = bm_acc_addr(pv[x], L);
    const float nw = idf * bm_weight(pv[x]);
    lds_stf(ao[x], nw);
    mx = fmaxf(mx, nw);
  }
  return mx;
}
__device__ __forceinline__ float bm_chunk_read(const u32x4 q, float idf, const BmLds& L, float mx, uint32_t (&ao)[4],
                                               float (&nw)[4]) {
  const uint32_t pv[4] = {q.x, q.y, q.z, q.w};
  float old[4];
#pragma unroll
  for (int x = 0; x < 4; x++) {
    ao[x] = bm_acc_addr(pv[x], L);
    old[x] = lds_ldf(ao[x]);
  }
#pragma unroll
  for (int x = 0; x < 4; x++) {
    nw[x] = old[x] + idf * bm_weight(pv[x]);
    mx = fmaxf(mx, nw[x]);
  }
  return mx;
}
__device__ __forceinline__ float bm_chunk_keep(const u32x4 q, float idf, const BmLds& L, float mx, uint32_t (&ao)[4]) {
  float nw[4];
  mx = bm_chunk_read(q, idf, L, mx, ao, nw);
#pragma unroll
  for (int x = 0; x < 4; x++) lds_stf(ao[x], nw[x]);
  return mx;
}

template <bool HAS_AND>
__device__ __forceinline__ void bm_clear_tile(const BmLds& L, bool is_and, int lane) {
  if (lane == 0) lds_stf(L.accb, 0.f);  // the dump slot: NULL postings decode to a tiny non-zero weight
#pragma unroll
  for (int i = 0; i < BM_SUB / 256; i++) {
    lds_stf4(L.tile + (i * 64 + lane) * 16, f32x4{0.f, 0.f, 0.f, 0.f});
    if (HAS_AND && is_and) lds_st32(L.cntw + (i * 64 + lane) * 4, 0u);
  }
}

// Wave-resident result state of one (query, partition) assignment
template <int KPL>
struct BmTop {
  u64 keys[KPL];   // sorted top-k, rank r*64+lane in keys[r]
  u64 worst;       // key at rank k-1 (0 while the list is not full)
  float wsc;       // its score (-1 while not full): the trigger threshold of the item loop
  u64 matched;     // exact match count (Count / TopkCount)
};

// Dense scan of the tile (only when some doc may enter the list, or exact counts are wanted): clears it, counts
// matches (union: any term, intersection: all terms) and offers survivors to the wave-resident top-k.  Kept OUT OF
// LINE: its ballots / lane loops need many scalar registers, and inlined into the item loop they push the loop's own
// scalar state (descriptors, boundaries) into spill lanes.  State crosses the call in registers (by value).
template <bool HAS_AND, int KPL>
__device__ __attribute__((noinline)) BmTop<KPL> bm_scan_tile(BmTop<KPL> T, uint32_t tile, uint32_t cntw, uint32_t nt_and,
                                                             uint32_t doc_base, uint32_t count_mode, uint32_t k, float thr,
                                                             uint32_t* tau_q, const uint32_t* __restrict__ del,
                                                             uint32_t del_words) {
  // thr = max(own k-th best score, the k-th best score some other partition of the query already holds): a doc
  // below it cannot be in the query's top-k.  Equal scores stay admissible (the final merge breaks ties by doc id).
  const float wsc_in = T.wsc;
  const int lane = __lane_id();
  const bool is_and = HAS_AND && nt_and != 0;  // nt_and = number of terms of an intersection, 0 for a union
  // all_terms_frequent shortcut: bit 7 of a match byte = "some term has tf < 10 here" -- the doc matches (is counted) on
  // its low 7 bits and is ranked only with bit 7 clear (decode_positions_multiterm_singlefield returns true -> counted,
  // never scored: add_result.rs:2091-2104, 3541-3556)
  const uint32_t cmask = (HAS_AND && (nt_and & BM_AND_FREQ)) ? 0x7Fu : 0xFFu;
  const bool gated = HAS_AND && (nt_and & BM_AND_GATED), touch = HAS_AND && (nt_and & BM_AND_TOUCH);  // a union under a field filter
  nt_and &= 0xFFu;
  // tombstones of this sub-block (128 words): lane l keeps words l and 64 + l; iteration i needs word 8 i + lane / 8.
  // A deleted doc neither counts nor ranks (add_result.rs:3435, union.rs:975).
  uint32_t dw0 = 0u, dw1 = 0u;
  if (del) {
    const uint32_t wb = doc_base >> 5;
    if (wb + (uint32_t)lane < del_words) dw0 = del[wb + lane];
    if (wb + 64u + (uint32_t)lane < del_words) dw1 = del[wb + 64u + lane];
  }
  const bool any_del = del && __ballot((dw0 | dw1) != 0u);
  if (lane == 0) lds_stf(tile - 4u, 0.f);  // the dump slot in front of the tile (NULL postings decode to a tiny non-zero weight)
#pragma unroll 2
  for (int i = 0; i < BM_SUB / 256; i++) {
    const int slot = i * 64 + lane;
    f32x4 x = lds_ldf4(tile + slot * 16);
    lds_stf4(tile + slot * 16, f32x4{0.f, 0.f, 0.f, 0.f});
    uint32_t nib = 0u;  // tombstone bits of this lane's four docs
    if (any_del) {
      const uint32_t wsel = __shfl(i < 8 ? dw0 : dw1, (i * 8 + (lane >> 3)) & 63);
      nib = (wsel >> ((lane & 7) * 4)) & 15u;
      if (nib & 1u) x.x = 0.f;
      if (nib & 2u) x.y = 0.f;
      if (nib & 4u) x.z = 0.f;
      if (nib & 8u) x.w = 0.f;
    }
    bool h0 = x.x > 0.f, h1 = x.y > 0.f, h2 = x.z > 0.f, h3 = x.w > 0.f;  // a doc of a NOT list has a negative score
    bool c0 = false, c1 = false, c2 = false, c3 = false;  // gated unions of > 2 terms: what is COUNTED (the unfiltered union)
    if (HAS_AND && is_and) {
      const uint32_t cw = lds_ld32(cntw + slot * 4);
      lds_st32(cntw + slot * 4, 0u);
      if (gated) {
        // h stays "score > 0": a doc none of whose terms occurs in a listed field has accumulated nothing.  touch: a doc any list
        // of the query holds counts unless it is deleted (score zeroed above) or in a NOT list (negative score)
        c0 = (cw & 0x80u) && !(nib & 1u) && !(x.x < 0.f);
        c1 = ((cw >> 8) & 0x80u) && !(nib & 2u) && !(x.y < 0.f);
        c2 = ((cw >> 16) & 0x80u) && !(nib & 4u) && !(x.z < 0.f);
        c3 = ((cw >> 24) & 0x80u) && !(nib & 8u) && !(x.w < 0.f);
      } else {
        h0 = h0 && (cw & cmask) == nt_and;  // h: score > 0 (not deleted, in no NOT list) and every term present
        h1 = h1 && ((cw >> 8) & cmask) == nt_and;
        h2 = h2 && ((cw >> 16) & cmask) == nt_and;
        h3 = h3 && ((cw >> 24) & cmask) == nt_and;
        if (!h0 || (cw & 0x80u & ~cmask)) x.x = 0.f;  // counted below, ranked only without the tf < 10 mark
        if (!h1 || ((cw >> 8) & 0x80u & ~cmask)) x.y = 0.f;
        if (!h2 || ((cw >> 16) & 0x80u & ~cmask)) x.z = 0.f;
        if (!h3 || ((cw >> 24) & 0x80u & ~cmask)) x.w = 0.f;
      }
    }
    if (count_mode && touch)
      T.matched += __popcll(__ballot(c0)) + __popcll(__ballot(c1)) + __popcll(__ballot(c2)) + __popcll(__ballot(c3));
    else if (count_mode)
      T.matched += __popcll(__ballot(h0)) + __popcll(__ballot(h1)) + __popcll(__ballot(h2)) + __popcll(__ballot(h3));
    if (k) {
      const float m = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
      if (__ballot(m > 0.f && m >= thr)) {
        const uint32_t d0 = doc_base + slot * 4;
        u64 k0 = ((u64)__float_as_uint(x.x) << 32) | (u64)(0xFFFFFFFFu - d0);
        u64 k1 = ((u64)__float_as_uint(x.y) << 32) | (u64)(0xFFFFFFFFu - (d0 + 1));
        u64 k2 = ((u64)__float_as_uint(x.z) << 32) | (u64)(0xFFFFFFFFu - (d0 + 2));
        u64 k3 = ((u64)__float_as_uint(x.w) << 32) | (u64)(0xFFFFFFFFu - (d0 + 3));
        k0 = (x.x > 0.f && x.x >= thr && k0 > T.worst) ? k0 : 0ull;
        k1 = (x.y > 0.f && x.y >= thr && k1 > T.worst) ? k1 : 0ull;
        k2 = (x.z > 0.f && x.z >= thr && k2 > T.worst) ? k2 : 0ull;
        k3 = (x.w > 0.f && x.w >= thr && k3 > T.worst) ? k3 : 0ull;
        if (__ballot((k0 | k1 | k2 | k3) != 0ull)) {
          T.worst = topk_offer<KPL>(T.keys, k0, k1, k2, k3, T.worst, k);
          if (T.worst) {
            T.wsc = __uint_as_float((uint32_t)(T.worst >> 32));
            thr = fmaxf(thr, T.wsc);
          }
        }
      }
    }
  }
  // publish a raised k-th best score (scores are positive floats: their bit patterns order like unsigned integers)
  if (tau_q && T.wsc > wsc_in && lane == 0) bm_publish_tau(tau_q, T.wsc);
  return T;
}

// one candidate key per lane (0 = none) offered to the wave-resident top-k; publishes a raised k-th best score
template <int KPL>
__device__ __attribute__((noinline)) BmTop<KPL> bm_offer_lane_keys(BmTop<KPL> T, u64 key, uint32_t k, uint32_t* tau_q) {
  const float wsc_in = T.wsc;
  T.worst = topk_offer<KPL>(T.keys, key, 0ull, 0ull, 0ull, T.worst, k);
  if (T.worst) T.wsc = __uint_as_float((uint32_t)(T.worst >> 32));
  if (tau_q && T.wsc > wsc_in && __lane_id() == 0) bm_publish_tau(tau_q, T.wsc);
  return T;
}


// partition lists -> the callers' outputs (bm25.hip)
int ssi_bm25_merge_lists(u64* bufA, u64* bufB, uint32_t nq, uint32_t P, uint32_t KS, uint32_t k, const u64* total, const uint32_t* tau,
                         uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, hipStream_t st);
// scan kernels (bm25_fast.hip): NT-specialised for <= 4 terms and k <= 128, grouped generic kernel otherwise
int ssi_bm25_launch_scan(const BmParams& p, uint32_t nt_max, bool has_and, int KPL, hipStream_t st);
// 16-bit-accumulator scan (bm25_scan16.hip): unions of <= 6 lists / intersections of 2-3, k <= 64, NOT lists, tombstones, exact counts; 16 waves per CU
bool ssi_bm25_scan16_serves(uint32_t nn_max, uint32_t np_max, bool has_and, bool count, bool tombstones, int KPL, uint32_t k, uint32_t and_exact_nt);
int ssi_bm25_launch_scan16(const BmParams& p, uint32_t np_max, uint32_t nn_max, bool is_and, int KPL, hipStream_t st);
// exact union counts from the probe index's bit records (bm25_probe.hip)
int ssi_bm25_launch_union_count(const BmParams& p, const uint2* probe, const uint32_t* probe_row, bool all_queries, hipStream_t st,
                                unsigned long long* match_bits = nullptr);
// pruned top-k over the probe index (bm25_probe.hip); SS_ENOTSUP if it cannot serve the request
// phrase queries (bm25_phrase.hip): intersection over the probe index + position check
int ssi_bm25_launch_phrase(const BmParams& p, const uint2* probe, const uint32_t* probe_z, const uint32_t* probe_row, const uint16_t* pos16,
                           const uint32_t* pos32, const uint32_t* pos_off, const unsigned long long* pos_base, uint32_t nt_max, int KPL,
                           hipStream_t st);
int ssi_bm25_launch_probe(const BmParams& p, const uint2* probe, const uint32_t* probe_z, const uint32_t* probe_row, const float* umax,
                          const float* submax, float* pmax_ws, uint32_t nt_max, int KPL, bool any_not, hipStream_t st);
