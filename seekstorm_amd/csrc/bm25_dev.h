// Device-side helpers shared by the BM25 scan kernels (generic and NT-specialised).  Internal.
#pragma once
#include "ss_common.h"

constexpr int BM_RC = 12;            // posting chunks (256 postings each) in flight per wave and round
constexpr float BM_K1P = 2.2f;       // K + 1.0 (add_result.rs:20)
// LDS layout per workgroup: [comp 256 f32][wlut 4096 f32][per wave: acc 4096 f32 (+64 dump slots) (+ 4096 match-count bytes)]
constexpr int BM_LUT_BYTES = (256 + 4096) * 4;

struct BmParams {
  const uint32_t* post;
  const unsigned long long* term_base;
  const uint32_t* sub_off;
  const float* comp;
  const ss_bm25_query* q;
  unsigned long long* part_keys;   // [nq][P][KS]
  unsigned long long* total;       // [nq] exact match counts
  uint32_t n_sub, n_terms, nq, P, k, count;
};

typedef unsigned long long u64;

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
__device__ __attribute__((noinline)) u64 topk_offer(u64 (&keys)[KPL], u64 k0, u64 k1, u64 k2, u64 k3, u64 worst,
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


// NT-specialised scan kernels (bm25_fast.hip); returns SS_ENOTSUP if there is no instantiation for (nt_max, KPL)
int ssi_bm25_launch_fast(const BmParams& p, uint32_t nt_max, bool has_and, int KPL, hipStream_t st);
