// Device-side pieces shared by the f32 (vec_scan.hip) and i8 (vec8_scan.hip) vector scans: per-batch state, the
// order-preserving score key, and the epilogue that appends the rare rows beating the running threshold.
#pragma once
#include "ss_common.h"

// cnt: one counter per query, each on its own 128-byte line -- the scan kernels' appends are device-scope atomics, and
// atomics to one line serialise: with the 64 counters packed in two lines a launch with a few hundred appends per query
// spent ~150 us on them (i8 scan, 10M rows), independent of the grid.
constexpr int VS_CNT_STRIDE = 32;
struct VState {
  float tau[64];
  uint32_t cnt[64 * VS_CNT_STRIDE];
  uint32_t kept[64];
  uint32_t ovf;
  uint32_t pad[63];
  unsigned long long total[64];
};

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}
// larger key = better: (score desc, row asc)
__device__ __forceinline__ unsigned long long mk_key(float s, uint32_t row) {
  return ((unsigned long long)f2ord(s) << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}


// Appends the scores of one lane's 16 rows that beat tau to query q's candidate buffer: ONE atomic per lane and tile
// (the first launches, where every row is a candidate, would otherwise serialise 16 x as many on 64 counters).
__device__ __forceinline__ void vs_append(const float (&f)[16], float tau, uint32_t q, unsigned long long row_base,
                                          unsigned long long n_rows, VState* __restrict__ st, unsigned long long* __restrict__ cand) {
  uint32_t n = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) n += (f[r] > tau && row_base + (r & 3) + 8 * (r >> 2) < n_rows) ? 1u : 0u;
  if (n == 0) return;
  uint32_t slot = atomicAdd(&st->cnt[q * VS_CNT_STRIDE], n);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const unsigned long long row = row_base + (r & 3) + 8 * (r >> 2);
    if (f[r] > tau && row < n_rows) {
      if (slot < VS_CAP) cand[(size_t)q * VS_CAP + slot] = mk_key(f[r], (uint32_t)row);
      slot++;
    }
  }
}
