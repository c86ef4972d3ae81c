// Device-side pieces shared by the f32 (vec_scan.hip) and i8 (vec8_scan.hip) vector scans: per-batch state, the
// order-preserving score key, and the epilogue that appends the rare rows beating the running threshold.
#pragma once
#include "ss_common.h"

struct VState {
  float tau[64];
  uint32_t cnt[64];
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

