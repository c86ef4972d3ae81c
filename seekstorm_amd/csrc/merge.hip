// Cross-shard top-k merge on device: the gather + sort + truncate of search.rs:1875-1940 / 2098-2119 for a batch of
// queries whose per-shard top-k lists were all-gathered over RCCL.  One workgroup per query: S*k (score, local id)
// pairs -> keys (score desc, concatenation order on ties = the reference's stable sort) -> bitonic sort in LDS ->
// global ids local*S + shard (search.rs:1671).
#include "ss_common.h"

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t mg_f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// stride_ds / stride_c: elements between two shards' blocks in doc / score and in cnt (separate gathered arrays: nq * k and
// nq; one packed gather [S][doc | score | cnt]: (2 k + 1) nq for all three)
__global__ void __launch_bounds__(256) topk_merge_kernel(uint32_t nq, uint32_t S, uint32_t k, const uint32_t* __restrict__ doc,
                                                        const float* __restrict__ score, const uint32_t* __restrict__ cnt,
                                                        size_t stride_ds, size_t stride_c,
                                                        u64* __restrict__ out_doc, float* __restrict__ out_score,
                                                        uint32_t* __restrict__ out_cnt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* keys = (u64*)smem;
  const uint32_t q = blockIdx.x;
  const uint32_t n = S * k;
  uint32_t np = 64;
  while (np < n) np <<= 1;
  for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
    u64 key = 0ull;
    if (i < n) {
      uint32_t s = i / k, r = i % k;
      uint32_t c = cnt[(size_t)s * stride_c + q];
      if (c == 0xFFFFFFFFu) c = 0;
      if (r < c) key = ((u64)mg_f2ord(score[(size_t)s * stride_ds + (size_t)q * k + r]) << 32) | (u64)(0xFFFFFFFFu - i);
    }
    keys[i] = key;
  }
  __syncthreads();
  for (uint32_t size = 2; size <= np; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = threadIdx.x; i < (np >> 1); i += blockDim.x) {
        uint32_t lo = 2 * i - (i & (stride - 1));
        uint32_t hi = lo + stride;
        bool desc = ((lo & size) == 0);
        u64 a = keys[lo], b = keys[hi];
        if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  __shared__ uint32_t total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  uint32_t local = 0;
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    u64 key = keys[i];
    u64 gd = ~0ull;
    float sc = 0.f;
    if (key) {
      uint32_t pos = 0xFFFFFFFFu - (uint32_t)key;
      uint32_t s = pos / k, r = pos % k;
      size_t at = (size_t)s * stride_ds + (size_t)q * k + r;
      gd = (u64)doc[at] * S + s;  // search.rs:1671
      sc = score[at];
      local++;
    }
    out_doc[(size_t)q * k + i] = gd;
    out_score[(size_t)q * k + i] = sc;
  }
  if (local) atomicAdd(&total, local);
  __syncthreads();
  if (threadIdx.x == 0) out_cnt[q] = total;
}

extern "C" int ss_topk_merge_dev(int device, uint32_t n_queries, uint32_t n_shards, uint32_t k, const uint32_t* d_doc,
                                 const float* d_score, const uint32_t* d_count, uint64_t* d_out_doc, float* d_out_score,
                                 uint32_t* d_out_count, void* stream) {
  if (!d_doc || !d_score || !d_count || !d_out_doc || !d_out_score || !d_out_count) return SS_EINVAL;
  if (n_shards == 0 || k == 0 || (uint64_t)n_shards * k > 8192) return SS_EINVAL;
  if (n_queries == 0) return SS_OK;
  SS_HIP(hipSetDevice(device));
  uint32_t np = 64;
  while (np < n_shards * k) np <<= 1;
  SS_SET_MAX_LDS(topk_merge_kernel, 8192 * 8);
  topk_merge_kernel<<<n_queries, 256, np * sizeof(u64), (hipStream_t)stream>>>(n_queries, n_shards, k, d_doc, d_score, d_count,
                                                                              (size_t)n_queries * k, (size_t)n_queries,
                                                                              (u64*)d_out_doc, d_out_score, d_out_count);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// The same over ONE gathered buffer: every shard contributes [nq * k doc ids | nq * k score bits | nq counts] (32-bit words),
// so that the multi-GPU path needs a single all-gather per batch instead of three.
extern "C" int ss_topk_merge_dev_packed(int device, uint32_t n_queries, uint32_t n_shards, uint32_t k, const uint32_t* d_packed,
                                        uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, void* stream) {
  if (!d_packed || !d_out_doc || !d_out_score || !d_out_count) return SS_EINVAL;
  if (n_shards == 0 || k == 0 || (uint64_t)n_shards * k > 8192) return SS_EINVAL;
  if (n_queries == 0) return SS_OK;
  SS_HIP(hipSetDevice(device));
  uint32_t np = 64;
  while (np < n_shards * k) np <<= 1;
  SS_SET_MAX_LDS(topk_merge_kernel, 8192 * 8);
  const size_t nk = (size_t)n_queries * k, stride = 2 * nk + n_queries;
  topk_merge_kernel<<<n_queries, 256, np * sizeof(u64), (hipStream_t)stream>>>(n_queries, n_shards, k, d_packed, (const float*)(d_packed + nk),
                                                                              d_packed + 2 * nk, stride, stride, (u64*)d_out_doc,
                                                                              d_out_score, d_out_count);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
