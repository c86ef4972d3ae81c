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
                                                        size_t stride_ds, size_t stride_c, uint32_t out_len,
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
  for (uint32_t i = threadIdx.x; i < out_len; i += blockDim.x) {  // out_len = k: the merged top-k; S * k: the whole concatenation, sorted
    u64 key = i < np ? keys[i] : 0ull;
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
    out_doc[(size_t)q * out_len + i] = gd;
    out_score[(size_t)q * out_len + i] = sc;
  }
  if (local) atomicAdd(&total, local);
  __syncthreads();
  if (threadIdx.x == 0) out_cnt[q] = total;
}

// The same merge for ANY n_shards * k (more than the 8192 keys the kernel above sorts in LDS: deep pages, k > SS_MAX_K, or more than 8 shards at
// k = 1024): no sort at all -- the lists arrive sorted, so every entry FINDS its rank in the merged order: its rank in its own list + for
// every other list the number of entries that precede it there (a binary search; equal scores: concatenation order, exactly the order
// of the keys above) -- and writes itself to that slot if it lies inside out_len.  One thread per entry; the slots no entry claims were
// filled with (UINT64_MAX, 0) before.
__global__ void __launch_bounds__(256) topk_merge_rank_kernel(uint32_t nq, uint32_t S, uint32_t k, const uint32_t* __restrict__ doc,
                                                             const float* __restrict__ score, const uint32_t* __restrict__ cnt,
                                                             size_t stride_ds, size_t stride_c, uint32_t out_len,
                                                             u64* __restrict__ out_doc, float* __restrict__ out_score,
                                                             uint32_t* __restrict__ out_cnt) {
  const uint32_t q = blockIdx.y;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto count_of = [&](uint32_t s) -> uint32_t {
    uint32_t c = cnt[(size_t)s * stride_c + q];
    if (c == 0xFFFFFFFFu) c = 0;
    return c < k ? c : k;
  };
  if (e == 0) {
    u64 total = 0;
    for (uint32_t s = 0; s < S; s++) total += count_of(s);
    out_cnt[q] = (uint32_t)(total < out_len ? total : out_len);
  }
  if (e >= (size_t)S * k) return;
  const uint32_t s = (uint32_t)(e / k), r = (uint32_t)(e % k);
  if (r >= count_of(s)) return;
  const size_t at = (size_t)s * stride_ds + (size_t)q * k + r;
  const uint32_t mine = mg_f2ord(score[at]);
  size_t rank = r;
  for (uint32_t o = 0; o < S; o++) {
    if (o == s) continue;
    const float* __restrict__ ls = score + (size_t)o * stride_ds + (size_t)q * k;
    // entries of list o that come before this one: score above it, or equal in an EARLIER list
    uint32_t lo = 0, hi = count_of(o);
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      const uint32_t x = mg_f2ord(ls[mid]);
      const bool before = o < s ? x >= mine : x > mine;
      if (before) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  if (rank < out_len) {
    out_doc[(size_t)q * out_len + rank] = (u64)doc[at] * S + s;  // search.rs:1671
    out_score[(size_t)q * out_len + rank] = score[at];
  }
}

// one launcher for both: the LDS sort up to 8192 keys, the rank merge beyond
static int topk_merge_any(int device, uint32_t nq, uint32_t S, uint32_t k, const uint32_t* d_doc, const float* d_score, const uint32_t* d_count,
                          size_t stride_ds, size_t stride_c, uint32_t out_len, uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count,
                          hipStream_t st) {
  if (S == 0 || k == 0 || out_len == 0 || (uint64_t)out_len > (uint64_t)S * k || (uint64_t)S * k > 0xFFFFFFFFull) return SS_EINVAL;
  if (nq == 0) return SS_OK;
  SS_HIP(hipSetDevice(device));
  if ((uint64_t)S * k <= 8192) {
    uint32_t np = 64;
    while (np < S * k) np <<= 1;
    SS_SET_MAX_LDS(topk_merge_kernel, 8192 * 8);
    topk_merge_kernel<<<nq, 256, np * sizeof(u64), st>>>(nq, S, k, d_doc, d_score, d_count, stride_ds, stride_c, out_len, (u64*)d_out_doc, d_out_score,
                                                        d_out_count);
  } else {
    SS_HIP(hipMemsetAsync(d_out_doc, 0xFF, (size_t)nq * out_len * sizeof(u64), st));
    SS_HIP(hipMemsetAsync(d_out_score, 0, (size_t)nq * out_len * sizeof(float), st));
    const dim3 grid((uint32_t)(((size_t)S * k + 255) / 256), nq);
    topk_merge_rank_kernel<<<grid, 256, 0, st>>>(nq, S, k, d_doc, d_score, d_count, stride_ds, stride_c, out_len, (u64*)d_out_doc, d_out_score, d_out_count);
  }
  SS_HIP(hipGetLastError());
  return SS_OK;
}

extern "C" int ss_topk_merge_dev(int device, uint32_t n_queries, uint32_t n_shards, uint32_t k, const uint32_t* d_doc,
                                 const float* d_score, const uint32_t* d_count, uint64_t* d_out_doc, float* d_out_score,
                                 uint32_t* d_out_count, void* stream) {
  if (!d_doc || !d_score || !d_count || !d_out_doc || !d_out_score || !d_out_count) return SS_EINVAL;
  return topk_merge_any(device, n_queries, n_shards, k, d_doc, d_score, d_count, (size_t)n_queries * k, (size_t)n_queries, k, d_out_doc, d_out_score,
                        d_out_count, (hipStream_t)stream);
}

// The same over ONE gathered buffer: every shard contributes [nq * k doc ids | nq * k score bits | nq counts] (32-bit words),
// so that the multi-GPU path needs a single all-gather per batch instead of three.
extern "C" int ss_topk_merge_dev_packed(int device, uint32_t n_queries, uint32_t n_shards, uint32_t k, const uint32_t* d_packed,
                                        uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, void* stream) {
  if (!d_packed || !d_out_doc || !d_out_score || !d_out_count) return SS_EINVAL;
  const size_t nk = (size_t)n_queries * k, stride = 2 * nk + n_queries;
  return topk_merge_any(device, n_queries, n_shards, k, d_packed, (const float*)(d_packed + nk), d_packed + 2 * nk, stride, stride, k, d_out_doc,
                        d_out_score, d_out_count, (hipStream_t)stream);
}

// the same kernel for the sharded searches (comm.hip): lists anywhere inside the gathered buffer, out_len = k or S * k entries
int ssi_topk_merge_launch(int device, uint32_t nq, uint32_t S, uint32_t k, const uint32_t* d_doc, const float* d_score, const uint32_t* d_count,
                          size_t stride_ds, size_t stride_c, uint32_t out_len, uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count,
                          hipStream_t st) {
  return topk_merge_any(device, nq, S, k, d_doc, d_score, d_count, stride_ds, stride_c, out_len, d_out_doc, d_out_score, d_out_count, st);
}

// the WHOLE concatenation, sorted: [n_queries][n_shards * k] -- what the RRF ranks of a hybrid search over several shards run
// over (search.rs:1962-2035 sorts the appended per-shard lists, untruncated); feeds ss_rrf_merge_dev with k_lex = n_shards * k
extern "C" int ss_topk_concat_dev_packed(int device, uint32_t n_queries, uint32_t n_shards, uint32_t k, const uint32_t* d_packed,
                                         uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, void* stream) {
  if (!d_packed || !d_out_doc || !d_out_score || !d_out_count) return SS_EINVAL;
  const size_t nk = (size_t)n_queries * k, stride = 2 * nk + n_queries;
  return ssi_topk_merge_launch(device, n_queries, n_shards, k, d_packed, (const float*)(d_packed + nk), d_packed + 2 * nk, stride, stride,
                               n_shards * k, d_out_doc, d_out_score, d_out_count, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Hybrid fusion on device: reciprocal rank fusion of a batch's lexical and vector result lists (search.rs:1962-2035) and
// the final sort / offset / length (2098-2119), so that a batched hybrid search never leaves the GPU between the two
// shard searches and its answer.  score(d) = sum over the lists holding d of 1 / (0.6 + rank), rank 0-based in the list;
// a doc of both lists is `Hybrid`, of one list keeps that list's source.  Equal fused scores: doc id ascending (the
// reference leaves them in hash order), as ss_merge_results does.  One workgroup per query; the union of the two lists
// (<= 8192 entries) lives in LDS: match the vector entries against the lexical ones, bitonic sort by (score desc, doc asc).
// Both lists must be sorted by score descending with unique doc ids -- what the searches and merges above produce.
__global__ void __launch_bounds__(256) rrf_merge_kernel(uint32_t k_lex, uint32_t k_vec, const void* __restrict__ lex_doc,
                                                       const uint32_t* __restrict__ lex_cnt, const void* __restrict__ vec_doc,
                                                       const uint32_t* __restrict__ vec_cnt, int wide_ids, uint32_t offset,
                                                       uint32_t length, u64* __restrict__ out_doc, float* __restrict__ out_score,
                                                       uint8_t* __restrict__ out_src, uint32_t* __restrict__ out_cnt, uint32_t np) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* docs = (u64*)smem;
  float* sc = (float*)(docs + np);
  uint8_t* src = (uint8_t*)(sc + np);
  const uint32_t q = blockIdx.x;
  uint32_t nl = lex_doc ? lex_cnt[q] : 0u, nv = vec_doc ? vec_cnt[q] : 0u;
  if (nl == 0xFFFFFFFFu) nl = 0;  // an overflowed vector batch (ss_vec_search_dev) carries no list
  if (nv == 0xFFFFFFFFu) nv = 0;
  nl = nl < k_lex ? nl : k_lex;
  nv = nv < k_vec ? nv : k_vec;
  auto id_at = [&](const void* base, size_t i) -> u64 {
    return wide_ids ? ((const u64*)base)[i] : (u64)((const uint32_t*)base)[i];
  };
  for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
    u64 d = ~0ull;
    float s = -INFINITY;
    uint8_t so = 0;
    if (i < nl) {
      d = id_at(lex_doc, (size_t)q * k_lex + i);
      s = __fdiv_rn(1.0f, ss_fadd(0.6f, (float)i));
      so = SS_SRC_LEXICAL;
    } else if (i < nl + nv) {
      d = id_at(vec_doc, (size_t)q * k_vec + (i - nl));
      s = __fdiv_rn(1.0f, ss_fadd(0.6f, (float)(i - nl)));
      so = SS_SRC_VECTOR;
    }
    docs[i] = d; sc[i] = s; src[i] = so;
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < nv; j += blockDim.x) {  // a vector entry whose doc is in the lexical list joins it
    const u64 d = docs[nl + j];
    for (uint32_t i = 0; i < nl; i++) {
      if (docs[i] == d) {
        sc[i] = ss_fadd(sc[i], sc[nl + j]);
        src[i] = SS_SRC_HYBRID;
        docs[nl + j] = ~0ull;
        sc[nl + j] = -INFINITY;
        break;
      }
    }
  }
  __syncthreads();
  for (uint32_t size = 2; size <= np; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = threadIdx.x; i < (np >> 1); i += blockDim.x) {
        const uint32_t lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool first = ((lo & size) == 0);  // this pair is in a run sorted "best first"
        const float sa = sc[lo], sb = sc[hi];
        const u64 da = docs[lo], db = docs[hi];
        const bool a_before_b = sa > sb || (sa == sb && da < db);
        const bool same = sa == sb && da == db;
        if (!same && a_before_b != first) {
          sc[lo] = sb; sc[hi] = sa; docs[lo] = db; docs[hi] = da;
          const uint8_t t = src[lo]; src[lo] = src[hi]; src[hi] = t;
        }
      }
      __syncthreads();
    }
  }
  __shared__ uint32_t written;
  if (threadIdx.x == 0) written = 0;
  __syncthreads();
  uint32_t local = 0;
  for (uint32_t i = threadIdx.x; i < length; i += blockDim.x) {
    const uint32_t at = offset + i;
    const bool live = at < np && docs[at] != ~0ull;
    out_doc[(size_t)q * length + i] = live ? docs[at] : ~0ull;
    out_score[(size_t)q * length + i] = live ? sc[at] : 0.f;
    if (out_src) out_src[(size_t)q * length + i] = live ? src[at] : (uint8_t)0;
    local += live ? 1u : 0u;
  }
  if (local) atomicAdd(&written, local);
  __syncthreads();
  if (threadIdx.x == 0) out_cnt[q] = written;
}

extern "C" int ss_rrf_merge_dev(int device, uint32_t n_queries, uint32_t k_lex, const void* d_lex_doc, const uint32_t* d_lex_count,
                                uint32_t k_vec, const void* d_vec_doc, const uint32_t* d_vec_count, int doc_ids_are_u64,
                                uint32_t offset, uint32_t length, uint64_t* d_out_doc, float* d_out_score, uint8_t* d_out_source,
                                uint32_t* d_out_count, void* stream) {
  if ((k_lex && (!d_lex_doc || !d_lex_count)) || (k_vec && (!d_vec_doc || !d_vec_count))) return SS_EINVAL;
  if (!d_out_doc || !d_out_score || !d_out_count || length == 0) return SS_EINVAL;
  if ((uint64_t)k_lex + k_vec == 0) return SS_EINVAL;
  if ((uint64_t)k_lex + k_vec > 8192) return SS_ENOTSUP;  // (the union of the two lists lives in LDS: 13 bytes an entry)
  if (n_queries == 0) return SS_OK;
  SS_HIP(hipSetDevice(device));
  uint32_t np = 64;
  while (np < k_lex + k_vec) np <<= 1;
  SS_SET_MAX_LDS(rrf_merge_kernel, 8192 * 13);
  rrf_merge_kernel<<<n_queries, 256, (size_t)np * 13u, (hipStream_t)stream>>>(
      k_lex, k_vec, k_lex ? d_lex_doc : nullptr, d_lex_count, k_vec ? d_vec_doc : nullptr, d_vec_count, doc_ids_are_u64 ? 1 : 0,
      offset, length, (u64*)d_out_doc, d_out_score, d_out_source, d_out_count, np);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
