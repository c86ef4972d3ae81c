// BM25 posting-list union / intersection with exact top-k for gfx950.
//
// Replaces the dispatch block of search_lexical_shard (search.rs:3374-3560): single_blockid (single.rs:292),
// union_docid_2/3 (union.rs:1168/1308), union_blockid -> union_scan (union.rs:265/403) and intersection_blockid
// -> intersection_docid (intersection.rs:2023/112), together with add_result_multiterm_singlefield
// (add_result.rs:3418), get_bm25f_multiterm_singlefield (add_result.rs:1429) and MinHeap::add_topk (min_heap.rs:1193).
//
// The reference walks compressed containers with galloping / bit tables on one core per shard.  Here the HBM image
// holds DECODED postings packed to one dword {doc-in-sub-block:13 | SmallFloat length byte:8 | tf:11}, CSR by
// (term, 4096-doc sub-block).  One WAVE owns a (query, partition-of-sub-blocks) assignment and, per sub-block:
//   phase 1  streams every query term's postings with coalesced dwordx4 loads and adds
//            idf * tf*(K+1)/(tf + comp[len])            (add_result.rs:1445-1447)
//            into a 4096-entry f32 accumulator tile in LDS (ds_add_f32); intersection additionally counts matches;
//   phase 2  revisits the same postings, atomically swaps each accumulator back to 0 (first visitor gets the full
//            score, so every matching doc is seen exactly once and the tile is clean for the next sub-block),
//            counts matches (union: any term, intersection: all terms) and merges survivors into a wave-resident,
//            register-held sorted top-k (strict '>' admission against the current k-th, ties -> lower doc id).
// Partition-local top-k lists are merged by a small bitonic kernel.  Integer / irregular work: no MFMA; the bound is
// HBM bandwidth (4 B per posting + CSR offsets).
#include "ss_common.h"

constexpr int BM_RC = 12;            // posting chunks (256 postings each) in flight per wave and round
constexpr float BM_K1P = 2.2f;       // K + 1.0 (add_result.rs:20)

struct BmParams {
  const uint32_t* post;
  const unsigned long long* term_base;
  const uint32_t* sub_off;
  const float* comp;
  const ss_bm25_query* q;
  unsigned long long* part_keys;   // [nq][P][KS]
  unsigned long long* total;       // [nq] exact match counts
  uint32_t n_sub, n_terms, nq, P, k;
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

template <bool HAS_AND, int KPL>
__global__ void __launch_bounds__(HAS_AND ? 192 : 256) bm25_scan_kernel(BmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WAVES = HAS_AND ? 3 : 4;
  constexpr int WAVE_LDS = BM_SUB * 4 + (HAS_AND ? BM_SUB : 0);
  float* comp = (float*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* acc = (float*)(smem + 1024 + w * WAVE_LDS);
  uint32_t* cntw = (uint32_t*)(smem + 1024 + w * WAVE_LDS + BM_SUB * 4);

  for (int i = tid; i < 256; i += WAVES * 64) comp[i] = p.comp[i];
  for (int i = lane; i < BM_SUB; i += 64) acc[i] = 0.f;
  if (HAS_AND)
    for (int i = lane; i < BM_SUB / 4; i += 64) cntw[i] = 0u;
  __syncthreads();

  const uint32_t total_waves = gridDim.x * WAVES;
  const uint32_t A = p.nq * p.P;
  const uint32_t row_len = p.n_sub + 1;
  const uint32_t k = p.k;

  for (uint32_t a = blockIdx.x * WAVES + w; a < A; a += total_waves) {
    const uint32_t qi = a % p.nq, part = a / p.nq;
    const ss_bm25_query* Q = p.q + qi;
    const uint32_t nt = __builtin_amdgcn_readfirstlane(Q->n_terms);
    const bool is_and = HAS_AND && (__builtin_amdgcn_readfirstlane(Q->op) == SS_OP_INTERSECTION) && nt > 1;
    // lane t < nt carries term t's constants
    uint32_t rowoff = 0;
    u64 tbase = 0;
    float idf_l = 0.f;
    if ((uint32_t)lane < nt) {
      uint32_t term = Q->term[lane];
      idf_l = Q->idf[lane];
      rowoff = term * row_len;
      tbase = p.term_base[term];
    }
    const uint32_t s_begin = (uint32_t)(((u64)p.n_sub * part) / p.P);
    const uint32_t s_end = (uint32_t)(((u64)p.n_sub * (part + 1)) / p.P);
    const uint32_t cpt = BM_RC / nt;  // chunks per term per round
    const uint32_t used = cpt * nt;

    u64 topk[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) topk[r] = 0ull;
    u64 worst = 0ull;
    u64 matched = 0;

    for (uint32_t s = s_begin; s < s_end; ++s) {
      uint32_t b = 0, e = 0;
      if ((uint32_t)lane < nt) {
        b = p.sub_off[rowoff + s];
        e = p.sub_off[rowoff + s + 1];
      }
      const uint32_t len = e - b;
      const u64 pb = tbase + b;
      uint32_t maxlen = 0;
      for (uint32_t t = 0; t < nt; ++t) {
        uint32_t l = __builtin_amdgcn_readlane(len, t);
        maxlen = l > maxlen ? l : maxlen;
      }
      if (maxlen == 0) continue;
      const uint32_t maxc = (maxlen + 255) >> 8;
      const uint32_t doc_base = s << BM_SUB_LOG2;

      uint4 v[BM_RC];
      uint32_t nv[BM_RC];
      float idfj[BM_RC];

      // PH: 1 = accumulate, 2 = collect (reload), 3 = both on the same registers
      auto round = [&](uint32_t c0, int PH) {
        uint32_t t = 0, c = c0;
#pragma unroll
        for (int j = 0; j < BM_RC; j++) {
          nv[j] = 0;
          if ((uint32_t)j < used) {
            uint32_t lj = __builtin_amdgcn_readlane(len, t);
            uint32_t start = c << 8;
            if (start < lj) {
              uint32_t n = lj - start;
              nv[j] = n > 256 ? 256 : n;
              idfj[j] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(idf_l), t));
              u64 base = rdlane64(pb, t) + start;
              if ((uint32_t)lane * 4 < nv[j]) v[j] = *(const uint4*)(p.post + base + lane * 4);
            }
            if (++t == nt) { t = 0; ++c; }
          }
        }
        if (PH & 1) {
#pragma unroll
          for (int j = 0; j < BM_RC; j++) {
            if (nv[j]) {
              const uint32_t pv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
              for (int x = 0; x < 4; x++) {
                if ((uint32_t)lane * 4 + x < nv[j]) {
                  uint32_t doc = pv[x] & 0x1FFFu;
                  float tf = (float)(pv[x] >> 21);
                  float cmp = comp[(pv[x] >> 13) & 0xFFu];
                  // idf * (tf*(K+1) / (tf + comp))   add_result.rs:1447 (SIGMA = 0)
                  float wgt = idfj[j] * (tf * BM_K1P * __builtin_amdgcn_rcpf(tf + cmp));
                  __hip_atomic_fetch_add(&acc[doc], wgt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  if (HAS_AND && is_and)
                    __hip_atomic_fetch_add(&cntw[doc >> 2], 1u << ((doc & 3) * 8), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
              }
            }
          }
        }
        if (PH & 2) {
#pragma unroll
          for (int j = 0; j < BM_RC; j++) {
            if (nv[j]) {
              const uint32_t pv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
              u64 ck[4];
              bool any_c = false;
#pragma unroll
              for (int x = 0; x < 4; x++) {
                const uint32_t doc = pv[x] & 0x1FFFu;
                float sc = 0.f;
                bool hit = false;
                if ((uint32_t)lane * 4 + x < nv[j]) {
                  sc = __hip_atomic_exchange(&acc[doc], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  hit = sc != 0.f;
                  if (HAS_AND && is_and) {
                    uint32_t sh = (doc & 3) * 8;
                    uint32_t old = __hip_atomic_fetch_and(&cntw[doc >> 2], ~(0xFFu << sh), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_WAVEFRONT);
                    hit = hit && (((old >> sh) & 0xFFu) == nt);
                  }
                }
                matched += __popcll(__ballot(hit));
                u64 key = ((u64)__float_as_uint(sc) << 32) | (u64)(0xFFFFFFFFu - (doc_base + doc));
                ck[x] = (hit && key > worst) ? key : 0ull;
                any_c |= ck[x] != 0ull;
              }
              if (k && __ballot(any_c)) worst = topk_offer<KPL>(topk, ck[0], ck[1], ck[2], ck[3], worst, k);
            }
          }
        }
      };

      if (maxc <= cpt) {
        round(0, 3);
      } else {
        for (uint32_t c0 = 0; c0 < maxc; c0 += cpt) round(c0, 1);
        for (uint32_t c0 = 0; c0 < maxc; c0 += cpt) round(c0, 2);
      }
    }

    // publish the partition-local list and the exact match count
    u64* out = p.part_keys + ((size_t)qi * p.P + part) * (64 * KPL);
#pragma unroll
    for (int r = 0; r < KPL; r++) out[r * 64 + lane] = topk[r];
    if (lane == 0 && matched) atomicAdd(&p.total[qi], matched);
  }
}

// ---------------------------------------------------------------- merge of partition-local lists (bitonic in LDS)
// in: [nq][n_lists][KS] sorted-desc lists; each workgroup merges `group` consecutive lists of one query into one
// sorted-desc list of KS keys: out [nq][ceil(n_lists/group)][KS].
__global__ void __launch_bounds__(1024) bm25_merge_kernel(const u64* __restrict__ in, u64* __restrict__ out,
                                                         uint32_t n_lists, uint32_t group, uint32_t KS) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* keys = (u64*)smem;
  const uint32_t q = blockIdx.y, g = blockIdx.x;
  const uint32_t n_groups = (n_lists + group - 1) / group;
  const uint32_t l0 = g * group;
  const uint32_t nl = (l0 + group <= n_lists) ? group : (n_lists - l0);
  const uint32_t n = nl * KS;
  uint32_t np = 64;
  while (np < n) np <<= 1;
  const u64* src = in + ((size_t)q * n_lists + l0) * KS;
  for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) keys[i] = i < n ? src[i] : 0ull;
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
  u64* dst = out + ((size_t)q * n_groups + g) * KS;
  for (uint32_t i = threadIdx.x; i < KS; i += blockDim.x) dst[i] = i < np ? keys[i] : 0ull;
}

__global__ void bm25_final_kernel(const u64* __restrict__ keys, const u64* __restrict__ total, uint32_t KS, uint32_t k,
                                  uint32_t* __restrict__ out_doc, float* __restrict__ out_score,
                                  uint32_t* __restrict__ out_count, u64* __restrict__ out_total) {
  const uint32_t q = blockIdx.x;
  __shared__ uint32_t cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  uint32_t local = 0;
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    u64 key = i < KS ? keys[(size_t)q * KS + i] : 0ull;
    uint32_t doc = SS_NO_DOC;
    float sc = 0.f;
    if (key) {
      doc = 0xFFFFFFFFu - (uint32_t)key;
      sc = __uint_as_float((uint32_t)(key >> 32));
      local++;
    }
    out_doc[(size_t)q * k + i] = doc;
    out_score[(size_t)q * k + i] = sc;
  }
  if (local) atomicAdd(&cnt, local);
  __syncthreads();
  if (threadIdx.x == 0) {
    out_count[q] = cnt;
    out_total[q] = total[q];
  }
}

// ---------------------------------------------------------------- host side
template <bool HAS_AND, int KPL>
static int launch_scan(const BmParams& p, uint32_t grid, hipStream_t st) {
  constexpr int WAVES = HAS_AND ? 3 : 4;
  constexpr int lds = 1024 + WAVES * (BM_SUB * 4 + (HAS_AND ? BM_SUB : 0));
  static bool done = false;
  if (!done) {
    SS_HIP(hipFuncSetAttribute((const void*)bm25_scan_kernel<HAS_AND, KPL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds));
    done = true;
  }
  bm25_scan_kernel<HAS_AND, KPL><<<grid, WAVES * 64, lds, st>>>(p);
  return SS_OK;
}

int ssi_bm25_search(ss_shard* s, uint32_t nq, const ss_bm25_query* d_q, uint32_t k, uint32_t rt, uint32_t* d_out_doc,
                    float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, bool has_and, hipStream_t st) {
  if (!s->d_post) return SS_ESTATE;
  if (nq == 0) return SS_OK;
  if (k > SS_MAX_K) return SS_EINVAL;
  if (rt == SS_RT_COUNT) k = 0;
  const uint32_t kk = k ? k : 1;
  const int KPL = kk <= 64 ? 1 : kk <= 128 ? 2 : kk <= 256 ? 4 : 16;
  const uint32_t KS = 64 * KPL;
  // partitions per query: enough assignments to load-balance ~2048 resident waves
  uint32_t P = (8192 + nq - 1) / nq;
  P = std::max<uint32_t>(1, std::min<uint32_t>(P, s->bm_n_sub));
  const size_t need = (size_t)nq * P * KS * 2 + nq;  // two ping-pong merge buffers + totals
  if (need > s->part_cap) {
    if (s->d_part) (void)hipFree(s->d_part);
    s->d_part = nullptr;
    s->part_cap = 0;
    SS_HIP(hipMalloc(&s->d_part, need * sizeof(u64)));
    s->part_cap = need;
  }
  u64* bufA = (u64*)s->d_part;
  u64* bufB = bufA + (size_t)nq * P * KS;
  u64* total = bufB + (size_t)nq * P * KS;
  SS_HIP(hipMemsetAsync(total, 0, nq * sizeof(u64), st));

  BmParams p;
  p.post = s->d_post;
  p.term_base = (const unsigned long long*)s->d_term_base;
  p.sub_off = s->d_sub_off;
  p.comp = s->d_comp;
  p.q = d_q;
  p.part_keys = bufA;
  p.total = total;
  p.n_sub = s->bm_n_sub;
  p.n_terms = s->bm_n_terms;
  p.nq = nq;
  p.P = P;
  p.k = k;
  const uint32_t A = nq * P;
  const int waves_per_wg = has_and ? 3 : 4;
  uint32_t grid = std::min<uint32_t>((A + waves_per_wg - 1) / waves_per_wg, 512);

  hipEvent_t e0 = nullptr, e1 = nullptr;
  ssi_prof_begin(s, 0, st, &e0, &e1);
  int rc;
#define SS_LAUNCH(AND_)                                   \
  switch (KPL) {                                          \
    case 1: rc = launch_scan<AND_, 1>(p, grid, st); break; \
    case 2: rc = launch_scan<AND_, 2>(p, grid, st); break; \
    case 4: rc = launch_scan<AND_, 4>(p, grid, st); break; \
    default: rc = launch_scan<AND_, 16>(p, grid, st); break; \
  }
  if (has_and) { SS_LAUNCH(true) } else { SS_LAUNCH(false) }
#undef SS_LAUNCH
  ssi_prof_end(s, 0, st, e0, e1);
  if (rc) return rc;

  // merge tree over the P partition lists
  static bool mattr = false;
  if (!mattr) {
    SS_HIP(hipFuncSetAttribute((const void*)bm25_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
    mattr = true;
  }
  uint32_t lists = P;
  u64 *src = bufA, *dst = bufB;
  const uint32_t group = 8192 / KS;
  while (lists > 1) {
    uint32_t ng = (lists + group - 1) / group;
    bm25_merge_kernel<<<dim3(ng, nq), 1024, 8192 * 8, st>>>(src, dst, lists, group, KS);
    std::swap(src, dst);
    lists = ng;
  }
  bm25_final_kernel<<<nq, 64, 0, st>>>(src, total, KS, k, d_out_doc, d_out_score, d_out_count,
                                       (u64*)d_out_total);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
