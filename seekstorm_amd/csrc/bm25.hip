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
#include "bm25_dev.h"


template <bool HAS_AND, int KPL>
__global__ void __launch_bounds__((HAS_AND ? BM_WAVES_AND : BM_WAVES_OR) * 64) bm25_scan_kernel(BmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WAVES = HAS_AND ? BM_WAVES_AND : BM_WAVES_OR;
  constexpr int WAVE_LDS = BM_SUB * 4 + (HAS_AND ? BM_SUB : 0);
  float* comp = (float*)smem;
  float* wlut = comp + 256;  // wlut[(tf<<8)|len] = tf*(K+1)/(tf+comp[len]) for tf < 16
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* wbase = smem + BM_LUT_BYTES + w * WAVE_LDS;
  float* acc = (float*)wbase;
  uint32_t* cntw = (uint32_t*)(wbase + BM_SUB * 4);

  for (int i = tid; i < 256 + 4096; i += WAVES * 64) comp[i] = p.comp[i];
  for (int i = lane; i < BM_SUB; i += 64) acc[i] = 0.f;
  if (HAS_AND)
    for (int i = lane; i < BM_SUB / 4; i += 64) cntw[i] = 0u;
  __syncthreads();

  const uint32_t total_waves = gridDim.x * WAVES;
  const uint32_t A = p.nq * p.P;
  const uint32_t row_len = p.n_sub + 1;
  const uint32_t k = p.k;
  const bool count_mode = p.count != 0;

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

    // sub-block boundaries of my term (lane < nt), rolling window; indices past s_end clamp -> empty items
    // (every lane loads -- lanes >= nt read row 0 -- so the load is unconditional and hipcc can count it)
    auto bnd = [&](uint32_t j) -> uint32_t { return p.sub_off[rowoff + (j < s_end ? j : s_end)]; };

    // issue the loads of one round (chunks c0 .. c0+cpt-1 of every term) of the item [b, b+len) into v[].
    // Always exactly BM_RC loads: inactive slots / lanes read the first posting (one cached line) so that the
    // compiler sees a fixed number of outstanding loads and emits COUNTED vmcnt waits -- the next item's loads
    // then stay in flight while the current item is processed.
    auto issue_loads = [&](uint4(&v)[BM_RC], uint32_t b, uint32_t len, uint32_t c0) {
      const u64 abs0 = tbase + b;
      uint32_t t = 0, c = c0;
#pragma unroll
      for (int j = 0; j < BM_RC; j++) {
        const uint32_t lj = __builtin_amdgcn_readlane(len, t);
        const u64 bj = rdlane64(abs0, t);
        const uint32_t lead = (uint32_t)bj & 3u;  // 16-byte aligned loads; leading elements masked in phase 1
        const uint32_t vs = c << 8;
        const bool on = ((uint32_t)j < used) && (vs + (uint32_t)lane * 4 < lead + lj);
        const uint32_t* src = on ? (p.post + (bj - lead) + vs + lane * 4) : p.post;
        v[j] = *(const uint4*)src;
        if (++t == nt) { t = 0; ++c; }
      }
    };

    // phase 1 on the registers of one round: acc[doc] += idf * wlut[tf,len]   (add_result.rs:1445-1447)
    auto phase1 = [&](uint4(&v)[BM_RC], uint32_t b, uint32_t len, uint32_t c0) {
      const u64 abs0 = tbase + b;
      uint32_t t = 0, c = c0;
#pragma unroll
      for (int j = 0; j < BM_RC; j++) {
        if ((uint32_t)j < used) {
          const uint32_t lj = __builtin_amdgcn_readlane(len, t);
          const uint32_t lead = __builtin_amdgcn_readlane((uint32_t)abs0, t) & 3u;
          const float idf = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(idf_l), t));
          const uint32_t vs = c << 8;
          if (vs < lead + lj) {
            const uint32_t i0 = vs + (uint32_t)lane * 4 - lead;  // index inside the term's span (wraps if before it)
            const uint32_t pv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            // Plain (non-atomic) read-modify-write: the 256 postings of a chunk belong to ONE term, so their docs are
            // distinct, and the tile is private to this wave whose LDS operations execute in order.  (LDS float
            // atomics measured ~3 clk per lane here; a gather + scatter is an order of magnitude cheaper.)
            bool valid[4];
            float old[4], wgt[4];
            uint32_t cold[4];
#pragma unroll
            for (int x = 0; x < 4; x++) {
              const uint32_t pp = pv[x];
              valid[x] = (i0 + x) < lj;
              const uint32_t doc = valid[x] ? (pp & 0xFFFu) : 0u;
              old[x] = acc[doc];
              wgt[x] = wlut[valid[x] ? ((pp >> 13) & 0xFFFu) : 0u];
              if (HAS_AND && is_and) cold[x] = ((const uint8_t*)cntw)[doc];
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
              const uint32_t pp = pv[x];
              if (valid[x]) {
                float wp = wgt[x];
                if (pp >> 25) {  // tf >= 16: outside the table (rare)
                  float tf = (float)(pp >> 21);
                  wp = tf * BM_K1P * __builtin_amdgcn_rcpf(tf + comp[(pp >> 13) & 0xFFu]);
                }
                acc[pp & 0xFFFu] = old[x] + idf * wp;
                if (HAS_AND && is_and) ((uint8_t*)cntw)[pp & 0xFFFu] = (uint8_t)(cold[x] + 1u);
              }
            }
          }
          if (++t == nt) { t = 0; ++c; }
        }
      }
    };

    // phase 2: dense scan of the 4096-entry tile (16 x ds_read_b128 per lane), clear it, collect matches.
    // doc = (i*64 + lane)*4 + e; for intersections byte e of cntw[i*64+lane] counts the terms that hit doc.
    auto phase2 = [&](uint32_t doc_base) {
      const float worst_sc = __uint_as_float((uint32_t)(worst >> 32));
      float wsc = worst_sc;
#pragma unroll 4
      for (int i = 0; i < BM_SUB / 256; i++) {
        const int slot = i * 64 + lane;
        float4 x = *(const float4*)(acc + slot * 4);
        *(float4*)(acc + slot * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        bool h0 = x.x != 0.f, h1 = x.y != 0.f, h2 = x.z != 0.f, h3 = x.w != 0.f;
        if (HAS_AND && is_and) {
          uint32_t cw = cntw[slot];
          cntw[slot] = 0u;
          h0 = (cw & 0xFFu) == nt;
          h1 = ((cw >> 8) & 0xFFu) == nt;
          h2 = ((cw >> 16) & 0xFFu) == nt;
          h3 = (cw >> 24) == nt;
          if (!h0) x.x = 0.f;
          if (!h1) x.y = 0.f;
          if (!h2) x.z = 0.f;
          if (!h3) x.w = 0.f;
        }
        if (count_mode)
          matched += __popcll(__ballot(h0)) + __popcll(__ballot(h1)) + __popcll(__ballot(h2)) + __popcll(__ballot(h3));
        if (k) {
          const float m = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
          if (__ballot(m > 0.f && m >= wsc)) {  // rare once the list is warm
            const uint32_t d0 = doc_base + slot * 4;
            u64 k0 = ((u64)__float_as_uint(x.x) << 32) | (u64)(0xFFFFFFFFu - d0);
            u64 k1 = ((u64)__float_as_uint(x.y) << 32) | (u64)(0xFFFFFFFFu - (d0 + 1));
            u64 k2 = ((u64)__float_as_uint(x.z) << 32) | (u64)(0xFFFFFFFFu - (d0 + 2));
            u64 k3 = ((u64)__float_as_uint(x.w) << 32) | (u64)(0xFFFFFFFFu - (d0 + 3));
            k0 = (x.x > 0.f && k0 > worst) ? k0 : 0ull;
            k1 = (x.y > 0.f && k1 > worst) ? k1 : 0ull;
            k2 = (x.z > 0.f && k2 > worst) ? k2 : 0ull;
            k3 = (x.w > 0.f && k3 > worst) ? k3 : 0ull;
            if (__ballot((k0 | k1 | k2 | k3) != 0ull)) {
              worst = topk_offer<KPL>(topk, k0, k1, k2, k3, worst, k);
              wsc = __uint_as_float((uint32_t)(worst >> 32));
            }
          }
        }
      }
    };

    // rounds needed by an item: max over terms of ceil((lead + len) / 256)
    auto rounds_of = [&](uint32_t b, uint32_t len) -> uint32_t {
      const uint32_t lead = ((uint32_t)(tbase + b)) & 3u;
      const uint32_t nch = len ? ((lead + len + 255) >> 8) : 0;
      uint32_t mx = 0;
      for (uint32_t t = 0; t < nt; ++t) {
        uint32_t n = __builtin_amdgcn_readlane(nch, t);
        mx = n > mx ? n : mx;
      }
      return mx;
    };

    uint4 vA[BM_RC], vB[BM_RC];
    uint32_t B0 = bnd(s_begin), B1 = bnd(s_begin + 1), B2 = bnd(s_begin + 2);
    issue_loads(vA, B0, B1 - B0, 0);

    // one item: prefetch the boundary 3 ahead and the postings of the next item, then process the current one
    auto body = [&](uint4(&cur)[BM_RC], uint4(&nxt)[BM_RC], uint32_t s) {
      const uint32_t B3 = bnd(s + 3);
      issue_loads(nxt, B1, B2 - B1, 0);
      const uint32_t len = B1 - B0;
      const uint32_t maxc = rounds_of(B0, len);
      if (maxc) {
        for (uint32_t c0 = 0; c0 < maxc; c0 += cpt) {
          if (c0) issue_loads(cur, B0, len, c0);  // oversized item: rounds after the prefetched one load synchronously
          phase1(cur, B0, len, c0);
        }
        phase2(s << BM_SUB_LOG2);
      }
      B0 = B1; B1 = B2; B2 = B3;
    };
    for (uint32_t s = s_begin; s < s_end; s += 2) {
      body(vA, vB, s);
      if (s + 1 < s_end) body(vB, vA, s + 1);
      else break;
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
  constexpr int WAVES = HAS_AND ? BM_WAVES_AND : BM_WAVES_OR;
  constexpr int lds = BM_LUT_BYTES + WAVES * (BM_SUB * 4 + (HAS_AND ? BM_SUB : 0));
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
                    float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, bool has_and, uint32_t nt_max, hipStream_t st) {
  if (!s->d_post) return SS_ESTATE;
  if (nq == 0) return SS_OK;
  if (k > SS_MAX_K) return SS_EINVAL;
  if (rt == SS_RT_COUNT) k = 0;
  const uint32_t kk = k ? k : 1;
  const int KPL = kk <= 64 ? 1 : kk <= 128 ? 2 : kk <= 256 ? 4 : 16;
  const uint32_t KS = 64 * KPL;
  // partitions per query: enough assignments to load-balance ~2048 resident waves
  uint32_t P = (4u * 256u * BM_WAVES_OR + nq - 1) / nq;
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
  p.count = (rt == SS_RT_TOPK) ? 0u : 1u;  // Topk: result_count_total is not required to be exact
  const uint32_t A = nq * P;
  const int waves_per_wg = has_and ? BM_WAVES_AND : BM_WAVES_OR;
  uint32_t grid = std::min<uint32_t>((A + waves_per_wg - 1) / waves_per_wg, 256);

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
  rc = ssi_bm25_launch_fast(p, nt_max, has_and, KPL, st);  // NT-specialised kernels for <= 4 terms, k <= 128
  if (rc == SS_ENOTSUP) {
    if (has_and) { SS_LAUNCH(true) } else { SS_LAUNCH(false) }
  }
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
