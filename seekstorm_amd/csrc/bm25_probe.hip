// BM25 top-k with MaxScore-style pruning over a probe index (no accumulator tile): the gfx950 counterpart of the
// reference's "touch as few postings as possible" strategy -- block-max ordered intersection (intersection.rs:2023-2301),
// sub-query decomposition of unions guarded by list maxima (union.rs:1308-1479: a subset is only evaluated if the sum of
// its max_list_score can beat the heap minimum) -- restated for a wave:
//
//   * terms are ordered by their score upper bound U_t = idf_t * max weight of the list (index.rs:3239 max_list_score);
//   * with thr = the k-th best score known for the query (own list or any other partition's, shared through tau[]), the
//     longest suffix of terms whose U sums to less than thr is NON-ESSENTIAL: a doc made only of them cannot enter the
//     top-k.  Only the posting streams of essential terms are read ("drivers"), 64 consecutive postings per load, one per
//     lane, regardless of sub-block boundaries (each lane derives its sub-block from the stream position);
//   * every other term is PROBED for the driver's doc: an 8-byte record of 64 doc bits tells membership; only on a hit
//     are the rank table (index of the group's first posting) and then the posting itself read for its weight.  Probing
//     stops as soon as partial score + remaining upper bounds < thr.  A doc present in several essential terms is evaluated by
//     the first; driver streams run one after the other (largest U first), so by the time the second stream would
//     start the threshold has usually made it non-essential and it is skipped altogether;
//   * intersections use the shortest list as the only driver and require a hit in every other term; their exact match
//     count is the number of surviving drivers, so Count / TopkCount need no exhaustive pass either.
//
// Scores are combined in query-term order with the same fma chain as the exhaustive kernels, so both produce bit-identical
// scores.  One wave per (query, partition); LDS holds the weight table and one survivor queue per wave (dense stage ->
// sparse stage); G chunks are evaluated together so that their gathers are in flight at the same time.  Exact union
// counts (bm25_union_count_kernel, bottom of the file) are popcounts over the same bit records.
#include <cstdlib>
#include <type_traits>

#include "bm25_probe_body.h"


// pmax[q][part][t] = largest weight of query term t (query order, t < 4) inside the sub-blocks of partition `part`: one wave
// per (query, partition), lanes strided over the partition's sub-blocks.  Reads nq * n_terms_of_query * n_sub floats per
// launch (29 MB for 1000 C2 queries): a few microseconds, and the probe kernel's registers stay what they were (computing
// the maxima inside it cost 3 VGPRs and pushed the kernel into scratch: 0.57 -> 1.06 ms per 1000 queries).
__global__ void __launch_bounds__(256) bm_partmax_kernel(const bm_vquery* __restrict__ qs, const float* __restrict__ submax, uint32_t n_sub,
                                                         uint32_t n_terms, uint32_t nq, uint32_t P, float* __restrict__ pmax) {
  const uint32_t a = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (a >= nq * P) return;
  const uint32_t qi = a % nq, part = a / nq;
  const bm_vquery* __restrict__ Q = qs + qi;
  const uint32_t s_begin = (uint32_t)(((u64)n_sub * part) / P), s_end = (uint32_t)(((u64)n_sub * (part + 1)) / P);
  for (uint32_t t = 0; t < 4u; t++) {
    const uint32_t term = t < Q->n_terms ? Q->term[t] : n_terms;
    float m = 0.f;
    for (uint32_t sb = s_begin + (uint32_t)lane; sb < s_end; sb += 64u) m = fmaxf(m, submax[(size_t)term * n_sub + sb]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) pmax[((size_t)qi * P + part) * 4u + t] = m;
  }
}

// qbound[q][s] = sum over the query's terms of idf * (largest weight of the term in sub-block s): no doc of sub-block s can
// score higher.  The driver streams of the probe kernel jump over sub-blocks whose bound lies below the current threshold
// -- the reference's block skip (single.rs:383-391, intersection.rs:2227-2233) at this image's block size.
__global__ void __launch_bounds__(256) bm_qbound_kernel(const bm_vquery* __restrict__ qs, const float* __restrict__ submax, uint32_t n_sub,
                                                        uint32_t nq, float* __restrict__ qbound) {
  const uint32_t q = blockIdx.y, sb = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq || sb >= n_sub) return;
  const bm_vquery* __restrict__ Q = qs + q;
  float b = 0.f;
  for (uint32_t t = 0; t < Q->n_terms; t++) b = fmaf(Q->idf[t], submax[(size_t)Q->term[t] * n_sub + sb], b);
  qbound[(size_t)q * n_sub + sb] = b * 1.00001f;  // the sum is rounded in another order than a doc's score
}

// (FILT / SKIP and the body itself: bm25_probe_body.h pb_wave)
template <int NT, int KPL, bool FILT, bool SKIP>
__global__ void __launch_bounds__(PB_WAVES * 64, 6) bm25_probe_kernel(
    const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base, const uint32_t* __restrict__ sub_off,
    const uint2* __restrict__ probe, const uint32_t* __restrict__ probe_z,
    const uint32_t* __restrict__ probe_row, const float* __restrict__ umax, const float* __restrict__ pmax, const float* __restrict__ qbound,
    const bm_vquery* __restrict__ qs, unsigned long long* __restrict__ part_keys, unsigned long long* __restrict__ total,
    uint32_t* tau, const uint32_t* __restrict__ del, uint32_t del_words, uint32_t n_sub, uint32_t n_terms,
    uint32_t nq, uint32_t P, uint32_t k, uint32_t count) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  const uint32_t a = blockIdx.x * PB_WAVES + w;
  if (a >= nq * P) return;
  const uint32_t qi = a % nq, part = a / nq;
#ifndef PB_STAGED_KTHB
#define PB_STAGED_KTHB 0
#endif
  // (PB_STAGED_KTHB: rank 0 of every partition's output list as its best-so-far key, zeroed by bm_expand_kernel -- measurement switch)
  const BmTop<KPL> T = pb_wave<NT, KPL, FILT, SKIP, false, 4, PB_STAGED_KTHB != 0>(post, term_base, sub_off, probe, probe_z, probe_row, umax, pmax, qbound, PbQueryMem{qs + qi}, tau, del,
                                                    del_words, n_sub, n_terms, P, k, count, qi, part, w, lane, 0.f,
                                                    (PB_STAGED_KTHB != 0 && k <= 64u && P >= k && k != 0u) ? part_keys + (size_t)qi * P * (64 * KPL) : nullptr, 64u * KPL);

  u64* out = part_keys + ((size_t)qi * P + part) * (64 * KPL);
#pragma unroll
  for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
  if (lane == 0 && T.matched) atomicAdd(&total[qi], T.matched);
}

template <int NT, int KPL, bool FILT, bool SKIP>
static int launch_probe(const BmParams& p, const uint2* probe, const uint32_t* probe_z, const uint32_t* probe_row, const float* umax,
                        const float* pmax, const float* qbound, hipStream_t st) {
  const uint32_t A = p.nq * p.P;
  bm25_probe_kernel<NT, KPL, FILT, SKIP><<<(A + PB_WAVES - 1) / PB_WAVES, PB_WAVES * 64, PB_WAVES * PB_QCAP * 12, st>>>(
      p.post, p.term_base, p.sub_off, probe, probe_z, probe_row, umax, pmax, qbound, p.q, p.part_keys, p.total, p.tau, p.del, p.del_words, p.n_sub, p.n_terms, p.nq, p.P, p.k,
      p.count);
  return SS_OK;
}

// Exact match counts from the probe index's bit records: |A u B u ...| = sum over 64-doc groups of
// popcount(bits_A | bits_B | ...), an intersection the popcount of the AND over the query terms (each the OR of its
// (term, field) lists), minus NOT lists and tombstones -- the reference's own way of counting (union_count over bitmaps,
// union.rs:807-; deleted docs cleared, union.rs:975; Bitmap x Bitmap intersections, intersection.rs:564-752).  One
// coalesced 8-byte load per term and group, no scoring: with it a TopkCount request is a top-k kernel (pruned, or the scan
// without its count mode) plus this count.  One wave per (query, partition of the group range).  all_queries = 0: only
// unions of > 1 lists (the pruned kernel counts intersections and single lists while it ranks them); 1: every query.
constexpr int CNT_WAVES = 4, CNT_UNROLL = 4;
__global__ void __launch_bounds__(CNT_WAVES * 64) bm25_union_count_kernel(
    const uint2* __restrict__ probe, const uint32_t* __restrict__ probe_row, const bm_vquery* __restrict__ qs,
    unsigned long long* __restrict__ total, const uint32_t* __restrict__ del, uint32_t del_words, uint32_t n_sub, uint32_t nq,
    uint32_t P, uint32_t all_queries, unsigned long long* __restrict__ match_bits /* the match sets themselves ([nq][groups]), or null */) {
  const int lane = threadIdx.x & 63;
  const uint32_t a = blockIdx.x * CNT_WAVES + (threadIdx.x >> 6);
  if (a >= nq * P) return;
  const uint32_t qi = a % nq, part = a / nq;
  const bm_vquery* __restrict__ Q = qs + qi;
  const uint32_t np = Q->n_terms, n_not = bm_q_nnot(Q->op);
  const bool is_and = Q->and_target != 0u;
  if (!all_queries && (is_and || np < 2)) return;
  const uint32_t n_groups = n_sub * (BM_SUB / 64);
  if (match_bits) match_bits += (size_t)qi * n_groups;
  const uint32_t g_begin = (uint32_t)(((u64)n_groups * part) / P), g_end = (uint32_t)(((u64)n_groups * (part + 1)) / P);
  // bit rows of the first lists, resolved once (term -> probe row -> address): the loop below then issues plain loads
  constexpr int CNT_FAST = 8;
  const uint2* rows[CNT_FAST];
#pragma unroll
  for (int t = 0; t < CNT_FAST; t++)
    rows[t] = probe + (size_t)probe_row[(uint32_t)t < np + n_not ? Q->term[t] : Q->term[0]] * n_groups;
  uint32_t cnt = 0;
  for (uint32_t g0 = g_begin; g0 < g_end; g0 += 64u * CNT_UNROLL) {
    u64 acc[CNT_UNROLL], cur[CNT_UNROLL], neg[CNT_UNROLL];
#pragma unroll
    for (int u = 0; u < CNT_UNROLL; u++) { acc[u] = is_and ? ~0ull : 0ull; cur[u] = 0ull; neg[u] = 0ull; }
    if (!is_and) {  // union: OR of the scored lists (the common case: kept free of the group bookkeeping)
#pragma unroll
      for (int t = 0; t < CNT_FAST; t++) {
        if ((uint32_t)t >= np + n_not) break;
#pragma unroll
        for (int u = 0; u < CNT_UNROLL; u++) {
          const uint32_t g = g0 + 64u * u + lane;
          uint2 r = make_uint2(0u, 0u);
          if (g < g_end) r = rows[t][g];
          const u64 b = ((u64)r.y << 32) | r.x;
          if ((uint32_t)t < np) acc[u] |= b; else neg[u] |= b;
        }
      }
      for (uint32_t t = CNT_FAST; t < np + n_not; t++) {
        const uint2* __restrict__ row = probe + (size_t)probe_row[Q->term[t]] * n_groups;
#pragma unroll
        for (int u = 0; u < CNT_UNROLL; u++) {
          const uint32_t g = g0 + 64u * u + lane;
          uint2 r = make_uint2(0u, 0u);
          if (g < g_end) r = row[g];
          const u64 b = ((u64)r.y << 32) | r.x;
          if (t < np) acc[u] |= b; else neg[u] |= b;
        }
      }
    } else {  // intersection: AND over the query terms, each the OR of its (term, field) lists
      uint32_t grp = Q->group[0], seen = np ? 1u : 0u;
      for (uint32_t t = 0; t < np + n_not; t++) {
        const uint2* __restrict__ row = probe + (size_t)probe_row[Q->term[t]] * n_groups;
        const uint32_t gt = Q->group[t];
        if (t < np && gt != grp) {  // next query term: the previous one's fields are complete
#pragma unroll
          for (int u = 0; u < CNT_UNROLL; u++) { acc[u] &= cur[u]; cur[u] = 0ull; }
          grp = gt;
          seen++;
        }
        // a list outside the query's field filter scores but cannot match its term (and_val = 0, bm_expand_kernel)
        const bool matches = t >= np || Q->and_val[t] != 0;
#pragma unroll
        for (int u = 0; u < CNT_UNROLL; u++) {
          const uint32_t g = g0 + 64u * u + lane;
          uint2 r = make_uint2(0u, 0u);
          if (g < g_end) r = row[g];
          const u64 b = ((u64)r.y << 32) | r.x;
          if (t < np) { if (matches) cur[u] |= b; } else neg[u] |= b;
        }
      }
      // a query term without a posting in any field has no list here at all: nothing can match the intersection
#pragma unroll
      for (int u = 0; u < CNT_UNROLL; u++) acc[u] = seen == Q->n_groups ? acc[u] & cur[u] : 0ull;
    }
    if (del) {
#pragma unroll
      for (int u = 0; u < CNT_UNROLL; u++) {
        const uint32_t g = g0 + 64u * u + lane;  // group g = bitmap words 2g, 2g + 1
        if (g < g_end && 2u * g + 1u < del_words) {
          const uint2 r = ((const uint2*)del)[g];
          neg[u] |= ((u64)r.y << 32) | r.x;
        } else if (g < g_end && 2u * g < del_words) {
          neg[u] |= (u64)del[2u * g];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < CNT_UNROLL; u++) cnt += (uint32_t)__popcll(acc[u] & ~neg[u]);
    if (match_bits) {  // facet counting (facet.hip) wants the docs, not only their number
#pragma unroll
      for (int u = 0; u < CNT_UNROLL; u++) {
        const uint32_t g = g0 + 64u * u + lane;
        if (g < g_end) match_bits[g] = acc[u] & ~neg[u];
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0 && cnt) atomicAdd(&total[qi], (unsigned long long)cnt);
}

int ssi_bm25_launch_union_count(const BmParams& p, const uint2* probe, const uint32_t* probe_row, bool all_queries, hipStream_t st,
                                unsigned long long* match_bits) {
  // one partition per ~4096 groups and at least enough waves for two rounds of a full chip
  const uint32_t n_groups = p.n_sub * (BM_SUB / 64);
  uint32_t P = std::max<uint32_t>(1u, std::min<uint32_t>((2u * 8192u + p.nq - 1) / p.nq, (n_groups + 1023u) / 1024u));
  const uint32_t A = p.nq * P;
  bm25_union_count_kernel<<<(A + CNT_WAVES - 1) / CNT_WAVES, CNT_WAVES * 64, 0, st>>>(probe, probe_row, p.q, p.total, p.del, p.del_words,
                                                                                     p.n_sub, p.nq, P, all_queries ? 1u : 0u, match_bits);
  return SS_OK;
}

// returns SS_ENOTSUP when there is no instantiation for (nt_max, KPL): the caller falls back to the exhaustive kernels
int ssi_bm25_launch_probe(const BmParams& p, const uint2* probe, const uint32_t* probe_z, const uint32_t* probe_row, const float* umax,
                          const float* submax, float* pmax_ws, uint32_t nt_max, int KPL, bool any_not, hipStream_t st) {
  const float* pmax = nullptr;
  const float* qbound = nullptr;
  if (submax && pmax_ws) {  // workspace: [nq][P][4] partition maxima, then [nq][n_sub] sub-block bounds
    const uint32_t A = p.nq * p.P;
    bm_partmax_kernel<<<(A + 3) / 4, 256, 0, st>>>(p.q, submax, p.n_sub, p.n_terms, p.nq, p.P, pmax_ws);
    float* qb = pmax_ws + (size_t)A * 4u;
    bm_qbound_kernel<<<dim3((p.n_sub + 255) / 256, p.nq), 256, 0, st>>>(p.q, submax, p.n_sub, p.nq, qb);
    pmax = pmax_ws;
    qbound = qb;
  }
  if (!probe || !probe_z || !probe_row || !umax || nt_max == 0 || nt_max > 4 || (KPL != 1 && KPL != 2)) return SS_ENOTSUP;
  const int NT = nt_max <= 2 ? 2 : (int)nt_max;
  const bool filt = any_not || p.del != nullptr;
#define SS_P(NT_, KPL_)                                                                          \
  if (NT == NT_ && KPL == KPL_)                                                                  \
    return qbound ? (filt ? launch_probe<NT_, KPL_, true, true>(p, probe, probe_z, probe_row, umax, pmax, qbound, st)   \
                          : launch_probe<NT_, KPL_, false, true>(p, probe, probe_z, probe_row, umax, pmax, qbound, st)) \
                  : (filt ? launch_probe<NT_, KPL_, true, false>(p, probe, probe_z, probe_row, umax, pmax, qbound, st)  \
                          : launch_probe<NT_, KPL_, false, false>(p, probe, probe_z, probe_row, umax, pmax, qbound, st));
  SS_P(2, 1) SS_P(3, 1) SS_P(4, 1) SS_P(2, 2) SS_P(3, 2) SS_P(4, 2)
#undef SS_P
  return SS_ENOTSUP;
}
