// Generic intersections and phrases by GALLOPING LOOKUPS -- the query shapes the specialised kernels leave out, so that no query
// the reference answers comes back SS_ENOTSUP (VERDICT r5 "next" 1; the shape router is ss_api.hip bm25_route_shapes):
//
//   * intersections (and single terms under a field filter) of ANY number of terms <= SS_MAX_QUERY_TERMS over any mix of the dense
//     and the sparse tier, one or several indexed fields with or without merged lists, under a field filter with any number of
//     (term, field) lists, with NOT terms, tombstones / a facet filter's bitmap, exact counts and any k <= SS_MAX_K;
//   * the all_terms_frequent shortcut (intersection.rs:198-209, add_result.rs:2091-2104) for any number of terms: the mark is bit 0
//     of a flagged list's weight code (ss_common.h), read from the posting a lookup finds;
//   * phrases (add_result.rs:3586-3684) of up to SS_MAX_PHRASE unique terms, any k, either tier, no probe rows needed.
//
// The reference's own intersection is this (intersection.rs:352-362 "galloping"; north_star: "galloping intersection"): the SHORTEST
// list of the query drives, every one of its docs is looked up in the other lists by binary search -- a dense list inside the doc's
// (term, 4096-doc sub-block) segment, a sparse list over the whole array (bm25_find.h).  One wave per (query, share of the driver's
// postings); a lane owns one driver posting.  A dense driver is walked by posting INDEX (shares are equal whatever the doc
// distribution); the posting's sub-block is found in the term's directory row by binary search.
// Scores: the fma chain in query-term order over the same 19-bit weight codes every other kernel reads -- bit-identical to the
// specialised kernels where both can answer (tests/test_gpu_shape_sweep.py compares them).  HBM-latency bound (dependent loads),
// which is what rare shapes may cost: 1023 sub-queries of a 10-term filtered union run as ONE launch of this kernel.
#include <algorithm>

#include "bm25_find.h"

namespace {

constexpr int GP_WAVES = 4;
constexpr int GP_NT = SS_MAX_PHRASE;  // unique terms of a phrase

struct GpArgs {
  const uint32_t* post;
  const unsigned long long* term_base;
  const uint32_t* sub_off;
  const float* boost;  // [L] (several indexed fields), else null
  const unsigned long long* sp_base;
  const unsigned long long* sp_post;
  const uint32_t* del;
  const ss_bm25_query* q;
  unsigned long long* part_keys;
  unsigned long long* total;
  // phrases: positions of the dense image (pos: u16 or u32 by PT) and of the sparse tier
  const void* pos;
  const uint32_t* pos_off;
  const unsigned long long* pos_base;
  const void* sp_pos;
  const unsigned long long* sp_pos_end;
  uint32_t n_sub, n_dense, L, RF, merged, del_words, nq, P, k, count;
};

// sub-block of the posting at 16-byte unit x4 of a term's image: the s with r[s] <= x4 < r[s + 1] (r: the term's directory row)
__device__ __forceinline__ uint32_t gp_sub_of(const uint32_t* __restrict__ r, uint32_t n_sub, uint32_t x4) {
  uint32_t lo = 0, hi = n_sub;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (r[mid + 1] <= x4) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// PHRASE: the position check of bm25_phrase.hip / bm25_sparse_phrase_kernel over up to GP_NT unique terms; PT: u16 (one indexed field)
// or u32 (merged lists: field << 20 | position)
template <int KPL, bool PHRASE, typename PT>
__global__ void __launch_bounds__(GP_WAVES * 64) bm25_gallop_kernel(const GpArgs A) {
  const int lane = threadIdx.x & 63;
  const uint32_t a = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (a >= A.nq * A.P) return;
  const uint32_t qi = a % A.nq, part = a / A.nq;
  const ss_bm25_query* __restrict__ Q = A.q + qi;
  const uint32_t nt = Q->n_terms, n_not = bm_q_nnot(Q->op), L = A.L, RF = A.RF;
  const uint32_t filt = RF > 1u ? bm_q_field_filter(Q->op) : 0u;
  constexpr bool MF = sizeof(PT) == 4;
  // a phrase's filter is a test on its start's field (bm25_phrase.hip); everything else asks every term for a listed field
  const uint32_t gate = PHRASE ? 0u : filt;
  const uint32_t fmask = (PHRASE && MF && filt) ? filt : 0xFFFFFFFFu;
  const bool freq = !PHRASE && bm_q_all_frequent(Q->op) && nt > 1u && gate == 0u;  // (the host cleared the bit where the rule does not hold)
  const bool field_lists = L > 1u && !A.merged;  // several indexed fields without merged lists: a term = its RF (term, field) lists
  const uint32_t row_len = A.n_sub + 1u;
  const float mscale = L > 1u && A.merged ? A.boost[L - 1u] : 1.0f;  // the merged lists' scale comes back through idf (bm_expand_kernel)

  // ---- the driver: the scored term with the fewest postings (dense lists by their padded image length)
  uint32_t drv = 0;
  {
    unsigned long long best = ~0ull;
    for (uint32_t t = 0; t < nt; t++) {
      const uint32_t term = Q->term[t];
      unsigned long long len = 0;
      if (term >= A.n_dense) len = A.sp_base[term - A.n_dense + 1u] - A.sp_base[term - A.n_dense];
      else if (field_lists) { for (uint32_t f = 0; f < RF; f++) len += (A.term_base[term * L + f + 1u] - A.term_base[term * L + f]) * 4ull; }
      else len = (A.term_base[term * L + L] - A.term_base[term * L + L - 1u]) * 4ull;
      if (len < best) { best = len; drv = t; }
    }
  }
  const uint32_t dterm = Q->term[drv];
  const bool dsparse = dterm >= A.n_dense;
  const uint32_t n_dlists = (!dsparse && field_lists) ? RF : 1u;

  unsigned long long wpack = 0ull;  // place i of the phrase -> unique term (15 = a place inside an n-gram key), 4 bits each
  if constexpr (PHRASE) {
#pragma unroll
    for (int i = 0; i < SS_MAX_PHRASE; i++) wpack |= (unsigned long long)(Q->phrase_seq[i] == SS_PHRASE_SKIP ? 15u : (Q->phrase_seq[i] & 15u)) << (4 * i);
  }
  auto wslot = [&](uint32_t i) -> uint32_t { return (uint32_t)(wpack >> (4u * i)) & 15u; };
  const uint32_t plen = PHRASE ? Q->phrase_len : 0u;

  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull;
  T.wsc = -1.0f;
  T.matched = 0;

  for (uint32_t dl = 0; dl < n_dlists; dl++) {
    const uint32_t drow = dsparse ? 0u : (field_lists ? dterm * L + dl : dterm * L + (L - 1u));
    const unsigned long long dbase = dsparse ? A.sp_base[dterm - A.n_dense] : A.term_base[drow] * 4ull;
    const unsigned long long dlen = dsparse ? A.sp_base[dterm - A.n_dense + 1u] - dbase : A.term_base[drow + 1u] * 4ull - dbase;
    const unsigned long long chunks = (dlen + 63ull) >> 6;
    const unsigned long long x_begin = ((chunks * part) / A.P) << 6, x_end = std::min<unsigned long long>(((chunks * (part + 1ull)) / A.P) << 6, dlen);
    const uint32_t* __restrict__ drow_dir = A.sub_off + (size_t)drow * row_len;
    for (unsigned long long x0 = x_begin; x0 < x_end; x0 += 64ull) {
      const unsigned long long x = x0 + (unsigned)lane;
      bool live = x < x_end;
      uint32_t doc = 0u;
      if (dsparse) {
        if (live) doc = (uint32_t)A.sp_post[dbase + x];
      } else {
        const uint32_t p = live ? A.post[dbase + x] : 0u;
        live = p != 0u;  // NULL padding at a segment's end
        if (live) doc = (gp_sub_of(drow_dir, A.n_sub, (uint32_t)(x >> 2)) << BM_SUB_LOG2) + bm_doc_field(p) - 1u;
        // (term, field) lists: a doc the term holds in several fields is met in each -- it is answered under the lowest
        for (uint32_t f = 0; f < dl && live; f++)
          if (dense_find(A.post, A.term_base, A.sub_off, A.n_sub, dterm * L + f, doc)) live = false;
      }
      if (live && A.del && (doc >> 5) < A.del_words && ((A.del[doc >> 5] >> (doc & 31u)) & 1u)) live = false;  // add_result.rs:3435
      if (__ballot(live) == 0ull) continue;
      float score = 0.f;
      uint32_t lt10 = 0u;
      const PT* pp[PHRASE ? GP_NT : 1];  // phrases: the doc's positions of every unique term
      uint32_t pn[PHRASE ? GP_NT : 1];
      if constexpr (PHRASE) {
#pragma unroll
        for (int t = 0; t < GP_NT; t++) { pp[t] = (const PT*)A.pos; pn[t] = 0u; }
      }
      // every term in query order: scored terms must be present (and stand in a listed field), NOT terms absent
#pragma unroll 1
      for (uint32_t t = 0; t < nt + n_not; t++) {
        if (__ballot(live) == 0ull) break;
        const uint32_t term = Q->term[t];
        const bool scored = t < nt;
        const float idf = scored ? Q->idf[t] : 0.f;
        bool present = false, pass = gate == 0u;
        const PT* tp = (const PT*)A.pos;
        uint32_t tn = 0u;
        if (live) {
          if (term >= A.n_dense) {
            const uint32_t j = term - A.n_dense;
            const unsigned long long b1 = A.sp_base[j + 1u];
            const unsigned long long p = sp_find(A.sp_post, A.sp_base[j], b1, doc);
            if (p < b1) {
              const unsigned long long e = A.sp_post[p];
              if ((uint32_t)e == doc) {
                const uint32_t code = (uint32_t)(e >> 32);
                present = true;
                pass = pass || ((code >> BM_SP_FIELD_SHIFT) & gate) != 0u;
                if (scored) score = fmaf(L > 1u ? mscale * idf : idf, bm_wdecode(code & BM_SP_CODE_MASK), score);
                if constexpr (PHRASE) if (scored) {
                  const unsigned long long st = p ? A.sp_pos_end[p - 1ull] : 0ull;
                  tp = (const PT*)A.sp_pos + st;
                  tn = (uint32_t)(A.sp_pos_end[p] - st);
                }
              }
            }
          } else if (!field_lists) {
            const uint32_t row = term * L + (L - 1u);
            uint32_t slot = 0u;
            const uint32_t code = dense_find(A.post, A.term_base, A.sub_off, A.n_sub, row, doc, &slot);
            if (code) {
              present = true;
              if (scored) {
                score = fmaf(L > 1u ? mscale * idf : idf, bm_wdecode(code), score);
                lt10 |= code & 1u;
                for (uint32_t f = 0; f < RF && !pass; f++)
                  if ((gate >> f) & 1u) pass = dense_find(A.post, A.term_base, A.sub_off, A.n_sub, term * L + f, doc) != 0u;
                if constexpr (PHRASE) {
                  const uint32_t* __restrict__ po = A.pos_off + A.term_base[row] * 4ull;
                  const uint32_t st = slot ? po[slot - 1u] : 0u;
                  tp = (const PT*)A.pos + A.pos_base[row] + st;
                  tn = po[slot] - st;
                }
              }
            }
          } else {
            for (uint32_t f = 0; f < RF; f++) {
              const uint32_t code = dense_find(A.post, A.term_base, A.sub_off, A.n_sub, term * L + f, doc);
              if (!code) continue;
              present = true;
              if (scored) score = fmaf(A.boost[f] * idf, bm_wdecode(code), score);  // weight * plo.idf, add_result.rs:1253-1261
              pass = pass || ((gate >> f) & 1u) != 0u;
            }
          }
        }
        live = live && (scored ? (present && pass) : !present);
        if constexpr (PHRASE) if (scored) {
#pragma unroll
          for (int u = 0; u < GP_NT; u++)
            if ((uint32_t)u == t) { pp[u] = tp; pn[u] = tn; }
        }
      }
      if (__ballot(live) == 0ull) continue;
      unsigned long long key = (live && score > 0.f) ? (((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xFFFFFFFFu - doc)) : 0ull;
      if constexpr (PHRASE) {
        // Topk: a doc whose BM25 cannot enter the list skips the position check, as the reference does (add_result.rs:3573-3583)
        if (A.k && !A.count) live = live && key > T.worst;
        if (live) {  // the phrase: start = a position of word 0, word i must sit at start + i
          auto range_of = [&](uint32_t sl, const PT*& base, uint32_t& n) {
            base = pp[0]; n = pn[0];
#pragma unroll
            for (int u = 1; u < GP_NT; u++)
              if (sl == (uint32_t)u) { base = pp[u]; n = pn[u]; }
          };
          const PT* b0p;
          uint32_t n0;
          range_of(wslot(0u), b0p, n0);
          bool match = false;
          for (uint32_t j = 0; j < n0 && !match; j++) {
            const uint32_t start = b0p[j];
            bool ok = !MF || ((fmask >> (start >> BM_POS_FIELD_SHIFT)) & 1u);
            for (uint32_t i = 1; i < plen && ok; i++) {
              if (wslot(i) == 15u) continue;
              const PT* bp;
              uint32_t n;
              range_of(wslot(i), bp, n);
              const uint32_t target = start + i;
              uint32_t lo = 0, hi = n;
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint32_t)bp[mid] < target) lo = mid + 1u; else hi = mid;
              }
              ok = lo < n && (uint32_t)bp[lo] == target;
            }
            match = ok;
          }
          live = match;
        }
        if (!live) key = 0ull;
      }
      if (A.count) T.matched += (unsigned long long)__popcll(__ballot(live));
      if (A.k) {
        if (freq && lt10) key = 0ull;  // counted, never ranked (add_result.rs:2091-2104, 3541-3556)
        key = key > T.worst ? key : 0ull;
        if (__ballot(key != 0ull)) T = bm_offer_lane_keys<KPL>(T, key, A.k, nullptr);
      }
    }
  }
  unsigned long long* out = A.part_keys + ((size_t)qi * A.P + part) * (64 * KPL);
#pragma unroll
  for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
  if (lane == 0 && T.matched) atomicAdd(&A.total[qi], T.matched);
}

template <int KPL>
int gp_launch(const GpArgs& A, bool phrase, bool mf, hipStream_t st) {
  const uint32_t waves = A.nq * A.P, grid = (waves + GP_WAVES - 1) / GP_WAVES;
  if (!phrase) bm25_gallop_kernel<KPL, false, uint16_t><<<grid, GP_WAVES * 64, 0, st>>>(A);
  else if (mf) bm25_gallop_kernel<KPL, true, uint32_t><<<grid, GP_WAVES * 64, 0, st>>>(A);
  else bm25_gallop_kernel<KPL, true, uint16_t><<<grid, GP_WAVES * 64, 0, st>>>(A);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

}  // namespace

// longest_driver: the most postings any query's driver list holds (the host has the queries and the lists' lengths in hand): sets the
// shares per query
int ssi_bm25_gallop_search(ss_shard* s, uint32_t nq, const ss_bm25_query* d_q, bool phrase, uint64_t longest_driver, uint32_t k, uint32_t rt,
                           uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, hipStream_t st) {
  if (!s->d_post) return SS_ESTATE;
  if (nq == 0) return SS_OK;
  if (k > SS_MAX_K) return SS_EINVAL;
  if (rt == SS_RT_COUNT) k = 0;
  const uint32_t kk = k ? k : 1;
  const int KPL = kk <= 64 ? 1 : kk <= 128 ? 2 : kk <= 256 ? 4 : 16;
  const uint32_t KS = 64 * KPL;
  const bool mf = s->bm_n_fields > 1;
  if (phrase) {
    if (mf && !s->bm_merged) return SS_ENOTSUP;  // (the CPU fall-through list)
    if (mf ? !s->d_pos32 : !s->d_pos) return SS_ESTATE;
  }
  // shares: >= 16 steps of 64 driver postings each, <= 64 per query (the merge's tournament), ~8192 waves in all
  uint32_t P = (uint32_t)std::min<uint64_t>(64u, std::max<uint64_t>(1u, longest_driver / 1024u));
  P = std::max<uint32_t>(1u, std::min<uint32_t>(P, std::max<uint32_t>(1u, 8192u / nq)));
  const size_t tau_words = (size_t)nq * BM_TAU_STRIDE / 2;
  const size_t need = (size_t)nq * P * KS * 2 + nq + tau_words;
  ss_bm_ws& W = s->bm_ws[st];
  if (need > W.part_cap) {
    SS_HIP(hipStreamSynchronize(st));
    if (W.d_part) (void)hipFree(W.d_part);
    W.d_part = nullptr;
    W.part_cap = 0;
    SS_HIP(hipMalloc(&W.d_part, need * sizeof(u64)));
    W.part_cap = need;
  }
  u64* bufA = (u64*)W.d_part;
  u64* bufB = bufA + (size_t)nq * P * KS;
  u64* total = bufB + (size_t)nq * P * KS;
  uint32_t* tau = (uint32_t*)(total + nq);
  SS_HIP(hipMemsetAsync(total, 0, ((size_t)nq + tau_words) * sizeof(u64), st));  // counts; "the query contradicted its batch" marks: none
  GpArgs A{};
  A.post = s->d_post;
  A.term_base = (const unsigned long long*)s->d_term_base;
  A.sub_off = s->d_sub_off;
  A.boost = s->d_boost;
  A.sp_base = (const unsigned long long*)s->d_sp_base;
  A.sp_post = (const unsigned long long*)s->d_sp_post;
  A.del = s->n_deleted ? s->d_deleted : nullptr;
  A.del_words = (uint32_t)s->deleted_words;
  A.q = d_q;
  A.part_keys = bufA;
  A.total = total;
  A.pos = mf ? (const void*)s->d_pos32 : (const void*)s->d_pos;
  A.pos_off = s->d_pos_off;
  A.pos_base = (const unsigned long long*)s->d_pos_base;
  A.sp_pos = s->d_sp_pos;
  A.sp_pos_end = (const unsigned long long*)s->d_sp_pos_end;
  A.n_sub = s->bm_n_sub;
  A.n_dense = s->bm_n_terms / s->bm_n_fields;
  A.L = s->bm_n_fields;
  A.RF = bm_real_fields(s);
  A.merged = s->bm_merged ? 1u : 0u;
  A.nq = nq;
  A.P = P;
  A.k = k;
  A.count = rt != SS_RT_TOPK ? 1u : 0u;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ssi_prof_begin(s, 0, st, &e0, &e1);
  int rc;
  switch (KPL) {
    case 1: rc = gp_launch<1>(A, phrase, mf, st); break;
    case 2: rc = gp_launch<2>(A, phrase, mf, st); break;
    case 4: rc = gp_launch<4>(A, phrase, mf, st); break;
    default: rc = gp_launch<16>(A, phrase, mf, st); break;
  }
  ssi_prof_end(s, 0, st, e0, e1);
  if (rc) return rc;
  return ssi_bm25_merge_lists(bufA, bufB, nq, P, KS, k, total, tau, d_out_doc, d_out_score, d_out_count, d_out_total, st);
}
