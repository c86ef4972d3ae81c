// Phrase queries (QueryType::Phrase, "..." operators) for gfx950.
//
// Reference path: intersection_blockid / intersection_docid find the docs that hold every unique term of the phrase, and
// add_result_multiterm_singlefield keeps a doc only if its positions carry the phrase (add_result.rs:3586-3684): the position
// lists of the phrase's words (decode_positions_multiterm_singlefield, add_result.rs:2036-2197; VINT positions,
// compress_postinglist.rs:949-977) are merged on "position - place in the phrase" until every word agrees on one start.
// The doc is then counted and scored like the intersection of the unique terms (get_bm25f_multiterm_singlefield).
//
// Here: one wave per (query, partition of sub-blocks).  The shortest list is the driver, 64 postings per step (one per
// lane); every other unique term is probed through the probe index's 64-doc bit records (membership + rank -> the index
// of the doc's posting in that term).  For the survivors -- the intersection -- each lane checks the phrase over the
// positions of ITS doc: the starts offered by the first word's positions, every other word looked up by binary search in
// its own (ascending) position list.  That decides exactly what the reference's merge decides (some start carries word i at
// start + i for every i; tests/test_oracle_kat.py pins the restated loop against this definition).  Under ResultType::Topk
// a doc whose BM25 cannot enter the list skips the position check, as the reference does (add_result.rs:3573-3583).
// Positions in HBM: d_pos (u16 pool in image order), d_pos_off (end offset per padded posting slot, relative to the term),
// d_pos_base (first position of a term) -- ss_bm25_upload_positions.
//
// SEVERAL indexed fields (add_result_multiterm_multifield, add_result.rs:2964-3414): the reference walks the fields in ascending
// order and runs the same merge inside every field in which all words of the phrase have positions (3259-3386), skipping the
// fields a field filter does not list (3285-3287); the first match ends it, and the doc is scored over ALL fields of its terms
// (get_bm25f_multiterm_multifield, 3140 / 3402).  Here (PT = uint32_t): the query reads its terms' MERGED lists (one posting per
// (term, doc), weight = the BM25F sum over the doc's fields), whose positions carry their field above bit 20 -- field f's positions
// of a posting follow field f - 1's, so the list is ascending as a whole, "word i at start + i" can only be satisfied inside the
// start's own field (a position is < 65 536, the phrase <= SS_MAX_PHRASE words), and the filter is a test of the START's field
// bits against Q->phrase_fields.  d_pos32 / ss_bm25_upload_fields_positions.
#include "bm25_dev.h"

constexpr int PH_WAVES = 4;

template <int NT, int KPL, typename PT>
__global__ void __launch_bounds__(PH_WAVES * 64) bm25_phrase_kernel(
    const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base, const uint32_t* __restrict__ sub_off,
    const uint2* __restrict__ probe, const uint32_t* __restrict__ probe_z, const uint32_t* __restrict__ probe_row,
    const PT* __restrict__ pos, const uint32_t* __restrict__ pos_off, const unsigned long long* __restrict__ pos_base,
    const bm_vquery* __restrict__ qs, unsigned long long* __restrict__ part_keys, unsigned long long* __restrict__ total,
    uint32_t* tau, const uint32_t* __restrict__ del, uint32_t del_words, uint32_t n_sub, uint32_t n_terms, uint32_t nq, uint32_t P,
    uint32_t k, uint32_t count) {
  const int lane = threadIdx.x & 63;
  const uint32_t a = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (a >= nq * P) return;
  const uint32_t qi = a % nq, part = a / nq;
  const bm_vquery* __restrict__ Q = qs + qi;
  const uint32_t nt = Q->n_terms, plen = Q->phrase_len, n_not = bm_q_nnot(Q->op);
  constexpr bool MF = sizeof(PT) == 4;  // several indexed fields: positions carry their field
  const uint32_t fmask = MF ? Q->phrase_fields : 0xFFFFFFFFu;
  const uint32_t row_len = n_sub + 1;

  // per unique term, in PROCESSING order (slot 0 = the driver = the shortest list); qpos = its place in the query
  const uint32_t* tptr[NT];
  const uint32_t* rowp[NT];
  const uint2* prow[NT];
  const uint32_t* zrow[NT];
  const uint32_t* po[NT];   // end offsets of the postings' positions (slot index = index inside the term)
  const PT* pp[NT];         // the term's positions
  float idf[NT];
  uint32_t qpos[NT];
  unsigned long long size[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const bool have = (uint32_t)t < nt;
    const uint32_t term = have ? Q->term[t] : n_terms;
    idf[t] = have ? Q->idf[t] : 0.f;
    qpos[t] = t;
    tptr[t] = post + term_base[term] * 4ull;
    rowp[t] = sub_off + (size_t)term * row_len;
    prow[t] = probe + (size_t)probe_row[term] * n_sub * (BM_SUB / 64);
    zrow[t] = probe_z + (size_t)probe_row[term] * n_sub * (BM_SUB / 64);
    po[t] = pos_off + term_base[term] * 4ull;
    pp[t] = pos + pos_base[have ? term : 0];
    size[t] = have ? term_base[term + 1] - term_base[term] : ~0ull;
  }
  // the shortest list drives (slot 0)
#pragma unroll
  for (int t = 1; t < NT; t++) {
    if (size[t] < size[0]) {
      { auto x = tptr[0]; tptr[0] = tptr[t]; tptr[t] = x; }
      { auto x = rowp[0]; rowp[0] = rowp[t]; rowp[t] = x; }
      { auto x = prow[0]; prow[0] = prow[t]; prow[t] = x; }
      { auto x = zrow[0]; zrow[0] = zrow[t]; zrow[t] = x; }
      { auto x = po[0]; po[0] = po[t]; po[t] = x; }
      { auto x = pp[0]; pp[0] = pp[t]; pp[t] = x; }
      { float x = idf[0]; idf[0] = idf[t]; idf[t] = x; }
      { uint32_t x = qpos[0]; qpos[0] = qpos[t]; qpos[t] = x; }
      { auto x = size[0]; size[0] = size[t]; size[t] = x; }
    }
  }
  // slot of every word of the phrase (uniform per wave): word i is unique term phrase_seq[i], which sits in slot slot_of(..)
  // (3 bits per word packed in one scalar: a register array indexed by the running word would live in scratch).
  // N-gram keys (the reference's default index, index.rs:1422-1424): a bigram / trigram key of the query is ONE entry of
  // non_unique_query_list whose positions are those of its first word and whose successor stands 2 / 3 places later
  // (term_index_nonunique = entries before it + preceding_ngram_count, search.rs:3305-3328): its first component term holds the
  // key's positions, the places of its other words carry no entry -- phrase_seq = SS_PHRASE_SKIP, slot 7 here.
  u64 wpack = 0ull;
#pragma unroll
  for (int i = 0; i < SS_MAX_PHRASE; i++) {
    uint32_t sl = Q->phrase_seq[i] == SS_PHRASE_SKIP ? 7u : 0u;
#pragma unroll
    for (int t = 0; t < NT; t++)
      if (qpos[t] == (uint32_t)Q->phrase_seq[i]) sl = t;
    wpack |= (u64)sl << (3 * i);
  }
  auto wslot = [&](uint32_t i) -> uint32_t { return (uint32_t)(wpack >> (3u * i)) & 7u; };

  const uint32_t s_begin = (uint32_t)(((u64)n_sub * part) / P);
  const uint32_t s_end = (uint32_t)(((u64)n_sub * (part + 1)) / P);
  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull;
  T.wsc = -1.0f;
  T.matched = 0;
  uint32_t* tau_q = tau + (size_t)qi * BM_TAU_STRIDE;

  for (uint32_t s = s_begin; s < s_end; s++) {
    const uint32_t b0 = rowp[0][s] * 4u, b1 = rowp[0][s + 1] * 4u;  // the driver's postings of this sub-block
    for (uint32_t x = b0; x < b1; x += 64u) {
      const uint32_t i0 = x + (uint32_t)lane;
      const uint32_t p0 = i0 < b1 ? tptr[0][i0] : 0u;
      bool alive = p0 != 0u;
      const uint32_t d = bm_doc_field(p0) - 1u;
      const uint32_t gidx = s * (uint32_t)(BM_SUB / 64) + (d >> 6);
      uint32_t idx[NT];  // index of the doc's posting inside each term
      float w[NT];
      idx[0] = i0;
      w[0] = bm_weight(p0);
      // membership + rank in every other unique term
#pragma unroll
      for (int t = 1; t < NT; t++) {
        idx[t] = 0u;
        w[t] = 0.f;
        if ((uint32_t)t < nt) {
          const uint2 r = prow[t][alive ? gidx : 0u];
          const u64 bits = ((u64)r.y << 32) | r.x;
          const bool hit = alive && ((bits >> (d & 63u)) & 1ull);
          const uint32_t rk = (uint32_t)__popcll(bits & ((1ull << (d & 63u)) - 1ull));
          const uint32_t z = zrow[t][hit ? gidx : 0u];
          alive = hit;
          idx[t] = z + rk;
        }
      }
      if (__ballot(alive) == 0ull) continue;
      // NOT terms (not_query_list applies to every query type, add_result.rs:3440-3497): a doc found in one of their lists is dropped
      // -- one bit record per NOT list and surviving lane
      for (uint32_t j = 0; j < n_not; j++) {
        const uint32_t nterm = Q->term[nt + j];
        const uint2 r = (probe + (size_t)probe_row[nterm] * n_sub * (BM_SUB / 64))[alive ? gidx : 0u];
        const u64 bits = ((u64)r.y << 32) | r.x;
        if ((bits >> (d & 63u)) & 1ull) alive = false;
      }
      if (__ballot(alive) == 0ull) continue;
      const uint32_t doc = (s << BM_SUB_LOG2) + d;
      if (del && alive) {
        const uint32_t wd = (doc >> 5) < del_words ? del[doc >> 5] : 0u;
        alive = !((wd >> (doc & 31u)) & 1u);
      }
      // BM25 of the unique terms, summed in query order (the intersection kernels' fma chain)
#pragma unroll
      for (int t = 1; t < NT; t++)
        if ((uint32_t)t < nt && alive) w[t] = bm_weight(tptr[t][idx[t]]);
      float score = 0.f;
#pragma unroll
      for (uint32_t qp = 0; qp < (uint32_t)NT; qp++) {
#pragma unroll
        for (int t = 0; t < NT; t++)
          if (qpos[t] == qp && (uint32_t)t < nt) score = fmaf(idf[t], w[t], score);
      }
      const float thr = fmaxf(T.wsc, __uint_as_float(__hip_atomic_load(tau_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
      if (k && !count) alive = alive && score >= thr;  // Topk: a doc that cannot enter the list is not checked (add_result.rs:3573-3583)
      // ---- the phrase: start = a position of word 0, word i must sit at start + i
      if (alive) {
        uint32_t st[NT], en[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
          st[t] = 0u; en[t] = 0u;
          if ((uint32_t)t < nt) {
            en[t] = po[t][idx[t]];
            st[t] = idx[t] ? po[t][idx[t] - 1u] : 0u;
          }
        }
        auto range_of = [&](uint32_t sl, uint32_t& lo, uint32_t& hi, const PT*& base) {
          lo = st[0]; hi = en[0]; base = pp[0];
#pragma unroll
          for (int t = 1; t < NT; t++)
            if (sl == (uint32_t)t) { lo = st[t]; hi = en[t]; base = pp[t]; }
        };
        uint32_t lo0, hi0;
        const PT* b0p;
        range_of(wslot(0u), lo0, hi0, b0p);
        bool match = false;
        for (uint32_t j = lo0; j < hi0 && !match; j++) {
          const uint32_t start = b0p[j];
          bool ok = !MF || ((fmask >> (start >> BM_POS_FIELD_SHIFT)) & 1u);  // the field the phrase would stand in is listed
          for (uint32_t i = 1; i < plen && ok; i++) {
            uint32_t lo, hi;
            const PT* bp;
            if (wslot(i) == 7u) continue;  // a place inside an n-gram key: no entry of its own
            range_of(wslot(i), lo, hi, bp);
            const uint32_t end = hi, target = start + i;
            while (lo < hi) {  // first position >= target (the list is ascending)
              const uint32_t mid = (lo + hi) >> 1;
              if ((uint32_t)bp[mid] < target) lo = mid + 1u; else hi = mid;
            }
            ok = lo < end && (uint32_t)bp[lo] == target;
          }
          match = ok;
        }
        alive = match;
      }
      if (__ballot(alive) == 0ull) continue;
      if (count) T.matched += __popcll(__ballot(alive));
      if (k) {
        const bool cand = alive && score >= thr && score > 0.f;
        if (__ballot(cand)) {
          const u64 key = cand ? (((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - doc)) : 0ull;
          const u64 key2 = key > T.worst ? key : 0ull;
          if (__ballot(key2 != 0ull)) T = bm_offer_lane_keys<KPL>(T, key2, k, tau_q);
        }
      }
    }
  }
  u64* out = part_keys + ((size_t)qi * P + part) * (64 * KPL);
#pragma unroll
  for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
  if (lane == 0 && T.matched) atomicAdd(&total[qi], T.matched);
}

template <int NT, int KPL, typename PT>
static int launch_phrase(const BmParams& p, const uint2* probe, const uint32_t* probe_z, const uint32_t* probe_row, const PT* pos,
                         const uint32_t* pos_off, const unsigned long long* pos_base, hipStream_t st) {
  const uint32_t A = p.nq * p.P;
  bm25_phrase_kernel<NT, KPL, PT><<<(A + PH_WAVES - 1) / PH_WAVES, PH_WAVES * 64, 0, st>>>(
      p.post, p.term_base, p.sub_off, probe, probe_z, probe_row, pos, pos_off, pos_base, p.q, p.part_keys, p.total, p.tau, p.del,
      p.del_words, p.n_sub, p.n_terms, p.nq, p.P, p.k, p.count);
  return SS_OK;
}

// pos16: one indexed field; pos32: the merged lists of an image with several (exactly one of the two is given)
int ssi_bm25_launch_phrase(const BmParams& p, const uint2* probe, const uint32_t* probe_z, const uint32_t* probe_row, const uint16_t* pos16,
                           const uint32_t* pos32, const uint32_t* pos_off, const unsigned long long* pos_base, uint32_t nt_max, int KPL,
                           hipStream_t st) {
  if (!probe || !probe_z || !probe_row || (!pos16 == !pos32) || !pos_off || !pos_base) return SS_ESTATE;
  if (nt_max == 0 || nt_max > 6 || (KPL != 1 && KPL != 2)) return SS_ENOTSUP;
  const int NT = nt_max <= 2 ? 2 : nt_max <= 4 ? (int)nt_max : 6;
#define SS_PH(NT_, KPL_)                                                                                             \
  if (NT == NT_ && KPL == KPL_)                                                                                      \
    return pos16 ? launch_phrase<NT_, KPL_, uint16_t>(p, probe, probe_z, probe_row, pos16, pos_off, pos_base, st)    \
                 : launch_phrase<NT_, KPL_, uint32_t>(p, probe, probe_z, probe_row, pos32, pos_off, pos_base, st);
  SS_PH(2, 1) SS_PH(3, 1) SS_PH(4, 1) SS_PH(6, 1) SS_PH(2, 2) SS_PH(3, 2) SS_PH(4, 2) SS_PH(6, 2)
#undef SS_PH
  return SS_ENOTSUP;
}
