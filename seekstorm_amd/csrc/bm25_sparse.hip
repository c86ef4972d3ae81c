// Sparse tier of the BM25 image: posting lists of RARE terms as plain sorted arrays, without a per-sub-block directory row and
// without probe rows.  A real vocabulary holds millions of keys (a segment's key_count, index.rs:3419-3740), almost all of them
// rare: in the dense tier every list costs a directory row of 4 B per 4096-doc sub-block whatever its length (9.8 KB at 10 M
// docs -- 10 GB per million terms); here a posting costs 8 bytes and a list nothing else.  north_star's "galloping intersection"
// (intersection.rs:352-362) is what such lists are read by: every doc of a sparse list is looked up in the query's other lists
// by binary search -- in another sparse list over the whole array, in a dense list inside the doc's (term, sub-block) segment.
//
// Term ids: dense lists keep 0 .. bm_n_terms - 1, sparse list i is term bm_n_terms + i (ss_bm25_append_sparse).
// A query that names a sparse term is answered in two parts (ss_api.hip bm25_search_tiered):
//   * unions: its DENSE terms alone through the ordinary kernels (any strategy), and every doc of its sparse lists scored in
//     FULL here (own posting + every other term probed).  A doc holding a sparse term is thus complete in this kernel's list and
//     at most partial in the dense list; a doc without one is complete in the dense list.  The union of the two lists, a doc kept
//     once with its larger (= full) score, holds the exact top-k: a doc of the true top-k that holds no sparse term is outranked in
//     the dense pass only by docs whose partial, hence full, scores are higher -- fewer than k of them;
//     exact count = the dense union's count + the docs found here in no dense list (each counted under the first sparse list
//     that holds it);
//   * intersections: the shortest sparse list drives, every other term must hit: this kernel alone answers (and counts).
// Scores: fma chain in query-term order, weights from the same 19-bit codes -- what the dense kernels compute for the same doc.
#include "bm25_dev.h"
#include "bm25_find.h"

namespace {

constexpr int SP_WAVES = 4;   // waves per workgroup of the merge kernel (one query per wave)
constexpr int SP_QWAVES = 8;  // waves per QUERY in the sparse kernel: a list of a few thousand postings is 64-posting steps of ~20 dependent
                              // loads each (binary searches) -- walked by one wave, the longest list of a batch set the kernel's time
                              // (1.0 ms at 5000 postings); the steps are dealt to 8 waves, whose lists the first one merges

template <int KPL, int QW>
__global__ void __launch_bounds__(QW * 64) bm25_sparse_kernel(
    const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base, const uint32_t* __restrict__ sub_off, uint32_t n_sub,
    uint32_t n_dense, uint32_t n_lists /* lists per dense term; > 1: the last one is the MERGED list, the one read here */,
    float idf_scale /* the merged lists' scale (d_boost[n_lists - 1]); one indexed field: unused */,
    const unsigned long long* __restrict__ sp_base, const unsigned long long* __restrict__ sp_post, uint32_t n_sparse,
    const ss_bm25_query* __restrict__ qs, uint32_t nq, uint32_t k, const uint32_t* __restrict__ del, uint32_t del_words,
    unsigned long long* __restrict__ out_keys /* [nq][64 KPL] */, unsigned long long* __restrict__ out_extra /* [nq] */) {
  __shared__ unsigned long long wkeys[QW][64 * KPL];
  __shared__ unsigned long long wcount[QW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t qi = blockIdx.x;  // one workgroup per query
  const ss_bm25_query* __restrict__ Q = qs + qi;
  const uint32_t nt = Q->n_terms, n_not = bm_q_nnot(Q->op);
  const bool is_and = bm_q_op(Q->op) == SS_OP_INTERSECTION && nt > 1;
  // a field filter (several indexed fields; intersections and single terms): every term must stand in a listed field -- a sparse
  // posting carries its fields, a dense term is looked up in its listed (term, field) lists; the score stays the merged weight's
  const uint32_t filt = n_lists > 1u ? bm_q_field_filter(Q->op) : 0u;
  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull;
  T.wsc = -1.0f;
  T.matched = 0;
  // the driver lists: a union walks every sparse list of the query, an intersection only its shortest one
  uint32_t first = 0, last = nt;
  if (is_and) {
    unsigned long long best = ~0ull;
    for (uint32_t t = 0; t < nt; t++)
      if (Q->term[t] >= n_dense) {
        const uint32_t i = Q->term[t] - n_dense;
        const unsigned long long len = sp_base[i + 1] - sp_base[i];
        if (len < best) { best = len; first = t; }
      }
    last = first + 1;
  }
  for (uint32_t s = first; s < last; s++) {
    if (Q->term[s] < n_dense) continue;
    const uint32_t si = Q->term[s] - n_dense;
    const unsigned long long b0 = sp_base[si], b1 = sp_base[si + 1];
    for (unsigned long long x = b0 + (unsigned)w * 64u; x < b1; x += 64ull * QW) {
      const bool live0 = x + (unsigned)lane < b1;
      const unsigned long long e = live0 ? sp_post[x + lane] : 0ull;
      const uint32_t doc = (uint32_t)e;
      bool live = live0, in_dense = false;
      float wv[SS_MAX_QUERY_TERMS];
      uint32_t pres = 0u;
#pragma unroll
      for (int t = 0; t < SS_MAX_QUERY_TERMS; t++) wv[t] = 0.f;
#pragma unroll
      for (int t = 0; t < SS_MAX_QUERY_TERMS; t++) {
        if ((uint32_t)t >= nt + n_not) break;
        uint32_t code = 0u;
        if ((uint32_t)t == s) {
          code = (uint32_t)(e >> 32);
          if (filt && !((code >> BM_SP_FIELD_SHIFT) & filt)) live = false;
        } else if (live) {
          const uint32_t term = Q->term[t];
          if (term >= n_dense) {
            const uint32_t j = term - n_dense;
            const unsigned long long p = sp_find(sp_post, sp_base[j], sp_base[j + 1], doc);
            if (p < sp_base[j + 1] && (uint32_t)sp_post[p] == doc) code = (uint32_t)(sp_post[p] >> 32);
            if (filt && (uint32_t)t < nt && !((code >> BM_SP_FIELD_SHIFT) & filt)) code = 0u;
            // a union scores a doc under the FIRST sparse list of the query that holds it
            if (code && !is_and && (uint32_t)t < s && (uint32_t)t < nt) live = false;
          } else {
            code = dense_find(post, term_base, sub_off, n_sub, term * n_lists + (n_lists - 1u), doc);
            if (code && filt && (uint32_t)t < nt) {
              bool listed = false;
              for (uint32_t f = 0; f + 1u < n_lists; f++)
                if (((filt >> f) & 1u) && !listed) listed = dense_find(post, term_base, sub_off, n_sub, term * n_lists + f, doc) != 0u;
              if (!listed) code = 0u;
            }
            if (code && (uint32_t)t < nt) in_dense = true;
          }
        }
        code &= BM_SP_CODE_MASK;
        if ((uint32_t)t >= nt) {  // NOT terms: a doc found in one is no result (add_result.rs:3440-3497)
          if (code) live = false;
        } else {
          if ((is_and || filt) && !code) live = false;
          if (code) { wv[t] = bm_wdecode(code); pres |= 1u << t; }
        }
      }
      if (live && del && (doc >> 5) < del_words && ((del[doc >> 5] >> (doc & 31u)) & 1u)) live = false;  // add_result.rs:3435
      // exact counts: an intersection's matches; of a union the docs the dense pass cannot have counted
      const unsigned long long cm = __ballot(is_and ? live : (live && !in_dense));
      T.matched += (unsigned long long)__popcll(cm);
      if (k && __ballot(live)) {
        float score = 0.f;
#pragma unroll
        for (int t = 0; t < SS_MAX_QUERY_TERMS; t++)
          if ((uint32_t)t < nt && ((pres >> t) & 1u)) score = fmaf(n_lists > 1u ? idf_scale * Q->idf[t] : Q->idf[t], wv[t], score);  // bm_expand_kernel's idf
        unsigned long long key = (live && score > 0.f) ? (((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xFFFFFFFFu - doc)) : 0ull;
        key = key > T.worst ? key : 0ull;
        if (__ballot(key != 0ull)) T = bm_offer_lane_keys<KPL>(T, key, k, nullptr);
      }
    }
  }
  // the waves' lists -> one: wave 0 merges the others' (sorted) keys into its own
#pragma unroll
  for (int r = 0; r < KPL; r++) wkeys[w][r * 64 + lane] = T.keys[r];
  if (lane == 0) wcount[w] = T.matched;
  __syncthreads();
  if (w != 0) return;
  unsigned long long matched = 0ull;
  for (int j = 0; j < QW; j++) matched += wcount[j];
  for (int j = 1; j < QW; j++) {
#pragma unroll
    for (int r = 0; r < KPL; r++) {
      const unsigned long long key = wkeys[j][r * 64 + lane];
      if (__ballot(key != 0ull)) T.worst = topk_merge64<KPL>(T.keys, key, max(k, 1u), lane);
    }
  }
  unsigned long long* out = out_keys + (size_t)qi * (64 * KPL);
#pragma unroll
  for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
  if (lane == 0) out_extra[qi] = matched;
}

// PHRASES naming a sparse term (QueryType::Phrase over a rare word: the common case of a quoted name).  A phrase is an intersection
// whose survivors pass the position check (add_result.rs:3586-3684, bm25_phrase.hip): the shortest SPARSE list of the phrase drives,
// every other unique term is looked up by binary search -- which yields the doc's posting, hence its positions: a sparse posting's
// in d_sp_pos (sp_pos_end), a dense posting's in the image's pool (pos_off by slot).  Each lane then checks ITS doc: a start offered
// by the first word's positions, every other word by binary search at start + its place (places inside an n-gram key: SS_PHRASE_SKIP).
// PT = uint32_t: several indexed fields -- merged lists, positions tagged with their field, the field filter a test on the start's tag.
template <int KPL, int QW, typename PT>
__global__ void __launch_bounds__(QW * 64) bm25_sparse_phrase_kernel(
    const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base, const uint32_t* __restrict__ sub_off, uint32_t n_sub,
    uint32_t n_dense, uint32_t n_lists, float idf_scale, const unsigned long long* __restrict__ sp_base,
    const unsigned long long* __restrict__ sp_post, const PT* __restrict__ sp_pos, const unsigned long long* __restrict__ sp_pos_end,
    const PT* __restrict__ pos, const uint32_t* __restrict__ pos_off, const unsigned long long* __restrict__ pos_base,
    const ss_bm25_query* __restrict__ qs, uint32_t nq, uint32_t k, const uint32_t* __restrict__ del, uint32_t del_words,
    unsigned long long* __restrict__ out_keys, unsigned long long* __restrict__ out_extra) {
  constexpr int NT = 6;  // unique terms of a phrase (the dense kernel's limit)
  constexpr bool MF = sizeof(PT) == 4;
  __shared__ unsigned long long wkeys[QW][64 * KPL];
  __shared__ unsigned long long wcount[QW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t qi = blockIdx.x;
  const ss_bm25_query* __restrict__ Q = qs + qi;
  const uint32_t nt = min(Q->n_terms, (uint32_t)NT), n_not = bm_q_nnot(Q->op), plen = Q->phrase_len;
  const uint32_t filt = MF ? bm_q_field_filter(Q->op) : 0u, fmask = filt ? filt : 0xFFFFFFFFu;
  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull;
  T.wsc = -1.0f;
  T.matched = 0;
  uint32_t drv = 0;
  {
    unsigned long long best = ~0ull;
    for (uint32_t t = 0; t < nt; t++)
      if (Q->term[t] >= n_dense) {
        const uint32_t i = Q->term[t] - n_dense;
        const unsigned long long len = sp_base[i + 1] - sp_base[i];
        if (len < best) { best = len; drv = t; }
      }
  }
  // place i of the phrase -> unique term (7 = a place inside an n-gram key), 3 bits each
  unsigned long long wpack = 0ull;
#pragma unroll
  for (int i = 0; i < SS_MAX_PHRASE; i++) wpack |= (unsigned long long)(Q->phrase_seq[i] == SS_PHRASE_SKIP ? 7u : (Q->phrase_seq[i] & 7u)) << (3 * i);
  auto wslot = [&](uint32_t i) -> uint32_t { return (uint32_t)(wpack >> (3u * i)) & 7u; };
  if (Q->term[drv] >= n_dense) {
    const uint32_t si = Q->term[drv] - n_dense;
    const unsigned long long b0 = sp_base[si], b1 = sp_base[si + 1];
    for (unsigned long long x = b0 + (unsigned)w * 64u; x < b1; x += 64ull * QW) {
      const bool live0 = x + (unsigned)lane < b1;
      const unsigned long long e = live0 ? sp_post[x + lane] : 0ull;
      const uint32_t doc = (uint32_t)e;
      bool live = live0;
      float wv[NT];
      const PT* pp[NT];  // the doc's positions of every unique term
      uint32_t pn[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        wv[t] = 0.f; pp[t] = sp_pos; pn[t] = 0u;
        if ((uint32_t)t >= nt || !live) continue;
        const uint32_t term = Q->term[t];
        uint32_t code = 0u;
        if (term >= n_dense) {
          unsigned long long p = x + lane;
          if ((uint32_t)t == drv) {
            code = (uint32_t)(e >> 32) & BM_SP_CODE_MASK;
          } else {
            const uint32_t j = term - n_dense;
            p = sp_find(sp_post, sp_base[j], sp_base[j + 1], doc);
            if (p < sp_base[j + 1] && (uint32_t)sp_post[p] == doc) code = (uint32_t)(sp_post[p] >> 32) & BM_SP_CODE_MASK;
          }
          if (code) {
            const unsigned long long st = p ? sp_pos_end[p - 1] : 0ull;
            pp[t] = sp_pos + st;
            pn[t] = (uint32_t)(sp_pos_end[p] - st);
          }
        } else {
          const uint32_t row = term * n_lists + (n_lists - 1u);
          uint32_t slot = 0u;
          code = dense_find(post, term_base, sub_off, n_sub, row, doc, &slot);
          if (code) {
            const uint32_t* po = pos_off + term_base[row] * 4ull;
            const uint32_t st = slot ? po[slot - 1u] : 0u;
            pp[t] = pos + pos_base[row] + st;
            pn[t] = po[slot] - st;
          }
        }
        if (!code) live = false;
        else wv[t] = bm_wdecode(code);
      }
      // NOT terms of either tier (add_result.rs:3440-3497)
      for (uint32_t j = 0; j < n_not && live; j++) {
        const uint32_t term = Q->term[Q->n_terms + j];
        if (term >= n_dense) {
          const uint32_t l = term - n_dense;
          const unsigned long long p = sp_find(sp_post, sp_base[l], sp_base[l + 1], doc);
          if (p < sp_base[l + 1] && (uint32_t)sp_post[p] == doc) live = false;
        } else if (dense_find(post, term_base, sub_off, n_sub, term * n_lists + (n_lists - 1u), doc)) {
          live = false;
        }
      }
      if (live && del && (doc >> 5) < del_words && ((del[doc >> 5] >> (doc & 31u)) & 1u)) live = false;
      if (live) {  // the phrase: start = a position of word 0, word i must sit at start + i
        auto range_of = [&](uint32_t sl, const PT*& base, uint32_t& n) {
          base = pp[0]; n = pn[0];
#pragma unroll
          for (int t = 1; t < NT; t++)
            if (sl == (uint32_t)t) { base = pp[t]; n = pn[t]; }
        };
        const PT* b0p;
        uint32_t n0;
        range_of(wslot(0u), b0p, n0);
        bool match = false;
        for (uint32_t j = 0; j < n0 && !match; j++) {
          const uint32_t start = b0p[j];
          bool ok = !MF || ((fmask >> (start >> BM_POS_FIELD_SHIFT)) & 1u);
          for (uint32_t i = 1; i < plen && ok; i++) {
            if (wslot(i) == 7u) continue;
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
      T.matched += (unsigned long long)__popcll(__ballot(live));
      if (k && __ballot(live)) {
        float score = 0.f;
#pragma unroll
        for (int t = 0; t < NT; t++)
          if ((uint32_t)t < nt) score = fmaf(n_lists > 1u ? idf_scale * Q->idf[t] : Q->idf[t], wv[t], score);
        unsigned long long key = (live && score > 0.f) ? (((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xFFFFFFFFu - doc)) : 0ull;
        key = key > T.worst ? key : 0ull;
        if (__ballot(key != 0ull)) T = bm_offer_lane_keys<KPL>(T, key, k, nullptr);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KPL; r++) wkeys[w][r * 64 + lane] = T.keys[r];
  if (lane == 0) wcount[w] = T.matched;
  __syncthreads();
  if (w != 0) return;
  unsigned long long matched = 0ull;
  for (int j = 0; j < QW; j++) matched += wcount[j];
  for (int j = 1; j < QW; j++) {
#pragma unroll
    for (int r = 0; r < KPL; r++) {
      const unsigned long long key = wkeys[j][r * 64 + lane];
      if (__ballot(key != 0ull)) T.worst = topk_merge64<KPL>(T.keys, key, max(k, 1u), lane);
    }
  }
  unsigned long long* out = out_keys + (size_t)qi * (64 * KPL);
#pragma unroll
  for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
  if (lane == 0) out_extra[qi] = matched;
}

// The answer of a tiered query: dense list (this query's dense terms through the ordinary kernels; none for an intersection or a
// query without dense terms: dense_row = ~0) and the sparse kernel's list, a doc kept once with its larger score, top-k by
// (score desc, doc asc); total = dense total + the sparse kernel's count.  Rows of queries without sparse terms are copied.
// One wave per query.
template <int KPL>
__global__ void __launch_bounds__(SP_WAVES * 64) bm25_tier_merge_kernel(
    uint32_t nq, uint32_t k, uint32_t kk /* row stride of the dense lists, max(k, 1) */, const uint32_t* __restrict__ dense_row /* [nq] row in the dense outputs or ~0 */,
    const uint32_t* __restrict__ sparse_row /* [nq] row in the sparse outputs or ~0 */, const uint32_t* __restrict__ d_doc,
    const float* __restrict__ d_score, const uint32_t* __restrict__ d_count, const unsigned long long* __restrict__ d_total,
    const unsigned long long* __restrict__ sp_keys, const unsigned long long* __restrict__ sp_extra, uint32_t* __restrict__ o_doc,
    float* __restrict__ o_score, uint32_t* __restrict__ o_count, unsigned long long* __restrict__ o_total) {
  __shared__ unsigned long long skeys[SP_WAVES][64 * KPL];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t qi = blockIdx.x * SP_WAVES + w;
  if (qi >= nq) return;
  const uint32_t dr = dense_row[qi], sr = sparse_row[qi];
  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull; T.wsc = -1.f; T.matched = 0;
  unsigned long long total = 0ull;
  if (sr != 0xFFFFFFFFu) {
#pragma unroll
    for (int r = 0; r < KPL; r++) {
      T.keys[r] = sp_keys[(size_t)sr * (64 * KPL) + r * 64 + lane];  // sorted, rank r * 64 + lane
      skeys[w][r * 64 + lane] = T.keys[r];
    }
    total += sp_extra[sr];
    T.worst = topk_finish<KPL>(T.keys, max(k, 1u), lane);
  } else {
#pragma unroll
    for (int r = 0; r < KPL; r++) skeys[w][r * 64 + lane] = 0ull;
  }
  __builtin_amdgcn_wave_barrier();
  if (dr != 0xFFFFFFFFu) {
    total += d_total[dr];
    uint32_t n = d_count[dr];
    if (n == 0xFFFFFFFFu) n = 0;
    n = min(n, k);
    for (uint32_t x = 0; x < n; x += 64) {
      unsigned long long key = 0ull;
      if (x + (uint32_t)lane < n) {
        const uint32_t doc = d_doc[(size_t)dr * kk + x + lane];
        key = ((unsigned long long)__float_as_uint(d_score[(size_t)dr * kk + x + lane]) << 32) | (unsigned long long)(0xFFFFFFFFu - doc);
        const uint32_t lowkey = 0xFFFFFFFFu - doc;
        for (int j = 0; j < 64 * KPL; j++) {  // the doc is in the sparse list as well: that entry carries the full score
          const unsigned long long sk = skeys[w][j];
          if (sk == 0ull) break;
          if ((uint32_t)sk == lowkey) { key = 0ull; break; }
        }
      }
      T.worst = topk_merge64<KPL>(T.keys, key, max(k, 1u), lane);
    }
  }
  uint32_t cnt = 0;
#pragma unroll
  for (int r = 0; r < KPL; r++) {
    const uint32_t rank = r * 64 + lane;
    if (rank < k) {
      const unsigned long long key = T.keys[r];
      o_doc[(size_t)qi * kk + rank] = key ? 0xFFFFFFFFu - (uint32_t)key : 0xFFFFFFFFu;
      o_score[(size_t)qi * kk + rank] = key ? __uint_as_float((uint32_t)(key >> 32)) : 0.f;
    }
    cnt += (uint32_t)__popcll(__ballot(rank < k && T.keys[r] != 0ull));
  }
  if (lane == 0) { o_count[qi] = cnt; o_total[qi] = total; }
}

// Threshold seeds for the dense sub-batch of a tiered batch (ss_api.hip bm25_search_tiered runs the sparse kernel FIRST): a union's sparse
// list holds docs with their FULL scores -- when it holds k of them, its k-th score is one k docs of the query reach, and the query's
// dense terms alone (frequent words, low idf) need not be read below it.  seed[dense_row[i]] = that score, 0 = none.
__global__ void sp_seed_kernel(uint32_t nq, uint32_t k, uint32_t keys_per_list, const uint32_t* __restrict__ dense_row, const uint32_t* __restrict__ sparse_row,
                               const ss_bm25_query* __restrict__ spq, const unsigned long long* __restrict__ sp_keys, float* __restrict__ seed) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const uint32_t dr = dense_row[i], sr = sparse_row[i];
  if (dr == 0xFFFFFFFFu) return;
  float v = 0.f;
  if (sr != 0xFFFFFFFFu && k >= 1u && k <= keys_per_list && bm_q_op(spq[sr].op) == SS_OP_UNION) {
    const unsigned long long key = sp_keys[(size_t)sr * keys_per_list + (k - 1u)];
    if (key) v = __uint_as_float((uint32_t)(key >> 32));
  }
  seed[dr] = v;
}

// The exclusion bitmap of ONE query that excludes sparse terms (ss_api.hip bm25_search_tiered_excl): the bitmap in force (tombstones
// or a facet filter's, or none) with the docs of the query's sparse NOT lists set on top.
struct SpLists { uint32_t id[SS_MAX_QUERY_TERMS]; uint32_t n; };
__global__ void sp_excl_init_kernel(const uint32_t* __restrict__ base, uint32_t base_words, uint32_t* __restrict__ out, uint32_t words) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < words) out[w] = (base && w < base_words) ? base[w] : 0u;
}
// unlisted != 0: only the postings that stand in NO field of that mask (several indexed fields: a sparse posting carries its fields)
__global__ void sp_excl_mark_kernel(const unsigned long long* __restrict__ sp_base, const unsigned long long* __restrict__ sp_post, SpLists L,
                                    uint32_t* __restrict__ out, uint32_t words, uint32_t unlisted) {
  const uint32_t li = blockIdx.y;
  if (li >= L.n) return;
  const unsigned long long b0 = sp_base[L.id[li]], b1 = sp_base[L.id[li] + 1];
  for (unsigned long long x = b0 + blockIdx.x * blockDim.x + threadIdx.x; x < b1; x += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long e = sp_post[x];
    const uint32_t doc = (uint32_t)e;
    if (unlisted && (((uint32_t)(e >> 32) >> BM_SP_FIELD_SHIFT) & unlisted)) continue;
    if ((doc >> 5) < words) atomicOr(out + (doc >> 5), 1u << (doc & 31u));
  }
}

}  // namespace

int ssi_bm25_sparse_excl_bits(const ss_shard* s, const uint32_t* d_base_bits, uint32_t base_words, const uint32_t* lists, uint32_t n_lists,
                              uint32_t* d_out, uint32_t words, hipStream_t st) {
  if (n_lists > (uint32_t)SS_MAX_QUERY_TERMS) return SS_EINVAL;
  SpLists L;
  L.n = n_lists;
  for (uint32_t i = 0; i < (uint32_t)SS_MAX_QUERY_TERMS; i++) L.id[i] = i < n_lists ? lists[i] : 0u;
  for (uint32_t i = 0; i < n_lists; i++)
    if (lists[i] >= s->sp_n) return SS_EINVAL;
  sp_excl_init_kernel<<<(words + 255) / 256, 256, 0, st>>>(d_base_bits, base_words, d_out, words);
  if (n_lists)
    sp_excl_mark_kernel<<<dim3(16, n_lists), 256, 0, st>>>((const unsigned long long*)s->d_sp_base, (const unsigned long long*)s->d_sp_post, L, d_out, words, 0u);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
// d_out |= the docs that hold a listed sparse term in fields outside `filter` only (ss_api.hip bm25_search_gated_scan_rule)
int ssi_bm25_sparse_mark_unlisted(const ss_shard* s, const uint32_t* lists, uint32_t n_lists, uint32_t filter, uint32_t* d_out, uint32_t words, hipStream_t st) {
  if (n_lists > (uint32_t)SS_MAX_QUERY_TERMS || filter == 0u) return SS_EINVAL;
  if (n_lists == 0) return SS_OK;
  SpLists L;
  L.n = n_lists;
  for (uint32_t i = 0; i < (uint32_t)SS_MAX_QUERY_TERMS; i++) L.id[i] = i < n_lists ? lists[i] : 0u;
  for (uint32_t i = 0; i < n_lists; i++)
    if (lists[i] >= s->sp_n) return SS_EINVAL;
  sp_excl_mark_kernel<<<dim3(16, n_lists), 256, 0, st>>>((const unsigned long long*)s->d_sp_base, (const unsigned long long*)s->d_sp_post, L, d_out, words, filter);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_bm25_launch_sparse(const ss_shard* s, const ss_bm25_query* d_q, uint32_t nq, uint32_t k, unsigned long long* d_keys,
                           unsigned long long* d_extra, hipStream_t st) {
  const uint32_t kk = std::max<uint32_t>(k, 1);
  const int KPL = ssi_bm25_sparse_kpl(kk);
  const uint32_t grid = nq;
  if (nq == 0) return SS_OK;
  const uint32_t* del = s->n_deleted ? s->d_deleted : nullptr;
  const float idf_scale = s->bm_n_fields > 1 && s->h_boost.size() == s->bm_n_fields ? s->h_boost[s->bm_n_fields - 1] : 1.0f;
#define SS_SP(KPL_, QW_)                                                                                                                          \
  bm25_sparse_kernel<KPL_, QW_><<<grid, QW_ * 64, 0, st>>>(s->d_post, (const unsigned long long*)s->d_term_base, s->d_sub_off, s->bm_n_sub,        \
                                                         s->bm_n_terms / s->bm_n_fields, s->bm_n_fields, idf_scale,                               \
                                                         (const unsigned long long*)s->d_sp_base,                                                 \
                                                         (const unsigned long long*)s->d_sp_post, s->sp_n, d_q, nq, k, del,                       \
                                                         (uint32_t)s->deleted_words, d_keys, d_extra)
  if (KPL == 1) SS_SP(1, SP_QWAVES);
  else if (KPL == 2) SS_SP(2, SP_QWAVES);
  else if (KPL == 4) SS_SP(4, SP_QWAVES);
  else SS_SP(16, 4);  // (the waves' lists are merged through LDS: 64 KB hold four of 1024 keys)
#undef SS_SP
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// the phrase queries of a tiered batch (rows of the same key / count arrays as ssi_bm25_launch_sparse's)
int ssi_bm25_launch_sparse_phrase(const ss_shard* s, const ss_bm25_query* d_q, uint32_t nq, uint32_t k, unsigned long long* d_keys,
                                  unsigned long long* d_extra, hipStream_t st) {
  if (nq == 0) return SS_OK;
  const uint32_t kk = std::max<uint32_t>(k, 1);
  const int KPL = ssi_bm25_sparse_kpl(kk);
  const uint32_t* del = s->n_deleted ? s->d_deleted : nullptr;
  const float idf_scale = s->bm_n_fields > 1 && s->h_boost.size() == s->bm_n_fields ? s->h_boost[s->bm_n_fields - 1] : 1.0f;
  const bool mf = s->bm_n_fields > 1;
  if (!s->d_sp_pos_end || s->sp_pos_elem != (mf ? 4u : 2u)) return SS_ESTATE;  // the tier carries no positions
#define SS_SPP(KPL_, QW_, PT_, POS_)                                                                                                              \
  bm25_sparse_phrase_kernel<KPL_, QW_, PT_><<<nq, QW_ * 64, 0, st>>>(                                                                            \
      s->d_post, (const unsigned long long*)s->d_term_base, s->d_sub_off, s->bm_n_sub, s->bm_n_terms / s->bm_n_fields, s->bm_n_fields, idf_scale, \
      (const unsigned long long*)s->d_sp_base, (const unsigned long long*)s->d_sp_post, (const PT_*)s->d_sp_pos,                                  \
      (const unsigned long long*)s->d_sp_pos_end, (const PT_*)(POS_), s->d_pos_off, (const unsigned long long*)s->d_pos_base, d_q, nq, k, del,     \
      (uint32_t)s->deleted_words, d_keys, d_extra)
#define SS_SPP2(KPL_, QW_) do { if (mf) SS_SPP(KPL_, QW_, uint32_t, s->d_pos32); else SS_SPP(KPL_, QW_, uint16_t, s->d_pos); } while (0)
  if (KPL == 1) SS_SPP2(1, SP_QWAVES);
  else if (KPL == 2) SS_SPP2(2, SP_QWAVES);
  else if (KPL == 4) SS_SPP2(4, SP_QWAVES);
  else SS_SPP2(16, 4);
#undef SS_SPP2
#undef SS_SPP
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_bm25_launch_tier_merge(uint32_t nq, uint32_t k, const uint32_t* d_dense_row, const uint32_t* d_sparse_row, const uint32_t* d_doc,
                               const float* d_score, const uint32_t* d_count, const unsigned long long* d_total, const unsigned long long* d_keys,
                               const unsigned long long* d_extra, uint32_t* o_doc, float* o_score, uint32_t* o_count, unsigned long long* o_total,
                               hipStream_t st) {
  const uint32_t kk = std::max<uint32_t>(k, 1);
  const int KPL = ssi_bm25_sparse_kpl(kk);
  const uint32_t grid = (nq + SP_WAVES - 1) / SP_WAVES;
#define SS_TM(KPL_)                                                                                                                              \
  bm25_tier_merge_kernel<KPL_><<<grid, SP_WAVES * 64, 0, st>>>(nq, k, kk, d_dense_row, d_sparse_row, d_doc, d_score, d_count, d_total, d_keys,    \
                                                              d_extra, o_doc, o_score, o_count, o_total)
  if (KPL == 1) SS_TM(1);
  else if (KPL == 2) SS_TM(2);
  else if (KPL == 4) SS_TM(4);
  else SS_TM(16);
#undef SS_TM
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_bm25_launch_sparse_seeds(uint32_t nq, uint32_t k, const uint32_t* d_dense_row, const uint32_t* d_sparse_row, const ss_bm25_query* d_spq,
                                 const unsigned long long* d_keys, float* d_seed, hipStream_t st) {
  if (nq == 0 || k == 0) return SS_OK;
  const uint32_t kpl = (uint32_t)ssi_bm25_sparse_kpl(k);
  sp_seed_kernel<<<(nq + 255) / 256, 256, 0, st>>>(nq, k, 64u * kpl, d_dense_row, d_sparse_row, d_spq, d_keys, d_seed);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
