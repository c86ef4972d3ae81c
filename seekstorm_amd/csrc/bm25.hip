// BM25 search: host-side launch logic, merge of the partition-local top-k lists and output formatting.  The scan
// kernels live in bm25_fast.hip (shared device code: bm25_dev.h).  Reference seam replaced: the dispatch block of
// search_lexical_shard (search.rs:3374-3560) down to MinHeap::add_topk (min_heap.rs:1193) and the final
// sort-by-score-descending of the heap array (search.rs:3565-3593).
#include <cstdlib>

#include "bm25_dev.h"


// ---------------------------------------------------------------- merge of partition-local lists (bitonic in LDS)
// in: [nq][n_lists][KS] sorted-desc lists; each workgroup merges `group` consecutive lists of one query into one
// sorted-desc list of KS keys: out [nq][ceil(n_lists/group)][KS].  Only the first `take` keys of a list can reach the
// final top-k (take >= k): the rest is not even read.
// is_last: the LAST pass (one group left): the merged list goes straight to the caller's outputs fin_* (what
// bm25_final_kernel does when there was nothing to merge); fin_doc / fin_score may be null when k == 0 (Count).
// A query that contradicted the caller's ops_mask (bm_expand_kernel) reports count = UINT32_MAX.
__global__ void __launch_bounds__(1024) bm25_merge_kernel(const u64* __restrict__ in, u64* __restrict__ out,
                                                         uint32_t n_lists, uint32_t group, uint32_t KS, uint32_t take,
                                                         const u64* __restrict__ fin_total, uint32_t k, uint32_t is_last,
                                                         uint32_t* __restrict__ fin_doc, float* __restrict__ fin_score,
                                                         uint32_t* __restrict__ fin_count, u64* __restrict__ fin_out_total,
                                                         const uint32_t* __restrict__ tau) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* keys = (u64*)smem;
  const uint32_t q = blockIdx.y, g = blockIdx.x;
  const uint32_t n_groups = (n_lists + group - 1) / group;
  const uint32_t l0 = g * group;
  const uint32_t nl = (l0 + group <= n_lists) ? group : (n_lists - l0);
  const uint32_t n = nl * take;
  uint32_t np = 64;
  while (np < n) np <<= 1;
  const u64* src = in + ((size_t)q * n_lists + l0) * KS;
  for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) keys[i] = i < n ? src[(i / take) * KS + (i % take)] : 0ull;
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
  if (is_last) {  // n_groups == 1
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    uint32_t local = 0;
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
      const u64 key = (i < KS && i < np) ? keys[i] : 0ull;
      uint32_t doc = SS_NO_DOC;
      float sc = 0.f;
      if (key) {
        doc = 0xFFFFFFFFu - (uint32_t)key;
        sc = __uint_as_float((uint32_t)(key >> 32));
        local++;
      }
      fin_doc[(size_t)q * k + i] = doc;
      fin_score[(size_t)q * k + i] = sc;
    }
    if (local) atomicAdd(&cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) {
      fin_count[q] = tau[(size_t)q * BM_TAU_STRIDE + 1] ? 0xFFFFFFFFu : cnt;
      fin_out_total[q] = fin_total[q];
    }
    return;
  }
  u64* dst = out + ((size_t)q * n_groups + g) * KS;
  for (uint32_t i = threadIdx.x; i < KS; i += blockDim.x) dst[i] = i < np ? keys[i] : 0ull;
}

// Small k and P <= 64 lists (the pruned kernel's usual shape: 24 partitions, k = 10): a tournament instead of a sorting
// network -- one wave per query, lane p holding the head of list p, k rounds of a wave-wide maximum; the winning lane
// advances.  Keys are unique (doc ids of different partitions differ).  Writes the caller's outputs like the last merge pass.
__global__ void __launch_bounds__(256) bm25_merge_small_kernel(const u64* __restrict__ in, uint32_t nq, uint32_t n_lists, uint32_t KS,
                                                              const u64* __restrict__ fin_total, uint32_t k,
                                                              uint32_t* __restrict__ fin_doc, float* __restrict__ fin_score,
                                                              uint32_t* __restrict__ fin_count, u64* __restrict__ fin_out_total,
                                                              const uint32_t* __restrict__ tau) {
  const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (q >= nq) return;
  const u64* lst = in + ((size_t)q * n_lists + lane) * KS;
  const bool have = lane < n_lists;
  uint32_t cur = 0;
  u64 head = have ? lst[0] : 0ull;
  u64 mine = 0ull;
  for (uint32_t r = 0; r < k; r++) {
    u64 m = head;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 x = shflx64(m, o); m = x > m ? x : m; }
    if (m == 0ull) break;  // wave-uniform: every list is exhausted
    if (lane == r) mine = m;
    if (have && head == m) {
      cur++;
      head = cur < KS ? lst[cur] : 0ull;
    }
  }
  if (lane < k) {
    fin_doc[(size_t)q * k + lane] = mine ? 0xFFFFFFFFu - (uint32_t)mine : SS_NO_DOC;
    fin_score[(size_t)q * k + lane] = mine ? __uint_as_float((uint32_t)(mine >> 32)) : 0.f;
  }
  const uint32_t n_res = (uint32_t)__popcll(__ballot(mine != 0ull));
  if (lane == 0) {
    fin_count[q] = tau[(size_t)q * BM_TAU_STRIDE + 1] ? 0xFFFFFFFFu : n_res;
    fin_out_total[q] = fin_total[q];
  }
}

__global__ void bm25_final_kernel(const u64* __restrict__ keys, const u64* __restrict__ total, uint32_t KS, uint32_t k,
                                  uint32_t* __restrict__ out_doc, float* __restrict__ out_score,
                                  uint32_t* __restrict__ out_count, u64* __restrict__ out_total, const uint32_t* __restrict__ tau) {
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
    out_count[q] = tau[(size_t)q * BM_TAU_STRIDE + 1] ? 0xFFFFFFFFu : cnt;
    out_total[q] = total[q];
  }
}

// ---------------------------------------------------------------- public queries -> virtual terms (one thread per query)
// `claim` = what the caller of ss_bm25_search_dev asserted about the batch in ops_mask, i.e. what the host chose the kernel
// variants by (BM_CLAIM_*).  A device-resident query cannot be inspected on the host, so the assertion is checked HERE: a
// query that contradicts it (an intersection in a batch declared union-only would run a variant without match counters,
// more terms than declared would overrun the NT-specialised kernel, ...) or that is malformed (term id out of range, no
// terms) is replaced by an empty query and reports d_out_count = UINT32_MAX instead of a silently wrong answer.
constexpr uint32_t BM_CLAIM_AND = 1u, BM_CLAIM_OR = 2u, BM_CLAIM_PROBED = 4u, BM_CLAIM_FREQ = 8u, BM_CLAIM_PHRASE = 16u;
// n_lists = posting lists per term (bm_n_fields); merged != 0: the last of them is the term's MERGED list (ss_common.h
// bm_merged) -- a query without a field filter reads only that one and is a single-field query from here on, a query with a
// field filter reads the n_lists - 1 (term, field) lists.
constexpr uint32_t BM_CLAIM_FILTER = 32u;  // some query of the batch carries a field filter (variants sized for term x field lists)
constexpr uint32_t BM_CLAIM_UNIFORM = 64u;  // every query has exactly np_claim terms (the 16-bit scan's intersection instance)
constexpr uint32_t BM_CLAIM_GATED = 128u;   // some query is a UNION of several terms under a field filter (exact counts then come from the scan)
__global__ void bm_expand_kernel(const ss_bm25_query* __restrict__ q, bm_vquery* __restrict__ vq, uint32_t nq, uint32_t n_lists,
                                 const unsigned long long* __restrict__ term_base, const float* __restrict__ boost,
                                 unsigned long long* __restrict__ total, uint32_t* __restrict__ tau, uint32_t claim,
                                 uint32_t n_vterms, const uint32_t* __restrict__ probe_row, uint32_t merged, uint32_t keep_tau = 0u,
                                 const float* __restrict__ kthw = nullptr, uint32_t kth_sel = 3u,
                                 unsigned long long* __restrict__ best_slots = nullptr, uint32_t best_P = 0u, uint32_t best_KS = 0u) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  // rank 0 of every partition's list doubles as the partition's BEST-SO-FAR key while the many-list scan runs (bm25_scan16m_kernel
  // derives the query's threshold from the k-th largest of them): they start from "none"
  if (best_slots)
    for (uint32_t p_ = 0; p_ < best_P; p_++) best_slots[((size_t)i * best_P + p_) * best_KS] = 0ull;
  total[i] = 0ull;                      // per-query match count and shared threshold start from zero (no separate memset launch)
  if (!keep_tau) tau[(size_t)i * BM_TAU_STRIDE] = 0u;  // (keep_tau: an experiment -- the same batch again with the thresholds it ended on)
                                                       // (a union without exclusions raises it to its seed at the end of the expansion)
  const ss_bm25_query& Q = q[i];  // (read in place: a local copy indexed at run time would live in scratch)
  bm_vquery& V = vq[i];  // written in place: a local copy indexed at run time would live in scratch
  const uint32_t np = Q.n_terms, n_not = bm_q_nnot(Q.op);
  const uint32_t n_fields = n_lists - (merged ? 1u : 0u);                      // indexed fields
  const uint32_t filt = n_fields > 1 ? bm_q_field_filter(Q.op) : 0u;
  const bool q_is_phrase = bm_q_op(Q.op) == SS_OP_PHRASE;
  const bool use_merged = merged != 0u && (filt == 0u || q_is_phrase);         // this query reads the merged lists only (a phrase always:
                                                                               // its filter is a test on the positions' field bits)
  const uint32_t f_begin = use_merged ? n_lists - 1u : 0u, f_end = use_merged ? n_lists : n_fields;  // the lists of a term it reads
  const uint32_t eff_fields = use_merged ? 1u : n_fields;                      // ... as how many fields the rules below see them
  {
    const uint32_t nt_claim = (claim >> 8) & 0xFFu, np_claim = (claim >> 16) & 0xFFu, ff = filt;
    bool bad = np == 0 || np + n_not > (uint32_t)SS_MAX_QUERY_TERMS || (np + n_not) * eff_fields > (uint32_t)BM_MAX_VTERMS;
    bad |= ff != 0u && merged != 0u && !q_is_phrase && !(claim & BM_CLAIM_FILTER);  // the variants were sized for one list per term
    bad |= np + n_not > nt_claim || np > np_claim;
    bad |= (claim & BM_CLAIM_UNIFORM) != 0u && np != np_claim;
    bad |= n_not != 0 && nt_claim == np_claim;  // nt == np declares a batch without NOT terms (unfiltered kernel variants)
    const bool q_and = ((bm_q_op(Q.op) == SS_OP_INTERSECTION || bm_q_op(Q.op) == SS_OP_PHRASE) && np > 1) || ff != 0u;
    const bool q_gated = ff != 0u && bm_q_op(Q.op) == SS_OP_UNION && np > 1;  // a union under a field filter: BM_AND_GATED
    bad |= q_gated && (np > 7u || !(claim & BM_CLAIM_GATED));
    bad |= q_and && !(claim & BM_CLAIM_AND);
    bad |= !q_and && np > 1 && !(claim & BM_CLAIM_OR);
    // a phrase batch holds phrase queries only (one indexed field or merged lists, no NOT terms, 2 .. SS_MAX_PHRASE words naming the unique terms)
    const bool q_phrase = bm_q_op(Q.op) == SS_OP_PHRASE;
    bad |= bm_q_op(Q.op) > (uint32_t)SS_OP_PHRASE || q_phrase != ((claim & BM_CLAIM_PHRASE) != 0u);
    if (q_phrase) {
      bad |= (n_lists != 1 && merged == 0u) || Q.phrase_len < 2u || Q.phrase_len > (uint32_t)SS_MAX_PHRASE || np > 6u;
      // (a word place inside an n-gram key carries SS_PHRASE_SKIP; the key's other component terms are scored, not placed)
      bad |= Q.phrase_seq[0] >= np;
      for (uint32_t j = 1; j < (uint32_t)SS_MAX_PHRASE && j < Q.phrase_len; j++) bad |= Q.phrase_seq[j] >= np && Q.phrase_seq[j] != SS_PHRASE_SKIP;
    }
    bad |= bm_q_all_frequent(Q.op) && bm_q_op(Q.op) == SS_OP_INTERSECTION && np > 1 && ff == 0u && !(claim & BM_CLAIM_FREQ);
    if (!bad)
      for (uint32_t t = 0; t < np + n_not; t++) {
        bad |= Q.term[t] >= n_vterms / n_lists;
        if (!bad && (claim & BM_CLAIM_PROBED) && probe_row)
          for (uint32_t f = f_begin; f < f_end; f++) {
            const uint32_t v = Q.term[t] * n_lists + f;
            bad |= probe_row[v] == BM_NO_PROBE_ROW && term_base[v + 1] != term_base[v];
          }
      }
    tau[(size_t)i * BM_TAU_STRIDE + 1] = bad ? 1u : 0u;
    if (bad) {  // the empty query: one absent term (the all-zero directory row n_vterms, idf 0)
      V.n_terms = 1; V.op = SS_OP_UNION; V.n_groups = 1; V.and_target = 0;
      for (uint32_t j = 0; j < (uint32_t)BM_MAX_VTERMS; j++) { V.term[j] = j ? 0 : n_vterms; V.idf[j] = 0.f; V.and_val[j] = 0; V.group[j] = j ? 0xFF : 0; }
      V.phrase_len = (claim & BM_CLAIM_PHRASE) ? 2u : 0u;  // an empty phrase query: both words are the absent term
      for (int j = 0; j < SS_MAX_PHRASE; j++) V.phrase_seq[j] = 0;
      V.phrase_fields = 0xFFFFFFFFu;
      return;
    }
  }
  // field_filter (several indexed fields): every term must occur in a listed field (add_result.rs:3124-3136) -- an
  // intersection whose match bits only the listed fields' lists may set; a single filtered term is an intersection of one
  const bool is_and = ((bm_q_op(Q.op) == SS_OP_INTERSECTION || bm_q_op(Q.op) == SS_OP_PHRASE) && np > 1) || filt != 0u;
  const bool mask = np <= 8;  // 9-10 terms (single field only, checked on the host): count instead of bits
  // all_terms_frequent (intersection.rs:198-209): the caller saw N > 256 k and df >= N / 2 for every term.  One field and
  // <= 7 terms (bit 7 of the match byte becomes the "some tf < 10" mark; the host refuses the rest).
  // (several indexed fields: over the merged lists only -- a field filter switches the shortcut off, add_result.rs:3116)
  const bool freq = bm_q_all_frequent(Q.op) && bm_q_op(Q.op) == SS_OP_INTERSECTION && np > 1 && np <= 7 && (n_lists == 1 || (use_merged && filt == 0u));
  // a UNION of several terms under a field filter (ss_common.h BM_AND_GATED): a term's listed fields first -- their postings add
  // and set the term's bit --, then its unlisted fields, which add only where the bit is set
  const bool gated = filt != 0u && bm_q_op(Q.op) == SS_OP_UNION && np > 1;
  uint32_t n = 0, n_scored = 0;
  for (uint32_t t = 0; t < np + n_not; t++) {
    if (t == np) n_scored = n;
    for (uint32_t pass = 0; pass < ((gated && t < np) ? 2u : 1u); pass++)
      for (uint32_t f = f_begin; f < f_end; f++) {
        const bool listed = !filt || ((filt >> f) & 1u);
        if (gated && t < np && listed != (pass == 0)) continue;
        const uint32_t v = Q.term[t] * n_lists + f;
        // a term without postings in a field contributes nothing there (one list per term: kept, the zero-length list is harmless)
        if (eff_fields > 1 && term_base[v + 1] == term_base[v]) continue;
        if (n >= (uint32_t)BM_MAX_VTERMS) break;
        V.term[n] = v;
        V.idf[n] = t < np ? (n_lists > 1 ? boost[f] * Q.idf[t] : Q.idf[t]) : 0.f;  // weight * plo.idf, add_result.rs:1253-1261
        if (gated && t < np) V.and_val[n] = (uint8_t)((listed ? 0u : 0x80u) | (1u << t));
        else V.and_val[n] = (is_and && t < np && listed) ? (uint8_t)(mask ? (1u << t) : 0xFFu) : (uint8_t)0;
        V.group[n] = (uint8_t)t;
        n++;
      }
  }
  if (n_not == 0) n_scored = n;
  V.n_terms = n_scored;
  V.op = (filt ? (uint32_t)SS_OP_INTERSECTION : np > 1 ? (bm_q_op(Q.op) == SS_OP_PHRASE ? (uint32_t)SS_OP_INTERSECTION : bm_q_op(Q.op)) : (uint32_t)SS_OP_UNION) |
         ((n - n_scored) << 8);
  V.n_groups = np;
  V.and_target = gated ? (1u | BM_AND_GATED | (np > 2 ? BM_AND_TOUCH : 0u)) : is_and ? ((mask ? (1u << np) - 1u : np) | (freq ? BM_AND_FREQ : 0u)) : 0u;
  for (uint32_t j = n; j < (uint32_t)BM_MAX_VTERMS; j++) { V.term[j] = 0; V.idf[j] = 0.f; V.and_val[j] = 0; V.group[j] = 0xFF; }
  V.phrase_len = bm_q_op(Q.op) == SS_OP_PHRASE ? Q.phrase_len : 0u;
  for (int j = 0; j < SS_MAX_PHRASE; j++) V.phrase_seq[j] = Q.phrase_seq[j];
  V.phrase_fields = (q_is_phrase && filt) ? filt : 0xFFFFFFFFu;
  // threshold seed (bm_kth_kernel): a union -- every list adds to the score of a doc that holds it -- with nothing that could take a
  // doc away again (the caller passes kthw only when no tombstone / filter bitmap is in force; NOT terms, field filters, a phrase's position test --
  // a phrase of ONE unique term is a "union" of one list here --: none of them)
  if (kthw && kth_sel < 3u && !keep_tau && !is_and && !q_is_phrase && !gated && n_not == 0u && filt == 0u && n == n_scored) {
    float seed = 0.f;
    for (uint32_t j = 0; j < n; j++) seed = fmaxf(seed, V.idf[j] * kthw[(size_t)V.term[j] * 4u + kth_sel]);
    tau[(size_t)i * BM_TAU_STRIDE] = __float_as_uint(seed);
  }
}

// a threshold seed from outside the batch's own lists (ss_common.h d_ext_seed): the query's shared threshold starts at least there
__global__ void bm_ext_seed_kernel(const float* __restrict__ seed, uint32_t nq, uint32_t* __restrict__ tau) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const uint32_t b = __float_as_uint(seed[i]);  // (scores are positive floats: their bit patterns order like unsigned integers)
  if (seed[i] > 0.f && b > tau[(size_t)i * BM_TAU_STRIDE]) tau[(size_t)i * BM_TAU_STRIDE] = b;
}

// ---------------------------------------------------------------- threshold seeds: the K-th largest weight of every list
// A union's score is a sum of positive terms, so the K docs holding a list's K largest weights all score at least idf * (the K-th
// largest weight): the query's k-th best score is at least the largest such product over its lists (k <= K) -- known before a single
// posting is read ("quantile" threshold estimation of MaxScore / WAND engines; the reference reaches the same state through its
// block-max ordering, intersection.rs:2225).  One workgroup per list, two passes of a radix select over the 19-bit weight codes
// (high 10 bits, then the low 9 bits inside the three bins the ranks fall into).
constexpr uint32_t KTH_RANKS[3] = {10u, 100u, 1000u};
__global__ void __launch_bounds__(256) bm_kth_kernel(const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base, uint32_t n_terms,
                                                     float* __restrict__ kthw) {
  __shared__ uint32_t hist[1024];
  __shared__ uint32_t hist2[3][512];
  __shared__ uint32_t bin_of[3], rank_in[3];
  const uint32_t t = blockIdx.x;
  if (t >= n_terms) return;
  const u64 d0 = term_base[t] * 4ull, d1 = term_base[t + 1] * 4ull;  // dword range of the list (padding postings are zero)
  for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) hist[i] = 0u;
  for (uint32_t i = threadIdx.x; i < 3u * 512u; i += blockDim.x) (&hist2[0][0])[i] = 0u;
  __syncthreads();
  for (u64 i = d0 + threadIdx.x; i < d1; i += blockDim.x) {
    const uint32_t p = post[i];
    if (p) atomicAdd(&hist[p >> 22], 1u);  // code = p >> 13 (19 bits): its high 10 bits
  }
  __syncthreads();
  if (threadIdx.x < 3u) {
    const uint32_t want = KTH_RANKS[threadIdx.x];
    uint32_t seen = 0u, b = 0xFFFFFFFFu, r = 0u;
    for (int i = 1023; i >= 0; i--) {
      if (seen + hist[i] >= want) { b = (uint32_t)i; r = want - seen; break; }
      seen += hist[i];
    }
    bin_of[threadIdx.x] = b;   // none: the list holds fewer postings
    rank_in[threadIdx.x] = r;  // the rank inside the bin, 1-based from the top
  }
  __syncthreads();
  const uint32_t b0 = bin_of[0], b1 = bin_of[1], b2 = bin_of[2];
  for (u64 i = d0 + threadIdx.x; i < d1; i += blockDim.x) {
    const uint32_t p = post[i];
    if (!p) continue;
    const uint32_t hi = p >> 22, lo = (p >> 13) & 511u;
    if (hi == b0) atomicAdd(&hist2[0][lo], 1u);
    if (hi == b1) atomicAdd(&hist2[1][lo], 1u);
    if (hi == b2) atomicAdd(&hist2[2][lo], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 3u) {
    float w = 0.f;
    if (bin_of[threadIdx.x] != 0xFFFFFFFFu) {
      uint32_t seen = 0u;
      for (int i = 511; i >= 0; i--) {
        seen += hist2[threadIdx.x][i];
        if (seen >= rank_in[threadIdx.x]) { w = bm_weight(((bin_of[threadIdx.x] << 9) | (uint32_t)i) << 13); break; }
      }
    }
    kthw[(size_t)t * 4u + threadIdx.x] = w;
  }
  if (threadIdx.x == 3u) kthw[(size_t)t * 4u + 3u] = 0.f;
}

void ssi_bm25_drop_kth(ss_shard* s) {
  if (s->d_kthw) (void)hipFree(s->d_kthw);
  s->d_kthw = nullptr;
  s->h_kthw.clear();
}
// built once per image, on the stream of the search that first wants it (the image does not change under a search: s->mu);
int ssi_bm25_ensure_kth(ss_shard* s, hipStream_t st) {
  if (s->d_kthw || !s->d_post || s->bm_n_terms == 0) return SS_OK;
  const size_t n = ((size_t)s->bm_n_terms + 1) * 4u;
  SS_HIP(hipMalloc(&s->d_kthw, n * sizeof(float)));
  SS_HIP(hipMemsetAsync(s->d_kthw, 0, n * sizeof(float), st));
  bm_kth_kernel<<<s->bm_n_terms, 256, 0, st>>>(s->d_post, (const unsigned long long*)s->d_term_base, s->bm_n_terms, s->d_kthw);
  SS_HIP(hipGetLastError());
  s->h_kthw.resize(n);
  SS_HIP(hipMemcpyAsync(s->h_kthw.data(), s->d_kthw, n * sizeof(float), hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

// ---------------------------------------------------------------- host side
// The match set of ONE query as a bitmap over the docs (bit d of word d / 64), from the probe index's bit records: what
// facet counting walks (facet.hip).  The caller has checked that every list of the query has a probe row.
int ssi_bm25_match_bits(ss_shard* s, const ss_bm25_query* d_q, unsigned long long* d_bits, unsigned long long* d_total, hipStream_t st, uint32_t nq) {
  if (!s->d_post || !s->d_probe) return SS_ESTATE;
  if (nq == 0 || nq > 64) return SS_EINVAL;
  ss_bm_ws& W = s->bm_ws[st];
  if (64 * sizeof(bm_vquery) > W.vq_cap) {
    SS_HIP(hipStreamSynchronize(st));
    if (W.d_vq) (void)hipFree(W.d_vq);
    W.d_vq = nullptr; W.vq_cap = 0;
    SS_HIP(hipMalloc(&W.d_vq, 64 * sizeof(bm_vquery)));
    W.vq_cap = 64 * sizeof(bm_vquery);
  }
  uint32_t* tau = (uint32_t*)(d_bits);  // the expansion zeroes one threshold line per query: scratch, overwritten below
  bm_expand_kernel<<<1, 128, 0, st>>>(d_q, (bm_vquery*)W.d_vq, nq, s->bm_n_fields, (const unsigned long long*)s->d_term_base,
                                      s->d_boost, d_total, tau, BM_CLAIM_AND | BM_CLAIM_OR | BM_CLAIM_FREQ | BM_CLAIM_FILTER | (0xFFu << 8) | (0xFEu << 16),
                                      s->bm_n_terms, nullptr, s->bm_merged ? 1u : 0u);  // the host entry point validated the query
  BmParams p{};
  p.q = (const bm_vquery*)W.d_vq;
  p.total = d_total;
  p.del = s->n_deleted ? s->d_deleted : nullptr;
  p.del_words = (uint32_t)s->deleted_words;
  p.n_sub = s->bm_n_sub;
  p.nq = nq;  // (nq > 1: d_bits holds nq match sets back to back, d_total nq counts)
  int rc = ssi_bm25_launch_union_count(p, s->d_probe, s->d_probe_row, true, st, d_bits);
  SS_HIP(hipGetLastError());
  return rc;
}

// The partition lists of a batch ([nq][P][KS] sorted keys in bufA; bufB: as much room again) -> the callers' outputs: a tournament per
// query for small k, else the merge tree.  tau: the batch's threshold lines (word 1 of a line = "this query contradicted the batch's
// declaration": its count reads UINT32_MAX).  Shared by every kernel family that answers in partition lists (bm25_gallop.hip too).
int ssi_bm25_merge_lists(u64* bufA, u64* bufB, uint32_t nq, uint32_t P, uint32_t KS, uint32_t k, const u64* total, const uint32_t* tau,
                         uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, hipStream_t st) {
  if (P > 1 && P <= 64 && k >= 1 && k <= 32 && k <= KS) {  // a tournament per query (one wave) instead of the sorting network
    bm25_merge_small_kernel<<<(nq + 3) / 4, 256, 0, st>>>(bufA, nq, P, KS, total, k, d_out_doc, d_out_score, d_out_count,
                                                         (u64*)d_out_total, tau);
    SS_HIP(hipGetLastError());
    return SS_OK;
  }
  // merge tree over the P partition lists
  SS_SET_MAX_LDS(bm25_merge_kernel, 8192 * 8);
  uint32_t lists = P;
  u64 *src = bufA, *dst = bufB;
  uint32_t take = 1;  // power of two >= k (the lists are sorted: deeper entries cannot reach the top-k)
  while (take < std::max<uint32_t>(k, 1u) && take < KS) take <<= 1;
  const uint32_t group = 8192 / take;
  while (lists > 1) {
    uint32_t ng = (lists + group - 1) / group;
    uint32_t np = 64;
    while (np < std::min(lists, group) * take) np <<= 1;
    const bool last = ng == 1;
    // LDS for the keys this pass really sorts (a flat 64 KB request kept all but two workgroups off a CU: 26 -> 12 us per
    // 1000 C2 queries)
    bm25_merge_kernel<<<dim3(ng, nq), std::min<uint32_t>(1024u, std::max<uint32_t>(64u, np / 2)), (size_t)np * sizeof(u64), st>>>(
        src, dst, lists, group, KS, take, total, k, last ? 1u : 0u, d_out_doc, d_out_score, d_out_count, (u64*)d_out_total, tau);
    std::swap(src, dst);
    lists = ng;
    if (last) {
      SS_HIP(hipGetLastError());
      return SS_OK;
    }
  }
  bm25_final_kernel<<<nq, 64, 0, st>>>(src, total, KS, k, d_out_doc, d_out_score, d_out_count,
                                       (u64*)d_out_total, tau);  // P == 1: nothing to merge
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_bm25_search(ss_shard* s, uint32_t nq, const ss_bm25_query* d_q, uint32_t k, uint32_t rt, uint32_t* d_out_doc,
                    float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, bool has_and, bool has_or,
                    uint32_t nt_max, uint32_t np_max, bool all_probed, hipStream_t st, bool any_frequent, bool phrase, bool any_field_filter,
                    bool uniform_terms, bool any_gated, uint32_t nn_max) {
  if (!s->d_post) return SS_ESTATE;
  if (nq == 0) return SS_OK;
  if (k > SS_MAX_K) return SS_EINVAL;
  if (rt == SS_RT_COUNT) k = 0;
  const uint32_t kk = k ? k : 1;
  const int KPL = kk <= 64 ? 1 : kk <= 128 ? 2 : kk <= 256 ? 4 : 16;
  const uint32_t KS = 64 * KPL;
  // Strategy.  Pruned (probe index, bm25_probe.hip): the top-k reads only the essential / shortest lists; exact counts
  // of intersections fall out of it, exact counts of unions are popcounts over the index's bit records
  // (bm25_union_count_kernel).  Exhaustive (bm25_fast.hip): > 4 scored terms, k > 128, no probe index, or
  // SS_BM25_EXHAUSTIVE selected.
  // Several indexed fields: every query term is a union of its (term, field) lists -- the pruned kernel ranks unions of
  // virtual terms as they are; an intersection of unions is left to the scan kernels' match masks.
  // With merged lists (ss_common.h bm_merged) a batch without field filters is a single-field batch to everything below.
  // phrase queries always read the merged lists (their field filter is a test on the positions, bm25_phrase.hip)
  const uint32_t F = (s->bm_merged && (!any_field_filter || phrase)) ? 1u : bm_real_fields(s);
  // nn_max: the most NOT terms of any query (0xFFFFFFFF = unknown, a device-resident batch that did not say: then whatever nt_max
  // leaves room for beside ONE scored term); in lists, like nt_max / np_max below
  if (nn_max == 0xFFFFFFFFu) nn_max = nt_max > np_max ? nt_max - 1u : 0u;
  nt_max *= F;
  np_max *= F;
  nn_max *= F;
  if (F > 1) has_or = true;
  // all_probed: every list the batch touches has a row in the probe index (rows are given to the longest lists first)
  const bool have_probe = s->d_probe && s->d_umax && all_probed;
  // an intersection under the all_terms_frequent shortcut ranks by a per-posting rule (tf >= 10): scan kernels only
  // SS_BM25_EXHAUSTIVE_F32: the exhaustive strategy held on the f32 tile (what the 16-bit scan replaced; measurements and cross-checks)
  const bool exhaustive = s->bm_strategy == SS_BM25_EXHAUSTIVE || s->bm_strategy == SS_BM25_EXHAUSTIVE_F32;
  const bool pruned = !exhaustive && have_probe && !(F > 1 && has_and) && !any_frequent &&
                      np_max >= 1 && np_max <= 4 && KPL <= 2;  // NOT terms are probed outside the template
  if (!pruned && !phrase && s->bm_strategy == SS_BM25_PRUNED) return SS_ENOTSUP;
  // phrase queries: their own kernel over the probe index and the positions (bm25_phrase.hip); every strategy
  // (one indexed field: d_pos; several: the merged lists and d_pos32)
  const bool have_pos = s->bm_n_fields == 1 ? s->d_pos != nullptr : (s->bm_merged && s->d_pos32 != nullptr);
  if (phrase && (!have_probe || !have_pos || KPL > 2 || np_max > 6)) return !have_pos ? SS_ESTATE : SS_ENOTSUP;  // (NOT lists: probed in the kernel)
  // Partitions per query (one wave each).  The grid is a whole number of "rounds" of resident waves: a partially
  // filled last round costs its full duration (measured on C2: 550K q/s at 1.95 rounds vs 477K at 2.44).  Exhaustive
  // scan: 2048 resident waves (LDS-bound), ~2 rounds; pruned: 6144 resident waves, ~4 rounds of shorter assignments
  // balance its more uneven work (driver streams differ 4x in length between queries).
  // Exact counts (Count / TopkCount).  Pruned: the probe kernel counts intersections and single lists while it ranks them,
  // unions are popcounted from the bit records.  Scan kernels under AUTO with a probe index: every count comes from the
  // bit records and the scan runs without its count mode (which scans every sub-block: 5.8 instead of 1.8 ms per 1000 C2
  // queries); a pure Count request then needs no scan at all.  SS_BM25_EXHAUSTIVE keeps the scan's own counts.
  const bool want_counts = rt != SS_RT_TOPK;
  // (a union under a field filter counts by its own rule, BM_AND_GATED / _TOUCH: from the scan)
  const bool bit_counts_all = want_counts && !pruned && !phrase && !exhaustive && have_probe && !any_gated;
  const bool scan_counts = want_counts && !bit_counts_all;
  // unions of <= 4 lists ranked by the scan: the 16-bit-accumulator kernel (16 waves per CU instead of 8)
  // ... and intersections of 2 or 3 terms when the batch holds nothing else: every query an intersection over one list per term
  // with exactly np_max terms, no NOT terms, no all_terms_frequent shortcut
  const uint32_t and_exact_nt = (has_and && !has_or && F == 1 && uniform_terms && !any_frequent && !any_field_filter) ? np_max : 0u;
  const bool scan16 = !pruned && !phrase && s->bm_strategy != SS_BM25_EXHAUSTIVE_F32 && !any_gated && !(F > 1 && has_and) && ssi_bm25_scan16_serves(nn_max, np_max, has_and, scan_counts, s->n_deleted != 0, KPL, k, and_exact_nt);
  // (16-bit scan, round 3: while the longest lists of C2 overflowed their register chunks -- a synchronous load per item for 17 % of
  // the queries -- 4 rounds beat 2 (1.415 against 1.496 ms per 1000 queries); with the chunk budgets following the sorted terms the
  // imbalance is gone and fewer, longer assignments win again: 1.08 ms at 8 partitions per query, 1.10 at 10 / 12, 1.11 at 16, 1.15 at
  // 24 (tools/probes/psweep.sh).  Workgroups made of the partitions of ONE query instead of 8 queries of one partition were tried as
  // well (profiles/r3_map_sweep.log): no better, for the pruned kernel neither.)
  constexpr bool staged_kthb = true;  // (only read by builds with -DPB_STAGED_KTHB=1)
  const bool scan16m = scan16 && (np_max > 6 || (np_max > 4 && KPL == 2));  // the many-list instance (its waves share their best keys)
  const uint32_t resident = (pruned || phrase) ? 6144u : scan16 ? 4096u : 2048u, rounds = (pruned || phrase) ? 4u : 2u;
  uint32_t P = (rounds * resident) / nq;
  // small batches: beyond ~150 waves the probe kernel gains nothing and every extra partition is one more list to merge
  // (single query on C2: 0.128 ms at P = n_sub = 2442, 0.086 ms at P = 128)
  if (pruned || phrase) P = std::min<uint32_t>(P, 160u);
  // Round 4, the pruned kernel re-measured over the batch sizes the coalescer forms and the headline's (tools/probes/small_batch_p.py,
  // pruned_p_sweep.py): at most 64 partitions -- the lists of <= 64 partitions are merged by ONE tournament launch --, and about 4096
  // waves in all, but never fewer than 16 partitions (150 sub-blocks per wave at 10 M docs): host-pointer calls of 8 / 64 / 145 / 256
  // queries 179 -> 159 / 280 -> 229 / 385 -> 290 / 488 -> 390 us, device-resident calls of 256 / 500 / 1000 queries 0.407 -> 0.275 /
  // 0.440 -> 0.39 / 0.625 -> 0.586 ms against the "four rounds, at most 160" rule above.
  // (intersections -- the shortest list drives, the others are probed: shorter assignments balance better -- keep at least 48: 1000
  // 2-term ANDs 1.035 ms at 16 partitions, 0.975 at 24, 0.947 at 48, 0.972 at 64)
  // (heavy queries want more partitions: a head-heavy mix of 4.2 M postings per query 3.68 ms per 1000 at 16 partitions, 3.34 at 48; C2's
  // 1.35 M per query is best at 16 -- where the host has seen the queries, about one partition per 84 K postings)
  if (pruned) {
    uint32_t floor_p = has_and ? 48u : 16u;
    if (s->bm_batch_postings) floor_p = std::max<uint32_t>(floor_p, (uint32_t)std::min<uint64_t>(64u, s->bm_batch_postings / 84000u));
    P = std::max<uint32_t>(floor_p, std::min<uint32_t>(64u, 4096u / std::max<uint32_t>(nq, 1u)));
  }
  if (const char* e = getenv("SS_BM25_P")) P = (uint32_t)atoi(e);  // tuning override
  P = std::max<uint32_t>(1, std::min<uint32_t>(P, s->bm_n_sub));
  const size_t tau_words = (size_t)nq * BM_TAU_STRIDE / 2;  // u64 words: one 128-byte line per query
  // u64 words: 4 floats per (query, partition) + one float per (query, sub-block) -- the pruned kernel's block-max bounds
  const size_t pmax_words = (size_t)nq * P * 2 + ((size_t)nq * s->bm_n_sub + 1) / 2;
  const size_t need = (size_t)nq * P * KS * 2 + nq + tau_words + pmax_words;  // two ping-pong merge buffers + totals + tau + bounds
  ss_bm_ws& W = s->bm_ws[st];  // this stream's workspace: searches on other streams of the shard run beside this one
  if (need > W.part_cap) {
    SS_HIP(hipStreamSynchronize(st));  // earlier searches of this stream may still use the old buffer
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
  float* pmax_ws = (float*)(total + nq + tau_words);

  // queries over (term, field) posting lists
  if ((size_t)nq * sizeof(bm_vquery) > W.vq_cap) {
    SS_HIP(hipStreamSynchronize(st));
    if (W.d_vq) (void)hipFree(W.d_vq);
    W.d_vq = nullptr;
    W.vq_cap = 0;
    SS_HIP(hipMalloc(&W.d_vq, (size_t)nq * sizeof(bm_vquery)));
    W.vq_cap = (size_t)nq * sizeof(bm_vquery);
  }
  if (k && !s->d_kthw) { const int rck = ssi_bm25_ensure_kth(s, st); if (rck) return rck; }
  // nt_max / np_max count (term, field) lists here; the claim is in public terms
  const uint32_t claim = (phrase ? BM_CLAIM_PHRASE : 0u) | (has_and ? BM_CLAIM_AND : 0u) | ((has_or || F > 1) ? BM_CLAIM_OR : 0u) | (all_probed ? BM_CLAIM_PROBED : 0u) |
                         (any_frequent ? BM_CLAIM_FREQ : 0u) | (any_field_filter ? BM_CLAIM_FILTER : 0u) | (uniform_terms ? BM_CLAIM_UNIFORM : 0u) | (any_gated ? BM_CLAIM_GATED : 0u) |
                         (std::min(nt_max / F, 255u) << 8) |
                         (std::min(np_max / F, 255u) << 16);
  bm_expand_kernel<<<(nq + 127) / 128, 128, 0, st>>>(d_q, (bm_vquery*)W.d_vq, nq, s->bm_n_fields,
                                                    (const unsigned long long*)s->d_term_base, s->d_boost, total, tau, claim,
                                                    s->bm_n_terms, s->d_probe_row, s->bm_merged ? 1u : 0u,
                                                    0u /* keep_tau: the ceiling experiment of round 3, tools/probes/tau_ceiling.py */,
                                                    (s->n_deleted || s->del_per_query || k == 0) ? nullptr : s->d_kthw, bm_kth_sel(k),
                                                    (scan16m || (pruned && staged_kthb)) ? bufA : nullptr, P, KS);

  if (s->d_ext_seed && s->ext_seed_n == nq && k) {  // (a sub-batch the tiered search split further runs without: its rows moved)
    bm_ext_seed_kernel<<<(nq + 255) / 256, 256, 0, st>>>(s->d_ext_seed, nq, tau);
    SS_HIP(hipGetLastError());
  }

  BmParams p;
  p.post = s->d_post;
  p.term_base = (const unsigned long long*)s->d_term_base;
  p.sub_off = s->d_sub_off;
  p.q = (const bm_vquery*)W.d_vq;
  p.part_keys = bufA;
  p.total = total;
  p.tau = tau;
  p.del = s->n_deleted ? s->d_deleted : nullptr;
  p.del_words = (uint32_t)s->deleted_words;
  if (s->del_per_query) {  // one exclusion bitmap per query (ss_bm25_search_sorted): the pruned kernel's filtered instances read them
    if (!pruned || phrase || want_counts || !p.del || (p.del_words >> 31)) return SS_ENOTSUP;
    p.del_words |= 0x80000000u;
  }
  p.n_sub = s->bm_n_sub;
  p.n_terms = s->bm_n_terms;
  p.nq = nq;
  p.P = P;
  p.k = k;
  p.count = scan_counts ? 1u : 0u;
  // per-partition block maxima as the pruned kernel's bounds: where the image's maxima vary over the doc ids (set at build),
  // SS_BM25_SUBMAX = 1 / 0 forces them on / off
  static const int force_partmax = [] { const char* e = getenv("SS_BM25_SUBMAX"); return e ? atoi(e) : -1; }();
  // (small batches are launch-latency bound: the two bound kernels cost more than they save -- 0.103 vs 0.118 ms at 32 queries)
  const bool use_partmax = force_partmax < 0 ? (s->bm_partmax && nq >= 128) : force_partmax != 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ssi_prof_begin(s, 0, st, &e0, &e1);
  int rc = SS_OK;
  if (bit_counts_all && k == 0) SS_HIP(hipMemsetAsync(bufA, 0, (size_t)nq * P * KS * sizeof(u64), st));  // no ranking wanted
  else if (phrase)
    rc = ssi_bm25_launch_phrase(p, s->d_probe, s->d_probe_z, s->d_probe_row, s->bm_n_fields == 1 ? s->d_pos : nullptr,
                                s->bm_n_fields == 1 ? nullptr : s->d_pos32, s->d_pos_off, (const unsigned long long*)s->d_pos_base,
                                np_max, KPL, st);
  else rc = pruned ? ssi_bm25_launch_probe(p, s->d_probe, s->d_probe_z, s->d_probe_row, s->d_umax, use_partmax ? s->d_submax : nullptr, pmax_ws, np_max, KPL, nt_max != np_max, st)
                   : scan16 ? ssi_bm25_launch_scan16(p, np_max, nn_max, has_and, KPL, st) : ssi_bm25_launch_scan(p, nt_max, has_and, KPL, st);
  ssi_prof_end(s, 0, st, e0, e1);
  if (rc) return rc;
  if (want_counts && ((pruned && has_or) || bit_counts_all)) {
    const int rcc = ssi_bm25_launch_union_count(p, s->d_probe, s->d_probe_row, bit_counts_all, st);
    if (rcc) return rcc;
  }

  return ssi_bm25_merge_lists(bufA, bufB, nq, P, KS, k, total, tau, d_out_doc, d_out_score, d_out_count, d_out_total, st);
}
