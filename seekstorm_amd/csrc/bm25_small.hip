// Small host-pointer batches (<= 64 queries) of the pruned strategy as ONE launch: the reference's real call shape is one query per
// call (search.rs:1637-1743, one task per shard and query), where the staged pipeline of bm25.hip -- copy the queries in, expand them,
// probe, merge the partition lists, four copies out, synchronise -- is all latency and no throughput (139 us for a
// single query of which the kernels themselves take about 50).  Here:
//
//   * the queries travel IN THE KERNEL ARGUMENTS (60 bytes each: <= 4 scored terms, <= 4 NOT terms; the host has validated and
//     "expanded" them -- one indexed field: there is nothing to expand) -- no copy in, no expansion launch;
//   * a workgroup = 8 partitions of ONE query (pb_wave, the body of bm25_probe_kernel); its 8 lists are merged in LDS;
//   * the LAST workgroup of a query to arrive (one atomic per workgroup) merges the per-workgroup lists and writes the answers
//     straight into the caller's PINNED host buffers; the last query to finish raises a flag word there, which the host thread
//     polls -- no merge launch, no copies out, no completion interrupt;
//   * exact union counts (TopkCount) are further workgroups of the SAME launch popcounting the probe index's bit records
//     (the job of bm25_union_count_kernel), arriving on the same per-query counter;
//   * the per-query state (shared threshold, match count, arrival counter) is left ZERO by whoever consumed it: the next launch
//     starts clean without a memset.
//
// Round 6 -- the TIERED image (a realistic vocabulary: a few thousand dense lists, a million rare ones in the sparse tier,
// bm25_sparse.hip; VERDICT r5 "next" 2: 85 of 96 text-shaped unions and 89 of 96 phrases name a sparse-tier term and used to fall back
// to the staged tiered pipeline of six launches and seven copies).  A query that names a sparse term gets ONE MORE workgroup of the
// same launch (role 3): the body of bm25_sparse_kernel -- every doc of its sparse lists scored in full by galloping lookups
// (bm25_find.h) -- or, for a phrase, of bm25_sparse_phrase_kernel.  Its list joins the query's partition lists; the last arriver's
// tournament drops a doc's dense (partial) entry where the sparse list holds the doc (bm25_tier_merge_kernel's rule), and the count
// is the dense parts' plus the docs role 3 found in no dense list.  Dense roles see the query's dense terms only (compacted on the
// device, in query order: the same fma chain as the staged pipeline's dense sub-query).
//
// Answers are bit-identical to the staged pipeline's: same bodies, same fma chain, same total order of the keys.
#include <cstring>

#include "bm25_probe_body.h"
#include "bm25_find.h"

constexpr uint32_t SM_MAX_Q = 64;   // queries per launch (their 60-byte forms must fit the 4 KB of kernel arguments)
constexpr uint32_t SM_MAX_PB = 64;  // workgroups (of 8 partitions) per query: the final tournament plays one list per lane
constexpr uint32_t SM_MAX_CB = 20;  // counting workgroups per query
constexpr uint32_t SM_MAX_SB = 4;   // role-3 workgroups per query (a sparse list is walked 64 postings per wave and step)
#ifndef SM_G
#define SM_G 8  // chunks of 64 driver postings per group (bm25_probe_body.h)
#endif
#ifndef SM_ARRIVE_ACQREL
#define SM_ARRIVE_ACQREL 0  // 1: acquire / release at agent scope on the arrival counter (see the arrival below)
#endif
#ifndef SM_DBG_SKIP
#define SM_DBG_SKIP 0  // experiment builds (tools/probes/tiered_roles.sh): bit 0 = role 1 does nothing, bit 1 = role 3 does nothing -- where a launch's time goes
#endif

constexpr uint32_t SM_SPARSE = 0x80000000u;  // pb_squery::term: a list of the SPARSE tier (low bits: its index), else the dense list's row
struct pb_squery {
  uint32_t n_terms;      // bits 0..7: scored terms; bits 8..15: words of the phrase (SS_OP_PHRASE)
  uint32_t op;           // SS_OP_* | NOT terms << 8 (bm_q_op / bm_q_nnot); a phrase: its places 10, 11 in bits 16..21 (3 bits each)
  uint32_t term[8];      // scored terms, then the NOT terms
  float idf[4];
  union {
    float thr0;          // threshold seed: a score k docs of the query reach for sure (0 = none; bm_kth_kernel)
    uint32_t places;     // a phrase: places 0 .. 9 -> unique term (7 = a place inside an n-gram key), 3 bits each
  };
};
static_assert(sizeof(pb_squery) == 60, "layout");

struct PbSmall {
  pb_squery q[SM_MAX_Q];
  const uint32_t* post;
  const unsigned long long* term_base;
  const uint32_t* sub_off;
  const uint2* probe;
  const uint32_t* probe_z;
  const uint32_t* probe_row;
  const float* umax;
  const uint32_t* del;
  unsigned long long* part_keys;  // [nq][PB + SB][64 * KPL]
  unsigned long long* total;      // [SM_MAX_Q]      zero between launches
  uint32_t* tau;                  // [SM_MAX_Q][BM_TAU_STRIDE]  zero between launches
  uint32_t* arrive;               // [SM_MAX_Q + 1]  zero between launches; the last word counts finished queries
  uint32_t* out_doc;              // caller's buffers (pinned host memory or device memory)
  float* out_score;
  uint32_t* out_count;
  unsigned long long* out_total;
  unsigned long long* bests;      // [SM_MAX_Q][SM_MAX_PB * 8 * 2] best key (k <= 64) / best two keys (k <= 128) of every partition (zero between launches): pb_publish_kth_best[2]
  uint32_t* flag;                 // pinned host word: = seq when every answer is in place
  // the sparse tier and (phrases) the positions of both tiers: role 3
  const unsigned long long* sp_base;
  const unsigned long long* sp_post;
  const void* pos;                // dense positions: u16 (one indexed field) or u32 (merged lists), by the kernel's TIER
  const uint32_t* pos_off;
  const unsigned long long* pos_base;
  const void* sp_pos;
  const unsigned long long* sp_pos_end;
  uint32_t del_words, n_sub, n_terms, nq, PB, CB, k, count, seq;
  uint32_t SB;                    // role-3 workgroups per query (those of a query without a sparse term leave at once)
};
static_assert(sizeof(PbSmall) <= 4096, "kernel arguments are limited to 4 KB");

typedef __attribute__((address_space(3))) unsigned long long bm_lds_u64;
__device__ __forceinline__ u64 lds_ld64(uint32_t off) { return *(bm_lds_u64*)(uintptr_t)off; }
__device__ __forceinline__ void lds_st64(uint32_t off, u64 v) { *(bm_lds_u64*)(uintptr_t)off = v; }

// Tournament over <= 64 sorted lists (descending, 0 = exhausted): lane p plays list p through ld(rank); rounds of a wave-wide maximum,
// the winning lane(s) advance -- equal keys in two lists win together and place once.  Four entries of every list are fetched ahead,
// so that the rounds do not wait on memory (a list that places more than four keys fetches the next four).  skip(m): the winner takes
// no place (a tiered query's dense entry of a doc its sparse list holds); at most `extra` such rounds.  Returns rank r of the merged
// list in lane r.
template <typename LD, typename SKIP>
__device__ __forceinline__ u64 pb_tournament(bool have, uint32_t k, LD ld, int lane, SKIP skip, uint32_t extra) {
  u64 h0 = have ? ld(0u) : 0ull, h1 = have ? ld(1u) : 0ull, h2 = have ? ld(2u) : 0ull, h3 = have ? ld(3u) : 0ull;
  uint32_t cur = 0, r = 0;
  u64 mine = 0ull;
  for (uint32_t it = 0; r < k && it < k + extra; it++) {
    u64 m = h0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 x = shflx64(m, o); m = x > m ? x : m; }
    if (m == 0ull) break;  // wave-uniform: every list is exhausted
    if (!skip(m)) {
      if ((uint32_t)lane == r) mine = m;
      r++;
    }
    if (h0 == m) {
      h0 = h1; h1 = h2; h2 = h3; h3 = 0ull;
      cur++;
      if ((cur & 3u) == 0u) { h0 = ld(cur); h1 = ld(cur + 1u); h2 = ld(cur + 2u); h3 = ld(cur + 3u); }
    }
  }
  return mine;
}
struct PbNoSkip { __device__ __forceinline__ bool operator()(u64) const { return false; } };

// element i (run-time, wave-uniform) of eight scalars without indexing a register array
__device__ __forceinline__ uint32_t sm_pick8(const uint32_t (&a)[8], uint32_t i) {
  uint32_t r = a[0];
#pragma unroll
  for (int j = 1; j < 8; j++) r = i == (uint32_t)j ? a[j] : r;
  return r;
}

// A doc's posting in a DENSE list through the probe index (every dense list of a one-launch batch has a row): the doc's 64-doc bit
// record says whether the list holds it and how many postings of the group stand before it, the group's z where its first posting lies --
// three dependent loads where the binary search inside the (term, sub-block) segment (bm25_find.h dense_find) takes ten.  Returns the
// posting's weight code (0 = absent); *slot = its index inside the term's image (what d_pos_off is indexed by).
__device__ __forceinline__ uint32_t sm_probe_find(const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base,
                                                 const uint2* __restrict__ probe, const uint32_t* __restrict__ probe_z,
                                                 const uint32_t* __restrict__ probe_row, uint32_t n_sub, uint32_t row, uint32_t doc, bool lanes,
                                                 uint32_t* slot = nullptr) {
  const size_t rb = (size_t)probe_row[row] * n_sub * (BM_SUB / 64);
  const uint32_t gidx = (doc >> BM_SUB_LOG2) * (uint32_t)(BM_SUB / 64) + ((doc & (BM_SUB - 1)) >> 6);
  const uint2 r = probe[rb + (lanes ? gidx : 0u)];
  const u64 bits = ((u64)r.y << 32) | r.x;
  const bool hit = lanes && ((bits >> (doc & 63u)) & 1ull);
  if (!hit) return 0u;
  const uint32_t at = probe_z[rb + gidx] + (uint32_t)__popcll(bits & ((1ull << (doc & 63u)) - 1ull));
  if (slot) *slot = at;
  const uint32_t p = post[term_base[row] * 4ull + at];
  return p >> 13 ? p >> 13 : 1u;
}

// ---- role 3, set queries: the body of bm25_sparse_kernel (bm25_sparse.hip has the story) for a query held in scalars -- tt: scored
// terms then NOT terms (SM_SPARSE | index, or a dense row), idf: of the scored terms (several indexed fields: already scaled).
// A union walks every sparse list of the query (a doc is scored under the FIRST one that holds it), an intersection its shortest one;
// every other term is looked up.  T.matched: an intersection's matches; of a union the docs NO dense list holds (the dense roles count
// the others).
template <int KPL>
__device__ __forceinline__ BmTop<KPL> sm_sparse_wave(const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base,
                                                    const uint2* __restrict__ probe, const uint32_t* __restrict__ probe_z,
                                                    const uint32_t* __restrict__ probe_row, uint32_t n_sub, const unsigned long long* __restrict__ sp_base,
                                                    const unsigned long long* __restrict__ sp_post, const uint32_t (&tt)[8], const float (&idf)[4],
                                                    uint32_t np, uint32_t n_not, bool is_and, uint32_t k, const uint32_t* __restrict__ del,
                                                    uint32_t del_words, uint32_t gw /* this wave among the query's role-3 waves */, uint32_t n_gw,
                                                    uint32_t* tau_q, int lane) {
  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull; T.wsc = -1.0f; T.matched = 0;
  uint32_t first = 0, last = np;
  if (is_and) {
    unsigned long long best = ~0ull;
#pragma unroll
    for (int t = 0; t < 4; t++)
      if ((uint32_t)t < np && (tt[t] & SM_SPARSE)) {
        const uint32_t i = tt[t] & ~SM_SPARSE;
        const unsigned long long len = sp_base[i + 1] - sp_base[i];
        if (len < best) { best = len; first = (uint32_t)t; }
      }
    last = first + 1u;
  }
  for (uint32_t s = first; s < last; s++) {
    const uint32_t ts = sm_pick8(tt, s);
    if (!(ts & SM_SPARSE)) continue;
    const uint32_t si = ts & ~SM_SPARSE;
    const unsigned long long b0 = sp_base[si], b1 = sp_base[si + 1];
    for (unsigned long long x = b0 + (unsigned long long)gw * 64u; x < b1; x += 64ull * n_gw) {
      const bool live0 = x + (unsigned)lane < b1;
      const unsigned long long e = live0 ? sp_post[x + lane] : 0ull;
      const uint32_t doc = (uint32_t)e;
      bool live = live0, in_dense = false;
      float wv[4] = {0.f, 0.f, 0.f, 0.f};
      uint32_t pres = 0u;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        if ((uint32_t)t >= np + n_not) break;
        uint32_t code = 0u;
        if ((uint32_t)t == s) {
          code = (uint32_t)(e >> 32);
        } else if (live) {
          const uint32_t term = tt[t];
          if (term & SM_SPARSE) {
            const uint32_t j = term & ~SM_SPARSE;
            const unsigned long long p = sp_find(sp_post, sp_base[j], sp_base[j + 1], doc);
            if (p < sp_base[j + 1] && (uint32_t)sp_post[p] == doc) code = (uint32_t)(sp_post[p] >> 32);
            if (code && !is_and && (uint32_t)t < s && (uint32_t)t < np) live = false;  // scored under the earlier sparse list
          } else {
            code = sm_probe_find(post, term_base, probe, probe_z, probe_row, n_sub, term, doc, live);
            if (code && (uint32_t)t < np) in_dense = true;
          }
        }
        code &= BM_SP_CODE_MASK;
        if ((uint32_t)t >= np) {  // NOT terms: a doc found in one is no result (add_result.rs:3440-3497)
          if (code) live = false;
        } else {
          if (is_and && !code) live = false;
          if (t < 4 && code) { wv[t < 4 ? t : 0] = bm_wdecode(code); pres |= 1u << t; }
        }
      }
      if (live && del && (doc >> 5) < del_words && ((del[doc >> 5] >> (doc & 31u)) & 1u)) live = false;  // add_result.rs:3435
      T.matched += (unsigned long long)__popcll(__ballot(is_and ? live : (live && !in_dense)));
      if (k && __ballot(live)) {
        float score = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++)
          if ((uint32_t)t < np && ((pres >> t) & 1u)) score = fmaf(idf[t], wv[t], score);
        u64 key = (live && score > 0.f) ? (((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - doc)) : 0ull;
        key = key > T.worst ? key : 0ull;
        // (tau_q: the scores here are FULL scores -- once k docs stand in this wave's list, its k-th is a score k docs of the query reach,
        // and the dense partitions, whose own threshold knows partial scores only, stop reading what cannot reach it)
        if (__ballot(key != 0ull)) T = bm_offer_lane_keys<KPL>(T, key, k, tau_q);
      }
    }
  }
  return T;
}

// ---- role 3, phrases naming a sparse term: the body of bm25_sparse_phrase_kernel for <= 4 unique terms held in scalars.  The shortest
// SPARSE list of the phrase drives; every other unique term is looked up, which yields its posting and so its positions (sparse: the
// tier's pool; dense: the image's); each lane checks the phrase over the positions of ITS doc.  places: 3 bits per place of the phrase
// (7 = a place inside an n-gram key).  PT = uint32_t: several indexed fields -- merged lists, positions tagged with their field.
template <int KPL, typename PT>
__device__ __forceinline__ BmTop<KPL> sm_phrase_wave(const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base,
                                                    const uint2* __restrict__ probe, const uint32_t* __restrict__ probe_z,
                                                    const uint32_t* __restrict__ probe_row, uint32_t n_sub, const unsigned long long* __restrict__ sp_base,
                                                    const unsigned long long* __restrict__ sp_post, const PT* __restrict__ sp_pos,
                                                    const unsigned long long* __restrict__ sp_pos_end, const PT* __restrict__ pos,
                                                    const uint32_t* __restrict__ pos_off, const unsigned long long* __restrict__ pos_base,
                                                    const uint32_t (&tt)[8], const float (&idf)[4], uint32_t np, uint32_t n_not, uint32_t plen,
                                                    unsigned long long places, uint32_t k, const uint32_t* __restrict__ del, uint32_t del_words,
                                                    uint32_t gw, uint32_t n_gw, int lane) {
  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull; T.wsc = -1.0f; T.matched = 0;
  uint32_t drv = 0;
  {
    unsigned long long best = ~0ull;
#pragma unroll
    for (int t = 0; t < 4; t++)
      if ((uint32_t)t < np && (tt[t] & SM_SPARSE)) {
        const uint32_t i = tt[t] & ~SM_SPARSE;
        const unsigned long long len = sp_base[i + 1] - sp_base[i];
        if (len < best) { best = len; drv = (uint32_t)t; }
      }
  }
  const uint32_t td = sm_pick8(tt, drv);
  if (!(td & SM_SPARSE)) return T;  // (the host sends only phrases that name a sparse term)
  auto wslot = [&](uint32_t i) -> uint32_t { return (uint32_t)(places >> (3u * i)) & 7u; };
  const uint32_t si = td & ~SM_SPARSE;
  const unsigned long long b0 = sp_base[si], b1 = sp_base[si + 1];
  for (unsigned long long x = b0 + (unsigned long long)gw * 64u; x < b1; x += 64ull * n_gw) {
    const bool live0 = x + (unsigned)lane < b1;
    const unsigned long long e = live0 ? sp_post[x + lane] : 0ull;
    const uint32_t doc = (uint32_t)e;
    bool live = live0;
    float wv[4];
    const PT* pp[4];  // the doc's positions of every unique term
    uint32_t pn[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      wv[t] = 0.f; pp[t] = sp_pos; pn[t] = 0u;
      if ((uint32_t)t >= np || !live) continue;
      const uint32_t term = tt[t];
      uint32_t code = 0u;
      if (term & SM_SPARSE) {
        unsigned long long p = x + lane;
        if ((uint32_t)t == drv) {
          code = (uint32_t)(e >> 32) & BM_SP_CODE_MASK;
        } else {
          const uint32_t j = term & ~SM_SPARSE;
          p = sp_find(sp_post, sp_base[j], sp_base[j + 1], doc);
          if (p < sp_base[j + 1] && (uint32_t)sp_post[p] == doc) code = (uint32_t)(sp_post[p] >> 32) & BM_SP_CODE_MASK;
        }
        if (code) {
          const unsigned long long st = p ? sp_pos_end[p - 1] : 0ull;
          pp[t] = sp_pos + st;
          pn[t] = (uint32_t)(sp_pos_end[p] - st);
        }
      } else {
        uint32_t slot = 0u;
        code = sm_probe_find(post, term_base, probe, probe_z, probe_row, n_sub, term, doc, live, &slot);
        if (code) {
          const uint32_t* po = pos_off + term_base[term] * 4ull;
          const uint32_t st = slot ? po[slot - 1u] : 0u;
          pp[t] = pos + pos_base[term] + st;
          pn[t] = po[slot] - st;
        }
      }
      if (!code) live = false;
      else wv[t] = bm_wdecode(code);
    }
    // NOT terms of either tier (add_result.rs:3440-3497)
#pragma unroll
    for (int j = 4; j < 8; j++) {
      const uint32_t at = np + (uint32_t)(j - 4);
      if ((uint32_t)(j - 4) >= n_not || !live) continue;
      const uint32_t term = sm_pick8(tt, at);
      if (term & SM_SPARSE) {
        const uint32_t l = term & ~SM_SPARSE;
        const unsigned long long p = sp_find(sp_post, sp_base[l], sp_base[l + 1], doc);
        if (p < sp_base[l + 1] && (uint32_t)sp_post[p] == doc) live = false;
      } else if (sm_probe_find(post, term_base, probe, probe_z, probe_row, n_sub, term, doc, live)) {
        live = false;
      }
    }
    if (live && del && (doc >> 5) < del_words && ((del[doc >> 5] >> (doc & 31u)) & 1u)) live = false;
    if (live) {  // the phrase: start = a position of word 0, word i must sit at start + i (inside the start's field: the tags differ otherwise)
      auto range_of = [&](uint32_t sl, const PT*& base, uint32_t& n) {
        base = pp[0]; n = pn[0];
#pragma unroll
        for (int t = 1; t < 4; t++)
          if (sl == (uint32_t)t) { base = pp[t]; n = pn[t]; }
      };
      const PT* b0p;
      uint32_t n0;
      range_of(wslot(0u), b0p, n0);
      bool match = false;
      for (uint32_t j = 0; j < n0 && !match; j++) {
        const uint32_t start = b0p[j];
        bool ok = true;
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
      for (int t = 0; t < 4; t++)
        if ((uint32_t)t < np) score = fmaf(idf[t], wv[t], score);
      u64 key = (live && score > 0.f) ? (((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - doc)) : 0ull;
      key = key > T.worst ? key : 0ull;
      if (__ballot(key != 0ull)) T = bm_offer_lane_keys<KPL>(T, key, k, nullptr);
    }
  }
  return T;
}

// TIER: 0 = every term of the batch is dense (the round-5 kernel); 1 = role 3 for set queries naming sparse terms; 2 / 3 = ... and for
// phrases, with u16 positions (one indexed field) / u32 (merged lists)
template <int NT, int KPL, bool FILT, int TIER>
__global__ void __launch_bounds__(PB_WAVES * 64, 4) bm25_small_kernel(const PbSmall fz_) {
  // the arguments are read where they lie (the kernel argument segment: constant address space, scalar loads with a run-time
  // index for the query) -- indexing the by-value copy would put it into scratch
  typedef __attribute__((address_space(4))) const PbSmall KArgs;
  KArgs* fz = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t nq = fz->nq, PB = fz->PB, CB = fz->CB, SB = TIER ? fz->SB : 0u, LPQ = PB + SB, k = fz->k, b = blockIdx.x;
  constexpr uint32_t KS = 64u * KPL;
  constexpr uint32_t WREG = pb_qcap(SM_G) * 12u;  // a wave's LDS region (its survivor queue while it probes, its list afterwards)
  // What one workgroup hands to another (partition lists, counts) travels in device-scope atomic accesses -- they meet at the
  // coherence point of the 8 XCDs' L2s by themselves.  No __threadfence(): on this part an agent-scope fence writes back and
  // invalidates the XCD's L2, and one per wave made a batch of 64 TopkCount queries take 595 us instead of 230.
  auto ldk = [&](const u64* p) -> u64 { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto stk = [&](u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // block order: role 3 first (its lists are short and its k-th score is what lets the dense partitions stop early), then roles 1, 2
  const uint32_t b1_ = b - nq * SB;  // (meaningful from role 1 on)
  const bool role3 = TIER && b < nq * SB, role1 = !role3 && b1_ < nq * PB;
  const bool role_list = role3 || role1;  // roles 1 and 3 end in a list of the query
  const uint32_t qi = role3 ? b % nq : role1 ? b1_ % nq : (b1_ - nq * PB) % nq;
  // the query's scalars; its DENSE view (the scored terms of the dense tier, compacted in query order) is what roles 1 and 2 read
  const uint32_t np = fz->q[qi].n_terms & 0xFFu, qop = fz->q[qi].op, n_not = bm_q_nnot(qop);
  uint32_t dm = 0u;  // bit t: scored term t is dense
  if (TIER) {
#pragma unroll
    for (int t = 0; t < 4; t++)
      if ((uint32_t)t < np && !(fz->q[qi].term[t] & SM_SPARSE)) dm |= 1u << t;
  } else {
    dm = (1u << np) - 1u;
  }
  const uint32_t nd = (uint32_t)__popc(dm);
  const bool tiered = TIER && nd != np;                                                  // names a sparse scored term: role 3 answers (part of) it
  const bool q_and = (bm_q_op(qop) == SS_OP_INTERSECTION || bm_q_op(qop) == SS_OP_PHRASE) && np > 1u;
  const bool dense_active = nd != 0u && !(tiered && (q_and || bm_q_op(qop) == SS_OP_PHRASE));  // an intersection with a sparse term is role 3's alone
  // Who counts a union's dense part (TopkCount)?  Two or more dense lists: role 2, from the bit records.  ONE dense list counts itself
  // while it is read -- unless reading it to its end is the price: under tombstones (a bitmap lookup per posting) or beside a sparse role
  // (whose threshold would let the list stop early): then role 2 counts it as well, and the list is free to stop.
  const bool count_by_bits = bm_q_op(qop) == SS_OP_UNION && (nd >= 2u || (nd == 1u && CB != 0u && (tiered || (FILT && fz->del != nullptr))));
  if (role_list) {
    BmTop<KPL> T;
#pragma unroll
    for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
    T.worst = 0ull; T.wsc = -1.0f; T.matched = 0;
    uint32_t slot;  // the list of the query this workgroup writes
    if (role1) {
      // ---- role 1: eight partitions of query qi
      const uint32_t pb = b1_ / nq, part = pb * PB_WAVES + (uint32_t)w;
      slot = pb;
      if (dense_active && !(SM_DBG_SKIP & 1)) {
        PbQueryRegs<NT> Q;
        Q.nt_ = nd;
        Q.op_ = (nd > 1u ? bm_q_op(qop) : (uint32_t)SS_OP_UNION) | (n_not << 8);  // a query of ONE term is always a union (bm_expand_kernel)
#pragma unroll
        for (int j = 0; j < NT; j++) {
          // the j-th dense scored term (TIER = 0: term j itself)
          uint32_t src = (uint32_t)j;
          if (TIER) {
            uint32_t c = 0u;
            src = 4u;
#pragma unroll
            for (int t = 0; t < 4; t++)
              if ((dm >> t) & 1u) { if (c == (uint32_t)j) src = (uint32_t)t; c++; }
          }
          uint32_t term = fz->n_terms;  // absent: the all-zero row
          float idf = 0.f;
#pragma unroll
          for (int t = 0; t < (TIER ? 4 : NT); t++)  // (its place in the query: any of the four when sparse terms stand between)
            if (src == (uint32_t)t && (uint32_t)j < nd) { term = fz->q[qi].term[t]; idf = fz->q[qi].idf[t]; }
          Q.term_[j] = term;
          Q.idf_[j] = idf;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) Q.not_[j] = FILT ? fz->q[qi].term[min(np + (uint32_t)j, 7u)] : 0u;
        T = pb_wave<NT, KPL, FILT, false, true, SM_G, true, TIER != 0>(fz->post, fz->term_base, fz->sub_off, fz->probe, fz->probe_z, fz->probe_row, fz->umax, nullptr, nullptr,
                                                   Q, fz->tau, fz->del, fz->del_words, fz->n_sub, fz->n_terms, PB * PB_WAVES, k, (fz->count & 1u) && !count_by_bits ? 1u : 0u, qi, part, w, lane, fz->q[qi].thr0,
                                                   // (the partitions' best keys: one each for k <= 64, two each -- KPL = 2 -- beyond: pb_publish_kth_best[2])
                                                   (fz->count & 4u) && PB * PB_WAVES * (uint32_t)KPL >= k && (KPL == 2 || k <= 64u) ? fz->bests + (size_t)qi * (SM_MAX_PB * PB_WAVES * 2u) : nullptr);
      }
    } else {
      // ---- role 3: a share of the sparse lists of query qi (a query without a sparse term has none: its workgroups leave, nobody
      // waits for them)
      const uint32_t sbi = b / nq, gw = sbi * PB_WAVES + (uint32_t)w, n_gw = SB * PB_WAVES;
      slot = PB + sbi;
      if (!tiered) return;
      if constexpr (TIER != 0) {
        uint32_t tt[8];
        float idf[4];
#pragma unroll
        for (int t = 0; t < 8; t++) tt[t] = fz->q[qi].term[t];
#pragma unroll
        for (int t = 0; t < 4; t++) idf[t] = fz->q[qi].idf[t];
        const uint32_t* del = FILT ? fz->del : nullptr;
        if (SM_DBG_SKIP & 2) {
        } else if (bm_q_op(qop) != SS_OP_PHRASE) {
          T = sm_sparse_wave<KPL>(fz->post, fz->term_base, fz->probe, fz->probe_z, fz->probe_row, fz->n_sub, fz->sp_base, fz->sp_post, tt, idf, np, n_not, q_and, k, del,
                                  fz->del_words, gw, n_gw, (q_and || !k) ? nullptr : fz->tau + (size_t)qi * BM_TAU_STRIDE, lane);
        } else if constexpr (TIER >= 2) {
          typedef typename std::conditional<TIER == 3, uint32_t, uint16_t>::type PT;
          const unsigned long long places = (unsigned long long)fz->q[qi].places | ((unsigned long long)((qop >> 16) & 0x3Fu) << 30);
          T = sm_phrase_wave<KPL, PT>(fz->post, fz->term_base, fz->probe, fz->probe_z, fz->probe_row, fz->n_sub, fz->sp_base, fz->sp_post, (const PT*)fz->sp_pos, fz->sp_pos_end,
                                      (const PT*)fz->pos, fz->pos_off, fz->pos_base, tt, idf, np, n_not, (fz->q[qi].n_terms >> 8) & 0xFFu, places, k, del,
                                      fz->del_words, gw, n_gw, lane);
        }
      }
    }
    // the workgroup's eight lists -> one (LDS: every wave's queue is empty by now and its region its own)
    const uint32_t lb = (uint32_t)w * WREG;
#pragma unroll
    for (int r = 0; r < KPL; r++) lds_st64(lb + ((uint32_t)r * 64u + (uint32_t)lane) * 8u, T.keys[r]);
    if (lane == 0) lds_st64(lb + KS * 8u, T.matched);
    // every wave's device-scope traffic -- the no-return atomicMax on the query's threshold, its best-key slots -- is PERFORMED before
    // the barrier: wave 0 arrives behind it, so whoever arrives last may zero that state for the next launch without a late atomic
    // landing on top (the workgroup barrier alone does not wait for vmcnt)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (w != 0) return;
    u64 matched = 0ull;
#pragma unroll
    for (int ww = 0; ww < PB_WAVES; ww++) matched += lds_ld64((uint32_t)ww * WREG + KS * 8u);
    u64* mine = fz->part_keys + ((size_t)qi * LPQ + slot) * KS;
    if (k <= 32u) {
      const uint32_t ob = (uint32_t)(lane & 7) * WREG;
      const u64 m = pb_tournament(lane < PB_WAVES, k, [&](uint32_t rank) -> u64 { return rank < KS ? lds_ld64(ob + rank * 8u) : 0ull; }, lane, PbNoSkip{}, 0u);
      stk(mine + lane, m);  // ranks >= k: 0 (KPL = 1: k <= 32 < 64)
      if (KPL == 2) stk(mine + 64 + lane, 0ull);
      if (TIER && role3 && !q_and) {  // k docs with FULL scores: their k-th is a score k docs of the query reach (see sm_sparse_wave)
        const u64 kth = rdlane64(m, (int)k - 1);
        if (kth && lane == 0) bm_publish_tau(fz->tau + (size_t)qi * BM_TAU_STRIDE, __uint_as_float((uint32_t)(kth >> 32)));
      }
    } else {
#pragma unroll 1
      for (int ww = 1; ww < PB_WAVES; ww++) {
        const uint32_t ob = (uint32_t)ww * WREG;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
          const u64 key = lds_ld64(ob + ((uint32_t)r * 64u + (uint32_t)lane) * 8u);
          if (__ballot(key > T.worst)) T = bm_offer_lane_keys<KPL>(T, key > T.worst ? key : 0ull, k, nullptr);
        }
      }
#pragma unroll
      for (int r = 0; r < KPL; r++) stk(mine + r * 64 + lane, T.keys[r]);
    }
    if (lane == 0 && matched) atomicAdd(&fz->total[qi], matched);
  } else {
    // ---- role 2: exact count of a union of >= 2 DENSE lists, popcounted from the bit records (bm25_union_count_kernel's job; the
    // bodies above count intersections, single lists and what the sparse lists add themselves)
    const uint32_t c = b1_ - nq * PB;
    const uint32_t cpart = (c / nq) * PB_WAVES + (uint32_t)w, CP = CB * PB_WAVES;
    if (count_by_bits) {
      const uint32_t n_groups = fz->n_sub * (uint32_t)(BM_SUB / 64);
      const uint32_t g_begin = (uint32_t)(((u64)n_groups * cpart) / CP), g_end = (uint32_t)(((u64)n_groups * (cpart + 1u)) / CP);
      const uint2* rows[8];
      // (a sparse scored term: the all-zero row -- the docs only it holds are role 3's to count; NOT terms of a union are dense)
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const uint32_t term = fz->q[qi].term[(uint32_t)t < np + n_not ? t : 0];
        rows[t] = fz->probe + (size_t)fz->probe_row[(TIER && (term & SM_SPARSE)) ? fz->n_terms : term] * n_groups;
      }
      const uint32_t* __restrict__ del = fz->del;
      const uint32_t del_words = fz->del_words;
      uint32_t cnt = 0;
      constexpr int U = 4;
      for (uint32_t g0 = g_begin; g0 < g_end; g0 += 64u * U) {
        u64 acc[U], neg[U];
#pragma unroll
        for (int u = 0; u < U; u++) { acc[u] = 0ull; neg[u] = 0ull; }
#pragma unroll
        for (int t = 0; t < 8; t++) {
          if ((uint32_t)t >= np + n_not) break;
#pragma unroll
          for (int u = 0; u < U; u++) {
            const uint32_t g = g0 + 64u * u + (uint32_t)lane;
            uint2 r = make_uint2(0u, 0u);
            if (g < g_end) r = rows[t][g];
            const u64 bts = ((u64)r.y << 32) | r.x;
            if ((uint32_t)t < np) acc[u] |= bts; else neg[u] |= bts;
          }
        }
        if (del) {
#pragma unroll
          for (int u = 0; u < U; u++) {
            const uint32_t g = g0 + 64u * u + (uint32_t)lane;  // group g = bitmap words 2g, 2g + 1
            if (g < g_end && 2u * g + 1u < del_words) {
              const uint2 r = ((const uint2*)del)[g];
              neg[u] |= ((u64)r.y << 32) | r.x;
            } else if (g < g_end && 2u * g < del_words) {
              neg[u] |= (u64)del[2u * g];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++) cnt += (uint32_t)__popcll(acc[u] & ~neg[u]);
      }
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
      if (lane == 0) lds_st32((uint32_t)w * WREG, cnt);
      __syncthreads();
      if (w != 0) return;
      uint32_t sum = 0u;
#pragma unroll
      for (int ww = 0; ww < PB_WAVES; ww++) sum += lds_ld32((uint32_t)ww * WREG);
      if (lane == 0 && sum) atomicAdd(&fz->total[qi], (unsigned long long)sum);
    } else if (w != 0) {
      return;
    }
  }

  // ---- arrival: one atomic per workgroup, behind its list and count (s_waitcnt: stores and atomics without return are counted
  // until they are performed); the last workgroup of the query merges and answers
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the compiler keeps the order; the hardware's part is the s_waitcnt)
  __builtin_amdgcn_s_waitcnt(0);
  uint32_t prev = 0u;
#if SM_ARRIVE_ACQREL
  // the memory model's own form: release what this workgroup wrote, acquire what the others released (one fence pair per WORKGROUP).
  // Measured against the relaxed form below (profiles/r6_small_arrive.log); the litmus probe tools/probes/small_litmus.hip is the
  // evidence the relaxed form rests on.
  if (lane == 0) prev = __hip_atomic_fetch_add(&fz->arrive[qi], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
  if (lane == 0) prev = atomicAdd(&fz->arrive[qi], 1u);
#endif
  prev = __builtin_amdgcn_readfirstlane(prev);
  if (prev + 1u != PB + CB + (tiered ? SB : 0u)) return;
  const uint32_t n_lists = PB + (tiered ? SB : 0u);
  const u64* lists = fz->part_keys + (size_t)qi * LPQ * KS;
  uint32_t* o_doc = fz->out_doc + (size_t)qi * k;
  float* o_score = fz->out_score + (size_t)qi * k;
  uint32_t n_res = 0u;
  if (k <= 32u) {
    const u64* lst = lists + (size_t)min((uint32_t)lane, n_lists - 1u) * KS;
    // a tiered union: the dense entry (a partial score) of a doc the sparse list holds takes no place -- the sparse entry carries the
    // full score (bm25_tier_merge_kernel's rule); lane j holds entry j of the sparse list (k <= 32: one key per lane)
    // (<= SM_MAX_SB sparse lists; a doc stands in at most one of them -- each workgroup of role 3 walks its own postings)
    u64 sk[SM_MAX_SB];
#pragma unroll
    for (int i = 0; i < (int)SM_MAX_SB; i++) sk[i] = (TIER && tiered && (uint32_t)i < SB) ? ldk(lists + (size_t)(PB + (uint32_t)i) * KS + (uint32_t)lane) : 0ull;
    auto skip = [&](u64 m) -> bool {
      bool dup = false;
#pragma unroll
      for (int i = 0; i < (int)SM_MAX_SB; i++) dup = dup || (sk[i] != 0ull && (uint32_t)sk[i] == (uint32_t)m && sk[i] != m);
      return TIER && tiered && __ballot(dup) != 0ull;
    };
    const u64 m = pb_tournament((uint32_t)lane < n_lists, k, [&](uint32_t rank) -> u64 { return rank < KS ? ldk(lst + rank) : 0ull; }, lane, skip,
                                (TIER && tiered) ? 64u * SM_MAX_SB : 0u);
    if ((uint32_t)lane < k) {
      o_doc[lane] = m ? 0xFFFFFFFFu - (uint32_t)m : SS_NO_DOC;
      o_score[lane] = m ? __uint_as_float((uint32_t)(m >> 32)) : 0.f;
    }
    n_res = (uint32_t)__popcll(__ballot(m != 0ull));
  } else {
    BmTop<KPL> F;
#pragma unroll
    for (int r = 0; r < KPL; r++) F.keys[r] = 0ull;
    F.worst = 0ull; F.wsc = -1.0f; F.matched = 0;
    // Every list costs KPL full 64-key merges when most of its keys beat the running k-th -- a single query's 64 lists made 150 us of its
    // 200 at k = 100.  The lists arrive SORTED: the k-th largest of their best two keys (<= 128 values, one list per lane) is a key k keys
    // of the answer reach, so nothing below it needs a look -- what is left of a list is the one or two keys it really contributes.
    // (one C2 query end to end, old / new: k = 33 130 / 109 us, 64 152 / 126, 65 200 / 160, 100 217 / 187, 128 232 / 225; the same bound in
    // the workgroups' own merge of their eight lists gains nothing: profiles/r6j_final_merge_bound.log)
    u64 bound = 0ull;
    if (PB * 2u >= k) {
      const u64* lst = lists + (size_t)min((uint32_t)lane, PB - 1u) * KS;
      const u64 b0 = (uint32_t)lane < PB ? ldk(lst) : 0ull, b1 = (uint32_t)lane < PB ? ldk(lst + 1) : 0ull;  // ranks 0 and 1 of list `lane`
      u64 kk2[2] = {0ull, 0ull};
      (void)topk_merge64<2>(kk2, b0, 128u, lane);
      (void)topk_merge64<2>(kk2, b1, 128u, lane);
      bound = k <= 64u ? rdlane64(kk2[0], (int)k - 1) : rdlane64(kk2[1], (int)k - 65);
    }
#pragma unroll 1
    for (uint32_t p = 0; p < PB; p++) {  // (k > 32: the host sends no tiered query here)
#pragma unroll
      for (int r = 0; r < KPL; r++) {
        const u64 key = ldk(lists + (size_t)p * KS + (uint32_t)r * 64u + (uint32_t)lane);
        const bool c = key > F.worst && key >= bound;
        if (__ballot(c)) F = bm_offer_lane_keys<KPL>(F, c ? key : 0ull, k, nullptr);
      }
    }
#pragma unroll
    for (int r = 0; r < KPL; r++) {
      const uint32_t rank = (uint32_t)r * 64u + (uint32_t)lane;
      const u64 m = F.keys[r];
      if (rank < k) {
        o_doc[rank] = m ? 0xFFFFFFFFu - (uint32_t)m : SS_NO_DOC;
        o_score[rank] = m ? __uint_as_float((uint32_t)(m >> 32)) : 0.f;
      }
      n_res += (uint32_t)__popcll(__ballot(m != 0ull && rank < k));
    }
  }
  if (lane == 0) {
    fz->out_count[qi] = n_res;
    fz->out_total[qi] = __hip_atomic_load(&fz->total[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // leave the query's state as the next launch expects it
    fz->total[qi] = 0ull;
    fz->tau[(size_t)qi * BM_TAU_STRIDE] = 0u;
    fz->arrive[qi] = 0u;
  }
  for (uint32_t p_ = (uint32_t)lane; p_ < PB * PB_WAVES * (uint32_t)KPL; p_ += 64u) fz->bests[(size_t)qi * (SM_MAX_PB * PB_WAVES * 2u) + p_] = 0ull;
  __threadfence_system();  // the answers (host memory) before the flag
  if (lane == 0) {
    const uint32_t done = atomicAdd(&fz->arrive[SM_MAX_Q], 1u);
    if (done + 1u == nq) {
      fz->arrive[SM_MAX_Q] = 0u;
      __hip_atomic_store(fz->flag, fz->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------- host side
size_t ssi_bm25_small_ws_bytes() {
  return (size_t)SM_MAX_Q * SM_MAX_PB * 128u * sizeof(u64) + SM_MAX_Q * sizeof(u64) + (size_t)SM_MAX_Q * BM_TAU_STRIDE * 4u + (SM_MAX_Q + 1u) * 4u + 64u +
         (size_t)SM_MAX_Q * SM_MAX_PB * PB_WAVES * 2u * sizeof(u64) + 8u;
}

// can this batch take the one-launch path?  (the caller has classified its queries: ss_api.hip bm25_small_try)
bool ssi_bm25_small_serves(const ss_shard* s, uint32_t nq, uint32_t k, uint32_t np_max, uint32_t nn_max) {
  static const int on = [] { const char* e = getenv("SS_BM25_SMALL"); return e ? atoi(e) : 1; }();
  // (several indexed fields: images with MERGED lists -- a query without a field filter reads one list per term, ss_common.h bm_merged)
  return on && nq >= 1 && nq <= SM_MAX_Q && (s->bm_n_fields == 1 || (s->bm_merged && s->h_boost.size() == s->bm_n_fields)) && s->d_probe && s->d_probe_z && s->d_probe_row && s->d_umax && k >= 1 && k <= 128 &&
         np_max <= 4 && nn_max <= 4 && !s->del_per_query && s->bm_strategy != SS_BM25_EXHAUSTIVE && s->bm_strategy != SS_BM25_EXHAUSTIVE_F32;
}

// ws: ssi_bm25_small_ws_bytes() of device memory, zero when first used.  Outputs / flag: device-visible addresses (pinned host memory).
// sh: what the batch holds (ss_common.h ss_small_shape; the caller has validated every query against it)
int ssi_bm25_small_launch(ss_shard* s, void* ws, uint32_t nq, const ss_bm25_query* hq, uint32_t k, const ss_small_shape& sh, uint32_t* out_doc,
                          float* out_score, uint32_t* out_count, uint64_t* out_total, uint32_t* flag, uint32_t seq, hipStream_t st) {
  if (nq == 0 || nq > SM_MAX_Q || !ws) return SS_EINVAL;
  if ((sh.any_sparse || sh.any_phrase) && (k > 32 || !s->d_sp_base || !s->d_sp_post)) return SS_EINVAL;
  PbSmall a;
  const uint32_t ksel = bm_kth_sel(k);
  const uint32_t L = s->bm_n_fields, n_dense = s->bm_n_terms / L;
  // the list a dense term reads: its only one, or -- several indexed fields -- its MERGED list (the last of the term's L lists), whose
  // weights carry sum_f boost_f w_f / S and whose "boost" S gives the scale back through idf (bm_expand_kernel does the same; the
  // sparse tier's codes are merged weights as well)
  const float scale = L > 1 ? s->h_boost[L - 1] : 1.0f;
  for (uint32_t i = 0; i < nq; i++) {
    const ss_bm25_query& Q = hq[i];
    const uint32_t np = Q.n_terms, n_not = bm_q_nnot(Q.op);
    const bool is_phrase = bm_q_op(Q.op) == SS_OP_PHRASE;
    if (np == 0 || np > 4 || n_not > 4) return SS_EINVAL;
    pb_squery& o = a.q[i];
    o.n_terms = np | (is_phrase ? Q.phrase_len << 8 : 0u);
    // a query of ONE term is always a union (bm_expand_kernel); nothing else of `op` reaches the kernel
    o.op = (is_phrase ? (uint32_t)SS_OP_PHRASE : np > 1 ? bm_q_op(Q.op) : (uint32_t)SS_OP_UNION) | (n_not << 8);
    bool all_dense = true;
    for (uint32_t t = 0; t < 8; t++) {
      if (t >= np + n_not) { o.term[t] = s->bm_n_terms; continue; }  // (absent: the all-zero row)
      const uint32_t term = Q.term[t];
      if (term >= n_dense) { o.term[t] = SM_SPARSE | (term - n_dense); all_dense = all_dense && t >= np; }
      else o.term[t] = term * L + (L - 1u);
    }
    for (uint32_t t = 0; t < 4; t++) o.idf[t] = t < np ? (L > 1 ? scale * Q.idf[t] : Q.idf[t]) : 0.f;
    o.thr0 = 0.f;
    if (is_phrase) {
      if (Q.phrase_len < 2 || Q.phrase_len > (uint32_t)SS_MAX_PHRASE || all_dense) return SS_EINVAL;
      unsigned long long places = 0ull;
      for (uint32_t j = 0; j < Q.phrase_len; j++) places |= (unsigned long long)(Q.phrase_seq[j] == SS_PHRASE_SKIP ? 7u : (Q.phrase_seq[j] & 3u)) << (3u * j);
      o.places = (uint32_t)(places & 0x3FFFFFFFull);
      o.op |= (uint32_t)((places >> 30) & 0x3Full) << 16;
    } else if (ksel < 3u && n_not == 0 && !s->n_deleted && !s->h_kthw.empty() && (np == 1 || bm_q_op(Q.op) == SS_OP_UNION)) {
      // threshold seed: a union with nothing that takes a doc away again (no NOT terms, no tombstones / filter bitmap); over its dense
      // lists -- k docs reach idf * (the k-th largest weight of a list) whatever else they hold
      for (uint32_t t = 0; t < np; t++)
        if (!(o.term[t] & SM_SPARSE)) o.thr0 = std::max(o.thr0, o.idf[t] * s->h_kthw[(size_t)o.term[t] * 4u + ksel]);
    }
  }
  for (uint32_t i = nq; i < SM_MAX_Q; i++) memset(&a.q[i], 0, sizeof(pb_squery));
  const int KPL = k <= 64 ? 1 : 2;
  char* w = (char*)ws;
  a.part_keys = (unsigned long long*)w;
  w += (size_t)SM_MAX_Q * SM_MAX_PB * 128u * sizeof(u64);
  a.total = (unsigned long long*)w;
  w += SM_MAX_Q * sizeof(u64);
  a.tau = (uint32_t*)w;
  w += (size_t)SM_MAX_Q * BM_TAU_STRIDE * 4u;
  a.arrive = (uint32_t*)w;
  w += ((SM_MAX_Q + 1u) * 4u + 64u + 7u) & ~(size_t)7u;
  a.bests = (unsigned long long*)w;
  a.post = s->d_post;
  a.term_base = (const unsigned long long*)s->d_term_base;
  a.sub_off = s->d_sub_off;
  a.probe = s->d_probe;
  a.probe_z = s->d_probe_z;
  a.probe_row = s->d_probe_row;
  a.umax = s->d_umax;
  a.del = s->n_deleted ? s->d_deleted : nullptr;
  a.del_words = (uint32_t)s->deleted_words;
  a.n_sub = s->bm_n_sub;
  a.n_terms = s->bm_n_terms;
  a.nq = nq;
  a.k = k;
  a.sp_base = (const unsigned long long*)s->d_sp_base;
  a.sp_post = (const unsigned long long*)s->d_sp_post;
  a.pos = L > 1 ? (const void*)s->d_pos32 : (const void*)s->d_pos;
  a.pos_off = s->d_pos_off;
  a.pos_base = (const unsigned long long*)s->d_pos_base;
  a.sp_pos = s->d_sp_pos;
  a.sp_pos_end = (const unsigned long long*)s->d_sp_pos_end;
  const int TIER = sh.any_phrase ? (L > 1 ? 3 : 2) : sh.any_sparse ? 1 : 0;
  // role-3 workgroups per query: a step of a wave is ~10 dependent loads for 64 postings -- the longest sparse list of the batch in one or
  // two steps per wave (a workgroup = 512 postings per step)
  a.SB = 0u;
  if (TIER) {
    uint64_t longest = 0;
    for (uint32_t i = 0; i < nq; i++)
      for (uint32_t t = 0; t < hq[i].n_terms; t++)
        if (hq[i].term[t] >= n_dense) { const uint32_t j = hq[i].term[t] - n_dense; longest = std::max<uint64_t>(longest, s->h_sp_base[j + 1] - s->h_sp_base[j]); }
    a.SB = (uint32_t)std::min<uint64_t>(SM_MAX_SB, std::max<uint64_t>(1u, (longest + 767u) / 768u));
  }
  // bit 2: the query's threshold from the partitions' best keys (pb_publish_kth_best).  From 16 queries per call on: 64 queries 164 -> 105 us,
  // 32 queries 112 -> 87 us; a call of 1 / 8 queries -- 512 partitions per query, a handful of groups each -- pays 6 / 13 us for the
  // re-computations and gains nothing (tools/probes/small_fused.py, profiles/r5_small_kth_best.log)
  // 64 < k <= 128 (two keys a partition, pb_publish_kth_best2), C2 unions at k = 100, ms per host-pointer call without / with it:
  // 8 queries 0.247 / 0.220, 16 0.294 / 0.192, 32 0.334 / 0.214, 64 0.449 / 0.233; one query 0.222 either way (profiles/r6i_k100_bests2.log)
#ifndef SM_BESTS2_FROM
#define SM_BESTS2_FROM 8u
#endif
  const bool use_bests = k <= 64u ? nq >= 16u : nq >= SM_BESTS2_FROM;
  a.count = (sh.want_counts ? 1u : 0u) | (use_bests ? 4u : 0u);
  a.seq = seq;
  a.out_doc = out_doc; a.out_score = out_score; a.out_count = out_count; a.out_total = (unsigned long long*)out_total; a.flag = flag;
  // partitions: about 4096 waves in all (the staged path's rule), 16 .. 256 per query; intersections at least 48 (the shortest list
  // drives, shorter assignments balance better); never more than the sub-blocks can feed
  // (measured on C2, tools/probes/small_fused.py: one query 54.9 us at 32 workgroups, 50.6 at 64; 8 queries 71.5 / 68.4; 32 queries best at 16)
  // (with the best-keys threshold 16 .. 31 queries are best at 2048 waves in all: 16 queries 84.5 us at 32 workgroups each, 77.7 at 16)
  uint32_t PB = std::max<uint32_t>(sh.has_and ? 6u : 2u, std::min<uint32_t>(SM_MAX_PB, ((use_bests && nq < 32u && !sh.has_and) ? 256u : 512u) / nq));
  PB = std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(PB, SM_MAX_PB - a.SB), (s->bm_n_sub + PB_WAVES - 1) / PB_WAVES));  // (the tournament plays PB + SB lists)
  a.PB = PB;
  // counting workgroups: a wave per >= 1024 groups of 64 docs, at most 2048 workgroups more in all
  uint32_t CB = 0;
  if (sh.want_counts && sh.has_or) {
    const uint32_t n_groups = s->bm_n_sub * (uint32_t)(BM_SUB / 64);
    CB = std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(SM_MAX_CB, (n_groups / 1024u + PB_WAVES - 1) / PB_WAVES), std::max<uint32_t>(1u, 2048u / nq)));
  }
  a.CB = CB;
  const uint32_t NT = sh.np_max <= 2 ? 2u : sh.np_max;
  const bool filt = sh.any_not || a.del != nullptr || TIER != 0;  // (the tiered instances are built with the filters in)
  const dim3 grid(nq * (PB + CB + a.SB)), block(PB_WAVES * 64);
  const size_t lds = (size_t)PB_WAVES * pb_qcap(SM_G) * 12;
  bool launched = false;
#define SS_S(NT_, KPL_)                                                                                  \
  if (!launched && TIER == 0 && NT == NT_ && KPL == KPL_) {                                              \
    if (filt) bm25_small_kernel<NT_, KPL_, true, 0><<<grid, block, lds, st>>>(a);                        \
    else bm25_small_kernel<NT_, KPL_, false, 0><<<grid, block, lds, st>>>(a);                            \
    launched = true;                                                                                     \
  }
  SS_S(2, 1) SS_S(3, 1) SS_S(4, 1) SS_S(2, 2) SS_S(3, 2) SS_S(4, 2)
#undef SS_S
#define SS_T(NT_, TIER_)                                                                                 \
  if (!launched && TIER == TIER_ && NT == NT_ && KPL == 1) {                                             \
    bm25_small_kernel<NT_, 1, true, TIER_><<<grid, block, lds, st>>>(a);                                 \
    launched = true;                                                                                     \
  }
  SS_T(2, 1) SS_T(3, 1) SS_T(4, 1) SS_T(2, 2) SS_T(3, 2) SS_T(4, 2) SS_T(2, 3) SS_T(3, 3) SS_T(4, 3)
#undef SS_T
  if (!launched) return SS_EINVAL;
  SS_HIP(hipGetLastError());
  return SS_OK;
}
