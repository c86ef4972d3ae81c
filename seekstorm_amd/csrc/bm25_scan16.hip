// Exhaustive BM25 union scan with 16-bit accumulators: top-k (k <= 64) of unions of <= 4 lists, ResultType::Topk, no NOT terms.
// Same reference path as bm25_fast.hip (union_docid_2/3, union_scan, single_blockid with add_result_multiterm_singlefield /
// get_bm25f_multiterm_singlefield / MinHeap::add_topk) and the same answers, bit for bit.
//
// Why a second scan kernel.  bm25_scan_fast_kernel keeps an f32 accumulator per doc of a 4096-doc sub-block: 16 KB of LDS
// per wave, so 8 waves per CU = 2 per SIMD, and at 2 waves per SIMD it is latency bound (at 1 wave per SIMD it takes 1.72x
// as long, while VALU and LDS are each about half busy).  The tile does not have to hold scores: it only has
// to say WHICH docs can reach the threshold.  Here a doc's accumulator is a u16 upper bound of its score in fixed point,
//     q(posting) = trunc(idf * scale * weight') + 1   >   idf * weight * scale,      scale = 65000 / (4 * sum idf)
// (weights are < 4 by their encoding, so a sum stays below 65 535), 8 KB of tile per wave, 16 waves per CU.  A doc whose
// exact score S reaches thr has sum q > S * scale * (1 - 4e-7) >= qthr = trunc(thr * scale * (1 - 1e-5)); the few docs at
// or above qthr (the real candidates and their closest neighbours) are re-scored EXACTLY from the item's postings, which
// are still in registers: the f32 sum in query-term order with the fma chain of the other kernels (s16_trigger).
// The tile is cleared densely (9 wide stores per item), so no address has to be remembered; the last term is only read
// (its sums reach the tile when the item has candidates).
// Structure of the item loop: the candidate path is NOT called from inside the streaming loop -- a hit leaves the loop,
// the candidates are evaluated, and the pipeline is primed again.  With the call inside, the prefetched postings are live
// across it: the allocator parks them in scratch in every item and the compiler's vmcnt bookkeeping merges the two
// histories into a wait for everything (measured: 7x slower).
#include "bm25_dev.h"

namespace {

constexpr int S16_WAVES = 8;                // waves per workgroup; 2 workgroups per CU
constexpr int S16_WAVE_LDS = 9 * 1024;      // [14 B pad][dump u16][4096 u16][pad to 9 KB: the dense clear is 9 full stores]
constexpr float S16_QMAX = 65000.0f;
constexpr float S16_WMAX = 4.0f;            // bm_wdecode(0x7FFFF) < 4

typedef __attribute__((address_space(3))) uint16_t bm_lds_u16;
typedef __attribute__((address_space(3))) u32x4 bm_lds_u32x4;
__device__ __forceinline__ uint32_t lds_ld16(uint32_t off) { return *(bm_lds_u16*)(uintptr_t)off; }
typedef __attribute__((address_space(3))) int16_t bm_lds_i16;
__device__ __forceinline__ uint32_t lds_ld16s(uint32_t off) { return (uint32_t)(int32_t)*(bm_lds_i16*)(uintptr_t)off; }  // sign-extending
__device__ __forceinline__ void lds_st16(uint32_t off, uint32_t v) { *(bm_lds_u16*)(uintptr_t)off = (uint16_t)v; }
__device__ __forceinline__ u32x4 lds_ld128(uint32_t off) { return *(bm_lds_u32x4*)(uintptr_t)off; }
__device__ __forceinline__ void lds_st128(uint32_t off, u32x4 v) { *(bm_lds_u32x4*)(uintptr_t)off = v; }

__device__ __forceinline__ float s16_uniform(float x) {  // a wave-uniform value, pinned to a scalar register
  return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x)));
}
// LDS address of a posting's accumulator: 2 VALU operations (and, shift-add; left alone the compiler shifts, masks and adds)
__device__ __forceinline__ uint32_t s16_addr(uint32_t p, uint32_t accb) {
  uint32_t r;
  asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(r) : "v"(bm_doc_field(p)), "v"(accb));
  return r;
}
// A posting's contribution to its doc's bound, 4 VALU operations: the weight is decoded WITHOUT masking the doc bits that the
// shift leaves below the 19-bit code -- they only make the float larger, by less than 2^-15 relative, and an upper bound is
// all the tile has to hold (the exact pass decodes with bm_weight).  q > idf * weight * scale, and
// sum q <= S * scale * (1 + 2^-15) + NT, which is what S16_SLACK accounts for.
__device__ __forceinline__ uint32_t s16_q(uint32_t p, float fidf) { return (uint32_t)(fidf * __uint_as_float((p >> 5) + BM_W_BASE)) + 1u; }
constexpr uint32_t S16_SLACK = 4u;  // + NT: how far a bound can lie above score * scale (NT roundings up, 65535 * 2^-15 from the decode)
// The unions' form of the same bound, two VALU operations cheaper per posting: the product is added to 2^23 + 1.5 in ONE fma,
// which leaves round(idf * scale * weight' + 1.5) in the low mantissa bits of the result -- an integer in (x + 1, x + 2], still
// an upper bound -- and the float's BITS are used as they are: the low 16 bits are the bound (ds_write_b16 stores just those), and
// since every such value carries the same upper bits 0x4B00, sums "old entry + bits" and their maxima compare like the bounds
// themselves (a tile sum stays below 2^16: no carry).  No v_cvt, no "+ 1".  Each posting can now lie up to 2 above its product
// (the slack of the cut counts 2 NT).
// Round 6: one VALU operation fewer again -- the "+ BM_W_BASE" of the decode is folded into the multiplier.  BM_W_BASE = 113 << 23 only
// moves the exponent: as_float(x + BM_W_BASE) = as_float(x) * 2^113 exactly for every x whose exponent field is not 0, so the unions
// carry idf * scale * 2^113 (S16_WFOLD; <= 16250 * 2^113 = 1.7e38: finite) and multiply the bits of p >> 5 as they are -- the same
// product, the same rounding.  Exponent field 0 (W19 < 2^15: weights below 2^-13, a doc some 20 000 times the average length) reads as
// a denormal, up to 2^-13 * 16250 < 2 below the true product x.  The addend is 2^23 + 3 (the "+ 1.5" above was never representable: a
// float of that size has no half, the literal was 2^23 + 2 all along), so a posting's bound is round(x') + 3 with x - 2 < x' <= x:
// above x by more than 0.5 and by at most 3.5 -- the slack of the cut counts 4 per list (S16_QM_UP; the 2 NT it counted before were
// short of the old form's 2.5 NT from five lists on).  C2 exhaustive: 1.088 -> 1.034 ms per 1000 queries (profiles/r6_qm_fold_ab.log).
constexpr uint32_t S16_MBITS = 0x4B000000u;  // bits of 2^23
#ifndef S16_QM_FOLD
#define S16_QM_FOLD 1  // 0: the round-2 form (experiment builds: A/B of the fold)
#endif
#if S16_QM_FOLD
constexpr float S16_WFOLD = 0x1p113f;
constexpr uint32_t S16_QM_UP = 4u;  // how far a union's bound of ONE posting can lie above idf * weight * scale
__device__ __forceinline__ uint32_t s16_qm(uint32_t p, float fidf_fold) {
  return __float_as_uint(__builtin_fmaf(fidf_fold, __uint_as_float(p >> 5), 8388611.0f));
}
#else
constexpr float S16_WFOLD = 1.0f;
constexpr uint32_t S16_QM_UP = 3u;
__device__ __forceinline__ uint32_t s16_qm(uint32_t p, float fidf) {
  return __float_as_uint(__builtin_fmaf(fidf, __uint_as_float((p >> 5) + BM_W_BASE), 8388610.0f));
}
#endif

// one 256-posting chunk: first = the tile holds nothing of this item yet (no read), keep = read / add / write,
// read = read / add, sums stay in registers (last term)
// CNT (ResultType::TopkCount / Count): the exact size of the union is counted while the bounds are accumulated -- a posting
// whose doc's entry is still 0 is the doc's FIRST posting in this item (every bound is >= 1, the tile is all zero when an item
// starts), so |A u B u ...| = sum over the postings of [entry was 0].  One ballot + scalar popcount per posting step; the NULL
// postings (dump slot) are kept out by p != 0.  The f32 kernel's count mode scans every tile densely instead (5.8 vs 1.4 ms).
// S16_LANE_COUNT: the count of a posting step kept per LANE (one v_addc) and summed over the wave once per partition, instead of a
// ballot + scalar popcount per step (a VALU -> SALU hand-over each)
#ifndef S16_LANE_COUNT
#define S16_LANE_COUNT 1
#endif
#if S16_LANE_COUNT
#define S16_CNT(c) ((c) ? 1u : 0u)
#else
#define S16_CNT(c) ((uint32_t)__popcll(__ballot(c)))
#endif
template <bool CNT>
__device__ __forceinline__ uint32_t s16_first(const u32x4 v, float fidf, uint32_t accb, uint32_t mx, uint32_t& cnt) {
  const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const uint32_t q = s16_qm(pv[x], fidf);
    lds_st16(s16_addr(pv[x], accb), q);
    mx = max(mx, q);
    if (CNT) cnt += S16_CNT(pv[x] != 0u);
  }
  return mx;
}
// SGN (the union count instances under exclusions, EXCL): entries are read SIGN-EXTENDED.  An excluded doc (NOT list, tombstone) carries
// the mark 0x8000 before the terms are accumulated: as a negative number it keeps "old + bound" below every threshold (no trigger),
// it is not zero (no first-touch count), and bounds added to it stay inside 0x8000..0xFFFF (the scale of these instances keeps a
// doc's sum below 2^15) -- an excluded doc rides along untouched by any extra instruction.
template <bool CNT, bool SGN = false>
__device__ __forceinline__ uint32_t s16_read(const u32x4 v, float fidf, uint32_t accb, uint32_t mx, uint32_t (&nw)[4], uint32_t& cnt) {
  const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
  uint32_t old[4];
#pragma unroll
  for (int x = 0; x < 4; x++) old[x] = SGN ? lds_ld16s(s16_addr(pv[x], accb)) : lds_ld16(s16_addr(pv[x], accb));
#pragma unroll
  for (int x = 0; x < 4; x++) {
    nw[x] = old[x] + s16_qm(pv[x], fidf);
    mx = max(mx, nw[x]);
    if (CNT) cnt += S16_CNT(old[x] == 0u && pv[x] != 0u);
  }
  return mx;
}
// (An LDS atomic for the middle term -- ds_add_rtn_u32 on the dword holding the 16-bit entry, one operation instead of read +
// write -- was measured: 1.78 ms against 1.08 ms per 1000 C2 queries.  Returning LDS atomics run far below the plain pipe's rate.)
template <bool CNT, bool SGN = false>
__device__ __forceinline__ uint32_t s16_keep(const u32x4 v, float fidf, uint32_t accb, uint32_t mx, uint32_t& cnt) {
  uint32_t nw[4];
  mx = s16_read<CNT, SGN>(v, fidf, accb, mx, nw, cnt);
  const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int x = 0; x < 4; x++) lds_st16(s16_addr(pv[x], accb), nw[x]);
  return mx;
}
__device__ __forceinline__ void s16_clear(uint32_t wb, int lane) {
#pragma unroll
  for (int i = 0; i < S16_WAVE_LDS / 1024; i++) lds_st128(wb + (uint32_t)(i * 64 + lane) * 16u, u32x4{0u, 0u, 0u, 0u});
}

// ---- exclusions: NOT terms (add_result.rs:3440-3497, union.rs:483-530: the docs of a NOT list are zeroed in the scan table) and
// tombstones (add_result.rs:3435, union.rs:975).  An excluded doc's entry goes back to 0 -- it is then no candidate, and in the count
// instances (EXCL) an entry that was NOT zero had been counted by its first posting: one decrement.
template <bool CNT>
__device__ __forceinline__ void s16_not_chunk(const u32x4 v, uint32_t accb, uint32_t& dec) {
  const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
  uint32_t ad[4], old[4];
#pragma unroll
  for (int x = 0; x < 4; x++) { ad[x] = s16_addr(pv[x], accb); old[x] = lds_ld16(ad[x]); }
#pragma unroll
  for (int x = 0; x < 4; x++) {
    if (CNT) dec += (uint32_t)__popcll(__ballot(old[x] != 0u && pv[x] != 0u));  // (a NULL posting reads the dump slot)
    lds_st16(ad[x], 0u);
  }
}
constexpr uint32_t S16_MARK = 0x8000u;  // an excluded doc's entry in the union count instances (s16_read<.., SGN>)
constexpr float S16_QMAX_EXCL = 32000.0f;  // ... whose bounds must leave bit 15 alone
__device__ __forceinline__ void s16_mark_chunk(const u32x4 v, uint32_t accb) {
  lds_st16(s16_addr(v.x, accb), S16_MARK);
  lds_st16(s16_addr(v.y, accb), S16_MARK);
  lds_st16(s16_addr(v.z, accb), S16_MARK);
  lds_st16(s16_addr(v.w, accb), S16_MARK);
}
__device__ __forceinline__ void s16_mark_bits(uint32_t w0, uint32_t w1, uint32_t accb, int lane) {
  u64 m = (u64)w0 | ((u64)w1 << 32);
  while (__ballot(m != 0ull)) {
    const bool act = m != 0ull;
    const uint32_t b = act ? (uint32_t)__ffsll((long long)m) - 1u : 0u;
    m &= m - 1ull;
    lds_st16(act ? accb + (((uint32_t)lane * 64u + b + 1u) << 1) : accb, S16_MARK);
  }
}
// the tombstone bits of the 64 docs this lane owns in the sub-block (docs 64 lane .. 64 lane + 63): lanes walk their set bits
// together, a lane that has run out parks on the dump slot
template <bool CNT>
__device__ __forceinline__ void s16_del_bits(uint32_t w0, uint32_t w1, uint32_t accb, int lane, uint32_t& dec) {
  u64 m = (u64)w0 | ((u64)w1 << 32);
  while (__ballot(m != 0ull)) {
    const bool act = m != 0ull;
    const uint32_t b = act ? (uint32_t)__ffsll((long long)m) - 1u : 0u;
    m &= m - 1ull;
    const uint32_t ad = act ? accb + (((uint32_t)lane * 64u + b + 1u) << 1) : accb;
    const uint32_t old = lds_ld16(ad);
    if (CNT) dec += (uint32_t)__popcll(__ballot(act && old != 0u));
    lds_st16(ad, 0u);
  }
}
// What the candidate path knows of a query's exclusions: everything is fetched when needed (the path is rare), nothing of it lives
// in registers across the streaming loop.
struct S16Excl {
  const bm_vquery* Q;              // NOT lists: Q->term[nt .. nt + nn)
  const uint32_t* post;
  const unsigned long long* term_base;
  const uint32_t* sub_off;
  const uint32_t* del;             // tombstone bitmap or null
  uint32_t nt, nn, row_len, del_words, item;
};
__device__ __forceinline__ void s16_exclude(const S16Excl& e, uint32_t accb, int lane) {
  uint32_t nodec = 0u;
  for (uint32_t j = 0; j < e.nn; j++) {
    const uint32_t term = e.Q->term[e.nt + j];
    const uint32_t* row = e.sub_off + (size_t)term * e.row_len;
    const uint32_t b0 = row[e.item], b1 = row[e.item + 1];
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(e.post + e.term_base[term] * 4ull), 0, (int)(b1 << 4), BM_RSRC_FLAGS);
    for (uint32_t u = b0; u < b1; u += 64u) s16_not_chunk<false>(__builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (int)(u << 4), 0), accb, nodec);
  }
  if (e.del) {
    const uint32_t w = e.item * (uint32_t)(BM_SUB / 32) + (uint32_t)lane * 2u;
    const uint32_t w0 = w < e.del_words ? e.del[w] : 0u, w1 = w + 1u < e.del_words ? e.del[w + 1u] : 0u;
    s16_del_bits<false>(w0, w1, accb, lane, nodec);
  }
}

// ---- intersections (AND template instance; 2 or 3 terms, every query of the batch with exactly NT terms).  An entry is
// (bound << 2) | level: level = how many of the query's terms the doc has been found in so far (1, 2), 3 = in ALL of them.
// Only the FIRST term creates entries; a later term adds to an entry only if the doc was in every term before it, the last
// term only reads: a doc matches when its entry stands at level NT - 1, and its bound is the entry's + the last posting's.
// Matches are counted on the way (exact intersection counts for TopkCount / Count).  Bounds: q as for unions at a quarter of
// the scale (14 bits).
constexpr float S16_QMAX_AND = 16000.0f;
__device__ __forceinline__ void s16a_first(const u32x4 v, float fidf, uint32_t accb) {
  const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int x = 0; x < 4; x++) lds_st16(s16_addr(pv[x], accb), (s16_q(pv[x], fidf) << 2) | 1u);
}
__device__ __forceinline__ void s16a_mid(const u32x4 v, float fidf, uint32_t accb, uint32_t level) {
  const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
  uint32_t ad[4], old[4];
#pragma unroll
  for (int x = 0; x < 4; x++) { ad[x] = s16_addr(pv[x], accb); old[x] = lds_ld16(ad[x]); }
#pragma unroll
  for (int x = 0; x < 4; x++) lds_st16(ad[x], (old[x] & 3u) == level ? old[x] + (s16_q(pv[x], fidf) << 2) + 1u : old[x]);
}
// WRITE = false: the streaming loop -- the largest bound of a matching doc and the number of matches; true: the candidate path
// -- a matching doc's entry becomes (bound << 2) | 3
template <bool CNT, bool WRITE>
__device__ __forceinline__ uint32_t s16a_last(const u32x4 v, float fidf, uint32_t accb, uint32_t level, uint32_t mx, uint32_t& cnt) {
  const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
  uint32_t ad[4], old[4];
#pragma unroll
  for (int x = 0; x < 4; x++) { ad[x] = s16_addr(pv[x], accb); old[x] = lds_ld16(ad[x]); }
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const bool m = (old[x] & 3u) == level && pv[x] != 0u;  // (a NULL posting reads whatever the dump slot holds)
    const uint32_t sum = (old[x] >> 2) + s16_q(pv[x], fidf);
    mx = max(mx, m ? sum : 0u);
    if (CNT) cnt += S16_CNT(m);
    if (WRITE) lds_st16(ad[x], m ? ((sum << 2) | 3u) : old[x]);
  }
  return mx;
}
// the entries of the first term's docs that did not make it to "in all terms" go back to 0
__device__ __forceinline__ void s16a_clean(const u32x4 v, uint32_t accb) {
  const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
  uint32_t ad[4], old[4];
#pragma unroll
  for (int x = 0; x < 4; x++) { ad[x] = s16_addr(pv[x], accb); old[x] = lds_ld16(ad[x]); }
#pragma unroll
  for (int x = 0; x < 4; x++) lds_st16(ad[x], (old[x] & 3u) == 3u ? old[x] : 0u);
}

constexpr uint32_t S16_LIST = 8320u;   // per-wave candidate list inside the slice's padding: up to S16_LIST_MAX doc-in-sub-block ids (u16)
constexpr uint32_t S16_LIST_MAX = 64u;
constexpr uint32_t S16_ACC = 8448u;    // their exact f32 scores while an item's candidates are evaluated

typedef unsigned short s16_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t s16_pkmax(uint32_t a, uint32_t b) {  // v_pk_max_u16
  const s16_u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16_u16x2, a), __builtin_bit_cast(s16_u16x2, b));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t s16_max8(const u32x4 e) {  // the largest of the eight bounds of one slot
  const uint32_t m = s16_pkmax(s16_pkmax(e.x, e.y), s16_pkmax(e.z, e.w));
  return max(m & 0xFFFFu, m >> 16);
}

// the accumulators alone (dump slot + 4096 entries: bytes 0 .. 8207 of the slice), leaving the candidate list / scores
__device__ __forceinline__ void s16_clear_tile(uint32_t wb, int lane) {
#pragma unroll
  for (int i = 0; i < 8; i++) lds_st128(wb + (uint32_t)(i * 64 + lane) * 16u, u32x4{0u, 0u, 0u, 0u});
  if (lane == 0) lds_st128(wb + 8192u, u32x4{0u, 0u, 0u, 0u});
}

// Chunks (256 postings = 64 lanes x 16 B) held in registers per term and item.  The terms are processed SORTED BY LIST LENGTH
// (shortest first; exact sums keep the query's order through qpos), so the budget follows the position: a 3-term union keeps
// 1 / 1 / 3 chunks -- 5 x 4 registers per buffer where round 2 kept 2 / 2 / 2 = 6 x 4 -- and the longest list of a C2 query (up to
// 15 % of a sub-block = 614 postings) fits without the synchronous remainder loop, which used to cost every item of the 17 % of
// the queries whose third term lies above 12.5 % a full memory round trip.
// Two terms: 1 / 4 (the 2-term intersections of the bench -- a 1-5 % list with a 5-20 % one -- went from 1.55 to 1.32 ms per 1000
// queries against 2 / 3, unions of two terms are unchanged); four terms: 1 / 1 / 1 / 3 (C2's three terms + a 0.2-0.5 % one: 1.88 ->
// 1.34 ms against 2 chunks per term, same answers).  S16_NT4_LAST / S16_NT1: experiment knobs (tools/probes/scan16_ab.py).
#ifndef S16_NT4_LAST
#define S16_NT4_LAST 3  // 0: 2 chunks per term; n: 1 / 1 / 1 / n
#endif
#ifndef S16_NT1
#define S16_NT1 2
#endif
// The part of a segment beyond its register chunks, streamed.  S16_REST4: four 64-lane chunks in flight at a time instead of one load
// per step (a dense segment is then no chain of dependent HBM round trips).  (Loads past the segment's end return NULL postings; they
// are not processed.)
#ifndef S16_REST4
#define S16_REST4 0
#endif
template <typename F>
__device__ __forceinline__ void s16_rest(__amdgpu_buffer_rsrc_t rs, int lane16, uint32_t u0, uint32_t u1, F f) {
#if S16_REST4
  for (uint32_t u = u0; u < u1; u += 256u) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (int)(u << 4), 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 1024, (int)(u << 4), 0);
    const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 2048, (int)(u << 4), 0);
    const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 3072, (int)(u << 4), 0);
    f(a);
    if (u + 64u < u1) f(b);
    if (u + 128u < u1) f(c);
    if (u + 192u < u1) f(d);
  }
#else
  for (uint32_t u = u0; u < u1; u += 64u) f(__builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (int)(u << 4), 0));
#endif
}

template <int NT> struct S16Cfg {
  static constexpr int cpt(int t) {
    return NT >= 5 ? (t == NT - 1 ? 2 : 1)  // five / six lists: 1 ... 1 / 2 (6 / 7 chunks: the registers of 2 per term would not fit)
           : NT == 3 ? (t == 2 ? 3 : 1) : NT == 2 ? (t == 1 ? 4 : 1) : NT == 1 ? S16_NT1 : (S16_NT4_LAST ? (t == NT - 1 ? S16_NT4_LAST : 1) : 2);
  }
  static constexpr int off(int t) { int o = 0; for (int i = 0; i < t; i++) o += cpt(i); return o; }
  static constexpr int RC = off(NT);
  static constexpr int CPTMAX = 4;
};
template <int RC> struct S16Cur { u32x4 v[RC]; };
template <int NT> struct S16Item {
  const uint32_t* tptr[NT];
  float idf[NT], fidf[NT];
  uint32_t b0[NT], b1[NT];  // the item's segments, 16-byte units relative to the list
  uint32_t qpos[NT];        // position of the term in the QUERY (the order a score is summed in)
};

// The candidate path of an item.  Out of line, and called from OUTSIDE the streaming loop (see the kernel): the item's
// postings arrive by value in registers, everything else is scalar.
//  1. the cut.  A doc the postings touched has a bound >= 1; qthr is what the query's threshold asks for.  While the list is
//     not full (or its threshold is still low) hundreds of docs of an item pass that, and only k of them can enter.  Every
//     lane owns 64 docs of the tile; if k lanes hold a bound >= x, k different docs have exact scores above
//     (x - NT - S16_SLACK) / scale, so a doc whose bound is below that is not among the item's k best: the cut rises to the
//     largest such x (bisection over the 64 lane maxima, ballots only).  k > 64 keeps the plain cut.
//  2. the docs at or above the cut go to the wave's list, at most S16_LIST_MAX at a time, ascending;
//  3. their EXACT scores are accumulated the way the bounds were -- by walking the item's postings -- but only for them:
//     the tile is cleared, a candidate's entry gets its list position + 1 as a marker, and a posting whose doc carries a
//     marker adds idf * weight to that candidate's f32 accumulator: terms in query order, one after the other (the docs
//     of one term are distinct and LDS operations execute in order) -- the other kernels' sum, bit for bit;
//  4. keys to the wave-resident top-k; more candidates than the list holds: the bounds are rebuilt and 2-4 repeat.
// Every segment of an item through one chunk routine: the chunks held in registers, then whatever of an oversized segment was
// streamed rather than kept (loaded again, synchronously: the candidate path is rare)
template <int NT, int TT, typename F>
__device__ __forceinline__ void s16_each_chunk(const S16Cur<S16Cfg<NT>::RC>& cur, const S16Item<NT>& it, int lane16, F f) {
  constexpr int CPT = S16Cfg<NT>::cpt(TT), OFF = S16Cfg<NT>::off(TT);
  const uint32_t n16 = it.b1[TT] - it.b0[TT];
#pragma unroll
  for (int c = 0; c < CPT; c++)
    if ((uint32_t)c * 64u < n16) f(cur.v[OFF + c]);
  if (n16 > (uint32_t)CPT * 64u) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)it.tptr[TT], 0, (int)(it.b1[TT] << 4), BM_RSRC_FLAGS);
    s16_rest(rs, lane16, it.b0[TT] + CPT * 64u, it.b1[TT], f);
  }
}
// ... for every term position TT in [A, B), f(chunk, TT)
template <int NT, int A, int B, typename F>
__device__ __forceinline__ void s16_each_term(const S16Cur<S16Cfg<NT>::RC>& cur, const S16Item<NT>& it, int lane16, F f) {
  if constexpr (A < B) {
    s16_each_chunk<NT, A>(cur, it, lane16, [&](const u32x4 v) { f(v, std::integral_constant<int, A>{}); });
    s16_each_term<NT, A + 1, B>(cur, it, lane16, f);
  }
}
// intersections: the tile as the candidate path wants it -- (bound << 2) | 3 for the docs found in every term, 0 elsewhere.
// from_scratch = false: the streaming loop has left the first NT - 1 terms accumulated (levels) and the last one unread.
// ... for the ONE term that stands at position qp of the query
template <int NT, int A, typename F>
__device__ __forceinline__ void s16_term_at(const S16Cur<S16Cfg<NT>::RC>& cur, const S16Item<NT>& it, int lane16, uint32_t qp, F f) {
  if constexpr (A < NT) {
    if (it.qpos[A] == qp) s16_each_chunk<NT, A>(cur, it, lane16, [&](const u32x4 v) { f(v, std::integral_constant<int, A>{}); });
    else s16_term_at<NT, A + 1>(cur, it, lane16, qp, f);
  }
}
template <int NT>
__device__ __forceinline__ void s16a_build(const S16Cur<S16Cfg<NT>::RC>& cur, const S16Item<NT>& it, uint32_t accb, int lane16, bool from_scratch) {
  uint32_t nocount = 0u;
  if (from_scratch) {
    s16_each_chunk<NT, 0>(cur, it, lane16, [&](const u32x4 v) { s16a_first(v, it.fidf[0], accb); });
    s16_each_term<NT, 1, NT - 1>(cur, it, lane16, [&](const u32x4 v, auto tc) { s16a_mid(v, it.fidf[decltype(tc)::value], accb, (uint32_t)decltype(tc)::value); });
  }
  s16_each_chunk<NT, NT - 1>(cur, it, lane16, [&](const u32x4 v) { (void)s16a_last<false, true>(v, it.fidf[NT - 1], accb, (uint32_t)NT - 1u, 0u, nocount); });
  s16_each_chunk<NT, 0>(cur, it, lane16, [&](const u32x4 v) { s16a_clean(v, accb); });
}

template <int NT, int KPL, bool AND>
__device__ __forceinline__ BmTop<KPL> s16_trigger(BmTop<KPL> T, S16Cur<S16Cfg<NT>::RC> cur, S16Item<NT> it, uint32_t wb, uint32_t qthr,
                                                            float thr, uint32_t doc_base, uint32_t k, uint32_t* tau_q, const S16Excl ex) {
  const uint32_t* __restrict__ del = ex.del;
  const uint32_t del_words = ex.del_words;
  const int lane = __lane_id();
  const int lane16 = lane * 16;
  const uint32_t tile = wb + 16u, accb = wb + 14u, accf = wb + S16_ACC;
  const float wsc_in = T.wsc;
  uint32_t start = 0u;  // docs below it were evaluated by an earlier round
  // intersections: entries are (bound << 2) | 3 -- thresholds and slack move up with them (the 3 sits below one unit of bound)
  constexpr uint32_t SH = AND ? 2u : 0u;
  if (AND) {
    qthr <<= 2;
    s16a_build<NT>(cur, it, accb, lane16, false);
  }
  // NOT lists and tombstones: their docs leave the tile BEFORE the cut looks at it (k lanes holding a bound >= x must mean k docs
  // that can be results).  The plain top-k instances never read a NOT list anywhere else: an excluded doc only ever made a bound
  // too large, i.e. a trigger too many.
  const bool excl = ex.nn != 0u || ex.del != nullptr;
  if (excl) s16_exclude(ex, accb, lane);
  for (;;) {
    // this lane's 8 slots (slot i * 64 + lane holds docs 8 * slot .. 8 * slot + 7): their largest bounds, packed 2 per register
    uint32_t sm[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
      sm[i] = s16_max8(lds_ld128(tile + (uint32_t)(2 * i * 64 + lane) * 16u)) |
              (s16_max8(lds_ld128(tile + (uint32_t)((2 * i + 1) * 64 + lane) * 16u)) << 16);
    const uint32_t lmp = s16_pkmax(s16_pkmax(sm[0], sm[1]), s16_pkmax(sm[2], sm[3]));
    const uint32_t lm = max(lmp & 0xFFFFu, lmp >> 16);  // the lane's largest bound
    uint32_t qcut = max(qthr, 1u);
    if (k <= 64u && (uint32_t)__popcll(__ballot(lm >= qcut)) > k) {
      uint32_t lo = qcut, hi = 65535u;  // invariant: at least k lanes reach lo
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if ((uint32_t)__popcll(__ballot(lm >= mid)) >= k) lo = mid; else hi = mid - 1u;
      }
      constexpr uint32_t SLACK = ((AND ? (uint32_t)NT : S16_QM_UP * (uint32_t)NT) + S16_SLACK) << SH;
      qcut = max(qcut, lo > SLACK ? lo - SLACK : 1u);
    } else if (KPL > 1 && k > 64u) {
      // k of 65 .. 128: more results than lanes -- the same cut over the 512 SLOTS (8 docs each, one maximum per slot): k slots holding a
      // bound >= x are k distinct docs that reach x
      auto slots_at = [&](uint32_t x) -> uint32_t {
        uint32_t c = 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) c += (uint32_t)__popcll(__ballot(((sm[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu) >= x));
        return c;
      };
      if (slots_at(qcut) > k) {
        uint32_t lo = qcut, hi = 65535u;  // invariant: at least k slots reach lo
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1u) >> 1;
          if (slots_at(mid) >= k) lo = mid; else hi = mid - 1u;
        }
        constexpr uint32_t SLACK2 = ((AND ? (uint32_t)NT : S16_QM_UP * (uint32_t)NT) + S16_SLACK) << SH;
        qcut = max(qcut, lo > SLACK2 ? lo - SLACK2 : 1u);
      }
    }
    uint32_t hotbits = 0u;  // bit i: slot i of this lane holds a bound at or above the cut
#pragma unroll
    for (int i = 0; i < 8; i++) hotbits |= (((sm[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu) >= qcut ? 1u : 0u) << i;
    uint32_t n = 0u, next = 0xFFFFu;
    // candidates in ascending doc order: slot-major, lane-minor -- only the (slot, lane) pairs that hold one are visited
#pragma unroll 1
    for (uint32_t i = start >> 9; i < (uint32_t)(BM_SUB / 512) && next == 0xFFFFu; i++) {
      u64 m = __ballot((hotbits >> i) & 1u);
      while (m && next == 0xFFFFu) {
        const uint32_t l = (uint32_t)__ffsll((long long)m) - 1u;
        m &= m - 1;
        const u32x4 g = lds_ld128(tile + (i * 64u + l) * 16u);  // one address for the whole wave
        const uint32_t dw[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)g.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)g.y),
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)g.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)g.w)};
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
          const uint32_t bound = (dw[j >> 1] >> ((j & 1u) * 16u)) & 0xFFFFu;
          const uint32_t din = (i * 64u + l) * 8u + j;
          if (bound >= qcut && din >= start && next == 0xFFFFu) {
            if (n == S16_LIST_MAX) next = din;
            else { lds_st16(wb + S16_LIST + n * 2u, din); n++; }
          }
        }
      }
    }
    start = next;
    s16_clear_tile(wb, lane);
    uint32_t din = 0u;
    if ((uint32_t)lane < n) {
      din = lds_ld16(wb + S16_LIST + (uint32_t)lane * 2u);
      lds_st16(accb + ((din + 1u) << 1), (uint32_t)lane + 1u);
      lds_stf(accf + (uint32_t)lane * 4u, 0.f);
    }
    auto exact = [&](const u32x4 v, float idf_t) {
      const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
      uint32_t mk[4];
#pragma unroll
      for (int x = 0; x < 4; x++) mk[x] = lds_ld16(s16_addr(pv[x], accb));
#pragma unroll
      for (int x = 0; x < 4; x++)
        if (mk[x]) {
          const uint32_t ad = accf + (mk[x] - 1u) * 4u;
          lds_stf(ad, __builtin_fmaf(idf_t, bm_weight(pv[x]), lds_ldf(ad)));
        }
    };
    if (n) {
      // terms in the QUERY's order (the fma chain of every other kernel), whatever order they are processed in: for each query
      // position the term that stands there (one of NT wave-uniform branches)
#pragma unroll
      for (int qp = 0; qp < NT; qp++)
        s16_term_at<NT, 0>(cur, it, lane16, (uint32_t)qp, [&](const u32x4 v, auto tc) { exact(v, it.idf[decltype(tc)::value]); });
      u64 key = 0ull;
      if ((uint32_t)lane < n) {
        const float sc = lds_ldf(accf + (uint32_t)lane * 4u);
        const uint32_t doc = doc_base + din;
        const bool gone = del && (doc >> 5) < del_words && ((del[doc >> 5] >> (doc & 31u)) & 1u);  // add_result.rs:3435, union.rs:975
        lds_st16(accb + ((din + 1u) << 1), 0u);  // the marker: the tile is all zero again
        if (!gone && sc > 0.f && sc >= thr) key = ((u64)__float_as_uint(sc) << 32) | (u64)(0xFFFFFFFFu - doc);
      }
      key = key > T.worst ? key : 0ull;
      if (__ballot(key != 0ull)) {
        T.worst = topk_offer<KPL>(T.keys, key, 0ull, 0ull, 0ull, T.worst, k);
        if (T.worst) {
          T.wsc = __uint_as_float((uint32_t)(T.worst >> 32));
          thr = fmaxf(thr, T.wsc);
        }
      }
    }
    if (start == 0xFFFFu) break;
    // more candidates than the list holds (ties, or a list that is still filling): rebuild the bounds and go on
    if (AND) {
      s16a_build<NT>(cur, it, accb, lane16, true);
    } else {
      uint32_t dummy = 0u, nocount = 0u;  // (the docs of this item were counted when its bounds were first accumulated)
      s16_each_term<NT, 0, NT>(cur, it, lane16, [&](const u32x4 v, auto tc) { dummy = s16_keep<false>(v, it.fidf[decltype(tc)::value], accb, dummy, nocount); });
    }
    if (excl) s16_exclude(ex, accb, lane);
  }
  s16_clear(wb, lane);
  if (tau_q && T.wsc > wsc_in && lane == 0) bm_publish_tau(tau_q, T.wsc);
  return T;
}



// EXCL (count instances only): the exclusions are applied INSIDE the streaming loop, because an exact count has to see every item --
// one NOT list streamed beside the query's terms (CN register chunks per item) and the sub-block's 128 tombstone words (8 bytes per
// lane).  Unions: the excluded docs are MARKED before the terms are accumulated (S16_MARK, s16_read<.., SGN>): a marked doc is no
// first touch and reaches no threshold.  Intersections: the exclusions follow the first term, the only one that creates entries
// (an excluded doc's entry goes back to 0: it can reach no level).  Without EXCL a query's NOT lists and the tombstones are only looked at by the candidate path.
template <int NT, int KPL, bool CNT, bool AND, bool EXCL>
__global__ void __launch_bounds__(S16_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4)))
bm25_scan16_kernel(const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base, const uint32_t* __restrict__ sub_off,
                   const bm_vquery* __restrict__ qs, unsigned long long* __restrict__ part_keys, unsigned long long* __restrict__ total, uint32_t* tau,
                   const uint32_t* __restrict__ del, uint32_t del_words, uint32_t n_sub, uint32_t n_terms, uint32_t nq, uint32_t P, uint32_t k) {
  using Cfg = S16Cfg<NT>;
  constexpr int RC = Cfg::RC;
  constexpr int CN = EXCL ? (NT <= 3 ? 2 : 1) : 0;  // register chunks of the streamed NOT list
  constexpr int DX = RC + CN;                        // EXCL: the chunk whose .x / .y hold this lane's two tombstone words
  constexpr int RCX = RC + CN + (EXCL ? 1 : 0);
  static_assert(!EXCL || CNT, "exclusions inside the streaming loop are what exact counts need");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();  // offsets below are absolute
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wb = (uint32_t)w * S16_WAVE_LDS, accb = wb + 14u;  // entry of doc field f at accb + 2 f; field 0 = dump
  s16_clear(wb, lane);  // every wave initialises and uses only its own slice: no barrier
  const uint32_t row_len = n_sub + 1;
  const int lane16 = lane * 16;

  const uint32_t a = blockIdx.x * S16_WAVES + w;
  if (a >= nq * P) return;
  const uint32_t qi = a % nq, part = a / nq;
  const bm_vquery* __restrict__ Q = qs + qi;
  const uint32_t nt = Q->n_terms;  // (a query's NOT lists follow in Q->term: candidate path, or the EXCL instances' stream)
  constexpr uint32_t BLK = 62;     // items per boundary block, as bm25_scan_fast_kernel
  const uint32_t* tptr[NT];
  float idf[NT];
  const uint32_t* rowp[NT];
  float fidf[NT];
  uint32_t qpos[NT];
  unsigned long long tlen[NT];
  float idf_sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const bool have = (uint32_t)t < nt;
    const uint32_t term = have ? Q->term[t] : n_terms;
    idf[t] = have ? Q->idf[t] : 0.f;
    idf_sum += idf[t];
    tptr[t] = post + term_base[term] * 4ull;
    rowp[t] = sub_off + (size_t)term * row_len;
    qpos[t] = (uint32_t)t;
    tlen[t] = have ? term_base[term + 1] - term_base[term] : 0ull;  // absent terms (a short query) sort to the front: empty lists
  }
  // processing order: by list length, shortest first (the chunk budgets of S16Cfg follow the position; an intersection's entries
  // are created by its first = shortest list); qpos keeps every term's place in the query for the exact sums
  auto cswap = [&](int x, int y) {
    if (tlen[y] < tlen[x]) {
      { auto t_ = tptr[x]; tptr[x] = tptr[y]; tptr[y] = t_; }
      { auto t_ = rowp[x]; rowp[x] = rowp[y]; rowp[y] = t_; }
      { float t_ = idf[x]; idf[x] = idf[y]; idf[y] = t_; }
      { uint32_t t_ = qpos[x]; qpos[x] = qpos[y]; qpos[y] = t_; }
      { auto t_ = tlen[x]; tlen[x] = tlen[y]; tlen[y] = t_; }
    }
  };
  if (NT == 2) { cswap(0, 1); }
  if (NT == 3) { cswap(0, 1); cswap(1, 2); cswap(0, 1); }
  if (NT == 4) { cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2); }
  if (NT == 5) { cswap(0, 1); cswap(3, 4); cswap(2, 4); cswap(2, 3); cswap(0, 3); cswap(0, 2); cswap(1, 4); cswap(1, 3); cswap(1, 2); }
  if (NT == 6) {
    cswap(1, 2); cswap(4, 5); cswap(0, 2); cswap(3, 5); cswap(0, 1); cswap(3, 4); cswap(2, 5); cswap(0, 3); cswap(1, 4); cswap(2, 4);
    cswap(1, 3); cswap(2, 3);
  }
  // EXCL: the one NOT list that is streamed (none: the empty list of the absent term); a query with more of them contradicts what
  // the host chose this instance by: flagged like any query that contradicts its batch's declaration (count = UINT32_MAX)
  if (EXCL && bm_q_nnot(Q->op) > 1u && lane == 0) tau[(size_t)qi * BM_TAU_STRIDE + 1] = 1u;
  const uint32_t nterm = (EXCL && bm_q_nnot(Q->op)) ? Q->term[nt] : n_terms;
  const uint32_t* ntptr = post + term_base[nterm] * 4ull;
  const uint32_t* nrowp = sub_off + (size_t)nterm * row_len;
  const float scale = (AND ? S16_QMAX_AND : EXCL ? S16_QMAX_EXCL : S16_QMAX) / (S16_WMAX * idf_sum);
#pragma unroll
  for (int t = 0; t < NT; t++) fidf[t] = s16_uniform(idf[t] * scale * (AND ? 1.0f : S16_WFOLD));  // unions: s16_qm's folded multiplier
  // wave-uniform constants of the item loop, pinned to scalar registers (left to the allocator, scale_thr went to scratch and
  // its reload brought a vmcnt(0) -- a wait for the whole prefetch -- into every item)
  const float scale_thr = s16_uniform(scale * (1.0f - 1e-5f));
  const uint32_t s_begin = (uint32_t)(((u64)n_sub * part) / P);
  const uint32_t s_end = (uint32_t)(((u64)n_sub * (part + 1)) / P);

  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull;
  T.wsc = -1.0f;
  T.matched = 0;

  auto issue_loads = [&](u32x4(&v)[RCX], const uint32_t (&b0)[NT], const uint32_t (&b1)[NT], uint32_t nb0, uint32_t nb1, uint32_t item) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[t], 0, (int)(b1[t] << 4), BM_RSRC_FLAGS);
#pragma unroll
      for (int c = 0; c < Cfg::CPTMAX; c++)
        if (c < Cfg::cpt(t)) v[Cfg::off(t) + c] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + c * 1024, (int)(b0[t] << 4), 0);
    }
    if constexpr (EXCL) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ntptr, 0, (int)(nb1 << 4), BM_RSRC_FLAGS);
#pragma unroll
      for (int c = 0; c < CN; c++) v[RC + c] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + c * 1024, (int)(nb0 << 4), 0);
      // tombstones: words 2 lane, 2 lane + 1 of the sub-block's 128; no bitmap / past its end: zeros without a memory access
      __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(del ? del : post), 0, del ? (int)(del_words << 2) : 0, BM_RSRC_FLAGS);
      const auto dw = __builtin_amdgcn_raw_buffer_load_b64(rd, lane * 8, (int)(item * (uint32_t)(BM_SUB / 8)), 0);
      v[DX] = u32x4{dw[0], dw[1], 0u, 0u};
    }
  };

  u32x4 vA[RCX], vB[RCX];
  uint32_t B0[NT], B1[NT], B2[NT];
  uint32_t vbnd[NT];
  uint32_t NB0 = 0u, NB1 = 0u, NB2 = 0u, vbndN = 0u, item0 = 0u;  // the NOT list's boundaries (EXCL); item0 = first item of the boundary block
  uint32_t* tau_q = tau + (size_t)qi * BM_TAU_STRIDE;

  // One item: prefetch the next one, accumulate the bounds of this one, decide.  Returns true when some doc may enter the
  // list -- the caller then LEAVES the streaming loop, runs the candidate path and primes the pipeline again.  The call is
  // kept out of the loop on purpose: with a call inside, the prefetched registers are live across it, the allocator parks
  // them in scratch in every item, and the vmcnt bookkeeping merges the two histories into a wait for everything.
  uint32_t qthr_hit = 0u;
  float thr_hit = 0.f;
  uint32_t hb0[NT], hb1[NT];
  auto body = [&](u32x4(&cur)[RCX], u32x4(&nxt)[RCX], uint32_t i) -> bool {
    const uint32_t tau_bits = __hip_atomic_load(tau_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int t = 0; t < NT; t++) B2[t] = __builtin_amdgcn_readlane(vbnd[t], i + 2);
    if constexpr (EXCL) NB2 = __builtin_amdgcn_readlane(vbndN, i + 2);
    issue_loads(nxt, B1, B2, NB1, NB2, item0 + i + 1u);
    // EXCL: the NOT list's chunks of this item and the tombstone words, applied where the instance wants them
    auto exclusions = [&](uint32_t& dec, bool& over) {
      if constexpr (EXCL) {
        const uint32_t nn16 = NB1 - NB0;
#pragma unroll
        for (int c = 0; c < CN; c++)
          if ((uint32_t)c * 64u < nn16) s16_not_chunk<CNT && !AND>(cur[RC + c], accb, dec);
        if (nn16 > (uint32_t)CN * 64u) {
          over = true;
          __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ntptr, 0, (int)(NB1 << 4), BM_RSRC_FLAGS);
          s16_rest(rs, lane16, NB0 + CN * 64u, NB1, [&](const u32x4 v) { s16_not_chunk<CNT && !AND>(v, accb, dec); });
        }
        if (del) s16_del_bits<CNT && !AND>(cur[DX].x, cur[DX].y, accb, lane, dec);
      }
    };
    uint32_t maxn = 0;
#pragma unroll
    for (int t = 0; t < NT; t++) maxn = max(maxn, B1[t] - B0[t]);
    bool hit = false;
    if constexpr (AND) {
      uint32_t minn = 0xFFFFFFFFu;
#pragma unroll
      for (int t = 0; t < NT; t++) minn = min(minn, B1[t] - B0[t]);
      if (minn) {  // a sub-block one of the terms has no doc in holds no match: nothing is touched
        uint32_t mx = 0u, cnt = 0u;
        auto seg = [&](auto tc, auto f) {
          constexpr int t = decltype(tc)::value;
          const uint32_t n16 = B1[t] - B0[t];
#pragma unroll
          for (int c = 0; c < Cfg::cpt(t); c++)
            if ((uint32_t)c * 64u < n16) f(cur[Cfg::off(t) + c]);
          if (n16 > (uint32_t)Cfg::cpt(t) * 64u) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[t], 0, (int)(B1[t] << 4), BM_RSRC_FLAGS);
            s16_rest(rs, lane16, B0[t] + Cfg::cpt(t) * 64u, B1[t], f);
          }
        };
        seg(std::integral_constant<int, 0>{}, [&](const u32x4 v) { s16a_first(v, fidf[0], accb); });
        if constexpr (EXCL) {  // an excluded doc loses the entry its first posting created: it can reach no level
          uint32_t nodec = 0u;
          bool noover = false;
          exclusions(nodec, noover);
        }
        if constexpr (NT == 3) seg(std::integral_constant<int, 1>{}, [&](const u32x4 v) { s16a_mid(v, fidf[1], accb, 1u); });
        seg(std::integral_constant<int, NT - 1>{}, [&](const u32x4 v) { mx = s16a_last<CNT, false>(v, fidf[NT - 1], accb, (uint32_t)NT - 1u, mx, cnt); });
        if (CNT) T.matched += cnt;
        const float thr = fmaxf(T.wsc, __uint_as_float(tau_bits));
        const uint32_t qthr = k ? (thr > 0.f ? (uint32_t)(thr * scale_thr) : 0u) : 0xFFFFFFFFu;
        if (__ballot(mx != 0u && mx >= qthr)) {
          hit = true;
          qthr_hit = qthr;
          thr_hit = thr;
#pragma unroll
          for (int t = 0; t < NT; t++) { hb0[t] = B0[t]; hb1[t] = B1[t]; }
        } else if (B1[0] - B0[0] <= (uint32_t)Cfg::cpt(0) * 64u) {  // only the first term's docs have entries
          const uint32_t n16 = B1[0] - B0[0];
#pragma unroll
          for (int c = 0; c < Cfg::cpt(0); c++)
            if ((uint32_t)c * 64u < n16) {
              const u32x4 v = cur[c];
              lds_st16(s16_addr(v.x, accb), 0u);
              lds_st16(s16_addr(v.y, accb), 0u);
              lds_st16(s16_addr(v.z, accb), 0u);
              lds_st16(s16_addr(v.w, accb), 0u);
            }
        } else {
          s16_clear(wb, lane);
        }
      }
    } else if (maxn) {
      uint32_t mx = 0u, cnt = 0u;
      bool over = false;  // some segment is longer than its register chunks: the remainder is streamed synchronously (and written)
#pragma unroll
      for (int t = 0; t < NT; t++) over = over || (B1[t] - B0[t]) > (uint32_t)Cfg::cpt(t) * 64u;
      if constexpr (EXCL) {
        // the exclusions FIRST, as marks (S16_MARK): the NOT list's docs and the tombstoned docs of the sub-block; the terms then
        // accumulate as always -- reading sign-extended, so a marked doc neither counts as a first touch nor reaches a threshold --
        // and the last term is only read, as in the plain instances
        {
          const uint32_t nn16 = NB1 - NB0;
#pragma unroll
          for (int c = 0; c < CN; c++)
            if ((uint32_t)c * 64u < nn16) s16_mark_chunk(cur[RC + c], accb);
          if (nn16 > (uint32_t)CN * 64u) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ntptr, 0, (int)(NB1 << 4), BM_RSRC_FLAGS);
            s16_rest(rs, lane16, NB0 + CN * 64u, NB1, [&](const u32x4 v) { s16_mark_chunk(v, accb); });
          }
          if (del) s16_mark_bits(cur[DX].x, cur[DX].y, accb, lane);
        }
#pragma unroll
        for (int t = 0; t + 1 < NT; t++) {
          const uint32_t n16 = B1[t] - B0[t];
#pragma unroll
          for (int c = 0; c < Cfg::CPTMAX; c++)
            if (c < Cfg::cpt(t) && (uint32_t)c * 64u < n16) mx = s16_keep<CNT, true>(cur[Cfg::off(t) + c], fidf[t], accb, mx, cnt);
          if (n16 > (uint32_t)Cfg::cpt(t) * 64u) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[t], 0, (int)(B1[t] << 4), BM_RSRC_FLAGS);
            s16_rest(rs, lane16, B0[t] + Cfg::cpt(t) * 64u, B1[t], [&](const u32x4 v) { mx = s16_keep<CNT, true>(v, fidf[t], accb, mx, cnt); });
          }
        }
        constexpr int CL = Cfg::cpt(NT - 1), OL = Cfg::off(NT - 1);
        const uint32_t nlast = B1[NT - 1] - B0[NT - 1];
        uint32_t nwL[CL][4];
#pragma unroll
        for (int c = 0; c < CL; c++)
          if ((uint32_t)c * 64u < nlast) mx = s16_read<CNT, true>(cur[OL + c], fidf[NT - 1], accb, mx, nwL[c], cnt);
        if (nlast > (uint32_t)CL * 64u) {
          __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[NT - 1], 0, (int)(B1[NT - 1] << 4), BM_RSRC_FLAGS);
          s16_rest(rs, lane16, B0[NT - 1] + CL * 64u, B1[NT - 1], [&](const u32x4 v) { mx = s16_keep<CNT, true>(v, fidf[NT - 1], accb, mx, cnt); });
        }
        T.matched += cnt;
        const float thr = fmaxf(T.wsc, __uint_as_float(tau_bits));
        const uint32_t qthr = k ? (thr > 0.f ? (uint32_t)(thr * scale_thr) : 0u) : 0xFFFFFFFFu;
        if (__ballot(mx >= (k ? qthr + S16_MBITS : 0xFFFFFFFFu))) {
          // the candidate path clears the excluded docs (marks and what was added to them) before it looks at the tile
          hit = true;
#pragma unroll
          for (int c = 0; c < CL; c++)
            if ((uint32_t)c * 64u < nlast) {
              const u32x4 v = cur[OL + c];
              lds_st16(s16_addr(v.x, accb), nwL[c][0]);
              lds_st16(s16_addr(v.y, accb), nwL[c][1]);
              lds_st16(s16_addr(v.z, accb), nwL[c][2]);
              lds_st16(s16_addr(v.w, accb), nwL[c][3]);
            }
          qthr_hit = qthr;
          thr_hit = thr;
#pragma unroll
          for (int t = 0; t < NT; t++) { hb0[t] = B0[t]; hb1[t] = B1[t]; }
        } else {
          s16_clear(wb, lane);  // (terms, marks: more narrow stores than the 9 wide ones of the whole slice)
        }
      } else {
#pragma unroll
      for (int t = 0; t + 1 < NT; t++) {  // the first term finds an empty tile: written without a read
        const uint32_t n16 = B1[t] - B0[t];
#pragma unroll
        for (int c = 0; c < Cfg::CPTMAX; c++)
          if (c < Cfg::cpt(t) && (uint32_t)c * 64u < n16) {
            if (t == 0) mx = s16_first<CNT>(cur[Cfg::off(t) + c], fidf[t], accb, mx, cnt);
            else mx = s16_keep<CNT>(cur[Cfg::off(t) + c], fidf[t], accb, mx, cnt);
          }
        if (n16 > (uint32_t)Cfg::cpt(t) * 64u) {  // the rest of the segment, loaded synchronously
          __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[t], 0, (int)(B1[t] << 4), BM_RSRC_FLAGS);
          s16_rest(rs, lane16, B0[t] + Cfg::cpt(t) * 64u, B1[t], [&](const u32x4 v) { mx = s16_keep<CNT>(v, fidf[t], accb, mx, cnt); });
        }
      }
      // the last term is only READ: its sums stay in registers and reach the tile when the item has candidates -- most
      // items have none, and the LDS pipe is what this kernel keeps busiest (a third of the scattered writes saved)
      constexpr int CL = Cfg::cpt(NT - 1), OL = Cfg::off(NT - 1);
      const uint32_t nlast = B1[NT - 1] - B0[NT - 1];
      uint32_t nwL[CL][4];
#pragma unroll
      for (int c = 0; c < CL; c++)
        if ((uint32_t)c * 64u < nlast) {
          mx = s16_read<CNT>(cur[OL + c], fidf[NT - 1], accb, mx, nwL[c], cnt);
        }
      if (nlast > (uint32_t)CL * 64u) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tptr[NT - 1], 0, (int)(B1[NT - 1] << 4), BM_RSRC_FLAGS);
        s16_rest(rs, lane16, B0[NT - 1] + CL * 64u, B1[NT - 1], [&](const u32x4 v) { mx = s16_keep<CNT>(v, fidf[NT - 1], accb, mx, cnt); });
      }
      if (CNT) T.matched += cnt;
      const float thr = fmaxf(T.wsc, __uint_as_float(tau_bits));
      const uint32_t qthr = k ? (thr > 0.f ? (uint32_t)(thr * scale_thr) : 0u) : 0xFFFFFFFFu;  // k = 0 (ResultType::Count): nothing is ranked
      if (__ballot(mx >= (k ? qthr + S16_MBITS : 0xFFFFFFFFu))) {  // mx = the bits of 2^23 + the largest bound (s16_qm)
        hit = true;
#pragma unroll
        for (int c = 0; c < CL; c++)
          if ((uint32_t)c * 64u < nlast) {
            const u32x4 v = cur[OL + c];
            lds_st16(s16_addr(v.x, accb), nwL[c][0]);
            lds_st16(s16_addr(v.y, accb), nwL[c][1]);
            lds_st16(s16_addr(v.z, accb), nwL[c][2]);
            lds_st16(s16_addr(v.w, accb), nwL[c][3]);
          }
        qthr_hit = qthr;
        thr_hit = thr;
#pragma unroll
        for (int t = 0; t < NT; t++) { hb0[t] = B0[t]; hb1[t] = B1[t]; }
      } else if (!over) {
        // no candidate (the common case): only the entries the first NT - 1 terms wrote are dirty (the last term was only
        // read) -- zero those through their postings, 4 narrow stores per chunk, instead of 9 wide ones for the whole tile
#pragma unroll
        for (int t = 0; t + 1 < NT; t++) {
          const uint32_t n16 = B1[t] - B0[t];
#pragma unroll
          for (int c = 0; c < Cfg::CPTMAX; c++)
            if (c < Cfg::cpt(t) && (uint32_t)c * 64u < n16) {
              const u32x4 v = cur[Cfg::off(t) + c];
              lds_st16(s16_addr(v.x, accb), 0u);
              lds_st16(s16_addr(v.y, accb), 0u);
              lds_st16(s16_addr(v.z, accb), 0u);
              lds_st16(s16_addr(v.w, accb), 0u);
            }
        }
      } else {
        s16_clear(wb, lane);
      }
      }  // !EXCL
    }
#pragma unroll
    for (int t = 0; t < NT; t++) { B0[t] = B1[t]; B1[t] = B2[t]; }
    if constexpr (EXCL) { NB0 = NB1; NB1 = NB2; }
    return hit;
  };

  for (uint32_t s0 = s_begin; s0 < s_end; s0 += BLK) {
    const uint32_t j = s0 + (uint32_t)lane;
#pragma unroll
    for (int t = 0; t < NT; t++) vbnd[t] = rowp[t][j < s_end ? j : s_end];
    if constexpr (EXCL) { vbndN = nrowp[j < s_end ? j : s_end]; item0 = s0; }
    const uint32_t cnt = min(BLK, s_end - s0);
    uint32_t i = 0;
    while (i < cnt) {
      // prime: the postings of item i (block start, or the item after a candidate path)
#pragma unroll
      for (int t = 0; t < NT; t++) {
        B0[t] = __builtin_amdgcn_readlane(vbnd[t], i);
        B1[t] = __builtin_amdgcn_readlane(vbnd[t], i + 1);
      }
      if constexpr (EXCL) { NB0 = __builtin_amdgcn_readlane(vbndN, i); NB1 = __builtin_amdgcn_readlane(vbndN, i + 1); }
      issue_loads(vA, B0, B1, NB0, NB1, s0 + i);
      bool hit, in_a;
      for (;;) {
        in_a = true;
        hit = body(vA, vB, i);
        i++;
        if (hit || i >= cnt) break;
        in_a = false;
        hit = body(vB, vA, i);
        i++;
        if (hit || i >= cnt) break;
      }
      if (hit) {
        S16Cur<RC> cc;
#pragma unroll
        for (int r = 0; r < RC; r++) cc.v[r] = in_a ? vA[r] : vB[r];
        S16Item<NT> it;
#pragma unroll
        for (int t = 0; t < NT; t++) { it.tptr[t] = tptr[t]; it.idf[t] = idf[t]; it.fidf[t] = fidf[t]; it.b0[t] = hb0[t]; it.b1[t] = hb1[t]; it.qpos[t] = qpos[t]; }
        const bm_vquery* __restrict__ Qh = qs + qi;  // (fetched here: nothing of the exclusions lives across the streaming loop)
        const S16Excl ex{Qh, post, term_base, sub_off, del, Qh->n_terms, bm_q_nnot(Qh->op), row_len, del_words, s0 + i - 1u};
        T = s16_trigger<NT, KPL, AND>(T, cc, it, wb, qthr_hit, thr_hit, (s0 + i - 1u) << BM_SUB_LOG2, k, tau_q, ex);
      }
    }
  }

  u64* out = part_keys + ((size_t)qi * P + part) * (64 * KPL);
#pragma unroll
  for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
#if S16_LANE_COUNT
  if (CNT) {
    uint32_t lo = (uint32_t)T.matched, hi = (uint32_t)(T.matched >> 32);
    for (int o = 32; o > 0; o >>= 1) {
      const u64 other = ((u64)__shfl_xor(hi, o) << 32) | __shfl_xor(lo, o);
      const u64 sum = (((u64)hi << 32) | lo) + other;
      lo = (uint32_t)sum; hi = (uint32_t)(sum >> 32);
    }
    T.matched = ((u64)hi << 32) | lo;
  }
#endif
  if (CNT && lane == 0 && T.matched) atomicAdd(&total[qi], T.matched);
}

template <int NT, int KPL, bool CNT, bool AND, bool EXCL = false>
int launch16(const BmParams& p, hipStream_t st) {
  constexpr int lds = S16_WAVES * S16_WAVE_LDS;
  SS_SET_MAX_LDS((bm25_scan16_kernel<NT, KPL, CNT, AND, EXCL>), lds);
  const uint32_t A = p.nq * p.P;
  bm25_scan16_kernel<NT, KPL, CNT, AND, EXCL><<<(A + S16_WAVES - 1) / S16_WAVES, S16_WAVES * 64, lds, st>>>(
      p.post, p.term_base, p.sub_off, p.q, p.part_keys, p.total, p.tau, p.del, p.del_words, p.n_sub, p.n_terms, p.nq, p.P, p.k);
  return SS_OK;
}


// ---------------------------------------------------------------- unions of MANY lists (5 .. 32): bm25_scan16m_kernel
// The reference scans unions of more than 10 terms with a 32-bit match mask per doc of the block (union_scan_32, union.rs:598-805) and
// sends unions of up to 10 through sub-query decomposition (union_docid_3); several indexed fields multiply a query's lists as well.
// The NT-specialised kernel above keeps an item's postings in registers for its candidate path -- 6 lists are what fits.  Here the
// number of lists is a RUN-TIME value: per-list constants live one per LANE (lane t: term, list address, idf, fixed-point idf, the
// list's boundary in the current and the next sub-block) and are fetched with v_readlane as the lists are walked; the lists of an item
// are streamed one after the other into the same 16-bit bound tile (the first chunk of list t + 1 in flight while list t is
// accumulated); the rare candidate path re-reads the item's segments (they are in the L2) to sum its candidates' exact scores in
// query order -- the fma chain of every other kernel.  Top-k only (exact counts come from the probe index's bit records; without one
// the f32 tile's count mode serves), NOT lists and tombstones through the candidate path as in the plain instances.
struct S16MLists {
  uint32_t v_tlo, v_thi;  // lane t: address of list t's postings
  uint32_t v_a, v_b;      // lane t: the item's segment of list t, [a, b) in 16-byte units relative to the list
  uint32_t v_idf, v_fidf; // lane t: idf (bits), idf * scale (bits)
  uint32_t nt;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t s16m_rsrc(const S16MLists& L, uint32_t t, uint32_t b1) {
  const u64 a = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)L.v_thi, t) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)L.v_tlo, t);
  return __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, (int)(b1 << 4), BM_RSRC_FLAGS);
}
// every chunk (64 lanes x 16 B) of the item's segment of list t
template <typename F>
__device__ __forceinline__ void s16m_each_chunk(const S16MLists& L, uint32_t t, int lane16, F f) {
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)L.v_a, t), b1 = (uint32_t)__builtin_amdgcn_readlane((int)L.v_b, t);
  if (b0 == b1) return;
  __amdgpu_buffer_rsrc_t rs = s16m_rsrc(L, t, b1);
  if (b1 - b0 <= 64u) {  // the usual case: one chunk
    f(__builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (int)(b0 << 4), 0));
    return;
  }
  for (uint32_t u = b0; u < b1; u += 256u) {  // a dense segment: four chunks in flight
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (int)(u << 4), 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 1024, (int)(u << 4), 0);
    const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 2048, (int)(u << 4), 0);
    const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 3072, (int)(u << 4), 0);
    f(a);
    if (u + 64u < b1) f(b);
    if (u + 128u < b1) f(c);
    if (u + 192u < b1) f(d);
  }
}
// the candidate path of an item (s16_trigger's steps with the postings re-read instead of held): cut, list, exact sums, keys
template <int KPL>
__device__ __attribute__((noinline)) BmTop<KPL> s16m_trigger(BmTop<KPL> T, const S16MLists L, uint32_t wb, uint32_t qthr, float thr, uint32_t doc_base,
                                                           uint32_t k, uint32_t* tau_q, const S16Excl ex) {
  const uint32_t* __restrict__ del = ex.del;
  const uint32_t del_words = ex.del_words;
  const int lane = __lane_id();
  const int lane16 = lane * 16;
  const uint32_t tile = wb + 16u, accb = wb + 14u, accf = wb + S16_ACC;
  const float wsc_in = T.wsc;
  uint32_t start = 0u;
  const bool excl = ex.nn != 0u || ex.del != nullptr;
  if (excl) s16_exclude(ex, accb, lane);
  const uint32_t SLACK = S16_QM_UP * L.nt + S16_SLACK;
  for (;;) {
    uint32_t sm[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
      sm[i] = s16_max8(lds_ld128(tile + (uint32_t)(2 * i * 64 + lane) * 16u)) |
              (s16_max8(lds_ld128(tile + (uint32_t)((2 * i + 1) * 64 + lane) * 16u)) << 16);
    const uint32_t lmp = s16_pkmax(s16_pkmax(sm[0], sm[1]), s16_pkmax(sm[2], sm[3]));
    const uint32_t lm = max(lmp & 0xFFFFu, lmp >> 16);
    uint32_t qcut = max(qthr, 1u);
    if (k <= 64u && (uint32_t)__popcll(__ballot(lm >= qcut)) > k) {
      uint32_t lo = qcut, hi = 65535u;  // invariant: at least k lanes reach lo
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if ((uint32_t)__popcll(__ballot(lm >= mid)) >= k) lo = mid; else hi = mid - 1u;
      }
      qcut = max(qcut, lo > SLACK ? lo - SLACK : 1u);
    } else if (KPL > 1 && k > 64u) {
      auto slots_at = [&](uint32_t x) -> uint32_t {
        uint32_t c = 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) c += (uint32_t)__popcll(__ballot(((sm[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu) >= x));
        return c;
      };
      if (slots_at(qcut) > k) {
        uint32_t lo = qcut, hi = 65535u;
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1u) >> 1;
          if (slots_at(mid) >= k) lo = mid; else hi = mid - 1u;
        }
        qcut = max(qcut, lo > SLACK ? lo - SLACK : 1u);
      }
    }
    uint32_t hotbits = 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) hotbits |= (((sm[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu) >= qcut ? 1u : 0u) << i;
    uint32_t n = 0u, next = 0xFFFFu;
#pragma unroll 1
    for (uint32_t i = start >> 9; i < (uint32_t)(BM_SUB / 512) && next == 0xFFFFu; i++) {
      u64 m = __ballot((hotbits >> i) & 1u);
      while (m && next == 0xFFFFu) {
        const uint32_t l = (uint32_t)__ffsll((long long)m) - 1u;
        m &= m - 1;
        const u32x4 g = lds_ld128(tile + (i * 64u + l) * 16u);
        const uint32_t dw[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)g.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)g.y),
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)g.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)g.w)};
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
          const uint32_t bound = (dw[j >> 1] >> ((j & 1u) * 16u)) & 0xFFFFu;
          const uint32_t din = (i * 64u + l) * 8u + j;
          if (bound >= qcut && din >= start && next == 0xFFFFu) {
            if (n == S16_LIST_MAX) next = din;
            else { lds_st16(wb + S16_LIST + n * 2u, din); n++; }
          }
        }
      }
    }
    start = next;
    s16_clear_tile(wb, lane);
    uint32_t din = 0u;
    if ((uint32_t)lane < n) {
      din = lds_ld16(wb + S16_LIST + (uint32_t)lane * 2u);
      lds_st16(accb + ((din + 1u) << 1), (uint32_t)lane + 1u);
      lds_stf(accf + (uint32_t)lane * 4u, 0.f);
    }
    if (n) {
      for (uint32_t t = 0; t < L.nt; t++) {  // lists stand in query order: the sum is the other kernels', bit for bit
        const float idf_t = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)L.v_idf, t));
        s16m_each_chunk(L, t, lane16, [&](const u32x4 v) {
          const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
          uint32_t mk[4];
#pragma unroll
          for (int x = 0; x < 4; x++) mk[x] = lds_ld16(s16_addr(pv[x], accb));
#pragma unroll
          for (int x = 0; x < 4; x++)
            if (mk[x]) {
              const uint32_t ad = accf + (mk[x] - 1u) * 4u;
              lds_stf(ad, __builtin_fmaf(idf_t, bm_weight(pv[x]), lds_ldf(ad)));
            }
        });
      }
      u64 key = 0ull;
      if ((uint32_t)lane < n) {
        const float sc = lds_ldf(accf + (uint32_t)lane * 4u);
        const uint32_t doc = doc_base + din;
        const bool gone = del && (doc >> 5) < del_words && ((del[doc >> 5] >> (doc & 31u)) & 1u);
        lds_st16(accb + ((din + 1u) << 1), 0u);
        if (!gone && sc > 0.f && sc >= thr) key = ((u64)__float_as_uint(sc) << 32) | (u64)(0xFFFFFFFFu - doc);
      }
      key = key > T.worst ? key : 0ull;
      if (__ballot(key != 0ull)) {
        T.worst = topk_offer<KPL>(T.keys, key, 0ull, 0ull, 0ull, T.worst, k);
        if (T.worst) {
          T.wsc = __uint_as_float((uint32_t)(T.worst >> 32));
          thr = fmaxf(thr, T.wsc);
        }
      }
    }
    if (start == 0xFFFFu) break;
    // more candidates than the list holds: the bounds again, the exclusions again, the next 64
    {
      uint32_t dummy = 0u, nocount = 0u;
      for (uint32_t t = 0; t < L.nt; t++) {
        const float fidf_t = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)L.v_fidf, t));
        s16m_each_chunk(L, t, lane16, [&](const u32x4 v) { dummy = s16_keep<false>(v, fidf_t, accb, dummy, nocount); });
      }
    }
    if (excl) s16_exclude(ex, accb, lane);
  }
  s16_clear(wb, lane);
  if (tau_q && T.wsc > wsc_in && lane == 0) bm_publish_tau(tau_q, T.wsc);
  return T;
}

template <int KPL>
__global__ void __launch_bounds__(S16_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4)))
bm25_scan16m_kernel(const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base, const uint32_t* __restrict__ sub_off,
                    const bm_vquery* __restrict__ qs, unsigned long long* __restrict__ part_keys, uint32_t* tau,
                    const uint32_t* __restrict__ del, uint32_t del_words, uint32_t n_sub, uint32_t n_terms, uint32_t nq, uint32_t P, uint32_t k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wb = (uint32_t)w * S16_WAVE_LDS, accb = wb + 14u;
  s16_clear(wb, lane);
  const uint32_t row_len = n_sub + 1;
  const int lane16 = lane * 16;
  const uint32_t a = blockIdx.x * S16_WAVES + w;
  if (a >= nq * P) return;
  const uint32_t qi = a % nq, part = a / nq;
  const bm_vquery* __restrict__ Q = qs + qi;
  const uint32_t nt = min(Q->n_terms, (uint32_t)BM_MAX_VTERMS);
  // lane t: everything about list t
  const bool mine = (uint32_t)lane < nt;
  const uint32_t term = mine ? Q->term[mine ? lane : 0] : n_terms;
  const float idf = mine ? Q->idf[mine ? lane : 0] : 0.f;
  float idf_sum = idf;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) idf_sum += __shfl_xor(idf_sum, o);
  const float scale = S16_QMAX / (S16_WMAX * idf_sum);
  const float scale_thr = s16_uniform(scale * (1.0f - 1e-5f));
  const u64 taddr = (u64)(uintptr_t)(post + term_base[term] * 4ull);
  const uint32_t* __restrict__ rowp = sub_off + (size_t)term * row_len;
  S16MLists L;
  L.v_tlo = (uint32_t)taddr; L.v_thi = (uint32_t)(taddr >> 32);
  L.v_idf = __float_as_uint(idf); L.v_fidf = __float_as_uint(idf * scale * S16_WFOLD);
  L.nt = nt;
  const uint32_t s_begin = (uint32_t)(((u64)n_sub * part) / P);
  const uint32_t s_end = (uint32_t)(((u64)n_sub * (part + 1)) / P);
  BmTop<KPL> T;
#pragma unroll
  for (int r = 0; r < KPL; r++) T.keys[r] = 0ull;
  T.worst = 0ull;
  T.wsc = -1.0f;
  T.matched = 0;
  uint32_t* tau_q = tau + (size_t)qi * BM_TAU_STRIDE;
  if (s_begin < s_end) {
    L.v_a = rowp[s_begin];
    L.v_b = rowp[s_begin + 1u];
    for (uint32_t s = s_begin; s < s_end; s++) {
      const uint32_t tau_bits = __hip_atomic_load(tau_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t v_c = rowp[min(s + 2u, n_sub)];  // the boundaries the NEXT item ends on: in flight under this item's lists
      uint32_t mx = 0u, nocount = 0u;
      bool any = false;
      // the first chunks of lists t + 1 .. t + 3 are in flight while list t is accumulated (a list past the query's last one: lane nt
      // and above hold the absent term's empty segment, the load returns NULL postings without touching memory)
      auto first_chunk = [&](uint32_t t) -> u32x4 {
        const uint32_t x0 = (uint32_t)__builtin_amdgcn_readlane((int)L.v_a, t), x1 = (uint32_t)__builtin_amdgcn_readlane((int)L.v_b, t);
        return __builtin_amdgcn_raw_buffer_load_b128(s16m_rsrc(L, t, x1), lane16, (int)(x0 << 4), 0);
      };
      u32x4 n1 = first_chunk(0), n2 = first_chunk(1), n3 = first_chunk(2);
      for (uint32_t t = 0; t < nt; t++) {
        const u32x4 cur = n1;
        n1 = n2; n2 = n3;
        n3 = first_chunk(min(t + 3u, 63u));
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)L.v_a, t), c1 = (uint32_t)__builtin_amdgcn_readlane((int)L.v_b, t);
        if (c0 == c1) continue;
        any = true;
        const float fidf_t = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)L.v_fidf, t));
        mx = s16_keep<false>(cur, fidf_t, accb, mx, nocount);
        if (c1 - c0 > 64u) {  // a dense segment: the rest, four chunks in flight
          __amdgpu_buffer_rsrc_t rs = s16m_rsrc(L, t, c1);
          for (uint32_t u = c0 + 64u; u < c1; u += 256u) {
            const u32x4 x0 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (int)(u << 4), 0);
            const u32x4 x1 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 1024, (int)(u << 4), 0);
            const u32x4 x2 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 2048, (int)(u << 4), 0);
            const u32x4 x3 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 3072, (int)(u << 4), 0);
            mx = s16_keep<false>(x0, fidf_t, accb, mx, nocount);
            if (u + 64u < c1) mx = s16_keep<false>(x1, fidf_t, accb, mx, nocount);
            if (u + 128u < c1) mx = s16_keep<false>(x2, fidf_t, accb, mx, nocount);
            if (u + 192u < c1) mx = s16_keep<false>(x3, fidf_t, accb, mx, nocount);
          }
        }
      }
      if (any) {
        const float thr = fmaxf(T.wsc, __uint_as_float(tau_bits));
        const uint32_t qthr = thr > 0.f ? (uint32_t)(thr * scale_thr) : 0u;
#ifdef S16M_NOTRIG  // measurement only: the streaming loop without its candidate path (answers are wrong)
        if (false) {
#else
        if (__ballot(mx >= qthr + S16_MBITS)) {  // mx = the bits of 2^23 + the largest bound (s16_qm)
#endif
          const S16Excl ex{Q, post, term_base, sub_off, del, Q->n_terms, bm_q_nnot(Q->op), row_len, del_words, s};
          T = s16m_trigger<KPL>(T, L, wb, qthr, thr, s << BM_SUB_LOG2, k, tau_q, ex);
          // A partition's own k-th best says little about the QUERY's: with 128 partitions the query's ten best docs sit in ten of them,
          // and the largest "tenth best of one partition" (what tau carried) stayed at 21 where the answer's tenth score was above 30 --
          // 40 % of the items of a 16-term union took the candidate path.  The partitions' BEST keys are k distinct docs as soon as k
          // partitions have one: the k-th largest of them bounds the query's k-th best score from below, and it is close.  Every
          // partition keeps its best key in rank 0 of its output list (zeroed by bm_expand_kernel); whoever has just been through the
          // candidate path reads them all and raises tau.
          if (KPL == 1 && k <= 64u) {
            u64* slots = part_keys + (size_t)qi * P * 64u;
            if (lane == 0) __hip_atomic_store(slots + (size_t)part * 64u, T.keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u64 m = 0ull;
            for (uint32_t p_ = (uint32_t)lane; p_ < P; p_ += 64u) {
              const u64 x = __hip_atomic_load(slots + (size_t)p_ * 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              m = x > m ? x : m;
            }
            m = wave_sort_desc(m, lane);
            const u64 kth = rdlane64(m, (int)k - 1);
            if (kth && lane == 0) bm_publish_tau(tau_q, __uint_as_float((uint32_t)(kth >> 32)));
          }
        } else {
          s16_clear(wb, lane);
        }
      }
      L.v_a = L.v_b;
      L.v_b = v_c;
    }
  }
  u64* out = part_keys + ((size_t)qi * P + part) * (64 * KPL);
#pragma unroll
  for (int r = 0; r < KPL; r++) out[r * 64 + lane] = T.keys[r];
}

template <int KPL>
int launch16m(const BmParams& p, hipStream_t st) {
  constexpr int lds = S16_WAVES * S16_WAVE_LDS;
  SS_SET_MAX_LDS((bm25_scan16m_kernel<KPL>), lds);
  const uint32_t A = p.nq * p.P;
  bm25_scan16m_kernel<KPL><<<(A + S16_WAVES - 1) / S16_WAVES, S16_WAVES * 64, lds, st>>>(
      p.post, p.term_base, p.sub_off, p.q, p.part_keys, p.tau, p.del, p.del_words, p.n_sub, p.n_terms, p.nq, p.P, p.k);
  return SS_OK;
}

}  // namespace

// What the 16-bit tile serves (everything else of the exhaustive strategy stays on bm25_scan_fast_kernel's f32 tile):
//  * unions of <= 4 lists, k <= 128 (five and six lists: top-k only, k <= 64), exact counts (TopkCount, and Count with k = 0);
//  * NOT lists (nn_max of them in some query): a top-k request never streams them -- the candidate path clears their docs from the tile before it
//    looks at it (s16_exclude); a count request streams ONE NOT list beside the terms (EXCL instances);
//  * tombstones: top-k in the candidate path, counts in the EXCL instances (the sub-block's 128 tombstone words per item);
//  * intersections (and_exact_nt != 0): batches of intersections only, every query with exactly and_exact_nt = 2 or 3 terms over one
//    list each, no all_terms_frequent shortcut (the dispatch checks); NOT lists / tombstones as for unions.
bool ssi_bm25_scan16_serves(uint32_t nn_max, uint32_t np_max, bool has_and, bool count, bool tombstones, int KPL, uint32_t k, uint32_t and_exact_nt) {
  // (the per-family switches of rounds 3-5 -- SS_BM25_SCAN16[_COUNT|_AND|_EXCL|_WIDE|_K128|_MANY]=0 -- are gone: SS_BM25_EXHAUSTIVE_F32 is
  // the strategy that holds a batch on the f32 tile)
  constexpr int off = 0, cnt_off = 0, and_off = 0, excl_off = 0, wide_off = 0;
  const uint32_t nn = nn_max;  // NOT lists of the query that has the most
  if (count && cnt_off) return false;
  if ((nn || tombstones) && excl_off) return false;
  if (count && nn > 1) return false;    // the count instances stream one NOT list
  if (nn > 8) return false;
  if (has_and && (and_off || and_exact_nt < 2 || and_exact_nt > 3 || and_exact_nt != np_max)) return false;
  if (np_max > 4 && (has_and || count || wide_off)) return false;  // five and more lists: plain top-k unions
  constexpr int k128_off = 0, many_off = 0;
  // 7 .. 32 lists, and 5 / 6 at k of 65 .. 128: bm25_scan16m_kernel (lists as a run-time loop)
  if (np_max > 6 || (np_max > 4 && KPL == 2)) return !off && !many_off && !k128_off && k != 0 && np_max <= (uint32_t)BM_MAX_VTERMS && KPL <= 2;
  if (KPL == 2 && k128_off) return false;  // k <= 128: two keys per lane in the candidate path (no k-lane cut)
  return !off && (k != 0 || count) && np_max >= 1 && np_max <= 6 && KPL <= 2;
}

int ssi_bm25_launch_scan16(const BmParams& p, uint32_t np_max, uint32_t nn_max, bool is_and, int KPL, hipStream_t st) {
  const int NT = np_max <= 2 ? 2 : (int)np_max;
  const bool excl = p.count && (p.del != nullptr || nn_max != 0);  // exclusions inside the streaming loop: exact counts only
  if (KPL != 1 && KPL != 2) return SS_ENOTSUP;
  if (is_and) {
#define SS_A(NT_, KPL_)                                                                                                    \
  if (NT == NT_ && KPL == KPL_)                                                                                            \
    return !p.count ? launch16<NT_, KPL_, false, true>(p, st) : excl ? launch16<NT_, KPL_, true, true, true>(p, st) : launch16<NT_, KPL_, true, true>(p, st);
    SS_A(2, 1) SS_A(3, 1) SS_A(2, 2) SS_A(3, 2)
#undef SS_A
    return SS_ENOTSUP;
  }
#define SS_F(NT_, KPL_)                                                                                                    \
  if (NT == NT_ && KPL == KPL_)                                                                                            \
    return !p.count ? launch16<NT_, KPL_, false, false>(p, st) : excl ? launch16<NT_, KPL_, true, false, true>(p, st) : launch16<NT_, KPL_, true, false>(p, st);
  SS_F(2, 1) SS_F(3, 1) SS_F(4, 1) SS_F(2, 2) SS_F(3, 2) SS_F(4, 2)
#undef SS_F
  if (NT == 5 && !p.count && KPL == 1) return launch16<5, 1, false, false>(p, st);
  if (NT == 6 && !p.count && KPL == 1) return launch16<6, 1, false, false>(p, st);
  if (NT >= 5 && !p.count) return KPL == 1 ? launch16m<1>(p, st) : launch16m<2>(p, st);
  return SS_ENOTSUP;
}
