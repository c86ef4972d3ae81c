// Lookups of ONE doc in a posting list of either tier: what north_star calls the galloping intersection (intersection.rs:352-362) --
// a doc of the list that drives a query is looked up in the query's other lists by binary search.  Shared by the sparse tier's kernels
// (bm25_sparse.hip) and the generic intersection / phrase kernels (bm25_gallop.hip).  Internal.
#pragma once
#include "bm25_dev.h"

// index of doc in the sparse list [lo, hi) (ascending docs in the low 32 bits), or ~0
__device__ __forceinline__ unsigned long long sp_find(const unsigned long long* __restrict__ sp, unsigned long long lo, unsigned long long hi, uint32_t doc) {
  while (lo < hi) {
    const unsigned long long mid = (lo + hi) >> 1;
    const uint32_t d = (uint32_t)sp[mid];
    if (d < doc) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// weight code of doc in a DENSE list, 0 = absent: binary search inside the doc's (term, sub-block) segment -- packed postings
// ascending by doc field, NULL (zero) padding at the segment's end ordering as +infinity.  *slot = the posting's index inside the
// term's image (what d_pos_off is indexed by).
__device__ __forceinline__ uint32_t dense_find(const uint32_t* __restrict__ post, const unsigned long long* __restrict__ term_base,
                                               const uint32_t* __restrict__ sub_off, uint32_t n_sub, uint32_t row, uint32_t doc,
                                               uint32_t* slot = nullptr) {
  const uint32_t sb = doc >> BM_SUB_LOG2, want = (doc & (BM_SUB - 1)) + 1u;
  const uint32_t* r = sub_off + (size_t)row * (n_sub + 1);
  const unsigned long long base = (term_base[row] + r[sb]) * 4ull;
  uint32_t lo = 0, hi = (r[sb + 1] - r[sb]) * 4u;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    const uint32_t f = bm_doc_field(post[base + mid]);
    if (f != 0u && f < want) lo = mid + 1; else hi = mid;
  }
  if (lo < (r[sb + 1] - r[sb]) * 4u) {
    const uint32_t p = post[base + lo];
    if (slot) *slot = r[sb] * 4u + lo;
    if (bm_doc_field(p) == want) return p >> 13 ? p >> 13 : 1u;  // (a code of 0 cannot occur: bm_wcode clamps to 1)
  }
  return 0u;
}
