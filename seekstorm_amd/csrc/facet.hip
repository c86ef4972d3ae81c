// Facet filter of the lexical search (search.rs FacetFilter -> FilterSparse, add_result.rs:341-482 is_facet_filter) for gfx950.
//
// The reference keeps one fixed-size record per doc in facet.bin (facets_size_sum bytes, each facet at its offset) and,
// per candidate doc, drops the doc unless EVERY filtered facet passes: a numeric value inside the half-open Rust range
// [start, end), a string id (String16 / String32) inside the set of wanted ids.  A filtered doc neither counts nor ranks
// (add_result.rs:3499-3501 returns before the count) -- exactly what a tombstone does.  So the filter is evaluated ONCE per
// call over all docs into an exclusion bitmap (OR-ed with the tombstones: ~n_docs x record bytes read, 20-50 us at 10 M
// docs) and the search kernels run with that bitmap in place of the tombstone bitmap; nothing in them changes.
#include <cstring>

#include "ss_common.h"

struct FacetFilters {
  uint32_t n;
  ss_facet_filter f[SS_MAX_FACET_FILTERS];
};

__device__ __forceinline__ unsigned long long facet_read(const uint8_t* p, uint32_t bytes) {
  unsigned long long v = 0;
  for (uint32_t b = 0; b < bytes; b++) v |= (unsigned long long)p[b] << (8u * b);  // little endian, any alignment
  return v;
}

__device__ bool facet_pass(const uint8_t* rec, const ss_facet_filter& f) {
  const uint8_t* p = rec + f.offset;
  switch (f.type) {
    case SS_FACET_U8: { const unsigned long long v = facet_read(p, 1); return v >= f.lo && v < f.hi; }
    case SS_FACET_U16: { const unsigned long long v = facet_read(p, 2); return v >= f.lo && v < f.hi; }
    case SS_FACET_U32: { const unsigned long long v = facet_read(p, 4); return v >= f.lo && v < f.hi; }
    case SS_FACET_U64: { const unsigned long long v = facet_read(p, 8); return v >= f.lo && v < f.hi; }
    case SS_FACET_I8: { const long long v = (int8_t)facet_read(p, 1); return v >= (long long)f.lo && v < (long long)f.hi; }
    case SS_FACET_I16: { const long long v = (int16_t)facet_read(p, 2); return v >= (long long)f.lo && v < (long long)f.hi; }
    case SS_FACET_I32: { const long long v = (int32_t)facet_read(p, 4); return v >= (long long)f.lo && v < (long long)f.hi; }
    case SS_FACET_I64: { const long long v = (long long)facet_read(p, 8); return v >= (long long)f.lo && v < (long long)f.hi; }
    case SS_FACET_F32: {
      const float v = __uint_as_float((uint32_t)facet_read(p, 4));
      return v >= __uint_as_float((uint32_t)f.lo) && v < __uint_as_float((uint32_t)f.hi);
    }
    case SS_FACET_F64: {
      const double v = __longlong_as_double((long long)facet_read(p, 8));
      return v >= __longlong_as_double((long long)f.lo) && v < __longlong_as_double((long long)f.hi);
    }
    case SS_FACET_STRING16:
    case SS_FACET_STRING32: {
      const uint32_t v = (uint32_t)facet_read(p, f.type == SS_FACET_STRING16 ? 2 : 4);
      for (uint32_t i = 0; i < f.n_values; i++)
        if (f.values[i] == v) return true;
      return false;
    }
    default: return true;
  }
}

// one thread per doc, one bitmap word per 32 docs: bit set = the doc is excluded (a failed facet or a tombstone)
__global__ void facet_filter_kernel(const uint8_t* __restrict__ records, uint32_t record_size, unsigned long long n_docs,
                                    FacetFilters F, const uint32_t* __restrict__ del, uint32_t del_words,
                                    uint32_t* __restrict__ out) {
  const unsigned long long d = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool excluded = false;
  if (d < n_docs) {
    const uint8_t* rec = records + d * record_size;
    for (uint32_t i = 0; i < F.n && !excluded; i++) excluded = !facet_pass(rec, F.f[i]);
  }
  const unsigned long long b = __ballot(excluded);
  const uint32_t lane = threadIdx.x & 63u;
  if ((lane & 31u) == 0) {
    const unsigned long long w = d >> 5;
    if (w < (n_docs + 31) / 32) {
      uint32_t bits = (uint32_t)(b >> (lane & 32u));
      if (w < del_words) bits |= del[w];
      out[w] = bits;
    }
  }
}

int ssi_facet_build(ss_shard* s, uint32_t n_filters, const ss_facet_filter* filters, hipStream_t st) {
  if (!s->d_facets) return SS_ESTATE;
  if (n_filters == 0 || n_filters > SS_MAX_FACET_FILTERS || !filters) return SS_EINVAL;
  FacetFilters F;
  F.n = n_filters;
  for (uint32_t i = 0; i < n_filters; i++) {
    const ss_facet_filter& f = filters[i];
    static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4};
    if (f.type > SS_FACET_STRING32 || f.offset + width[f.type] > s->facet_record_size) return SS_EINVAL;
    if ((f.type == SS_FACET_STRING16 || f.type == SS_FACET_STRING32) && f.n_values > 8) return SS_EINVAL;
    F.f[i] = f;
  }
  const uint64_t words = (s->facet_docs + 31) / 32;
  if (words > s->filter_words_cap) {
    if (s->d_filter_bits) (void)hipFree(s->d_filter_bits);
    s->d_filter_bits = nullptr;
    s->filter_words_cap = 0;
    SS_HIP(hipMalloc(&s->d_filter_bits, words * sizeof(uint32_t)));
    s->filter_words_cap = words;
  }
  facet_filter_kernel<<<(unsigned)((s->facet_docs + 255) / 256), 256, 0, st>>>(
      s->d_facets, s->facet_record_size, (unsigned long long)s->facet_docs, F, s->n_deleted ? s->d_deleted : nullptr,
      (uint32_t)s->deleted_words, s->d_filter_bits);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Facet counting (facet_count, add_result.rs:484-640): every counted doc adds one to the bucket of its facet value -- a
// string facet's id, or for a numeric facet the range whose lower bound is the last one <= the value (ranges sorted by
// lower bound, binary_search_by_key ... map_or_else(|idx| idx - 1, |idx| idx)).  The counted docs are the query's match
// set after NOT terms, tombstones and the facet filter: the bitmap bm25_union_count_kernel writes from the bit records.
// One thread per 64-doc group walks its set bits; the histogram lives in global memory (64-bit atomics).
__device__ __forceinline__ bool facet_le(uint32_t type, unsigned long long bound, unsigned long long v) {  // bound <= v
  switch (type) {
    case SS_FACET_I8: case SS_FACET_I16: case SS_FACET_I32: case SS_FACET_I64: return (long long)bound <= (long long)v;
    case SS_FACET_F32: return __uint_as_float((uint32_t)bound) <= __uint_as_float((uint32_t)v);
    case SS_FACET_F64: return __longlong_as_double((long long)bound) <= __longlong_as_double((long long)v);
    default: return bound <= v;
  }
}
__global__ void facet_count_kernel(const unsigned long long* __restrict__ bits, unsigned long long n_docs,
                                   const uint8_t* __restrict__ records, uint32_t record_size, uint32_t offset, uint32_t type,
                                   uint32_t n_buckets, const unsigned long long* __restrict__ bounds,
                                   unsigned long long* __restrict__ counts) {
  const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g * 64ull >= n_docs) return;
  unsigned long long m = bits[g];
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4};
  while (m) {
    const unsigned long long d = g * 64ull + (unsigned long long)__builtin_ctzll(m);
    m &= m - 1;
    if (d >= n_docs) break;
    unsigned long long v = facet_read(records + d * record_size + offset, width[type]);
    if (type == SS_FACET_I8) v = (unsigned long long)(long long)(int8_t)v;      // sign-extend for the comparisons
    else if (type == SS_FACET_I16) v = (unsigned long long)(long long)(int16_t)v;
    else if (type == SS_FACET_I32) v = (unsigned long long)(long long)(int32_t)v;
    uint32_t b;
    if (type >= SS_FACET_STRING16) {
      b = v < n_buckets ? (uint32_t)v : n_buckets;
    } else {
      uint32_t lo = 0, hi = n_buckets;  // number of bounds <= v
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (facet_le(type, bounds[mid], v)) lo = mid + 1; else hi = mid;
      }
      b = lo ? lo - 1 : n_buckets;  // below the first bound: the reference's index underflows; reported as "other"
    }
    atomicAdd(&counts[b], 1ull);
  }
}

int ssi_facet_count(ss_shard* s, const unsigned long long* d_bits, uint64_t n_docs, uint32_t offset, uint32_t type, uint32_t n_buckets,
                    const uint64_t* d_bounds, unsigned long long* d_counts, hipStream_t st) {
  const uint64_t groups = (n_docs + 63) / 64;
  facet_count_kernel<<<(unsigned)((groups + 255) / 256), 256, 0, st>>>(d_bits, (unsigned long long)n_docs, s->d_facets,
                                                                      s->facet_record_size, offset, type, n_buckets,
                                                                      (const unsigned long long*)d_bounds, d_counts);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
