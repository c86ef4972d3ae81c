// Facet filter of the lexical search (search.rs FacetFilter -> FilterSparse, add_result.rs:341-482 is_facet_filter) for gfx950.
//
// The reference keeps one fixed-size record per doc in facet.bin (facets_size_sum bytes, each facet at its offset) and,
// per candidate doc, drops the doc unless EVERY filtered facet passes: a numeric value inside the half-open Rust range
// [start, end), a string id (String16 / String32) inside the set of wanted ids.  A filtered doc neither counts nor ranks
// (add_result.rs:3499-3501 returns before the count) -- exactly what a tombstone does.  So the filter is evaluated ONCE per
// call over all docs into an exclusion bitmap (OR-ed with the tombstones: ~n_docs x record bytes read, 20-50 us at 10 M
// docs) and the search kernels run with that bitmap in place of the tombstone bitmap; nothing in them changes.
#include <algorithm>
#include <cstring>
#include <vector>

#include "facet_point.h"

struct FacetFilters {
  uint32_t n;
  ss_facet_filter f[SS_MAX_FACET_FILTERS];
};

__device__ __forceinline__ unsigned long long facet_read(const uint8_t* p, uint32_t bytes) {
  unsigned long long v = 0;
  for (uint32_t b = 0; b < bytes; b++) v |= (unsigned long long)p[b] << (8u * b);  // little endian, any alignment
  return v;
}

// ---- Point facets (FieldType::Point, index.rs:1050-1056): the stored value is the 64-bit Morton code of (lat, lon) x 1e7
// as i32 (geo_search.rs:27-41: lat in the even bits, lon in the odd ones).  What filters, counts and sorts is a DISTANCE
// to a base point, in f64 like the reference, operation by operation in its order (no fused multiply-add):
//   unit km / miles: euclidian_distance(base, doc) = R * sqrt(x^2 + y^2), x = DEG2RAD*(dlon)*cos(DEG2RAD*(lat1+lat2)/2),
//                    y = DEG2RAD*(dlat)                                   (geo_search.rs:115-124; facet filter, facet count)
//   sort key:        simplified_distance(doc, base) = x^2 + y^2 without DEG2RAD and R (geo_search.rs:82-87, morton_ordering)
// cos / sqrt are the device library's f64 routines: a last-place difference from the host's libm can move a doc that sits
// within an ulp of a range bound, nothing else.
struct FacetPoint {
  double lat, lon, radius;  // radius 0: the sort key
};
__host__ __device__ inline uint32_t morton_compact(unsigned long long x) {  // the even bits of x, packed
  x &= 0x5555555555555555ull;
  x = (x ^ (x >> 1)) & 0x3333333333333333ull;
  x = (x ^ (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x ^ (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x ^ (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x ^ (x >> 16)) & 0x00000000FFFFFFFFull;
  return (uint32_t)x;
}
__host__ inline unsigned long long morton_spread(uint32_t v) {
  unsigned long long x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}
__host__ inline uint32_t morton_coord(double degrees) {  // (v * 1e7) as i32 as u32: Rust's cast truncates, saturates, NaN -> 0
  const double v = degrees * 10000000.0;
  int32_t i;
  if (v != v) i = 0;
  else if (v >= 2147483647.0) i = INT32_MAX;
  else if (v <= -2147483648.0) i = INT32_MIN;
  else i = (int32_t)v;
  return (uint32_t)i;
}
__host__ inline unsigned long long morton_encode(double lat, double lon) { return (morton_spread(morton_coord(lon)) << 1) | morton_spread(morton_coord(lat)); }
static constexpr double kDeg2Rad = 0.017453292519943295, kEarthKm = 6371.0087714, kEarthMi = 3958.761315801475;

__device__ double facet_point_distance(const FacetPoint& b, unsigned long long code) {
#pragma clang fp contract(off)
  const double lat = (double)(int32_t)morton_compact(code) / 10000000.0;
  const double lon = (double)(int32_t)morton_compact(code >> 1) / 10000000.0;
  const double c = cos(kDeg2Rad * (b.lat + lat) / 2.0);
  if (b.radius != 0.0) {
    const double x = kDeg2Rad * (lon - b.lon) * c, y = kDeg2Rad * (lat - b.lat);
    return b.radius * sqrt(x * x + y * y);
  }
  const double x = (b.lon - lon) * c, y = b.lat - lat;
  return x * x + y * y;
}
// a facet's value as the bits the comparisons below work on: Point -> the f64 distance (from then on an F64 facet)
__device__ __forceinline__ unsigned long long facet_load(const uint8_t* p, uint32_t bytes, uint32_t& type, const FacetPoint& pt) {
  const unsigned long long v = facet_read(p, bytes);
  if (type != SS_FACET_POINT) return v;
  type = SS_FACET_F64;
  return (unsigned long long)__double_as_longlong(facet_point_distance(pt, v));
}
__host__ inline int facet_point_of(const ss_facet_point* p, FacetPoint* out) {
  if (!p || p->unit > SS_POINT_MILES) return SS_EINVAL;
  *out = FacetPoint{p->lat, p->lon, p->unit == SS_POINT_KM ? kEarthKm : p->unit == SS_POINT_MILES ? kEarthMi : 0.0};
  return SS_OK;
}

// lo <= v < hi as Rust's Range::contains; the two flag bits turn the ends around (the pivots of a result sort need
// "strictly above" and "equal to" for every type, and a 64-bit type has no value above its largest to write as an end)
template <typename T>
__device__ __forceinline__ bool facet_in(T v, T lo, T hi, uint32_t flags) {
  const bool above = (flags & SS_FACET_LO_EXCLUSIVE) ? v > lo : v >= lo;
  const bool below = (flags & SS_FACET_HI_INCLUSIVE) ? v <= hi : v < hi;
  return above && below;
}
__device__ bool facet_pass(const uint8_t* rec, const ss_facet_filter& f) {
  const uint8_t* p = rec + f.offset;
  const uint32_t fl = f.reserved;
  switch (f.type) {
    case SS_FACET_U8: return facet_in<unsigned long long>(facet_read(p, 1), f.lo, f.hi, fl);
    case SS_FACET_U16: return facet_in<unsigned long long>(facet_read(p, 2), f.lo, f.hi, fl);
    case SS_FACET_U32: return facet_in<unsigned long long>(facet_read(p, 4), f.lo, f.hi, fl);
    case SS_FACET_U64: return facet_in<unsigned long long>(facet_read(p, 8), f.lo, f.hi, fl);
    case SS_FACET_I8: return facet_in<long long>((int8_t)facet_read(p, 1), (long long)f.lo, (long long)f.hi, fl);
    case SS_FACET_I16: return facet_in<long long>((int16_t)facet_read(p, 2), (long long)f.lo, (long long)f.hi, fl);
    case SS_FACET_I32: return facet_in<long long>((int32_t)facet_read(p, 4), (long long)f.lo, (long long)f.hi, fl);
    case SS_FACET_I64: return facet_in<long long>((long long)facet_read(p, 8), (long long)f.lo, (long long)f.hi, fl);
    case SS_FACET_F32:
      return facet_in<float>(__uint_as_float((uint32_t)facet_read(p, 4)), __uint_as_float((uint32_t)f.lo), __uint_as_float((uint32_t)f.hi), fl);
    case SS_FACET_F64:
      return facet_in<double>(__longlong_as_double((long long)facet_read(p, 8)), __longlong_as_double((long long)f.lo),
                              __longlong_as_double((long long)f.hi), fl);
    case SS_FACET_STRING16:
    case SS_FACET_STRING32: {
      const uint32_t v = (uint32_t)facet_read(p, f.type == SS_FACET_STRING16 ? 2 : 4);
      if (f.n_values == SS_FACET_IDS_EXTERN)  // a set of any size: one bit per id (ssi_facet_build put it behind the exclusion bitmap)
        return v < f.hi && ((((const uint32_t*)(uintptr_t)f.lo)[v >> 5] >> (v & 31u)) & 1u) != 0u;
      for (uint32_t i = 0; i < f.n_values; i++)
        if (f.values[i] == v) return true;
      return false;
    }
    case SS_FACET_POINT: {  // FilterSparse::Point, add_result.rs:462-478: inside the Morton range AND inside the distance range
      const unsigned long long code = facet_read(p, 8);
      unsigned long long w[4];
      for (int i = 0; i < 4; i++) w[i] = (unsigned long long)f.values[2 * i] | ((unsigned long long)f.values[2 * i + 1] << 32);
      const FacetPoint b{__longlong_as_double((long long)w[0]), __longlong_as_double((long long)w[1]),
                         f.n_values == SS_POINT_KM ? kEarthKm : f.n_values == SS_POINT_MILES ? kEarthMi : 0.0};
      if (f.n_values != SS_POINT_SORTKEY && !(code >= w[2] && code < w[3])) return false;
      return facet_in<double>(facet_point_distance(b, code), __longlong_as_double((long long)f.lo), __longlong_as_double((long long)f.hi), fl);
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
  uint64_t set_words = 0, set_begin[SS_MAX_FACET_FILTERS] = {0}, set_bits[SS_MAX_FACET_FILTERS] = {0};
  for (uint32_t i = 0; i < n_filters; i++) {
    const ss_facet_filter& f = filters[i];
    static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4, 8};
    if (f.type > SS_FACET_POINT || f.offset + width[f.type] > s->facet_record_size) return SS_EINVAL;
    const bool strings = f.type == SS_FACET_STRING16 || f.type == SS_FACET_STRING32;
    if (strings && f.n_values > 8 && f.n_values != SS_FACET_IDS_EXTERN) return SS_EINVAL;
    F.f[i] = f;
    if (strings && f.n_values == SS_FACET_IDS_EXTERN) {  // lo = host array of hi ids -> a bitmap over [0, largest id]
      const uint32_t* ids = (const uint32_t*)(uintptr_t)f.lo;
      if (f.hi && !ids) return SS_EINVAL;
      uint64_t bits = 0;
      for (uint64_t j = 0; j < f.hi; j++) bits = std::max<uint64_t>(bits, (uint64_t)ids[j] + 1);
      if (f.type == SS_FACET_STRING16) bits = std::min<uint64_t>(bits, 65536);  // ids beyond the facet's width match nothing
      if (bits > (1ull << 30)) return SS_EINVAL;
      set_begin[i] = set_words;
      set_bits[i] = bits;
      set_words += (bits + 31) / 32;
    }
    if (f.type == SS_FACET_POINT) {
      if (f.n_values > SS_POINT_MILES) return SS_EINVAL;
      if (f.n_values != SS_POINT_SORTKEY) {
        // point_distance_to_morton_range(base, distance_range.end, unit), search.rs:2712-2722 / geo_search.rs:128-144: the
        // codes of the two corners of the box around the base -- a range of the Z-ORDER, as the reference compares it
        double lat, lon, end;
        unsigned long long w0 = (unsigned long long)f.values[0] | ((unsigned long long)f.values[1] << 32),
                           w1 = (unsigned long long)f.values[2] | ((unsigned long long)f.values[3] << 32);
        memcpy(&lat, &w0, 8); memcpy(&lon, &w1, 8); memcpy(&end, &f.hi, 8);
        const double radius = f.n_values == SS_POINT_KM ? kEarthKm : kEarthMi;
        const double lat_delta = end / (kDeg2Rad * radius), lon_delta = end / (kDeg2Rad * radius * cos(kDeg2Rad * lat));
        const unsigned long long m0 = morton_encode(lat - lat_delta, lon - lon_delta), m1 = morton_encode(lat + lat_delta, lon + lon_delta);
        F.f[i].values[4] = (uint32_t)m0; F.f[i].values[5] = (uint32_t)(m0 >> 32);
        F.f[i].values[6] = (uint32_t)m1; F.f[i].values[7] = (uint32_t)(m1 >> 32);
      }
    }
  }
  const uint64_t words = (s->facet_docs + 31) / 32;
  if (words + set_words > s->filter_words_cap) {
    if (s->d_filter_bits) (void)hipFree(s->d_filter_bits);
    s->d_filter_bits = nullptr;
    s->filter_words_cap = 0;
    SS_HIP(hipMalloc(&s->d_filter_bits, (words + set_words) * sizeof(uint32_t)));
    s->filter_words_cap = words + set_words;
  }
  if (set_words) {  // the id sets as bitmaps behind the exclusion bitmap
    // staging that outlives the call (the copy is asynchronous on st; calls that share a shard are stream-ordered)
    static thread_local std::vector<uint32_t> stage;
    stage.assign(set_words, 0u);
    for (uint32_t i = 0; i < n_filters; i++) {
      const ss_facet_filter& f = filters[i];
      if (!((f.type == SS_FACET_STRING16 || f.type == SS_FACET_STRING32) && f.n_values == SS_FACET_IDS_EXTERN)) continue;
      const uint32_t* ids = (const uint32_t*)(uintptr_t)f.lo;
      for (uint64_t j = 0; j < f.hi; j++)
        if (ids[j] < set_bits[i]) stage[set_begin[i] + (ids[j] >> 5)] |= 1u << (ids[j] & 31u);
      F.f[i].lo = (uint64_t)(uintptr_t)(s->d_filter_bits + words + set_begin[i]);
      F.f[i].hi = set_bits[i];
    }
    SS_HIP(hipMemcpyAsync(s->d_filter_bits + words, stage.data(), set_words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
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
                                   unsigned long long* __restrict__ counts, uint32_t stored_type, FacetPoint pt) {
  const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g * 64ull >= n_docs) return;
  unsigned long long m = bits[g];
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4, 8};
  while (m) {
    const unsigned long long d = g * 64ull + (unsigned long long)__builtin_ctzll(m);
    m &= m - 1;
    if (d >= n_docs) break;
    type = stored_type;
    unsigned long long v = facet_load(records + d * record_size + offset, width[stored_type], type, pt);
    if (type == SS_FACET_I8) v = (unsigned long long)(long long)(int8_t)v;      // sign-extend for the comparisons
    else if (type == SS_FACET_I16) v = (unsigned long long)(long long)(int16_t)v;
    else if (type == SS_FACET_I32) v = (unsigned long long)(long long)(int32_t)v;
    uint32_t b;
    if (type == SS_FACET_STRING16 || type == SS_FACET_STRING32) {
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
                    const uint64_t* d_bounds, unsigned long long* d_counts, const ss_facet_point* point, hipStream_t st) {
  FacetPoint pt{0, 0, 0};
  if (type == SS_FACET_POINT && facet_point_of(point, &pt) != SS_OK) return SS_EINVAL;
  const uint64_t groups = (n_docs + 63) / 64;
  facet_count_kernel<<<(unsigned)((groups + 255) / 256), 256, 0, st>>>(d_bits, (unsigned long long)n_docs, s->d_facets,
                                                                      s->facet_record_size, offset, type, n_buckets,
                                                                      (const unsigned long long*)d_bounds, d_counts, type, pt);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Result sort (search.rs ResultSort, min_heap.rs:574-1050 result_ordering_shard): results ordered by facet values -- each
// field ascending or descending, the score last -- instead of by score alone.  The top-k under such an order is found from
// its PIVOT: the k-th best value of the first sort field among the query's matches (a byte-wise radix select over the match
// set: one histogram pass per byte of the field), after which "strictly better than the pivot" and "equal to the pivot" are
// ordinary facet filters on ordinary searches (the hosts compose them: seekstorm_amd/search.py search_lexical_sorted).
// key: the stored value mapped to an unsigned integer that orders like the value (sign bit flipped for integers, the usual
// transform for floats), complemented for an ascending sort, so that a larger key is always the better one.
__host__ __device__ inline unsigned long long facet_order_key(unsigned long long v, uint32_t type, uint32_t bits, bool descending) {
  const unsigned long long mask = bits == 64 ? ~0ull : ((1ull << bits) - 1ull), top = 1ull << (bits - 1);
  unsigned long long k = v & mask;
  if (type >= SS_FACET_I8 && type <= SS_FACET_I64) k ^= top;
  else if (type == SS_FACET_F32 || type == SS_FACET_F64) k = (k & top) ? (~k & mask) : (k | top);
  return descending ? k : (~k & mask);
}
__host__ inline unsigned long long facet_order_value(unsigned long long k, uint32_t type, uint32_t bits, bool descending) {
  const unsigned long long mask = bits == 64 ? ~0ull : ((1ull << bits) - 1ull), top = 1ull << (bits - 1);
  if (!descending) k = ~k & mask;
  if (type >= SS_FACET_I8 && type <= SS_FACET_I64) k ^= top;
  else if (type == SS_FACET_F32 || type == SS_FACET_F64) k = (k & top) ? (k & ~top) : (~k & mask);
  return k;
}
// histogram of byte `byte_index` (0 = most significant) of the keys that share `prefix` in the bytes above it
__global__ void facet_radix_kernel(const unsigned long long* __restrict__ bits, unsigned long long n_docs,
                                   const uint8_t* __restrict__ records, uint32_t record_size, uint32_t offset, uint32_t type,
                                   uint32_t key_bits, uint32_t descending, unsigned long long prefix, uint32_t byte_index,
                                   unsigned long long* __restrict__ hist, FacetPoint pt) {
  __shared__ unsigned int h[256];
  h[threadIdx.x & 255u] = 0u;
  __syncthreads();
  const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g * 64ull < n_docs) {
    unsigned long long m = bits[g];
    const uint32_t shift = key_bits - 8u * (byte_index + 1u);
    while (m) {
      const unsigned long long d = g * 64ull + (unsigned long long)__builtin_ctzll(m);
      m &= m - 1;
      if (d >= n_docs) break;
      uint32_t ty = type;
      const unsigned long long v = facet_load(records + d * record_size + offset, key_bits / 8u, ty, pt);
      const unsigned long long k = facet_order_key(v, ty, key_bits, descending != 0u);
      if (byte_index == 0u || (k >> (shift + 8u)) == prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
  }
  __syncthreads();
  if (h[threadIdx.x & 255u]) atomicAdd(&hist[threadIdx.x & 255u], (unsigned long long)h[threadIdx.x & 255u]);
}

// k-th best key of the match set `d_bits` (k >= 1; with fewer than k matches: the worst match).  *n_better = matches
// strictly better than it, *n_equal = matches equal to it.  d_hist: 256 counters of workspace.
int ssi_facet_kth(ss_shard* s, const unsigned long long* d_bits, uint64_t n_docs, uint64_t n_matches, uint32_t offset, uint32_t type,
                  bool descending, uint64_t k, unsigned long long* d_hist, uint64_t* value_bits, uint64_t* n_better, uint64_t* n_equal,
                  const ss_facet_point* point, hipStream_t st) {
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 0, 0, 8};
  if (type == SS_FACET_STRING16 || type == SS_FACET_STRING32) return SS_ENOTSUP;  // they sort by their strings, which live in the host's facet.json
  FacetPoint pt{0, 0, 0};
  if (type == SS_FACET_POINT && facet_point_of(point, &pt) != SS_OK) return SS_EINVAL;  // the key is the f64 distance
  *n_better = 0; *n_equal = 0; *value_bits = 0;
  if (n_matches == 0) return SS_OK;
  const uint32_t key_bits = 8u * width[type];
  uint64_t want = std::min<uint64_t>(k, n_matches);  // rank (1-based) still to be found inside the current prefix
  unsigned long long prefix = 0;
  const uint64_t groups = (n_docs + 63) / 64;
  unsigned long long hist[256];
  uint64_t equal = 0;
  for (uint32_t b = 0; b < key_bits / 8u; b++) {
    SS_HIP(hipMemsetAsync(d_hist, 0, sizeof(hist), st));
    facet_radix_kernel<<<(unsigned)((groups + 255) / 256), 256, 0, st>>>(d_bits, (unsigned long long)n_docs, s->d_facets, s->facet_record_size,
                                                                        offset, type, key_bits, descending ? 1u : 0u, prefix, b, d_hist, pt);
    SS_HIP(hipGetLastError());
    SS_HIP(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
    int v = 255;
    for (; v > 0; v--) {  // from the best byte value down to the bucket that holds the wanted rank
      if (hist[v] >= want) break;
      want -= hist[v];
      *n_better += hist[v];
    }
    prefix = (prefix << 8) | (unsigned long long)v;
    equal = hist[v];
  }
  *n_equal = equal;
  *value_bits = facet_order_value(prefix, type == SS_FACET_POINT ? (uint32_t)SS_FACET_F64 : type, key_bits, descending);
  return SS_OK;
}

__global__ void facet_values_kernel(const uint8_t* __restrict__ records, uint32_t record_size, uint32_t offset, uint32_t bytes,
                                    const uint32_t* __restrict__ docs, uint32_t n, unsigned long long n_docs, unsigned long long* __restrict__ out,
                                    uint32_t type, FacetPoint pt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = docs[i] < n_docs ? facet_load(records + (size_t)docs[i] * record_size + offset, bytes, type, pt) : 0ull;
}
// point == nullptr: the stored bits (a Point facet's Morton code); with a point: the f64 distance to it
int ssi_facet_values(ss_shard* s, const uint32_t* d_docs, uint32_t n, uint32_t offset, uint32_t type, unsigned long long* d_out,
                     const ss_facet_point* point, hipStream_t st) {
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 2, 4, 8};
  FacetPoint pt{0, 0, 0};
  uint32_t load_as = type;
  if (type == SS_FACET_POINT) {
    if (!point) load_as = SS_FACET_U64;
    else if (facet_point_of(point, &pt) != SS_OK) return SS_EINVAL;
  }
  if (n) facet_values_kernel<<<(n + 255) / 256, 256, 0, st>>>(s->d_facets, s->facet_record_size, offset, width[type], d_docs, n,
                                                               (unsigned long long)s->facet_docs, d_out, load_as, pt);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Result sort for a BATCH, pivots on the device (ss_bm25_search_sorted).  The composition above -- pivot of the first field, a search
// filtered to "better", a search filtered to "equal" that recurses on the next field -- cost a host round trip per radix byte and per
// search.  Here the whole chain stays on the device, per query:
//   E = the match set (bits), B = {} ; per sort field: radix select over E for the k_left-th best key (histogram pass + a one-block
//   decide kernel per byte, the prefix kept in device memory), then a classify pass: docs of E strictly better than the pivot move to
//   B, worse ones leave E; k_left -= |moved|.  After the last field B holds the (< k) docs that are in the answer for sure, E the tie
//   group of the last pivot, of which the best by SCORE fill the rest.  Two ordinary searches under exclusion bitmaps ~B and ~E (the
//   bitmap the kernels honour for tombstones / facet filters) give both lists with their scores; a compose kernel orders B's docs by
//   (field 1, ..., field m, score, doc) -- a bitonic sort of <= 1024 tuples in LDS -- and appends the first k - |B| of E's.
struct SortState { unsigned long long prefix, want, n_better, n_equal, count_e, k_left, better_total, pad; };
struct SortFieldsDev {
  uint32_t n;
  uint32_t offset[SS_MAX_SORT_FIELDS], type[SS_MAX_SORT_FIELDS], bytes[SS_MAX_SORT_FIELDS], desc[SS_MAX_SORT_FIELDS];
  FacetPoint pt[SS_MAX_SORT_FIELDS];
};

// (every kernel below: blockIdx.y / blockIdx.x of the one-block kernels = the query of the batch; bit sets [nq][groups], states [nq],
// histograms [nq][256])
__global__ void sort_begin_kernel(SortState* st, const unsigned long long* total, unsigned long long k) {
  st += blockIdx.x;
  if (threadIdx.x == 0) { st->count_e = total[blockIdx.x]; st->k_left = k; st->better_total = 0ull; }
}
__global__ void sort_level_begin_kernel(SortState* st, unsigned long long* hist) {
  st += blockIdx.x;
  hist[(size_t)blockIdx.x * 256u + (threadIdx.x & 255u)] = 0ull;
  if (threadIdx.x == 0) { st->want = st->k_left < st->count_e ? st->k_left : st->count_e; st->prefix = 0ull; st->n_better = 0ull; st->n_equal = 0ull; }
}
// One byte of the select: the histogram of byte `byte_index` over the docs of E that share the prefix decided so far -- and, on the way,
// the classification by that prefix: a doc above it moves to B, a doc below it leaves E.  E thus shrinks 256-fold per byte, and only
// the first two passes of a field touch every match (each doc's record is a scattered read: they are what a sort by a wide field costs).
__global__ void sort_radix_kernel(unsigned long long* __restrict__ bits, unsigned long long* __restrict__ B, unsigned long long n_docs,
                                  unsigned long long groups, const uint8_t* __restrict__ records, uint32_t record_size, uint32_t offset,
                                  uint32_t type, uint32_t key_bits, uint32_t descending, const SortState* __restrict__ st, uint32_t byte_index,
                                  unsigned long long* __restrict__ hist, FacetPoint pt) {
  __shared__ unsigned int h[256];
  st += blockIdx.y;
  if (st->want == 0ull) return;
  bits += (size_t)blockIdx.y * groups;
  B += (size_t)blockIdx.y * groups;
  hist += (size_t)blockIdx.y * 256u;
  const unsigned long long prefix = st->prefix;
  h[threadIdx.x & 255u] = 0u;
  __syncthreads();
  const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g * 64ull < n_docs) {
    const unsigned long long m0 = bits[g];
    unsigned long long m = m0, keep = m0, better = 0ull;
    const uint32_t shift = key_bits - 8u * (byte_index + 1u);
    while (m) {
      const int b = __builtin_ctzll(m);
      const unsigned long long d = g * 64ull + (unsigned long long)b;
      m &= m - 1;
      if (d >= n_docs) break;
      uint32_t ty = type;
      const unsigned long long v = facet_load(records + d * record_size + offset, key_bits / 8u, ty, pt);
      const unsigned long long k = facet_order_key(v, ty, key_bits, descending != 0u);
      if (byte_index == 0u) {
        atomicAdd(&h[(k >> shift) & 255u], 1u);
      } else {
        const unsigned long long kp = k >> (shift + 8u);
        if (kp == prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
        else { keep &= ~(1ull << b); if (kp > prefix) better |= 1ull << b; }
      }
    }
    if (keep != m0) bits[g] = keep;
    if (better) B[g] |= better;
  }
  __syncthreads();
  if (h[threadIdx.x & 255u]) atomicAdd(&hist[threadIdx.x & 255u], (unsigned long long)h[threadIdx.x & 255u]);
}
// one byte decided: from the best byte value down to the bucket that holds the wanted rank (ssi_facet_kth's host loop)
__global__ void sort_decide_kernel(SortState* st, unsigned long long* hist) {
  st += blockIdx.x;
  hist += (size_t)blockIdx.x * 256u;
  if (threadIdx.x == 0) {
    unsigned long long want = st->want, nb = st->n_better;
    int v = 255;
    if (want != 0ull)
      for (; v > 0; v--) {
        if (hist[v] >= want) break;
        want -= hist[v];
        nb += hist[v];
      }
    st->prefix = (st->prefix << 8) | (unsigned long long)v;
    st->n_equal = st->want != 0ull ? hist[v] : 0ull;
    st->want = want;
    st->n_better = nb;
  }
  __syncthreads();
  hist[threadIdx.x & 255u] = 0ull;
}
// E: docs equal to the pivot stay; strictly better ones move to B; worse ones leave
__global__ void sort_classify_kernel(unsigned long long* __restrict__ E, unsigned long long* __restrict__ B, unsigned long long n_docs,
                                     unsigned long long groups, const uint8_t* __restrict__ records, uint32_t record_size, uint32_t offset,
                                     uint32_t type, uint32_t key_bits, uint32_t descending, const SortState* __restrict__ st, FacetPoint pt) {
  const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g * 64ull >= n_docs) return;
  st += blockIdx.y;
  E += (size_t)blockIdx.y * groups;
  B += (size_t)blockIdx.y * groups;
  const unsigned long long pivot = st->prefix;
  unsigned long long m = E[g], keep = 0ull, better = 0ull;
  while (m) {
    const int b = __builtin_ctzll(m);
    const unsigned long long d = g * 64ull + (unsigned long long)b;
    m &= m - 1;
    if (d >= n_docs) break;
    uint32_t ty = type;
    const unsigned long long v = facet_load(records + d * record_size + offset, key_bits / 8u, ty, pt);
    const unsigned long long k = facet_order_key(v, ty, key_bits, descending != 0u);
    if (k > pivot) better |= 1ull << b;
    else if (k == pivot) keep |= 1ull << b;
  }
  E[g] = keep;
  if (better) B[g] |= better;
}
__global__ void sort_level_end_kernel(SortState* st) {
  st += blockIdx.x;
  if (threadIdx.x == 0) { st->k_left -= st->n_better; st->count_e = st->n_equal; st->better_total += st->n_better; }
}
__global__ void sort_excl_kernel(const unsigned long long* __restrict__ E, const unsigned long long* __restrict__ B,
                                 unsigned long long* __restrict__ ex_b, unsigned long long* __restrict__ ex_e, unsigned long long groups) {
  const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t o = (size_t)blockIdx.y * groups + g;
  if (g < groups) { ex_b[o] = ~B[o]; ex_e[o] = ~E[o]; }
}
// the answer of one query (blockIdx.x): list A (the docs of B with their scores, any order) ordered by (sort keys, score desc, doc asc),
// then the first k - |A| of list C (the tie group of the last pivot, by score)
__global__ void __launch_bounds__(1024) sort_compose_kernel(const uint32_t* __restrict__ a_doc, const float* __restrict__ a_score,
                                                            const uint32_t* __restrict__ a_cnt, const uint32_t* __restrict__ c_doc,
                                                            const float* __restrict__ c_score, const uint32_t* __restrict__ c_cnt,
                                                            const unsigned long long* __restrict__ total, const uint8_t* __restrict__ records,
                                                            uint32_t record_size, unsigned long long n_facet_docs, SortFieldsDev F, uint32_t k,
                                                            uint32_t* __restrict__ out_doc, float* __restrict__ out_score,
                                                            uint32_t* __restrict__ out_count, unsigned long long* __restrict__ out_total) {
  __shared__ unsigned long long key[SS_MAX_SORT_FIELDS][1024];
  __shared__ float sc[1024];
  __shared__ uint32_t dc[1024];
  __shared__ uint16_t perm[1024];
  const uint32_t i = threadIdx.x, q = blockIdx.x;
  a_doc += (size_t)q * k; a_score += (size_t)q * k; c_doc += (size_t)q * k; c_score += (size_t)q * k;
  out_doc += (size_t)q * k; out_score += (size_t)q * k;
  const uint32_t na = min(a_cnt[q] == 0xFFFFFFFFu ? 0u : a_cnt[q], k);
  const uint32_t nc = min(c_cnt[q] == 0xFFFFFFFFu ? 0u : c_cnt[q], k - na);
  perm[i] = (uint16_t)i;
  dc[i] = i < na ? a_doc[i] : 0xFFFFFFFFu;
  sc[i] = i < na ? a_score[i] : -INFINITY;
  for (uint32_t f = 0; f < (uint32_t)SS_MAX_SORT_FIELDS; f++) {
    unsigned long long kk = 0ull;
    if (i < na && f < F.n && dc[i] < n_facet_docs) {
      uint32_t ty = F.type[f];
      const unsigned long long v = facet_load(records + (size_t)dc[i] * record_size + F.offset[f], F.bytes[f], ty, F.pt[f]);
      kk = facet_order_key(v, ty, 8u * F.bytes[f], F.desc[f] != 0u);
    }
    key[f][i] = kk;
  }
  __syncthreads();
  auto before = [&](uint32_t a, uint32_t b) -> bool {  // a stands before b in the answer
    const bool la = a < na, lb = b < na;
    if (la != lb) return la;
    if (!la) return a < b;
    for (uint32_t f = 0; f < F.n; f++)
      if (key[f][a] != key[f][b]) return key[f][a] > key[f][b];
    if (sc[a] != sc[b]) return sc[a] > sc[b];
    return dc[a] < dc[b];
  };
  for (uint32_t size = 2; size <= 1024u; size <<= 1)
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      const uint32_t j = i ^ stride;
      if (j > i) {
        const uint32_t a = perm[i], b = perm[j];
        const bool up = (i & size) == 0u;  // ascending block: the better one first
        if (up ? before(b, a) : before(a, b)) { perm[i] = (uint16_t)b; perm[j] = (uint16_t)a; }
      }
      __syncthreads();
    }
  if (i < na) { out_doc[i] = dc[perm[i]]; out_score[i] = sc[perm[i]]; }
  else if (i < na + nc) { out_doc[i] = c_doc[i - na]; out_score[i] = c_score[i - na]; }
  else if (i < k) { out_doc[i] = 0xFFFFFFFFu; out_score[i] = 0.f; }
  if (i == 0) { out_count[q] = na + nc; out_total[q] = total[q]; }
}

// the device chain of nq queries up to the exclusion bitmaps (the caller then runs the two searches and ssi_sort_compose);
// bit sets [nq][groups], d_total / d_state [nq], d_hist [nq][256]
int ssi_sort_select(ss_shard* s, uint32_t nq, unsigned long long* d_E, unsigned long long* d_B, unsigned long long* d_ex_b, unsigned long long* d_ex_e,
                    const unsigned long long* d_total, unsigned long long* d_hist, void* d_state, uint32_t n_sorts, const ss_result_sort* sorts,
                    uint32_t k, hipStream_t st) {
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 0, 0, 8};
  const unsigned long long n_docs = s->bm_n_docs, groups = (unsigned long long)s->bm_n_sub * (BM_SUB / 64);
  const dim3 grid((unsigned)((groups + 255) / 256), nq);
  SortState* state = (SortState*)d_state;
  SS_HIP(hipMemsetAsync(d_B, 0, (size_t)nq * groups * 8, st));
  sort_begin_kernel<<<nq, 64, 0, st>>>(state, d_total, (unsigned long long)k);
  for (uint32_t f = 0; f < n_sorts; f++) {
    const uint32_t type = sorts[f].facet_type;
    FacetPoint pt{0, 0, 0};
    if (type == SS_FACET_POINT) {
      const ss_facet_point base{sorts[f].base_lat, sorts[f].base_lon, SS_POINT_SORTKEY, 0};
      if (facet_point_of(&base, &pt) != SS_OK) return SS_EINVAL;
    }
    const uint32_t key_bits = 8u * width[type];
    sort_level_begin_kernel<<<nq, 256, 0, st>>>(state, d_hist);
    for (uint32_t b = 0; b < key_bits / 8u; b++) {
      sort_radix_kernel<<<grid, 256, 0, st>>>(d_E, d_B, n_docs, groups, s->d_facets, s->facet_record_size, sorts[f].facet_offset, type, key_bits,
                                              sorts[f].descending ? 1u : 0u, state, b, d_hist, pt);
      sort_decide_kernel<<<nq, 256, 0, st>>>(state, d_hist);
    }
    sort_classify_kernel<<<grid, 256, 0, st>>>(d_E, d_B, n_docs, groups, s->d_facets, s->facet_record_size, sorts[f].facet_offset, type, key_bits,
                                               sorts[f].descending ? 1u : 0u, state, pt);
    sort_level_end_kernel<<<nq, 64, 0, st>>>(state);
  }
  sort_excl_kernel<<<grid, 256, 0, st>>>(d_E, d_B, d_ex_b, d_ex_e, groups);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
// lists [nq][k], counts / totals / outputs per query
int ssi_sort_compose(ss_shard* s, uint32_t nq, const uint32_t* a_doc, const float* a_score, const uint32_t* a_cnt, const uint32_t* c_doc,
                     const float* c_score, const uint32_t* c_cnt, const unsigned long long* d_total, uint32_t n_sorts, const ss_result_sort* sorts,
                     uint32_t k, uint32_t* out_doc, float* out_score, uint32_t* out_count, unsigned long long* out_total, hipStream_t st) {
  static const uint32_t width[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 0, 0, 8};
  SortFieldsDev F;
  memset(&F, 0, sizeof(F));
  F.n = n_sorts;
  for (uint32_t f = 0; f < n_sorts; f++) {
    F.offset[f] = sorts[f].facet_offset; F.type[f] = sorts[f].facet_type; F.bytes[f] = width[sorts[f].facet_type]; F.desc[f] = sorts[f].descending ? 1u : 0u;
    if (sorts[f].facet_type == SS_FACET_POINT) {
      const ss_facet_point base{sorts[f].base_lat, sorts[f].base_lon, SS_POINT_SORTKEY, 0};
      if (facet_point_of(&base, &F.pt[f]) != SS_OK) return SS_EINVAL;
    }
  }
  sort_compose_kernel<<<nq, 1024, 0, st>>>(a_doc, a_score, a_cnt, c_doc, c_score, c_cnt, d_total, s->d_facets, s->facet_record_size,
                                           (unsigned long long)s->facet_docs, F, k, out_doc, out_score, out_count, out_total);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// the signatures ss_common.h declares: no base point
int ssi_facet_kth(ss_shard* s, const unsigned long long* d_bits, uint64_t n_docs, uint64_t n_matches, uint32_t offset, uint32_t type,
                  bool descending, uint64_t k, unsigned long long* d_hist, uint64_t* value_bits, uint64_t* n_better, uint64_t* n_equal,
                  hipStream_t st) {
  return ssi_facet_kth(s, d_bits, n_docs, n_matches, offset, type, descending, k, d_hist, value_bits, n_better, n_equal, nullptr, st);
}
int ssi_facet_values(ss_shard* s, const uint32_t* d_docs, uint32_t n, uint32_t offset, uint32_t type, unsigned long long* d_out, hipStream_t st) {
  return ssi_facet_values(s, d_docs, n, offset, type, d_out, nullptr, st);
}
int ssi_facet_count(ss_shard* s, const unsigned long long* d_bits, uint64_t n_docs, uint32_t offset, uint32_t type, uint32_t n_buckets,
                    const uint64_t* d_bounds, unsigned long long* d_counts, hipStream_t st) {
  return ssi_facet_count(s, d_bits, n_docs, offset, type, n_buckets, d_bounds, d_counts, nullptr, st);
}
