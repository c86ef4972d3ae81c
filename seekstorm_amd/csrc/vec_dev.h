// Device-side pieces shared by the f32 (vec_scan.hip) and i8 (vec8_scan.hip) vector scans: per-batch state, the
// order-preserving score key, and the epilogue that appends the rare rows beating the running threshold.
#pragma once
#include "ss_common.h"

// cnt: one counter per query, each on its own 128-byte line -- the scan kernels' appends are device-scope atomics, and
// atomics to one line serialise: with the 64 counters packed in two lines a launch with a few hundred appends per query
// spent ~150 us on them (i8 scan, 10M rows), independent of the grid.
constexpr int VS_CNT_STRIDE = 32;
struct VState {
  float tau[64];
  uint32_t cnt[64 * VS_CNT_STRIDE];
  uint32_t kept[64];
  uint32_t ovf;
  uint32_t pad[63];
  unsigned long long total[64];
};

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}
// larger key = better: (score desc, row asc)
__device__ __forceinline__ unsigned long long mk_key(float s, uint32_t row) {
  return ((unsigned long long)f2ord(s) << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}


// Appends the scores of one lane's 16 rows that beat tau to query q's candidate buffer: ONE atomic per lane and tile
// (the first launches, where every row is a candidate, would otherwise serialise 16 x as many on 64 counters).
__device__ __forceinline__ void vs_append(const float (&f)[16], float tau, uint32_t q, unsigned long long row_base,
                                          unsigned long long n_rows, VState* __restrict__ st, unsigned long long* __restrict__ cand) {
  uint32_t n = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) n += (f[r] > tau && row_base + (r & 3) + 8 * (r >> 2) < n_rows) ? 1u : 0u;
  if (n == 0) return;
  uint32_t slot = atomicAdd(&st->cnt[q * VS_CNT_STRIDE], n);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const unsigned long long row = row_base + (r & 3) + 8 * (r >> 2);
    if (f[r] > tau && row < n_rows) {
      if (slot < VS_CAP) cand[(size_t)q * VS_CAP + slot] = mk_key(f[r], (uint32_t)row);
      slot++;
    }
  }
}

// ---------------------------------------------------------------- ANN modes (vec_ann.hip)
// What a scan launch needs to visit only the clusters some query of the batch selected (AnnMode::Nprobe /
// Similaritythreshold, vector.rs:1300-1392): the list of 128-row tiles that hold a selected cluster, the
// cluster of every row and one bit per (query, cluster).  A row is a candidate of query q only if q selected its cluster.
// The same admission test carries the field filter of search_vector_shard (vector.rs:1397-1400: a record of a field that
// is not listed is skipped); without an ANN mode the tile list is absent (= every tile) and only the field test runs.
struct VAnn {
  const uint32_t* tiles;        // [n_tiles] tile ids, each once, interleaved (vec_ann.hip); null = every tile in order
  const uint32_t* n_tiles;      // device scalar (null with tiles)
  const uint32_t* row_cluster;  // [n_rows] shard-wide cluster index
  const uint32_t* sel;          // [64][sel_words] bit c of row q: query q visits cluster c; null = every cluster
  uint32_t sel_words;
  const uint16_t* row_field;    // [n_rows] indexed field of each record; null = no field filter
  unsigned long long field_mask;
};
__device__ __forceinline__ void vs_append_ann(const float (&f)[16], float tau, uint32_t q, unsigned long long row_base,
                                              unsigned long long n_rows, VState* __restrict__ st,
                                              unsigned long long* __restrict__ cand, const VAnn& ann) {
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const unsigned long long row = row_base + (r & 3) + 8 * (r >> 2);
    if (f[r] > tau && row < n_rows) {
      uint32_t ok = 1u;
      if (ann.sel) {
        const uint32_t c = ann.row_cluster[row];
        ok = (ann.sel[(size_t)q * ann.sel_words + (c >> 5)] >> (c & 31u)) & 1u;
      }
      if (ann.row_field) {
        const uint32_t fld = ann.row_field[row];
        ok &= fld < 64u ? (uint32_t)(ann.field_mask >> fld) & 1u : 0u;
      }
      m |= ok << r;
    }
  }
  if (m == 0) return;
  uint32_t slot = atomicAdd(&st->cnt[q * VS_CNT_STRIDE], (uint32_t)__popc(m));
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if ((m >> r) & 1u) {
      const unsigned long long row = row_base + (r & 3) + 8 * (r >> 2);
      if (slot < VS_CAP) cand[(size_t)q * VS_CAP + slot] = mk_key(f[r], (uint32_t)row);
      slot++;
    }
  }
}

// ---------------------------------------------------------------- i8 image geometry (vec8_scan.hip)
constexpr int V8_WAVES = 4;     // x 32 rows = one 128-row tile per workgroup step (same tile unit as the f32 scan)
constexpr int V8_D = 3;         // lines (128 bytes of a row) in flight per lane (6 measured the same)
constexpr int V8_LINE = 128;
// byte offset of X8[row][k] in the fragment-ordered image; L = lines (128 bytes) per row
__host__ __device__ inline size_t v8_index(unsigned long long row, uint32_t k, uint32_t L) {
  const unsigned long long blk = row >> 5;  // 32-row block = (tile, wave)
  const uint32_t lane = (uint32_t)(row & 31u) + 32u * ((k >> 6) & 1u);
  return ((((size_t)blk * L + (k >> 7)) * 4u + ((k >> 4) & 3u)) * 64u + lane) * 16u + (k & 15u);
}
