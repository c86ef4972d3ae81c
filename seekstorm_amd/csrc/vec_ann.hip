// The ANN modes of search_vector_shard (AnnMode::Nprobe / Similaritythreshold / NprobeSimilaritythreshold,
// vector.rs:1300-1392) for gfx950.
//
// The reference stores the records of a level (65 536 docs) cluster after cluster; the first record of a cluster is its
// medoid.  Per query and level it scores every medoid, pushes (cluster, score) through TopK::new(min(n_probe, clusters),
// cluster threshold) and then visits only the records of the surviving clusters.  Here, per batch of <= 64 queries:
//
//   1. ann_medoid_*_kernel   score[q][c] of every medoid, in the REFERENCE's summation order (dot_f32_avx2: 8 fmadd lanes
//                            summed in order, vector_similarity.rs:1118-1142; dot_f32 for dim % 8 != 0, 1006-1008; the
//                            i8 dot is an exact integer) -- a cluster chosen on a last-bit difference would change the
//                            result set, so this is bit-exact rather than "close";
//   2. ann_select_kernel     one wave per (query, level): the set TopK::push leaves behind over the level's clusters in file
//                            order (by rank when no tie reaches the last slot, else by replaying the pushes); sets bit c
//                            of the query's selection row and of the batch's union row;
//   3. ann_tiles_*_kernel    the list of 128-row tiles that hold a row of a cluster in the union, evenly interleaved.
//
// The scan kernels (vec_scan.hip / vec8_scan.hip, ANN instantiation) then walk that tile list instead of the whole image
// and admit a row only for the queries whose bit is set for the row's cluster.  One pass serves the whole batch: a
// single query reads n_probe / clusters of the image; a batch of 64 reads the union of 64 selections, never more than
// AnnMode::All.
#include <float.h>

#include <algorithm>

#include "ss_common.h"
#include "vec_dev.h"

constexpr int AQ = 8;  // queries per thread in the medoid kernels

// float offset in Qf (vec_qprep_kernel order) of Q[q][k .. k+3], k % 4 == 0
__device__ __forceinline__ uint32_t qf_off(uint32_t q, uint32_t k) {
  const uint32_t g = (k & 31u) >> 2;
  return (k >> 5) * 2048u + (q >> 5) * 1024u + (g >> 1) * 256u + ((q & 31u) + 32u * (g & 1u)) * 4u;
}
// byte offset in Qf8 (vec8_qprep_kernel order) of Q8[q][k .. k+15], k % 16 == 0
__device__ __forceinline__ uint32_t qf8_off(uint32_t q, uint32_t k) {
  return ((((k >> 7) * 4u + ((k >> 4) & 3u)) * 2u + (q >> 5)) * 64u + (q & 31u) + 32u * ((k >> 6) & 1u)) * 16u;
}

// The medoid records are copied once (ss_vec_set_clusters) into a matrix of their own, transposed so that the lanes of a
// wave (= 64 consecutive clusters) read consecutive memory: Mt[k / 8][cluster][8 floats] (i8: Mt8[k / 16][cluster][16 B]).
// Gathering them from the image, one 3 KB row per lane, measured 74 us for one query at 39 K clusters and 570 us for 64.
__global__ void ann_gather_medoids_f32_kernel(const float* __restrict__ X, uint32_t dim_pad, const uint32_t* __restrict__ cluster_first,
                                              uint32_t nc, float* __restrict__ Mt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (step, cluster)
  if (i >= (size_t)(dim_pad / 8u) * nc) return;
  const uint32_t c = (uint32_t)(i % nc), step = (uint32_t)(i / nc);
  const float4* src = (const float4*)(X + (size_t)cluster_first[c] * dim_pad + step * 8u);
  float4* dst = (float4*)(Mt + i * 8u);
  dst[0] = src[0];
  dst[1] = src[1];
}
__global__ void ann_gather_medoids_i8_kernel(const int8_t* __restrict__ X8, uint32_t dim_pad8, const uint32_t* __restrict__ cluster_first,
                                             uint32_t nc, int8_t* __restrict__ Mt8) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)(dim_pad8 / 16u) * nc) return;
  const uint32_t c = (uint32_t)(i % nc), step = (uint32_t)(i / nc);
  *(int4*)(Mt8 + i * 16u) = *(const int4*)(X8 + v8_index(cluster_first[c], step * 16u, dim_pad8 / 128u));
}

// thread = cluster, blockIdx.y = group of AQ queries
// euclid: Qf holds [2 q, -1, -|q|^2] (vec_qprep_kernel): q = 0.5 * Qf exactly, and the medoid's similarity is
// -euclidean_f32_avx2 / -euclidean_f32 (vector_similarity.rs:938-966 / 912-918): sub, mul, add each rounded on its own
__global__ void __launch_bounds__(64) ann_medoid_f32_kernel(const float* __restrict__ Mt, uint32_t dim, uint32_t nc,
                                                           const float* __restrict__ Qf, uint32_t nq, float* __restrict__ score,
                                                           int euclid) {
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  const uint32_t q0 = blockIdx.y * AQ;
  const float* x = Mt + (size_t)(c < nc ? c : nc - 1) * 8u;  // element k at x[(k / 8) * nc * 8 + k % 8]
  const size_t xs = (size_t)nc * 8u;
  float s[AQ];
  if ((dim & 7u) == 0) {  // dot_f32_avx2: lane j accumulates q[8 i + j] * e[8 i + j] with fmadd, then lanes 0..7 are summed
    float l[AQ][8];
#pragma unroll
    for (int a = 0; a < AQ; a++)
#pragma unroll
      for (int j = 0; j < 8; j++) l[a][j] = 0.f;
#pragma unroll 4
    for (uint32_t k = 0; k < dim; k += 8) {
      const float4 xa = *(const float4*)(x + (k >> 3) * xs), xb = *(const float4*)(x + (k >> 3) * xs + 4);
#pragma unroll
      for (int a = 0; a < AQ; a++) {
        const float4 qa = *(const float4*)(Qf + qf_off(q0 + a, k)), qb = *(const float4*)(Qf + qf_off(q0 + a, k + 4));
        if (euclid) {
          const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
          const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const float d = ss_fsub(0.5f * qv[j], xv[j]);
            l[a][j] = ss_fadd(l[a][j], ss_fmul(d, d));
          }
        } else {
          l[a][0] = fmaf(qa.x, xa.x, l[a][0]); l[a][1] = fmaf(qa.y, xa.y, l[a][1]);
          l[a][2] = fmaf(qa.z, xa.z, l[a][2]); l[a][3] = fmaf(qa.w, xa.w, l[a][3]);
          l[a][4] = fmaf(qb.x, xb.x, l[a][4]); l[a][5] = fmaf(qb.y, xb.y, l[a][5]);
          l[a][6] = fmaf(qb.z, xb.z, l[a][6]); l[a][7] = fmaf(qb.w, xb.w, l[a][7]);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < AQ; a++) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) t = ss_fadd(t, l[a][j]);
      s[a] = euclid ? -t : t;
    }
  } else {  // dot_f32: sequential, product and sum rounded separately
#pragma unroll
    for (int a = 0; a < AQ; a++) s[a] = 0.f;
    for (uint32_t k = 0; k < dim; k++) {
      const float xv = x[(k >> 3) * xs + (k & 7u)];
#pragma unroll
      for (int a = 0; a < AQ; a++) {
        const float qv = Qf[qf_off(q0 + a, k & ~3u) + (k & 3u)];
        if (euclid) { const float d = ss_fsub(0.5f * qv, xv); s[a] = ss_fadd(s[a], ss_fmul(d, d)); }
        else s[a] = ss_fadd(s[a], ss_fmul(qv, xv));
      }
    }
    if (euclid)
#pragma unroll
      for (int a = 0; a < AQ; a++) s[a] = -s[a];
  }
  if (c < nc)
#pragma unroll
    for (int a = 0; a < AQ; a++)
      if (q0 + a < nq) score[(size_t)(q0 + a) * nc + c] = s[a];
}

__global__ void __launch_bounds__(64) ann_medoid_i8_kernel(const int8_t* __restrict__ Mt8, uint32_t dim_pad8,
                                                          const uint32_t* __restrict__ cluster_first, uint32_t nc,
                                                          const int8_t* __restrict__ Qf8, uint32_t nq,
                                                          const float* __restrict__ row_scale, const float* __restrict__ q_scale,
                                                          int scaled, float* __restrict__ score, int euc_mode,
                                                          const float* __restrict__ row_norm, const int32_t* __restrict__ row_sq,
                                                          const float* __restrict__ qaux) {
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  const uint32_t q0 = blockIdx.y * AQ;
  const uint32_t cc = c < nc ? c : nc - 1;
  const uint32_t row = cluster_first[cc];
  int acc[AQ];
#pragma unroll
  for (int a = 0; a < AQ; a++) acc[a] = 0;
#pragma unroll 4
  for (uint32_t k = 0; k < dim_pad8; k += 16) {
    const int4 xv = *(const int4*)(Mt8 + ((size_t)(k >> 4) * nc + cc) * 16u);
    const int xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int a = 0; a < AQ; a++) {
      const int4 qv = *(const int4*)(Qf8 + qf8_off(q0 + a, k));
      const int qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
      for (int w = 0; w < 4; w++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a] += (int)(int8_t)(xw[w] >> (8 * b)) * (int)(int8_t)(qw[w] >> (8 * b));
    }
  }
  if (c < nc) {
    const float es = (scaled && row_scale) ? row_scale[row] : 1.f;
#pragma unroll
    for (int a = 0; a < AQ; a++)
      if (q0 + a < nq) {
        float f = (float)acc[a];
        if (euc_mode == 1) f = -(float)(__float_as_int(qaux[64 + q0 + a]) + row_sq[row] - 2 * acc[a]);  // -euclidean_i8
        else if (euc_mode == 2) {  // -euclidean_i8_quantized
          const float d = ss_fmul(ss_fmul(f, q_scale ? q_scale[q0 + a] : 1.f), es);
          f = -fmaxf(ss_fsub(ss_fadd(qaux[q0 + a], row_norm ? row_norm[row] : 0.f), ss_fmul(2.0f, d)), 0.0f);
        } else if (scaled) f = f * (q_scale ? q_scale[q0 + a] : 1.f) * es;  // dot_i8_quantized: dot as f32 * scale1 * scale2
        score[(size_t)(q0 + a) * nc + c] = f;
      }
  }
}

// TopK::new(k = min(n_probe, clusters), threshold) and TopK::push (vector.rs:366-496) over a level's medoid scores in
// cluster order; only the surviving SET matters downstream.  One WAVE per (query, level), the level's scores in LDS.
//
// Fast path: push-and-replace-the-minimum keeps the k best scores seen, whatever the order, as long as no two candidates
// for the last slot are EQUAL -- a score equal to the current minimum is rejected (`score > minimum` admits), so only a
// tie at the final boundary makes the result depend on the order of the pushes.  Each lane counts, for its clusters, the
// level's scores strictly above its own (LDS broadcast reads): fewer than k above = among the k best.  Equal scores have
// equal counts, so a tie group that straddles the boundary is selected as a whole and shows as MORE than k selected: only
// then the wave falls back to the replay.
// Replay: the reference's array, push by push (which of the tied entries survive depends on the order of the pushes and
// on which minimum is evicted: the first one in array order).  The reference's pre-filter `score <=
// lowest_similarity_score` (the value evicted last) never rejects what `score > current minimum` would admit, so the
// replay tests against the current minimum directly.  Sequential over the pushes, parallel inside one: the first-minimum
// search of an admission is a strided pass over the LDS array + a wave reduction.
// (One THREAD per (query, level) replaying in global memory measured 1.15 ms at n_probe = 64 of 256 clusters x 153
// levels, the wave replay alone 0.1-0.3 ms: more than the scans they prepare.)
constexpr uint32_t ASEL_KMAX = 1024;  // n_probe the LDS replay holds; beyond it the slow kernel below
constexpr uint32_t ASEL_CMAX = 4096;  // clusters per level the LDS score + rank arrays hold
constexpr int ASEL_E = 4;             // clusters ranked per lane at a time
__device__ __forceinline__ void wave_min_first(float& v, uint32_t& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o);
    const uint32_t oi = __shfl_xor(i, o);
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}
__global__ void __launch_bounds__(64) ann_select_kernel(const float* __restrict__ score, const uint32_t* __restrict__ level_off,
                                                       uint32_t n_levels, uint32_t nc, uint32_t n_probe, float cthr,
                                                       uint32_t* __restrict__ sel, uint32_t W, uint32_t* __restrict__ ncl,
                                                       uint32_t cmax4) {
  extern __shared__ __attribute__((aligned(16))) float ss[];  // the level's scores (-inf = below the cluster threshold) | ranks
  __shared__ float vs[ASEL_KMAX];
  __shared__ uint32_t vc[ASEL_KMAX];
  const uint32_t q = blockIdx.x / n_levels, l = blockIdx.x % n_levels;
  const uint32_t lane = threadIdx.x;
  const uint32_t c0 = level_off[l], C = level_off[l + 1] - c0;
  const uint32_t k = (n_probe == 0 || n_probe > C) ? C : n_probe;
  const float* sc = score + (size_t)q * nc + c0;
  auto select = [&](uint32_t c) {
    const uint32_t cg = c0 + c;
    atomicOr(&sel[(size_t)q * W + (cg >> 5)], 1u << (cg & 31u));  // (the batch's union row is ORed together afterwards: ann_union_kernel)
  };
  if (k == C) {  // every cluster that passes the threshold
    uint32_t n = 0;
    for (uint32_t c = lane; c < C; c += 64)
      if (!(sc[c] < cthr)) { select(c); n++; }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0 && n) atomicAdd(&ncl[q], n);
    return;
  }
  const uint32_t C4 = (C + 3u) & ~3u;
  uint32_t* above = (uint32_t*)(ss + cmax4);  // per cluster: number of eligible scores strictly above its own
  for (uint32_t c = lane; c < C4; c += 64) {
    const float v = c < C ? sc[c] : -INFINITY;
    ss[c] = v < cthr ? -INFINITY : v;
  }
  __syncthreads();
  // ---- fast path.  ASEL_E clusters per lane at a time against four scores per LDS read: the loop is a chain of LDS
  // latencies otherwise (one cluster per lane, one score per read and a second counter for equal scores measured
  // 120-550 us per launch)
  uint32_t n_in = 0;
  for (uint32_t cb = 0; cb < C; cb += 64 * ASEL_E) {
    float s[ASEL_E];
    uint32_t gt[ASEL_E];
#pragma unroll
    for (int e = 0; e < ASEL_E; e++) {
      const uint32_t c = cb + e * 64 + lane;
      s[e] = c < C ? ss[c] : -INFINITY;
      gt[e] = 0;
    }
    for (uint32_t c2 = 0; c2 < C4; c2 += 4) {
      const float4 v = *(const float4*)(ss + c2);
#pragma unroll
      for (int e = 0; e < ASEL_E; e++)
        gt[e] += (v.x > s[e] ? 1u : 0u) + (v.y > s[e] ? 1u : 0u) + (v.z > s[e] ? 1u : 0u) + (v.w > s[e] ? 1u : 0u);
    }
#pragma unroll
    for (int e = 0; e < ASEL_E; e++) {
      const uint32_t c = cb + e * 64 + lane;
      const bool in = s[e] != -INFINITY && gt[e] < k;
      n_in += in ? 1u : 0u;
      if (c < C) above[c] = in ? gt[e] : 0xFFFFFFFFu;
    }
  }
  for (int o = 32; o > 0; o >>= 1) n_in += __shfl_xor(n_in, o);
  if (n_in <= k) {  // no tie across the boundary: the clusters with fewer than k scores above them ARE the k best
    for (uint32_t c = lane; c < C; c += 64)  // own writes only: no barrier needed
      if (above[c] < k) select(c);
    if (lane == 0 && n_in) atomicAdd(&ncl[q], n_in);
    return;
  }
  // ---- replay
  uint32_t len = 0;
  float cur_min = 0.f;
  auto array_min = [&](float& mv, uint32_t& mi) {  // first minimum of vs[0 .. k)
    mv = INFINITY; mi = 0xFFFFFFFFu;
    for (uint32_t i = lane; i < k; i += 64) {
      const float v = vs[i];
      if (v < mv) { mv = v; mi = i; }
    }
    wave_min_first(mv, mi);
  };
  for (uint32_t c = 0; c < C; c++) {
    const float s = ss[c];
    if (s == -INFINITY) continue;
    if (len < k) {
      if (lane == 0) { vs[len] = s; vc[len] = c; }
      len++;
      if (len == k) {
        __syncthreads();
        uint32_t mi;
        array_min(cur_min, mi);
      }
      continue;
    }
    if (!(s > cur_min)) continue;
    float mv; uint32_t mi;
    array_min(mv, mi);
    __syncthreads();
    if (lane == 0) { vs[mi] = s; vc[mi] = c; }
    __syncthreads();
    array_min(cur_min, mi);
  }
  __syncthreads();
  for (uint32_t i = lane; i < len; i += 64) select(vc[i]);
  if (lane == 0 && len) atomicAdd(&ncl[q], len);
}

// n_probe > ASEL_KMAX: one thread per (query, level), the array in global memory
__global__ void ann_select_slow_kernel(const float* __restrict__ score, const uint32_t* __restrict__ level_off, uint32_t n_levels,
                                       uint32_t nc, uint32_t nq, uint32_t n_probe, float cthr, float* __restrict__ its,
                                       uint32_t* __restrict__ itc, uint32_t* __restrict__ sel, uint32_t W, uint32_t* __restrict__ ncl) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nq * n_levels) return;
  const uint32_t q = idx / n_levels, l = idx % n_levels;
  const uint32_t c0 = level_off[l], C = level_off[l + 1] - c0;
  const uint32_t k = (n_probe == 0 || n_probe > C) ? C : n_probe;
  const float* sc = score + (size_t)q * nc + c0;
  float* is = its + (size_t)q * nc + c0;
  uint32_t* ic = itc + (size_t)q * nc + c0;
  uint32_t len = 0;
  for (uint32_t c = 0; c < C; c++) {
    const float s = sc[c];
    if (s < cthr) continue;
    if (len < k) { is[len] = s; ic[len] = c; len++; continue; }
    uint32_t min_i = 0;
    float min_v = is[0];
    for (uint32_t i = 1; i < len; i++) {
      const float v = is[i];
      if (v < min_v) { min_v = v; min_i = i; }
    }
    if (s > min_v) { is[min_i] = s; ic[min_i] = c; }
  }
  for (uint32_t i = 0; i < len; i++) {
    const uint32_t cg = c0 + ic[i];
    atomicOr(&sel[(size_t)q * W + (cg >> 5)], 1u << (cg & 31u));
  }
  if (len) atomicAdd(&ncl[q], len);
}
// the batch's union row (row 64) = OR of the queries' selection rows
__global__ void ann_union_kernel(uint32_t* __restrict__ sel, uint32_t W, uint32_t nb) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  uint32_t u = 0;
  for (uint32_t q = 0; q < nb; q++) u |= sel[(size_t)q * W + w];
  sel[(size_t)64 * W + w] = u;
}

// Tile list of the batch: tile t is listed if a cluster in [cluster of its first row, cluster of its last row] is in the
// union.  Two launches: flag + rank inside each group of 1024 tiles, then place.  The list is NOT left ascending: a
// query's clusters are contiguous runs of rows, so in file order its candidates would arrive in bursts -- a query with
// nothing in the first launches keeps tau = -inf and then floods its candidate buffer (measured: a batch of 64 with a
// quarter of the clusters each overflowed every time).  Position j holds the ascending list's entry (j * s) mod n, s ~
// 0.618 n coprime to n: every prefix of the list is spread evenly over the image (three-distance theorem), so each
// query meets its rows at its average rate and the chunk schedule's bound (vec_scan.hip) holds as in AnnMode::All.
// The permutation is a pure function of n: results and totals do not depend on timing.
constexpr int AT = 1024;
__global__ void __launch_bounds__(AT) ann_tiles_flag_kernel(const uint32_t* __restrict__ row_cluster, unsigned long long n_rows,
                                                           uint32_t T, const uint32_t* __restrict__ sel_union,
                                                           uint32_t* __restrict__ rank, uint32_t* __restrict__ group_count) {
  __shared__ uint32_t wsum[AT / 64];
  const uint32_t t = blockIdx.x * AT + threadIdx.x;
  bool act = false;
  if (t < T) {
    const unsigned long long r_lo = (unsigned long long)t * VS_TR;
    const unsigned long long r_hi = (r_lo + VS_TR < n_rows ? r_lo + VS_TR : n_rows) - 1;
    const uint32_t c_lo = row_cluster[r_lo], c_hi = row_cluster[r_hi];
    for (uint32_t w = c_lo >> 5; w <= (c_hi >> 5) && !act; w++) {
      uint32_t m = 0xFFFFFFFFu;
      if (w == (c_lo >> 5)) m &= 0xFFFFFFFFu << (c_lo & 31u);
      if (w == (c_hi >> 5)) m &= 0xFFFFFFFFu >> (31u - (c_hi & 31u));
      act = (sel_union[w] & m) != 0;
    }
  }
  const unsigned long long b = __ballot(act);
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = (uint32_t)__popcll(b);
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (uint32_t i = 0; i < AT / 64; i++) {
    if (i < wv) base += wsum[i];
    total += wsum[i];
  }
  if (t < T) rank[t] = act ? base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
  if (threadIdx.x == 0) group_count[blockIdx.x] = total;
}
__global__ void __launch_bounds__(AT) ann_tiles_place_kernel(const uint32_t* __restrict__ rank, const uint32_t* __restrict__ group_count,
                                                            uint32_t T, uint32_t* __restrict__ tiles) {
  __shared__ uint32_t sbase, sn, sinv;
  if (threadIdx.x < 64) {
    uint32_t v = 0, all = 0;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += 64) {
      const uint32_t g = group_count[i];
      all += g;
      if (i < blockIdx.x) v += g;
    }
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o); all += __shfl_down(all, o); }
    if (threadIdx.x == 0) {
      sbase = v;
      sn = all;
      uint32_t inv = 1;
      if (all > 2) {
        uint32_t s = (uint32_t)((double)all * 0.6180339887498949);
        auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
        while (s < 2 || gcd(s, all) != 1) s++;  // all - 1 is coprime to all: terminates below all
        // inverse of s modulo all (extended Euclid): rank R sits at position R * inv mod all, i.e. position j holds rank j * s mod all
        long long r0 = all, r1 = s, t0 = 0, t1 = 1;
        while (r1) { const long long qq = r0 / r1, r2 = r0 - qq * r1, t2 = t0 - qq * t1; r0 = r1; r1 = r2; t0 = t1; t1 = t2; }
        inv = (uint32_t)(t0 < 0 ? t0 + all : t0);
      }
      sinv = inv;
    }
  }
  __syncthreads();
  const uint32_t t = blockIdx.x * AT + threadIdx.x;
  if (t < T) {
    const uint32_t r = rank[t];
    if (r != 0xFFFFFFFFu) tiles[(uint32_t)(((unsigned long long)(sbase + r) * sinv) % sn)] = t;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) tiles[T] = sn;
}

// ---------------------------------------------------------------- host side
void ssi_vec_free_clusters(ss_shard* s) {
  void* ptrs[] = {s->d_row_cluster, s->d_cluster_first, s->d_level_off, s->d_ann_score, s->d_ann_its,
                  s->d_ann_itc,     s->d_ann_sel,       s->d_ann_tiles, s->d_ann_ncl,   s->d_medoids,   s->d_ann_live};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  s->d_row_cluster = nullptr; s->d_cluster_first = nullptr; s->d_level_off = nullptr; s->d_ann_score = nullptr;
  s->d_ann_its = nullptr; s->d_ann_itc = nullptr; s->d_ann_sel = nullptr; s->d_ann_tiles = nullptr; s->d_ann_ncl = nullptr; s->d_medoids = nullptr;
  s->d_ann_live = nullptr; s->ann_live_cap = 0;
  s->vec_n_clusters = 0; s->vec_n_levels = 0; s->vec_max_level_clusters = 0;
  s->h_level_clusters.clear(); s->h_child_count.clear();
}

// caller holds the shard lock and has selected the device
int ssi_vec_set_clusters(ss_shard* s, uint32_t n_levels, const uint32_t* level_clusters, const uint32_t* child_count) {
  if (!s->d_X && !s->d_X8) return SS_ESTATE;
  if (n_levels == 0 || !level_clusters || !child_count) return SS_EINVAL;
  uint64_t nc = 0;
  for (uint32_t l = 0; l < n_levels; l++) nc += level_clusters[l];
  if (nc == 0 || nc > 0x7FFFFFFFull) return SS_EINVAL;
  // host copies of the declared structure (ss_vec_append_rows extends it by a level and declares it again)
  std::vector<uint32_t> lc(level_clusters, level_clusters + n_levels), cc(child_count, child_count + nc);
  level_clusters = lc.data();
  child_count = cc.data();
  std::vector<uint32_t> level_off(n_levels + 1), first(nc), row_cluster(s->n_rows);
  uint64_t row = 0, c = 0;
  for (uint32_t l = 0; l < n_levels; l++) {
    level_off[l] = (uint32_t)c;
    for (uint32_t i = 0; i < level_clusters[l]; i++, c++) {
      // an empty cluster has no medoid: the reference would score whatever bytes follow (vector.rs:1311-1316)
      if (child_count[c] == 0) return SS_ENOTSUP;
      if (row + child_count[c] > s->n_rows) return SS_EINVAL;
      first[c] = (uint32_t)row;
      std::fill(row_cluster.begin() + row, row_cluster.begin() + row + child_count[c], (uint32_t)c);
      row += child_count[c];
    }
  }
  level_off[n_levels] = (uint32_t)c;
  if (row != s->n_rows) return SS_EINVAL;
  SS_HIP(hipStreamSynchronize(s->stream));
  ssi_vec_free_clusters(s);
  const uint32_t W = (uint32_t)((nc + 31) / 32);
  const uint32_t T = (uint32_t)(s->n_rows_pad / VS_TR);
  const uint32_t groups = (T + AT - 1) / AT;
  SS_HIP(hipMalloc(&s->d_row_cluster, s->n_rows * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_cluster_first, nc * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_level_off, (n_levels + 1) * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_ann_score, (size_t)SS_VEC_BATCH * nc * sizeof(float)));
  SS_HIP(hipMalloc(&s->d_ann_its, (size_t)SS_VEC_BATCH * nc * sizeof(float)));
  SS_HIP(hipMalloc(&s->d_ann_itc, (size_t)SS_VEC_BATCH * nc * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_ann_sel, (size_t)(SS_VEC_BATCH + 1) * W * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_ann_tiles, ((size_t)T + 1 + T + groups) * sizeof(uint32_t)));  // list + count | rank | group counts
  SS_HIP(hipMalloc(&s->d_ann_ncl, SS_VEC_BATCH * sizeof(uint32_t)));
  SS_HIP(hipMemcpy(s->d_row_cluster, row_cluster.data(), s->n_rows * sizeof(uint32_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_cluster_first, first.data(), nc * sizeof(uint32_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_level_off, level_off.data(), (n_levels + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
  {  // the medoid records, transposed (ann_gather_medoids_*_kernel)
    const size_t steps = s->d_X8 ? s->dim_pad8 / 16u : s->dim_pad / 8u;
    const size_t bytes = steps * nc * (s->d_X8 ? 16u : 32u);
    SS_HIP(hipMalloc(&s->d_medoids, bytes));
    const unsigned grid = (unsigned)((steps * nc + 255) / 256);
    if (s->d_X8) ann_gather_medoids_i8_kernel<<<grid, 256, 0, s->stream>>>(s->d_X8, s->dim_pad8, s->d_cluster_first, (uint32_t)nc, (int8_t*)s->d_medoids);
    else ann_gather_medoids_f32_kernel<<<grid, 256, 0, s->stream>>>(s->d_X, s->dim_pad, s->d_cluster_first, (uint32_t)nc, (float*)s->d_medoids);
    SS_HIP(hipGetLastError());
    SS_HIP(hipStreamSynchronize(s->stream));
  }
  s->h_level_clusters.swap(lc);
  s->h_child_count.swap(cc);
  s->vec_n_clusters = (uint32_t)nc;
  s->vec_n_levels = n_levels;
  s->vec_max_level_clusters = *std::max_element(level_clusters, level_clusters + n_levels);
  return SS_OK;
}

// ---------------------------------------------------------------- observed_vector_count (SS_ANN_REPORT_OBSERVED)
// TopK::push counts every record it is handed (vector.rs:421): the records of the visited clusters whose field the filter lists
// (1397-1400) and whose doc is not tombstoned (1450-1452).  Two small passes: the live records of every cluster (rows of a
// cluster are contiguous and row_cluster ascends, so a wave counts its segments with ballots), then per query the sum over its
// selected clusters.  Without clusters (or a mode that skips none): one count over the whole image.
__global__ void __launch_bounds__(256) ann_live_kernel(const uint32_t* __restrict__ row_cluster, const uint16_t* __restrict__ row_field,
                                                       unsigned long long field_mask, const uint32_t* __restrict__ row_doc,
                                                       const uint32_t* __restrict__ del, uint32_t del_words, unsigned long long n_rows,
                                                       uint32_t* __restrict__ cluster_live, unsigned long long* __restrict__ all_live) {
  const unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  bool pass = r < n_rows;
  if (pass && field_mask) {
    const uint32_t f = row_field[r];
    pass = f < 64u && ((field_mask >> f) & 1ull);
  }
  if (pass && del) {
    const uint32_t doc = row_doc ? row_doc[r] : (uint32_t)r;
    pass = !((doc >> 5) < del_words && ((del[doc >> 5] >> (doc & 31u)) & 1u));
  }
  const unsigned long long pb = __ballot(pass);
  if (lane == 0 && pb) atomicAdd(all_live, (unsigned long long)__popcll(pb));
  if (!row_cluster) return;
  const uint32_t c = r < n_rows ? row_cluster[r] : 0xFFFFFFFFu;
  const uint32_t prev = __shfl_up(c, 1);
  const bool head = lane == 0 || prev != c;
  const unsigned long long hb = __ballot(head);
  if (head && c != 0xFFFFFFFFu) {
    const unsigned long long above = lane == 63 ? 0ull : (hb >> (lane + 1)) << (lane + 1);  // heads after this lane
    const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : 64u;                     // first lane of the next segment
    const unsigned long long seg = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
    const uint32_t n = (uint32_t)__popcll(pb & seg);
    if (n) atomicAdd(&cluster_live[c], n);
  }
}
// one wave per query: out[3 q] = clusters visited, out[3 q + 1 .. 2] = observed records
__global__ void __launch_bounds__(64) ann_observed_kernel(const uint32_t* __restrict__ sel, uint32_t W, const uint32_t* __restrict__ cluster_live,
                                                          const uint32_t* __restrict__ ncl, const unsigned long long* __restrict__ all_live,
                                                          uint32_t* __restrict__ out) {
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  unsigned long long n = 0;
  if (sel) {
    for (uint32_t w = lane; w < W; w += 64u) {
      uint32_t m = sel[(size_t)q * W + w];
      while (m) {
        const uint32_t b = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        n += cluster_live[w * 32u + b];
      }
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  } else {
    n = *all_live;
  }
  if (lane == 0) {
    out[3u * q] = ncl ? ncl[q] : 0u;
    out[3u * q + 1u] = (uint32_t)n;
    out[3u * q + 2u] = (uint32_t)(n >> 32);
  }
}
// live counts of the image under (field_mask, tombstones): once per call, before the batches
int ssi_vec_observed_prepare(ss_shard* s, unsigned long long field_mask, hipStream_t st) {
  const size_t words = (size_t)s->vec_n_clusters + 2;
  if (s->ann_live_cap < words) {
    SS_HIP(hipStreamSynchronize(st));
    if (s->d_ann_live) (void)hipFree(s->d_ann_live);
    s->d_ann_live = nullptr; s->ann_live_cap = 0;
    SS_HIP(hipMalloc(&s->d_ann_live, words * sizeof(uint32_t) + 8));
    s->ann_live_cap = words;
  }
  SS_HIP(hipMemsetAsync(s->d_ann_live, 0, words * sizeof(uint32_t) + 8, st));
  if (s->n_rows)
    ann_live_kernel<<<(uint32_t)((s->n_rows + 255) / 256), 256, 0, st>>>(
        s->d_row_cluster, s->d_row_field, s->d_row_field ? field_mask : 0ull, s->d_row_doc, s->n_deleted ? s->d_deleted : nullptr,
        (uint32_t)s->deleted_words, (unsigned long long)s->n_rows, s->d_ann_live + 2, (unsigned long long*)s->d_ann_live);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
// the triples of one batch; clusters_selected: ssi_vec_ann_prepare ran for it (s->d_ann_sel / d_ann_ncl hold its selection)
int ssi_vec_observed_report(ss_shard* s, uint32_t nb, bool clusters_selected, uint32_t* d_out3, hipStream_t st) {
  const uint32_t W = (s->vec_n_clusters + 31) / 32;
  ann_observed_kernel<<<nb, 64, 0, st>>>(clusters_selected ? s->d_ann_sel : nullptr, W, s->d_ann_live + 2, clusters_selected ? s->d_ann_ncl : nullptr,
                                         (const unsigned long long*)s->d_ann_live, d_out3);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_vec_ann_prepare(ss_shard* s, uint32_t nb, const float* d_qscale, const ss_ann_mode* mode, VAnn* out,
                        uint32_t* d_out_clusters, hipStream_t st, const float* d_qnorm) {
  (void)d_qnorm;  // already in s->d_qaux (ssi_vec8_qaux ran before the scan preparation)
  const bool euclid = s->vec_similarity == SS_SIM_EUCLIDEAN;
  const uint32_t nc = s->vec_n_clusters, W = (nc + 31) / 32;
  const uint32_t T = (uint32_t)(s->n_rows_pad / VS_TR), groups = (T + AT - 1) / AT;
  uint32_t* tiles = s->d_ann_tiles;
  uint32_t* rank = tiles + T + 1;
  uint32_t* gcount = rank + T;
  SS_HIP(hipMemsetAsync(s->d_ann_sel, 0, (size_t)(SS_VEC_BATCH + 1) * W * sizeof(uint32_t), st));
  SS_HIP(hipMemsetAsync(s->d_ann_ncl, 0, SS_VEC_BATCH * sizeof(uint32_t), st));
  const dim3 grid((nc + 63) / 64, (nb + AQ - 1) / AQ);
  if (s->d_X8) {
    const bool scaled = s->d_row_scale != nullptr || d_qscale != nullptr;
    const int euc_mode = euclid ? (scaled ? 2 : 1) : 0;
    ann_medoid_i8_kernel<<<grid, 64, 0, st>>>((const int8_t*)s->d_medoids, s->dim_pad8, s->d_cluster_first, nc,
                                               (const int8_t*)s->d_Qf, nb, s->d_row_scale, d_qscale, scaled ? 1 : 0, s->d_ann_score,
                                               euc_mode, s->d_row_norm, s->d_row_sq, s->d_qaux);
  } else {
    ann_medoid_f32_kernel<<<grid, 64, 0, st>>>((const float*)s->d_medoids, s->dim, nc, s->d_Qf, nb, s->d_ann_score, euclid ? 1 : 0);
  }
  const uint32_t nsel = nb * s->vec_n_levels;
  if (mode->n_probe <= ASEL_KMAX && s->vec_max_level_clusters <= ASEL_CMAX)
  {
    const uint32_t cmax4 = (s->vec_max_level_clusters + 3u) & ~3u;
    ann_select_kernel<<<nsel, 64, (size_t)cmax4 * 2 * sizeof(float), st>>>(
        s->d_ann_score, s->d_level_off, s->vec_n_levels, nc, mode->n_probe, mode->cluster_threshold_raw, s->d_ann_sel, W,
        s->d_ann_ncl, cmax4);
  }
  else
    ann_select_slow_kernel<<<(nsel + 63) / 64, 64, 0, st>>>(s->d_ann_score, s->d_level_off, s->vec_n_levels, nc, nb,
                                                            mode->n_probe, mode->cluster_threshold_raw, s->d_ann_its,
                                                            s->d_ann_itc, s->d_ann_sel, W, s->d_ann_ncl);
  ann_union_kernel<<<(W + 255) / 256, 256, 0, st>>>(s->d_ann_sel, W, nb);
  ann_tiles_flag_kernel<<<groups, AT, 0, st>>>(s->d_row_cluster, (unsigned long long)s->n_rows, T,
                                               s->d_ann_sel + (size_t)SS_VEC_BATCH * W, rank, gcount);
  ann_tiles_place_kernel<<<groups, AT, 0, st>>>(rank, gcount, T, tiles);
  if (d_out_clusters) SS_HIP(hipMemcpyAsync(d_out_clusters, s->d_ann_ncl, nb * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  SS_HIP(hipGetLastError());
  out->tiles = tiles;
  out->n_tiles = tiles + T;
  out->row_cluster = s->d_row_cluster;
  out->sel = s->d_ann_sel;
  out->sel_words = W;
  return SS_OK;
}
