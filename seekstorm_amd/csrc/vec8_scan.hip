// Dense-vector brute-force scan over i8 embeddings (AnnMode::All; the reference's quantised similarity:
// dot_i8 / dot_i8_quantized, vector_similarity.rs:1011-1016, 1754-1758; quantize_f32_to_i8, 1226-1232) for gfx950.
//
//   scores[N x 64] = X8[N x dim] . Q8^T  with v_mfma_i32_32x32x32_i8: an exact integer dot product, so the score
//   (dot as f32 [* query_scale * embedding_scale]) is bit-identical to the reference's whatever the summation order.
//
// At a batch of 64 queries this scan is HBM-bound (128 int ops per byte of X against ~500 for the machine): the kernel
// is a stream.  Every wave owns 32 rows at a time and loads them STRAIGHT INTO MFMA A-fragments -- a dot product does
// not care about the order of k as long as A and B agree, so lane (row r, half h) takes the bytes [64 h, 64 h + 64) of
// each 128-byte line of its row (four 16-byte pieces), and step j of the line multiplies piece 4 h + j of both operands.
// The IMAGE is stored in that fragment order (v8_index): [128-row tile][wave][line][j][lane][16 B], so each of a wave's
// load instructions reads one contiguous KB (eight full 128-byte lines) and its whole 32-row block is one linear stream;
// with row-major rows every instruction would touch 64 different lines (measured: 2.6 TB/s instead of the figure in
// DESIGN.md).  The 64 queries (64 x dim bytes, 48 KB at dim 768) live in LDS for the whole launch in the matching
// order.  Three lines per lane are in flight (prefetch ring in registers), three workgroups per CU.
// Threshold filter, candidate buffer and the refine / final kernels are those of the f32 scan (vec_scan.hip).
#include "ss_common.h"
#include "vec_dev.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// row-major [n_rows][dim] (device staging) -> fragment order; one thread per 16-byte piece of the padded image
__global__ void vec8_permute_kernel(const int8_t* __restrict__ src, unsigned long long n_rows, uint32_t dim, uint32_t dim_pad,
                                    unsigned long long n_rows_pad, int8_t* __restrict__ dst) {
  const uint32_t L = dim_pad / 128u;
  const unsigned long long piece = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long total = n_rows_pad * (dim_pad / 16u);
  if (piece >= total) return;
  const unsigned long long row = piece / (dim_pad / 16u);
  const uint32_t k0 = (uint32_t)(piece % (dim_pad / 16u)) * 16u;
  int8_t v[16];
#pragma unroll
  for (int b = 0; b < 16; b++) v[b] = (row < n_rows && k0 + b < dim) ? src[row * dim + k0 + b] : (int8_t)0;
  int8_t* o = dst + v8_index(row, k0, L);
#pragma unroll
  for (int b = 0; b < 16; b++) o[b] = v[b];
}

// appended rows: row-major [n][dim] staging -> rows [r0, r0 + n) of the image (the padding around them is zero already)
__global__ void vec8_permute_range_kernel(const int8_t* __restrict__ src, unsigned long long r0, unsigned long long n, uint32_t dim,
                                          uint32_t dim_pad, int8_t* __restrict__ dst) {
  const uint32_t L = dim_pad / 128u;
  const unsigned long long piece = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (piece >= n * (dim_pad / 16u)) return;
  const unsigned long long row = piece / (dim_pad / 16u);
  const uint32_t k0 = (uint32_t)(piece % (dim_pad / 16u)) * 16u;
  int8_t v[16];
#pragma unroll
  for (int b = 0; b < 16; b++) v[b] = k0 + b < dim ? src[row * dim + k0 + b] : (int8_t)0;
  int8_t* o = dst + v8_index(r0 + row, k0, L);
#pragma unroll
  for (int b = 0; b < 16; b++) o[b] = v[b];
}

// rows [r0, r0 + n) back to row-major [n][dim]
__global__ void vec8_gather_rows_kernel(const int8_t* __restrict__ img, uint32_t dim, uint32_t dim_pad, unsigned long long r0,
                                        unsigned long long n, int8_t* __restrict__ out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim) return;
  const unsigned long long r = i / dim;
  const uint32_t k = (uint32_t)(i % dim);
  out[i] = img[v8_index(r0 + r, k, dim_pad / 128u)];
}

// Qf8[line][j(4)][nt(2)][lane(64)] 16 bytes = Q8[q = nt*32 + (lane & 31)][128 line + 64 (lane >> 5) + 16 j .. + 16)
__global__ void vec8_qprep_kernel(const int8_t* __restrict__ Q, uint32_t nq, uint32_t dim, int8_t* __restrict__ Qf8) {
  const uint32_t line = blockIdx.x;
  for (uint32_t e = threadIdx.x; e < 8192; e += blockDim.x) {
    const uint32_t b = e & 15, lane = (e >> 4) & 63, nt = (e >> 10) & 1, j = e >> 11;
    const uint32_t q = nt * 32 + (lane & 31);
    const uint32_t k = line * V8_LINE + 64 * (lane >> 5) + 16 * j + b;
    Qf8[(size_t)line * 8192 + e] = (q < nq && k < dim) ? Q[(size_t)q * dim + k] : (int8_t)0;
  }
}

// quantize_f32_to_i8 (vector_similarity.rs:1226-1232): (v * 127).round().clamp(-127, 127), round half away from zero
__global__ void vec8_quantize_kernel(const float* __restrict__ X, uint32_t dim, uint32_t dim_pad_f, unsigned long long n_rows,
                                     int8_t* __restrict__ X8, uint32_t dim_pad8) {
  const unsigned long long r = blockIdx.x;
  if (r >= n_rows) return;
  for (uint32_t c = threadIdx.x; c < dim; c += blockDim.x) {
    const float v = roundf(X[r * dim_pad_f + c] * 127.0f);
    X8[v8_index(r, c, dim_pad8 / 128u)] = (int8_t)fminf(fmaxf(v, -127.0f), 127.0f);
  }
}

// ---- VectorSimilarity::Euclidean on i8 records (the SCALED instantiation carries it; mode 0 = dot product)
//   mode 1  -euclidean_i8 (vector_similarity.rs:921-932): -(sum (a - b)^2) = -(|a|^2 + |b|^2 - 2 a.b), all exact integers
//   mode 2  -euclidean_i8_quantized (1721-1735): -max(0, norm1 + norm2 - 2 * (dot_i32 as f32 * scale1 * scale2)), every
//           operation rounded on its own in the reference's order
// qaux: [0..63] the queries' norms (mode 2), [64..127] their sums of squares as i32 bits (mode 1)
struct V8Euc {
  int mode;
  const float* row_norm;
  const int32_t* row_sq;
  const float* qaux;
};
__global__ void vec8_row_sq_kernel(const int8_t* __restrict__ X8, uint32_t dim, uint32_t dim_pad8, unsigned long long r0, unsigned long long n_rows,
                                   int32_t* __restrict__ row_sq) {
  const unsigned long long r = r0 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  int32_t ss = 0;
  for (uint32_t c = 0; c < dim; c++) { const int32_t v = X8[v8_index(r, c, dim_pad8 / 128u)]; ss += v * v; }
  row_sq[r] = ss;
}
__global__ void vec8_qaux_kernel(const int8_t* __restrict__ Q, uint32_t nq, uint32_t dim, const float* __restrict__ q_norm,
                                 float* __restrict__ qaux) {
  const uint32_t q = threadIdx.x;
  if (q >= 64) return;
  int32_t ss = 0;
  if (q < nq) for (uint32_t c = 0; c < dim; c++) { const int32_t v = Q[(size_t)q * dim + c]; ss += v * v; }
  qaux[q] = (q < nq && q_norm) ? q_norm[q] : 0.f;
  qaux[64 + q] = __int_as_float(ss);
}

// EVEN: the number of lines per row is a multiple of the ring depth -> the steady state has no conditional loads (the
// compiler's s_waitcnt insertion then counts the ring exactly instead of draining it with vmcnt(0) at every merge)
// ANN: the launch walks the batch's list of selected tiles and admits a row only for the queries that selected its cluster
template <bool SCALED, bool EVEN, bool ANN>
__global__ void __launch_bounds__(V8_WAVES * 64, 3)
vec8_scan_kernel(const int8_t* __restrict__ X, uint32_t dim_pad, unsigned long long n_rows, const int8_t* __restrict__ Qf8,
                 uint32_t L, uint32_t tile0, uint32_t ntiles, const float* __restrict__ row_scale,
                 const float* __restrict__ q_scale, VState* __restrict__ st, unsigned long long* __restrict__ cand, VAnn ann,
                 V8Euc euc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float tau0 = st->tau[lane & 31];
  float tau1 = st->tau[32 + (lane & 31)];
  if (st->ovf) return;
  for (uint32_t i0 = 0; i0 < L * 512u; i0 += V8_WAVES * 64 * 4) {  // 48 KB at dim 768: loads of four rounds in flight together
    v4i t[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t i = i0 + u * V8_WAVES * 64 + tid;
      t[u] = i < L * 512u ? ((const v4i*)Qf8)[i] : v4i{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t i = i0 + u * V8_WAVES * 64 + tid;
      if (i < L * 512u) ((v4i*)smem)[i] = t[u];
    }
  }
  float qs0 = 1.f, qs1 = 1.f;
  if (SCALED && q_scale) { qs0 = q_scale[lane & 31]; qs1 = q_scale[32 + (lane & 31)]; }
  float qn0 = 0.f, qn1 = 0.f;
  int qq0 = 0, qq1 = 0;
  if (SCALED && euc.mode) {
    qn0 = euc.qaux[lane & 31]; qn1 = euc.qaux[32 + (lane & 31)];
    qq0 = __float_as_int(euc.qaux[64 + (lane & 31)]); qq1 = __float_as_int(euc.qaux[96 + (lane & 31)]);
  }
  __syncthreads();

  if (ANN && ann.tiles) {
    const uint32_t na = *ann.n_tiles;
    ntiles = na > tile0 ? min(ntiles, na - tile0) : 0u;
  }
  const uint32_t first = blockIdx.x;
  if (first >= ntiles) return;
  const uint32_t my_tiles = (ntiles - first + gridDim.x - 1) / gridDim.x;
  const uint32_t G = my_tiles * L;  // lines of this wave's row blocks, flattened

  // fragment-ordered image: my 32-row block of tile t starts at (4 t + w) * L * 4096, line c at + 4096 c, piece j at + 1024 j
  const size_t lane_off = (size_t)w * L * 4096u + (size_t)lane * 16u;
  const size_t tile_stride = (size_t)(V8_WAVES * 32) * dim_pad;
  uint32_t i_tile = 0, i_line = 0, i_tix = 0;
  // ANN: the id of a tile comes from the batch's list -- a load the first line of the tile would have to wait for.  It is fetched
  // one tile ahead (i_tix_next), and the ids of the tiles in flight stay in a small ring for the epilogue (tix_ring).
  uint32_t i_tix_next = (ANN && ann.tiles) ? ann.tiles[tile0 + first] : 0u;
  uint32_t tix_ring[4] = {0u, 0u, 0u, 0u};
  v4i xa[V8_D][4];
  auto issue = [&](v4i(&buf)[4]) {
    const uint32_t t = min(i_tile, my_tiles - 1);  // past the end: re-read a line of the last tile (never consumed)
    if (ANN) {
      if (i_line == 0) {
        if (ann.tiles) {
          i_tix = i_tix_next;
          i_tix_next = ann.tiles[tile0 + first + min(i_tile + 1u, my_tiles - 1u) * gridDim.x];
        } else {
          i_tix = tile0 + first + t * gridDim.x;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) tix_ring[r] = (i_tile & 3u) == (uint32_t)r ? i_tix : tix_ring[r];
      }
    }
    const size_t tix = ANN ? (size_t)i_tix : (size_t)(tile0 + first + (size_t)t * gridDim.x);
    const v4i* p = (const v4i*)(X + tix * tile_stride + lane_off + (size_t)i_line * 4096u);
#pragma unroll
    for (int j = 0; j < 4; j++) buf[j] = __builtin_nontemporal_load(p + j * 64);
    if (++i_line == L) { i_line = 0; ++i_tile; }
  };

  v16i acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
#pragma unroll
  for (int d = 0; d < V8_D; d++) issue(xa[d]);

  uint32_t c_tile = 0, c_line = 0;
  for (uint32_t g0 = 0; g0 < G; g0 += V8_D) {
#pragma unroll
    for (int d = 0; d < V8_D; d++) {
      if (!EVEN && g0 + d >= G) break;
      const char* qb = smem + (size_t)c_line * 8192u + lane * 16;
      v4i b[8];
#pragma unroll
      for (int j = 0; j < 8; j++) b[j] = *(const v4i*)(qb + j * 1024);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[d][j], b[2 * j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[d][j], b[2 * j + 1], acc1, 0, 0, 0);
      }
      issue(xa[d]);
      if (++c_line == L) {
        // ---- threshold filter: lane owns query (lane & 31) + {0, 32}, 16 rows per accumulator
        uint32_t c_ring = tix_ring[0];
#pragma unroll
        for (int r = 1; r < 4; r++) c_ring = (c_tile & 3u) == (uint32_t)r ? tix_ring[r] : c_ring;
        const unsigned long long c_tix = ANN ? (unsigned long long)c_ring : (unsigned long long)(tile0 + first + (unsigned long long)c_tile * gridDim.x);
        const unsigned long long row_base = c_tix * (V8_WAVES * 32) + 32u * w + 4u * (lane >> 5);
        float f0[16], f1[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          f0[r] = (float)acc0[r];
          f1[r] = (float)acc1[r];
          if (SCALED) {  // dot_i8_quantized: dot as f32 * scale1 (query) * scale2 (embedding)
            const unsigned long long row = row_base + (r & 3) + 8 * (r >> 2);
            const float es = (row_scale && row < n_rows) ? row_scale[row] : 1.f;
            if (euc.mode == 1) {  // exact integers
              const int rs = row < n_rows ? euc.row_sq[row] : 0;
              f0[r] = -(float)(qq0 + rs - 2 * acc0[r]);
              f1[r] = -(float)(qq1 + rs - 2 * acc1[r]);
            } else if (euc.mode == 2) {
              const float rn = (euc.row_norm && row < n_rows) ? euc.row_norm[row] : 0.f;
              const float d0 = ss_fmul(ss_fmul(f0[r], qs0), es), d1 = ss_fmul(ss_fmul(f1[r], qs1), es);
              f0[r] = -fmaxf(ss_fsub(ss_fadd(qn0, rn), ss_fmul(2.0f, d0)), 0.0f);
              f1[r] = -fmaxf(ss_fsub(ss_fadd(qn1, rn), ss_fmul(2.0f, d1)), 0.0f);
            } else {
              f0[r] = f0[r] * qs0 * es;
              f1[r] = f1[r] * qs1 * es;
            }
          }
        }
        float m0 = f0[0], m1 = f1[0];
#pragma unroll
        for (int r = 1; r < 16; r++) { m0 = fmaxf(m0, f0[r]); m1 = fmaxf(m1, f1[r]); }
        if (ANN) {
          if (m0 > tau0) vs_append_ann(f0, tau0, lane & 31, row_base, n_rows, st, cand, ann);
          if (m1 > tau1) vs_append_ann(f1, tau1, 32 + (lane & 31), row_base, n_rows, st, cand, ann);
        } else {
          if (m0 > tau0) vs_append(f0, tau0, lane & 31, row_base, n_rows, st, cand);
          if (m1 > tau1) vs_append(f1, tau1, 32 + (lane & 31), row_base, n_rows, st, cand);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
        c_line = 0;
        ++c_tile;
      }
    }
  }
}

// ---------------------------------------------------------------- host side
int ssi_vec8_qprep(ss_shard* s, const int8_t* d_queries, uint32_t nb, hipStream_t st) {
  vec8_qprep_kernel<<<s->dim_pad8 / V8_LINE, 512, 0, st>>>(d_queries, nb, s->dim, (int8_t*)s->d_Qf);
  return SS_OK;
}
int ssi_vec8_qaux(ss_shard* s, const int8_t* d_queries, uint32_t nb, const float* d_qnorm, hipStream_t st) {
  if (!s->d_qaux) SS_HIP(hipMalloc(&s->d_qaux, 128 * sizeof(float)));
  vec8_qaux_kernel<<<1, 64, 0, st>>>(d_queries, nb, s->dim, d_qnorm, s->d_qaux);
  return SS_OK;
}
int ssi_vec8_row_sq(ss_shard* s, hipStream_t st, uint64_t r0) {
  if (!s->d_X8) return SS_ESTATE;
  if (!s->d_row_sq) SS_HIP(hipMalloc(&s->d_row_sq, (size_t)std::max<uint64_t>(s->n_rows, s->vec_rows_cap) * sizeof(int32_t)));
  if (r0 >= s->n_rows) return SS_OK;
  vec8_row_sq_kernel<<<(unsigned)((s->n_rows - r0 + 255) / 256), 256, 0, st>>>(s->d_X8, s->dim, s->dim_pad8, (unsigned long long)r0, (unsigned long long)s->n_rows, s->d_row_sq);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
// Euclidean sub-mode of the image in flight: quantised (scales / norms present) or plain integer
static V8Euc v8_euc(const ss_shard* s, const float* d_qscale) {
  V8Euc e{0, nullptr, nullptr, nullptr};
  if (s->vec_similarity != SS_SIM_EUCLIDEAN) return e;
  e.mode = (s->d_row_scale || d_qscale) ? 2 : 1;
  e.row_norm = s->d_row_norm;
  e.row_sq = s->d_row_sq;
  e.qaux = s->d_qaux;
  return e;
}

template <bool SCALED, bool EVEN, bool ANN>
static int launch_vec8(ss_shard* s, uint32_t tile0, uint32_t ntiles, const float* d_qscale, const VAnn& ann, hipStream_t st) {
  const uint32_t L = s->dim_pad8 / V8_LINE;
  const uint32_t gmax = 768;  // (3 workgroups per CU; profiles/r5: 512 / 768 / 1024 measured)
  const uint32_t grid = std::min<uint32_t>(ntiles, gmax);
  SS_SET_MAX_LDS((vec8_scan_kernel<SCALED, EVEN, ANN>), 160 * 1024);
  vec8_scan_kernel<SCALED, EVEN, ANN><<<grid, V8_WAVES * 64, L * 8192u, st>>>(
      s->d_X8, s->dim_pad8, (unsigned long long)s->n_rows, (const int8_t*)s->d_Qf, L, tile0, ntiles, s->d_row_scale, d_qscale,
      (VState*)s->d_vstate, (unsigned long long*)s->d_cand, ann, v8_euc(s, d_qscale));
  return SS_OK;
}

template <bool ANN>
static int launch_vec8_ann(ss_shard* s, uint32_t tile0, uint32_t ntiles, const float* d_qscale, const VAnn& ann, hipStream_t st) {
  const bool scaled = s->d_row_scale != nullptr || d_qscale != nullptr || s->vec_similarity == SS_SIM_EUCLIDEAN;
  const bool even = (s->dim_pad8 / V8_LINE) % V8_D == 0;
  if (scaled) return even ? launch_vec8<true, true, ANN>(s, tile0, ntiles, d_qscale, ann, st) : launch_vec8<true, false, ANN>(s, tile0, ntiles, d_qscale, ann, st);
  return even ? launch_vec8<false, true, ANN>(s, tile0, ntiles, d_qscale, ann, st) : launch_vec8<false, false, ANN>(s, tile0, ntiles, d_qscale, ann, st);
}

int ssi_vec8_launch_scan(ss_shard* s, uint32_t tile0, uint32_t ntiles, const float* d_qscale, const VAnn* ann, hipStream_t st) {
  if (ann) return launch_vec8_ann<true>(s, tile0, ntiles, d_qscale, *ann, st);
  return launch_vec8_ann<false>(s, tile0, ntiles, d_qscale, VAnn{}, st);
}

// row-major device staging -> the fragment-ordered image
int ssi_vec8_permute(ss_shard* s, const int8_t* d_rows_row_major, hipStream_t st) {
  const unsigned long long total = (unsigned long long)s->n_rows_pad * (s->dim_pad8 / 16u);
  vec8_permute_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(d_rows_row_major, s->n_rows, s->dim, s->dim_pad8,
                                                                       s->n_rows_pad, s->d_X8);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_vec8_permute_range(ss_shard* s, const int8_t* d_rows_row_major, uint64_t r0, uint64_t n, hipStream_t st) {
  const unsigned long long total = (unsigned long long)n * (s->dim_pad8 / 16u);
  if (!total) return SS_OK;
  vec8_permute_range_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(d_rows_row_major, r0, n, s->dim, s->dim_pad8, s->d_X8);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_vec8_gather_rows(ss_shard* s, uint64_t r0, uint64_t n, int8_t* d_out, hipStream_t st) {
  const unsigned long long total = (unsigned long long)n * s->dim;
  vec8_gather_rows_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(s->d_X8, s->dim, s->dim_pad8, r0, n, d_out);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_vec8_quantize(ss_shard* s, hipStream_t st) {
  vec8_quantize_kernel<<<(unsigned)s->n_rows, 256, 0, st>>>(s->d_X, s->dim, s->dim_pad, (unsigned long long)s->n_rows, s->d_X8, s->dim_pad8);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
