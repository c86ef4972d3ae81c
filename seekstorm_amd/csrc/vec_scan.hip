// Dense-vector brute-force scan (AnnMode::All, F32 dot / cosine) for gfx950.
//
// Replaces search_vector_shard's hot loop (vector.rs:1397-1466: read_record -> dot_f32_avx2 ->
// TopK::push) for a batch of up to 64 queries per pass over the matrix.
//
//   scores[N x 64] = X[N x dim] . Q^T  computed with v_mfma_f32_32x32x2_f32 (exact f32 fma chain),
//   never materialised: each wave holds a 32-row x 64-query accumulator tile, compares it in
//   registers against the per-query running threshold tau (the current k-th best score) and appends
//   the rare survivors to a per-query candidate buffer.  A tiny "refine" kernel between geometrically
//   growing row chunks re-selects the exact top-k and raises tau (TopK::push semantics, vector.rs:410-496:
//   strict '>' against the current minimum; ties keep the earlier row).
//
// Data movement: X streams HBM -> LDS with global_load_lds_dwordx4 in full 128-byte lines (8 lanes per
// row-line, XOR-swizzled on the SOURCE piece so ds_read_b128 of the MFMA A fragments is conflict-free);
// Q is pre-permuted into MFMA B-fragment order once per batch and re-streamed from L2 per K-chunk.
// 3-stage LDS ring, counted vmcnt, one raw s_barrier per K-chunk, persistent across tiles.
#include <cstdio>
#include <cstdlib>

#include "ss_common.h"
#include "vec_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------- Q -> MFMA B-fragment order
// Qf[kc][nt(2)][t(4)][lane(64)][j(4)] = Q[q = nt*32 + (lane&31)][k = kc*32 + (2t + (lane>>5))*4 + j]
// euclid: the image rows are [x, |x|^2, 1] and the query becomes [2 q, -1, -|q|^2], so that the scan's dot product is
// 2 q.x - |x|^2 - |q|^2 = -|q - x|^2 (VectorSimilarity::Euclidean: minus the squared distance, larger = closer).  2 q is
// exact, so the ANN medoid kernel recovers q as 0.5 * Qf.  The scan's value only SELECTS: the scores returned are
// recomputed in the reference's own summation order (vec_rescore_euclid_kernel).
__global__ void vec_qprep_kernel(const float* __restrict__ Q, uint32_t nq, uint32_t dim, float* __restrict__ Qf, int euclid) {
  const uint32_t kc = blockIdx.x;
  for (uint32_t e = threadIdx.x; e < 2048; e += blockDim.x) {
    uint32_t j = e & 3, lane = (e >> 2) & 63, t = (e >> 8) & 3, nt = e >> 10;
    uint32_t q = nt * 32 + (lane & 31);
    uint32_t k = kc * 32 + (2 * t + (lane >> 5)) * 4 + j;
    float v = 0.0f;
    if (q < nq) {
      if (k < dim) v = euclid ? 2.0f * Q[(size_t)q * dim + k] : Q[(size_t)q * dim + k];
      else if (euclid && k == dim) v = -1.0f;
      else if (euclid && k == dim + 1) {
        float ss = 0.0f;
        for (uint32_t i = 0; i < dim; i++) ss = fmaf(Q[(size_t)q * dim + i], Q[(size_t)q * dim + i], ss);
        v = -ss;
      }
    }
    Qf[(size_t)kc * 2048 + e] = v;
  }
}

// f32 Euclidean image: column dim = |x|^2, column dim + 1 = 1 (one thread per row)
__global__ void vec_augment_kernel(float* __restrict__ X, unsigned long long r0, unsigned long long n_rows, uint32_t dim, uint32_t dim_pad) {
  const unsigned long long r = r0 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  float* row = X + r * dim_pad;
  float ss = 0.0f;
  for (uint32_t i = 0; i < dim; i++) ss = fmaf(row[i], row[i], ss);
  row[dim] = ss;
  row[dim + 1] = 1.0f;
}
int ssi_vec_augment(ss_shard* s, hipStream_t st, uint64_t r0) {
  if (!s->d_X || s->dim_pad < s->dim + 2) return SS_ESTATE;
  if (r0 >= s->n_rows) return SS_OK;
  vec_augment_kernel<<<(unsigned)((s->n_rows - r0 + 255) / 256), 256, 0, st>>>(s->d_X, (unsigned long long)r0, (unsigned long long)s->n_rows, s->dim, s->dim_pad);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// The kept candidates of a Euclidean f32 search get the score the reference computes: -euclidean_f32_avx2 (eight lanes of
// sub / mul / add over dim / 8 steps, lanes summed 0..7, vector_similarity.rs:938-966) when dim % 8 == 0, else
// -euclidean_f32 (sequential, 912-918) -- every operation rounded on its own, no fma.  One thread per candidate.
// thr: the search's raw threshold is applied HERE, on the exact score (`score < threshold -> reject`, vector.rs:423): the scan
// runs without it, because its expanded form 2 q.x - |x|^2 - |q|^2 may land on the other side of a threshold the exact
// distance meets.  A rejected candidate's key becomes 0 (vec_final_kernel drops it).
__global__ void vec_rescore_euclid_kernel(const VState* __restrict__ st, unsigned long long* __restrict__ cand, const float* __restrict__ X,
                                          uint32_t dim, uint32_t dim_pad, const float* __restrict__ Q, uint32_t nq, uint32_t k, float thr) {
  const uint32_t q = blockIdx.x;
  if (q >= nq) return;
  const uint32_t n = st->cnt[q * VS_CNT_STRIDE] < k ? st->cnt[q * VS_CNT_STRIDE] : k;
  const float* qv = Q + (size_t)q * dim;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned long long key = cand[(size_t)q * VS_CAP + i];
    const uint32_t row = 0xFFFFFFFFu - (uint32_t)key;
    const float* x = X + (size_t)row * dim_pad;
    float d2;
    if ((dim & 7u) == 0) {
      float l[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (uint32_t c = 0; c < dim; c += 8)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float d = ss_fsub(qv[c + j], x[c + j]);
          l[j] = ss_fadd(l[j], ss_fmul(d, d));
        }
      d2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) d2 = ss_fadd(d2, l[j]);
    } else {
      d2 = 0.f;
      for (uint32_t c = 0; c < dim; c++) {
        const float d = ss_fsub(qv[c], x[c]);
        d2 = ss_fadd(d2, ss_fmul(d, d));
      }
    }
    cand[(size_t)q * VS_CAP + i] = (-d2 < thr) ? 0ull : mk_key(-d2, row);
  }
}

__global__ void vec_init_kernel(VState* st, float tau_init) {
  int i = threadIdx.x;
  if (i < 64) {
    st->tau[i] = tau_init;
    st->cnt[i * VS_CNT_STRIDE] = 0;
    st->kept[i] = 0;
    st->total[i] = 0ull;
  }
  if (i == 0) st->ovf = 0;
}

// ---------------------------------------------------------------- the scan
// TWO: queries 32..63 are in use.  A batch of <= 32 queries skips their half of the MFMA work, which turns the scan from
// MFMA-bound (9.2 ms per pass at 10 M x 768) into HBM-bound: the latency of small batches and single queries.
// ANN: the launch walks the batch's list of selected tiles (AnnMode::Nprobe / Similaritythreshold, vec_ann.hip) and admits
// a row only for the queries that selected its cluster.
template <bool TWO, bool ANN>
__global__ void __launch_bounds__(VS_WAVES * 64, 2)
vec_scan_kernel(const float* __restrict__ X, uint32_t dim_pad, unsigned long long n_rows,
                const float* __restrict__ Qf, uint32_t nch, uint32_t tile0, uint32_t ntiles, VState* __restrict__ st,
                unsigned long long* __restrict__ cand, VAnn ann) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  // per-launch constants: my two queries' thresholds (tau only changes between launches)
  float tau0 = st->tau[lane & 31];
  float tau1 = st->tau[32 + (lane & 31)];
  if (st->ovf) return;
  // pin the thresholds in registers NOW: a compiler-placed wait at their first use (the per-tile epilogue)
  // would be a vmcnt(0) that drains the LDS-DMA ring once per tile
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(tau0), "+v"(tau1)::"memory");

  if (ANN && ann.tiles) {
    const uint32_t na = *ann.n_tiles;
    ntiles = na > tile0 ? min(ntiles, na - tile0) : 0u;
  }
  // tiles of this workgroup: tile0 + blockIdx.x + i * gridDim.x
  const uint32_t first = blockIdx.x;
  if (first >= ntiles) return;
  const uint32_t my_tiles = (ntiles - first + gridDim.x - 1) / gridDim.x;
  const uint32_t G = my_tiles * nch;

  // ---- producer addressing (global -> LDS, 16 B per lane, 8 lanes per 128-B row line)
  uint32_t xsrc[4];  // float offset of my piece inside the tile's chunk, per issue i
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t rl = 32u * w + 8u * i + (lane >> 3);     // row inside tile
    uint32_t f = (rl >> 1) & 7u;                      // swizzle key (conflict-free ds_read_b128)
    uint32_t piece = (uint32_t)(lane & 7) ^ f;        // logical 16-B piece fetched into physical slot lane&7
    xsrc[i] = rl * dim_pad + piece * 4u;
  }
  const size_t tile_stride = (size_t)VS_TR * dim_pad;  // floats per tile

  // ---- consumer addressing
  uint32_t aoff[4];
  {
    uint32_t rw = 32u * w + (lane & 31);
    uint32_t f = (rw >> 1) & 7u;
#pragma unroll
    for (int t = 0; t < 4; t++) aoff[t] = rw * 128u + (((2u * t + (lane >> 5)) ^ f) << 4);
  }
  const uint32_t boff = VS_XS + lane * 16u;

  uint32_t i_tile = 0, i_kc = 0, i_stage = 0;  // issue cursor
  uint32_t i_tix = 0;
  auto issue = [&]() {
    if (ANN) { if (i_kc == 0) i_tix = ann.tiles ? ann.tiles[tile0 + first + i_tile * gridDim.x] : tile0 + first + i_tile * gridDim.x; }
    const size_t tix = ANN ? (size_t)i_tix : (size_t)(tile0 + first + (size_t)i_tile * gridDim.x);
    const float* xt = X + tix * tile_stride + i_kc * VS_KC;
    char* sb = smem + i_stage * VS_STAGE;
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_global_load_lds(GPTR(xt + xsrc[i]), LPTR(sb + (32 * w + 8 * i) * 128), 16, 0, 0);
    const float* qs = Qf + (size_t)i_kc * 2048 + lane * 4;
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_global_load_lds(GPTR(qs + (2 * w + i) * 256), LPTR(sb + VS_XS + (2 * w + i) * 1024), 16, 0, 0);
    if (++i_kc == nch) { i_kc = 0; ++i_tile; }
    if (++i_stage == VS_STAGES) i_stage = 0;
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }

  issue();
  if (G > 1) issue();

  uint32_t c_tile = 0, c_kc = 0, c_stage = 0;  // consume cursor
  for (uint32_t g = 0; g < G; ++g) {
    // my own 6 LDS-DMA pieces of chunk g have landed (chunk g+1's 6 may still be in flight)
    if (g + 1 < G) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's pieces landed; everyone finished reading stage (g-1)%3
    if (g + 2 < G) issue();        // refill the stage consumed in iteration g-1

    const char* sb = smem + c_stage * VS_STAGE;
    f32x4 xa[4], qb0[4], qb1[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      xa[t] = *(const f32x4*)(sb + aoff[t]);
      qb0[t] = *(const f32x4*)(sb + boff + t * 1024);
      if (TWO) qb1[t] = *(const f32x4*)(sb + boff + (4 + t) * 1024);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[t][j], qb0[t][j], acc0, 0, 0, 0);
        if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[t][j], qb1[t][j], acc1, 0, 0, 0);
      }
    }

    if (++c_kc == nch) {
      // ---- fused top-k filter: lane owns query (lane&31)+{0,32}, 16 rows per accumulator
      const unsigned long long c_tix = (ANN && ann.tiles) ? (unsigned long long)ann.tiles[tile0 + first + c_tile * gridDim.x]
                                           : (unsigned long long)(tile0 + first + (unsigned long long)c_tile * gridDim.x);
      const unsigned long long row_base = c_tix * VS_TR + 32u * w + 4u * (lane >> 5);
      float m0 = acc0[0], m1 = TWO ? acc1[0] : -INFINITY;
#pragma unroll
      for (int r = 1; r < 16; r++) { m0 = fmaxf(m0, acc0[r]); if (TWO) m1 = fmaxf(m1, acc1[r]); }
      if (m0 > tau0) {
        float f[16];
#pragma unroll
        for (int r = 0; r < 16; r++) f[r] = acc0[r];
        if (ANN) vs_append_ann(f, tau0, lane & 31, row_base, n_rows, st, cand, ann);
        else vs_append(f, tau0, lane & 31, row_base, n_rows, st, cand);
      }
      if (TWO && m1 > tau1) {
        float f[16];
#pragma unroll
        for (int r = 0; r < 16; r++) f[r] = acc1[r];
        if (ANN) vs_append_ann(f, tau1, 32 + (lane & 31), row_base, n_rows, st, cand, ann);
        else vs_append(f, tau1, 32 + (lane & 31), row_base, n_rows, st, cand);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
      c_kc = 0;
      ++c_tile;
    }
    if (++c_stage == VS_STAGES) c_stage = 0;
  }
}


// ---------------------------------------------------------------- ANN batches: the sparse instantiation
// A batch of independent queries under AnnMode::Nprobe selects, per query, n_probe of every level's clusters; the batch's UNION of
// tiles is nearly the whole image (64 queries x 16 of 256 clusters: 98 %), but a given tile concerns only a few of the queries
// (4 of 64 on average).  The MFMA kernel above still multiplies every tile with all 64 queries -- the whole brute-force work plus
// the preparation.  This kernel does the arithmetic only for the queries a tile concerns, on the VALU: per tile the interested
// queries are found from the selection bitmaps; up to 8 at a time are staged in LDS (3 KB each at dim 768, one copy per
// workgroup); the tile's rows stream from HBM straight into registers (a wave reads one row per instruction group, fully
// coalesced, dim / 64 components per lane), four rows at a time, and every query vector read from LDS is used for the four of
// them.  The 4 x 8 partial sums are summed over the wave by a transposing butterfly -- 16 + 8 + 4 + 2 + 1 exchanges leave lane L
// with the sum over its 32-lane half of value L & 31, one more adds the halves: 8 exchanges per row instead of 48 -- after which
// lane L < 32 holds the score of (row L >> 3, query slot L & 7).  The pass is then bound by HBM, not by the matrix cores.
// f32 dot / cosine, dim = dim_pad a multiple of 256; everything else (and batches of <= 32 queries, whose MFMA pass is HBM bound
// already) keeps the kernel above.  The summation order differs from the MFMA kernel's (and from the reference's avx2 order), inside
// the same 1e-4 tolerance.
template <int NV>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) vec_ann_sparse_kernel(const float* __restrict__ X, uint32_t dim_pad, unsigned long long n_rows,
                                                            const float* __restrict__ Q, uint32_t nb, uint32_t tile0, uint32_t ntiles,
                                                            VState* __restrict__ st, unsigned long long* __restrict__ cand, VAnn ann) {
  __shared__ __attribute__((aligned(16))) float qs[8 * NV * 256];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (st->ovf) return;
  {
    const uint32_t na = *ann.n_tiles;
    ntiles = na > tile0 ? min(ntiles, na - tile0) : 0u;
  }
  const float tau_l = st->tau[lane];  // lane q holds query q's threshold (fixed during a launch)
  const uint32_t W = ann.sel_words;
  const uint32_t slot = (uint32_t)lane & 7u;
  for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const unsigned long long tix = ann.tiles[tile0 + t];
    const unsigned long long row0 = tix * VS_TR;
    if (row0 >= n_rows) continue;  // (uniform over the workgroup)
    const unsigned long long rlast = min(row0 + (unsigned long long)VS_TR, n_rows) - 1ull;
    const uint32_t c_lo = ann.row_cluster[row0], c_hi = ann.row_cluster[rlast];
    // the queries this tile concerns: some selected cluster among the tile's (the same mask in all four waves)
    bool intr = false;
    if ((uint32_t)lane < nb)
      for (uint32_t x = c_lo >> 5; x <= (c_hi >> 5); x++) {
        uint32_t m = 0xFFFFFFFFu;
        if (x == (c_lo >> 5)) m &= 0xFFFFFFFFu << (c_lo & 31u);
        if (x == (c_hi >> 5)) m &= 0xFFFFFFFFu >> (31u - (c_hi & 31u));
        intr = intr || (ann.sel[(size_t)lane * W + x] & m) != 0u;
      }
    unsigned long long im = __ballot(intr);
    while (im) {
      // up to 8 of them: their indices are uniform
      uint32_t qidx[8];
      uint32_t nqk = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (im) { qidx[j] = (uint32_t)__builtin_ctzll(im); im &= im - 1ull; nqk++; }
        else qidx[j] = qidx[0];
      }
      __syncthreads();  // the previous chunk's readers are done with qs
#pragma unroll
      for (int j = 0; j < 8; j++)
        for (uint32_t e = (uint32_t)tid; e < (uint32_t)NV * 64u; e += 256u)
          *(f32x4*)(qs + (uint32_t)j * NV * 256u + e * 4u) = *(const f32x4*)(Q + (size_t)qidx[j] * dim_pad + e * 4u);
      __syncthreads();
      uint32_t my_q = qidx[0];
#pragma unroll
      for (int j = 1; j < 8; j++) my_q = slot == (uint32_t)j ? qidx[j] : my_q;
      const float my_tau = __shfl(tau_l, (int)my_q);
      const unsigned long long wrow0 = row0 + 32ull * (unsigned long long)w;
      auto load4 = [&](f32x4(&x)[4][NV], uint32_t g) {  // rows 4 g .. 4 g + 3 of my 32
#pragma unroll
        for (int r = 0; r < 4; r++) {
          unsigned long long rr = wrow0 + 4u * g + (uint32_t)r;
          rr = rr < n_rows ? rr : n_rows - 1ull;  // rows past the end are computed and dropped
          const float* xr = X + rr * dim_pad + (uint32_t)lane * 4u;
#pragma unroll
          for (int v = 0; v < NV; v++) x[r][v] = *(const f32x4*)(xr + (uint32_t)v * 256u);
        }
      };
      auto group = [&](const f32x4(&x)[4][NV], uint32_t g) {
        float a[32];  // a[r * 8 + j]: row r of the group, query slot j -- my lane's part of the dot product
#pragma unroll
        for (int j = 0; j < 8; j++) {
          f32x4 qv[NV];
#pragma unroll
          for (int v = 0; v < NV; v++) qv[v] = *(const f32x4*)(qs + (uint32_t)j * NV * 256u + (uint32_t)v * 256u + (uint32_t)lane * 4u);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < NV; v++) {
              acc = __builtin_fmaf(x[r][v][0], qv[v][0], acc);
              acc = __builtin_fmaf(x[r][v][1], qv[v][1], acc);
              acc = __builtin_fmaf(x[r][v][2], qv[v][2], acc);
              acc = __builtin_fmaf(x[r][v][3], qv[v][3], acc);
            }
            a[r * 8 + j] = acc;
          }
          __builtin_amdgcn_sched_barrier(0);  // one query vector live at a time (the scheduler would hoist all 8 x NV LDS reads: 96 registers)
        }
        // 32 values -> 1 over the lane bits 4 .. 0: the lane with the bit set keeps the upper half
#pragma unroll
        for (int h = 16; h >= 1; h >>= 1) {
          const bool up = ((uint32_t)lane & (uint32_t)h) != 0u;
#pragma unroll
          for (int i = 0; i < h; i++) {
            const float keep = up ? a[i + h] : a[i];
            const float send = up ? a[i] : a[i + h];
            a[i] = keep + __shfl_xor(send, h);
          }
        }
        const float score = a[0] + __shfl_xor(a[0], 32);  // value index lane & 31 = (row (lane >> 3) & 3, slot lane & 7)
        const unsigned long long row = wrow0 + 4u * g + (((uint32_t)lane >> 3) & 3u);
        bool ok = lane < 32 && slot < nqk && row < n_rows && score > my_tau;
        if (ok) {
          const uint32_t c = ann.row_cluster[row];
          ok = (ann.sel[(size_t)my_q * W + (c >> 5)] >> (c & 31u)) & 1u;
          if (ok && ann.row_field) {
            const uint32_t fld = ann.row_field[row];
            ok = fld < 64u && ((ann.field_mask >> fld) & 1ull);
          }
        }
        if (ok) {
          const uint32_t at = atomicAdd(&st->cnt[my_q * VS_CNT_STRIDE], 1u);
          if (at < VS_CAP) cand[(size_t)my_q * VS_CAP + at] = mk_key(score, (uint32_t)row);
        }
      };
      // no software prefetch: four waves per SIMD, each with its four rows (dim x 16 B) in flight, cover the memory latency
      // between them, and the registers a second buffer would take are what keeps them at four
#pragma unroll 1
      for (uint32_t g = 0; g < 8u; g++) {
        f32x4 xa[4][NV];
        load4(xa, g);
        group(xa, g);
      }
    }
  }
}

// ---------------------------------------------------------------- refine: exact top-k of the candidates, raise tau
// One workgroup per query.  Sorts the (<= VS_CAP) candidate keys descending in LDS (bitonic), keeps the best k
// at the front of the buffer, sets tau = k-th best score (TopK::push admits only score > current minimum).
// With several records per doc (one per indexed field x chunk, vector.rs:561-576) TopK::push keeps ONE entry per doc
// holding its best score (vector.rs:441-452, 462-473): candidates are first sorted by (doc, key desc), every entry
// but the first of a doc is dropped, then the survivors are sorted by key.
__device__ __forceinline__ void vr_bitonic(unsigned long long* keys, uint32_t* docs, uint32_t np, bool by_doc) {
  for (uint32_t size = 2; size <= np; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = threadIdx.x; i < (np >> 1); i += blockDim.x) {
        uint32_t lo = 2 * i - (i & (stride - 1));
        uint32_t hi = lo + stride;
        bool desc = ((lo & size) == 0);
        unsigned long long a = keys[lo], b = keys[hi];
        bool a_before_b;  // order: by_doc ? (doc asc, key desc) : (key desc)
        if (by_doc) {
          uint32_t da = docs[lo], db = docs[hi];
          a_before_b = da != db ? da < db : a > b;
          if (a_before_b != desc && !(da == db && a == b)) { keys[lo] = b; keys[hi] = a; docs[lo] = db; docs[hi] = da; }
        } else {
          if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
        }
      }
      __syncthreads();
    }
  }
}

constexpr int VR_THREADS = 1024;  // (256 threads measured twice as slow on the ~1700 candidates a growth-16 chunk leaves)
__global__ void __launch_bounds__(VR_THREADS) vec_refine_kernel(VState* __restrict__ st, unsigned long long* __restrict__ cand,
                                                         uint32_t k, const uint32_t* __restrict__ row_doc /* non-null: dedup */,
                                                         const uint32_t* __restrict__ doc_map /* row -> doc, null = identity */,
                                                         const uint32_t* __restrict__ del, uint32_t del_words,
                                                         uint32_t del_stride /* words; != 0: one bitmap per query (deep pages) */,
                                                         uint32_t del_rows /* ... of the first del_rows queries: the batch's own */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = (unsigned long long*)smem;
  uint32_t* docs = (uint32_t*)(smem + VS_CAP * sizeof(unsigned long long));
  const uint32_t q = blockIdx.x;
  const uint32_t raw = st->cnt[q * VS_CNT_STRIDE];
  const uint32_t kept = st->kept[q];
  if (raw == kept) return;  // nothing new since the last refine
  if (raw > VS_CAP) {
    if (threadIdx.x == 0) st->ovf = 1;  // candidate buffer overflow: host re-runs the batch in safe mode
    return;
  }
  const uint32_t n = raw;
  uint32_t np = 64;
  while (np < n) np <<= 1;
  unsigned long long* base = cand + (size_t)q * VS_CAP;
  if (del && del_stride) del = q < del_rows ? del + (size_t)q * del_stride : nullptr;  // (slots past the batch hold no query of anybody's)
  __shared__ uint32_t live, dropped;
  if (threadIdx.x == 0) { live = 0; dropped = 0; }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
    unsigned long long key = i < n ? base[i] : 0ull;
    if (del && key && i >= kept) {  // tombstones: the record was scored but is never pushed (vector.rs:1450-1452)
      const uint32_t row = 0xFFFFFFFFu - (uint32_t)key;
      const uint32_t doc = doc_map ? doc_map[row] : row;
      if ((doc >> 5) < del_words && ((del[doc >> 5] >> (doc & 31u)) & 1u)) {
        key = 0ull;
        atomicAdd(&dropped, 1u);
      }
    }
    keys[i] = key;
    if (row_doc) docs[i] = key ? row_doc[0xFFFFFFFFu - (uint32_t)key] : 0xFFFFFFFFu;  // empty slots sort last
  }
  __syncthreads();
  if (!row_doc) {
    // ---- one record per doc: SELECT instead of sort.  Between launches only the SET of the k best candidates and the
    // k-th score (tau) matter -- their order is made once, in vec_final_kernel.  Radix select over the 64-bit keys, most
    // significant byte first: histogram of the keys that share the prefix found so far, the bucket holding the k-th
    // largest, next byte; it stops as soon as that bucket holds one key.  Keys are unique (the row is part of the key),
    // so "key >= k-th key" keeps exactly k.  ~3 passes of 3 barriers for ~1700 candidates instead of the 66 barrier steps
    // of a 2048-key bitonic network.
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long s_prefix, s_mask, s_kth;
    __shared__ uint32_t s_want, s_single, s_slot;
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) c += keys[i] != 0ull;
    if (c) atomicAdd(&live, c);
    if (threadIdx.x == 0) { s_prefix = 0ull; s_mask = 0ull; s_kth = 0ull; s_want = k; s_single = 0u; s_slot = 0u; }
    __syncthreads();
    const uint32_t nl = live;
    unsigned long long kth = 0ull;  // keep every live key
    if (nl > k) {
      for (int pass = 7; pass >= 0; pass--) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix, mask = s_mask;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
          const unsigned long long key = keys[i];
          if (key && (key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // bucket of the want-th largest: buckets from 255 downwards
          const uint32_t lane = threadIdx.x;
          const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
          uint32_t x = h0 + h1 + h2 + h3;
          const uint32_t own = x;
          for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_down(x, o);
            if (lane + o < 64) x += y;
          }
          uint32_t above = x - own;  // keys in the buckets of higher lanes
          const uint32_t want = s_want;
          if (above < want && want <= above + own) {
            const uint32_t hs[4] = {h0, h1, h2, h3};
            for (int j = 3; j >= 0; j--) {
              if (want <= above + hs[j]) {
                s_prefix = prefix | ((unsigned long long)(4 * lane + j) << (8 * pass));
                s_mask = mask | (0xFFull << (8 * pass));
                s_want = want - above;
                s_single = hs[j] == 1u ? 1u : 0u;
                break;
              }
              above += hs[j];
            }
          }
        }
        __syncthreads();
        if (s_single || pass == 0) break;
      }
      // the k-th key itself: the one key left under the prefix (rank s_want = 1 among them)
      const unsigned long long prefix = s_prefix, mask = s_mask;
      for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long key = keys[i];
        if (key && (key & mask) == prefix) s_kth = key;
      }
      __syncthreads();
      kth = s_kth;
    }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long key = keys[i];
      if (key && key >= kth) base[atomicAdd(&s_slot, 1u)] = key;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t keep = s_slot;  // = min(nl, k)
      st->total[q] += (unsigned long long)(n - kept - dropped);
      st->cnt[q * VS_CNT_STRIDE] = keep;
      st->kept[q] = keep;
      if (nl > k) st->tau[q] = ord2f((uint32_t)(kth >> 32));
      else if (nl == k && k > 0) {  // exactly k candidates: tau = the smallest of them
        unsigned long long m = ~0ull;
        for (uint32_t i = 0; i < n; i++) if (keys[i] && keys[i] < m) m = keys[i];
        st->tau[q] = ord2f((uint32_t)(m >> 32));
      }
    }
    return;
  }
  // ---- several records per doc: sort by (doc, key), keep the best record of every doc, sort by key
  vr_bitonic(keys, docs, np, true);
  {
    // (doc asc, key desc): an entry whose predecessor has the same doc is a worse record of that doc
    unsigned long long mine[VS_CAP / VR_THREADS];
    for (uint32_t j = 0, i = threadIdx.x; i < np; i += blockDim.x, j++)
      mine[j] = (i > 0 && keys[i] && docs[i] == docs[i - 1]) ? 0ull : keys[i];
    __syncthreads();
    for (uint32_t j = 0, i = threadIdx.x; i < np; i += blockDim.x, j++) keys[i] = mine[j];
    __syncthreads();
  }
  vr_bitonic(keys, docs, np, false);
  // number of live entries after the dedup
  uint32_t cnt = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) cnt += keys[i] != 0ull;
  if (cnt) atomicAdd(&live, cnt);
  __syncthreads();
  const uint32_t nl = live;
  const uint32_t keep = nl < k ? nl : k;
  for (uint32_t i = threadIdx.x; i < keep; i += blockDim.x) base[i] = keys[i];
  if (threadIdx.x == 0) {
    st->total[q] += (unsigned long long)(n - kept - dropped);
    st->cnt[q * VS_CNT_STRIDE] = keep;
    st->kept[q] = keep;
    if (nl >= k && k > 0) st->tau[q] = ord2f((uint32_t)(keys[k - 1] >> 32));
  }
}

// The kept candidates (<= k, in no particular order after a select-refine) sorted by (score desc, row asc) and written out.
__global__ void vec_final_kernel(const VState* __restrict__ st, const unsigned long long* __restrict__ cand,
                                 const uint32_t* __restrict__ row_doc, uint32_t nq, uint32_t k,
                                 uint32_t* __restrict__ out_doc, float* __restrict__ out_score,
                                 uint32_t* __restrict__ out_count, unsigned long long* __restrict__ out_total) {
  __shared__ unsigned long long keys[SS_MAX_K];
  const uint32_t q = blockIdx.x;
  if (q >= nq) return;
  const uint32_t n = st->cnt[q * VS_CNT_STRIDE] < k ? st->cnt[q * VS_CNT_STRIDE] : k;
  uint32_t np = 64;
  while (np < n) np <<= 1;
  for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) keys[i] = i < n ? cand[(size_t)q * VS_CAP + i] : 0ull;
  __syncthreads();
  for (uint32_t size = 2; size <= np; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = threadIdx.x; i < (np >> 1); i += blockDim.x) {
        const uint32_t lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  __shared__ uint32_t n_valid;  // keys zeroed by the Euclidean rescoring (below the threshold) sorted to the end
  if (threadIdx.x == 0) n_valid = 0;
  __syncthreads();
  {
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) c += keys[i] != 0ull;
    if (c) atomicAdd(&n_valid, c);
  }
  __syncthreads();
  const uint32_t nv = n_valid;
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    uint32_t doc = SS_NO_DOC;
    float sc = 0.f;
    if (i < nv) {
      unsigned long long key = keys[i];
      uint32_t row = 0xFFFFFFFFu - (uint32_t)key;
      doc = row_doc ? row_doc[row] : row;
      sc = ord2f((uint32_t)(key >> 32));
    }
    out_doc[(size_t)q * k + i] = doc;
    out_score[(size_t)q * k + i] = sc;
  }
  if (threadIdx.x == 0) {
    out_count[q] = st->ovf ? 0xFFFFFFFFu : nv;
    out_total[q] = st->total[q];
  }
}

// ---------------------------------------------------------------- host side
int ssi_vec_alloc_ws(ss_shard* s) {
  if (!s->d_Qf) {  // queries in MFMA B-fragment order: f32 [dim_pad / 32][2048], i8 [dim_pad8 / 128][8192]
    const size_t bytes = s->d_X8 ? (size_t)s->dim_pad8 * 64 : (size_t)(s->dim_pad / VS_KC) * 2048 * sizeof(float);
    SS_HIP(hipMalloc(&s->d_Qf, bytes));
  }
  if (!s->d_vstate) SS_HIP(hipMalloc(&s->d_vstate, sizeof(VState)));
  if (!s->d_cand) SS_HIP(hipMalloc(&s->d_cand, (size_t)64 * VS_CAP * sizeof(unsigned long long)));
  SS_SET_MAX_LDS((vec_scan_kernel<true, false>), VS_LDS);
  SS_SET_MAX_LDS((vec_scan_kernel<false, false>), VS_LDS);
  SS_SET_MAX_LDS((vec_scan_kernel<true, true>), VS_LDS);
  SS_SET_MAX_LDS((vec_scan_kernel<false, true>), VS_LDS);
  SS_SET_MAX_LDS(vec_refine_kernel, VS_CAP * (sizeof(unsigned long long) + sizeof(uint32_t)));
  return SS_OK;
}

int ssi_vec_search(ss_shard* s, uint32_t nq, const void* d_queries, const float* d_qscale, uint32_t k, float thr,
                   uint32_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, uint64_t* d_out_total, hipStream_t st,
                   bool safe_mode, const ss_ann_mode* ann_mode, uint32_t* d_out_clusters, const float* d_qnorm) {
  if (!s->d_X && !s->d_X8) return SS_ESTATE;
  const bool euclid = s->vec_similarity == SS_SIM_EUCLIDEAN;
  // an ANN mode proper (some clusters are skipped) or only the field filter riding on the same admission test
  const bool ann_clusters = ann_mode && (ann_mode->n_probe != 0 || ann_mode->cluster_threshold_raw > -3.4028234663852886e38f);
  if (ann_clusters && !s->d_row_cluster) return SS_ESTATE;  // the image carries no cluster structure (ss_vec_set_clusters)
  if (ann_mode && ann_mode->field_mask && !s->d_row_field) return SS_ESTATE;  // nor field ids (ss_vec_set_fields)
  const bool i8 = s->d_X8 != nullptr;
  if (k == 0 || k > SS_MAX_K) return SS_EINVAL;
  int rc = ssi_vec_alloc_ws(s);
  if (rc) return rc;
  // SS_ANN_REPORT_OBSERVED: d_out_clusters holds three words per query (clusters, observed records low / high)
  const bool observed = ann_mode && (ann_mode->flags & SS_ANN_REPORT_OBSERVED) && d_out_clusters;
  const uint32_t ocw = observed ? 3u : 1u;
  if (observed) { rc = ssi_vec_observed_prepare(s, ann_mode->field_mask, st); if (rc) return rc; }
  // a mode that only asks for the report scans like AnnMode::All
  if (ann_mode && !ann_clusters && !ann_mode->field_mask) ann_mode = nullptr;
  const uint32_t nch = s->dim_pad / VS_KC;
  const uint32_t T = (uint32_t)(s->n_rows_pad / VS_TR);
  VState* vst = (VState*)s->d_vstate;
  unsigned long long* cand = (unsigned long long*)s->d_cand;
  // `score < threshold -> reject` (vector.rs:423)  ==  admit score > nextafter(threshold, -inf)
  // f32 Euclidean: the threshold is applied on the exact rescored values (vec_rescore_euclid_kernel), the scan runs without
  const float tau_init = (thr <= -3.4028234663852886e38f || (euclid && !i8)) ? -INFINITY : nextafterf(thr, -INFINITY);

  // chunk schedule (data independent): first chunk small (everything is a candidate), then geometric growth
  // so that the expected number of survivors per chunk stays ~ k * growth << VS_CAP.
  std::vector<uint32_t> chunks;
  if (safe_mode) {
    uint32_t step = (VS_CAP - k) / VS_TR;
    for (uint32_t d = 0; d < T; d += step) chunks.push_back(std::min(step, T - d));
  } else {
    // tau is fixed during a launch: rows in random order, a chunk g times the rows seen so far leaves ~ g k candidates per
    // query; keep that 2.5 times below the free slots (an adversarial order overflows and re-runs in safe mode)
    double g_cap = 16.0, margin = 2.5;
    uint32_t first = VS_FIRST_TILES;
    double growth = std::max(1.5, std::min(g_cap, (double)(VS_CAP - k) / (margin * k)));
    uint32_t done = std::min<uint32_t>(T, first);
    chunks.push_back(done);
    while (done < T) {
      uint32_t c = (uint32_t)std::min<double>((double)(T - done), std::max(1.0, done * growth));
      chunks.push_back(c);
      done += c;
    }
  }

  for (uint32_t g0 = 0; g0 < nq; g0 += SS_VEC_BATCH) {
    const uint32_t nb = std::min<uint32_t>(SS_VEC_BATCH, nq - g0);
    if (i8) {
      ssi_vec8_qprep(s, (const int8_t*)d_queries + (size_t)g0 * s->dim, nb, st);
      if (euclid) { rc = ssi_vec8_qaux(s, (const int8_t*)d_queries + (size_t)g0 * s->dim, nb, d_qnorm ? d_qnorm + g0 : nullptr, st); if (rc) return rc; }
    } else vec_qprep_kernel<<<nch, 512, 0, st>>>((const float*)d_queries + (size_t)g0 * s->dim, nb, s->dim, s->d_Qf, euclid ? 1 : 0);
    vec_init_kernel<<<1, 64, 0, st>>>(vst, tau_init);
    VAnn ann{};
    if (ann_clusters) {  // medoid scores -> per-query cluster selection -> the batch's tile list
      rc = ssi_vec_ann_prepare(s, nb, d_qscale ? d_qscale + g0 : nullptr, ann_mode, &ann,
                               (d_out_clusters && !observed) ? d_out_clusters + g0 : nullptr, st, d_qnorm ? d_qnorm + g0 : nullptr);
      if (rc) return rc;
    } else if (ann_mode && d_out_clusters && !observed) {
      SS_HIP(hipMemsetAsync(d_out_clusters + g0, 0, nb * sizeof(uint32_t), st));
    }
    if (observed) { rc = ssi_vec_observed_report(s, nb, ann_clusters, d_out_clusters + (size_t)g0 * ocw, st); if (rc) return rc; }
    if (ann_mode && ann_mode->field_mask) {
      ann.row_field = s->d_row_field;
      ann.field_mask = ann_mode->field_mask;
    }
    uint32_t tile0 = 0;
    // the sparse ANN instantiation (vec_ann_sparse_kernel): a batch of more than 32 queries under Nprobe whose tiles concern few
    // of its queries each -- expected interested queries per tile = nb x n_probe / (clusters per level) <= 8
    uint32_t sparse_nv = 0;
    {
      if (ann_clusters && !i8 && !euclid && nb > 32 && ann_mode->n_probe != 0 && s->dim == s->dim_pad && s->dim % 256u == 0 &&
          s->dim <= 1024u && s->vec_n_clusters && (double)nb * ann_mode->n_probe * s->vec_n_levels <= 8.0 * s->vec_n_clusters)
        sparse_nv = s->dim / 256u;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ssi_prof_begin(s, 1, st, &e0, &e1);
    for (uint32_t c : chunks) {
      // workgroups per launch: 2 per CU at a time (72 KB of LDS each).  A grid of exactly 512 would be PERSISTENT -- every workgroup lives as
      // long as the chunk, 8 ms for the last one of a 10 M-row pass, and nothing else gets a CU's LDS meanwhile: a lexical launch on the
      // shard's high-priority stream (a hybrid caller's first half) then waits for the pass to end (profiles/r6_hybrid_hist.log: 7.6 ms of
      // device wait per lexical batch under T = 256 hybrid callers).  VS_GRID_MULT x as many workgroups, each walking 1 / VS_GRID_MULT of
      // the tiles: a workgroup retires every few hundred microseconds and the dispatcher hands its CU to the higher-priority queue first.
      uint32_t grid = std::min<uint32_t>(c, 512u * VS_GRID_MULT);
      if (i8) ssi_vec8_launch_scan(s, tile0, c, d_qscale ? d_qscale + g0 : nullptr, ann_mode ? &ann : nullptr, st);
      else if (ann_mode && sparse_nv) {
        const uint32_t sg = std::min<uint32_t>(c, 1024);
        const float* qraw = (const float*)d_queries + (size_t)g0 * s->dim;
#define SS_SPARSE(NV_) vec_ann_sparse_kernel<NV_><<<sg, 256, 0, st>>>(s->d_X, s->dim_pad, (unsigned long long)s->n_rows, qraw, nb, tile0, c, vst, cand, ann)
        if (sparse_nv == 1) SS_SPARSE(1); else if (sparse_nv == 2) SS_SPARSE(2); else if (sparse_nv == 3) SS_SPARSE(3); else SS_SPARSE(4);
#undef SS_SPARSE
      } else if (ann_mode) {
        if (nb > 32)
          vec_scan_kernel<true, true><<<grid, VS_WAVES * 64, VS_LDS, st>>>(s->d_X, s->dim_pad, (unsigned long long)s->n_rows,
                                                                           s->d_Qf, nch, tile0, c, vst, cand, ann);
        else
          vec_scan_kernel<false, true><<<grid, VS_WAVES * 64, VS_LDS, st>>>(s->d_X, s->dim_pad, (unsigned long long)s->n_rows,
                                                                            s->d_Qf, nch, tile0, c, vst, cand, ann);
      } else if (nb > 32)
        vec_scan_kernel<true, false><<<grid, VS_WAVES * 64, VS_LDS, st>>>(s->d_X, s->dim_pad, (unsigned long long)s->n_rows,
                                                                          s->d_Qf, nch, tile0, c, vst, cand, ann);
      else
        vec_scan_kernel<false, false><<<grid, VS_WAVES * 64, VS_LDS, st>>>(s->d_X, s->dim_pad, (unsigned long long)s->n_rows,
                                                                           s->d_Qf, nch, tile0, c, vst, cand, ann);
      vec_refine_kernel<<<SS_VEC_BATCH, VR_THREADS, VS_CAP * (sizeof(unsigned long long) + sizeof(uint32_t)), st>>>(
          vst, cand, k, s->vec_multi_record ? s->d_row_doc : nullptr, s->d_row_doc, s->n_deleted ? s->d_deleted : nullptr,
          (uint32_t)s->deleted_words, s->n_deleted ? s->vec_del_stride : 0u, nb);
      tile0 += c;
    }
    ssi_prof_end(s, 1, st, e0, e1);
    if (euclid && !i8)
      vec_rescore_euclid_kernel<<<nb, 256, 0, st>>>(vst, cand, s->d_X, s->dim, s->dim_pad, (const float*)d_queries + (size_t)g0 * s->dim, nb, k, thr);
    vec_final_kernel<<<nb, 256, 0, st>>>(vst, cand, s->d_row_doc, nb, k, d_out_doc + (size_t)g0 * k,
                                         d_out_score + (size_t)g0 * k, d_out_count + g0,
                                         (unsigned long long*)d_out_total + g0);
  }
  SS_HIP(hipGetLastError());
  return SS_OK;
}
