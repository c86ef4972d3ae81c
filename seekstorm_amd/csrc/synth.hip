// HBM image builders: (a) from host arrays (ss_bm25_upload), (b) device-side synthetic corpora whose
// counter-based generator is bit-identical to the CPU oracle's (oracle/ss_oracle.c so_lex_* / so_vec_gen), so a
// 10M-doc / 10M x 768 corpus never crosses PCIe and the oracle can still regenerate any slice of it.
#include "ss_common.h"
#include "ss_threads.h"
#include "bm25_build.h"
#include "sparse_levels.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

typedef unsigned long long u64;

__host__ __device__ inline u64 ss_splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  u64 z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline u64 ss_h(u64 seed, u64 a, u64 b) {
  return ss_splitmix64(seed ^ (a * 0x9E3779B97F4A7C15ull) ^ (b * 0xC2B2AE3D27D4EB4Full));
}

// ---------------------------------------------------------------- vectors
// one thread per row: uniform(-1,1) from the hash, then normalize_f32 semantics (vector_similarity.rs:70-74):
// sequential sum of squares (unfused), factor = 1/sqrt(sum), multiply.
// gs / go: the shard holds rows go, go + gs, go + 2 gs, ... of ONE generator stream (doc g -> shard g % S, local id g / S,
// index.rs:5284): local row r is global row r * gs + go (ss_synth_set_partition; 1 / 0 = the whole stream)
__global__ void vec_synth_kernel(float* __restrict__ X, u64 seed, u64 n_rows, uint32_t dim, uint32_t dim_pad, u64 gs, u64 go) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  float* row = X + r * dim_pad;
  r = r * gs + go;
  float s = 0.f;
  for (uint32_t c = 0; c < dim; c++) {
    int iv = (int)(uint32_t)(ss_h(seed, r, c) >> 32);
    float v = ss_fmul((float)iv, 4.656612873077392578125e-10f);
    s = ss_fadd(s, ss_fmul(v, v));
  }
  float f = __fdiv_rn(1.0f, __fsqrt_rn(s));
  for (uint32_t c = 0; c < dim; c++) {
    int iv = (int)(uint32_t)(ss_h(seed, r, c) >> 32);
    row[c] = ss_fmul(ss_fmul((float)iv, 4.656612873077392578125e-10f), f);
  }
}

// SURVEY 8d's own generator for the rows -- Box-Muller N(0, 1) components, then normalize_f32 -- as a device-only EXPERIMENT
// (SS_VEC_SYNTH_BOXMULLER=1): f32 log / cos differ between the host's libm and the device, so no oracle can regenerate these rows
// and no parity test uses them; tools/probes/generators_ab.sh shows that the scan's throughput does not depend on the choice.
__global__ void vec_synth_boxmuller_kernel(float* __restrict__ X, u64 seed, u64 n_rows, uint32_t dim, uint32_t dim_pad, u64 gs, u64 go) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  float* row = X + r * dim_pad;
  r = r * gs + go;
  auto comp = [&](uint32_t c) {
    const u64 h = ss_h(seed, r, c);
    const float u1 = ((float)(uint32_t)(h >> 32) + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)(uint32_t)h * 2.3283064365386963e-10f;
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
  };
  float s = 0.f;
  for (uint32_t c = 0; c < dim; c++) { const float v = comp(c); s += v * v; }
  const float f = 1.0f / sqrtf(s);
  for (uint32_t c = 0; c < dim; c++) row[c] = comp(c) * f;
}

int ssi_vec_synth(ss_shard* s, uint64_t seed, hipStream_t st) {
  u64 n = s->n_rows;
  uint32_t grid = (uint32_t)((n + 255) / 256);
  static const bool boxmuller = [] { const char* e = getenv("SS_VEC_SYNTH_BOXMULLER"); return e && atoi(e) != 0; }();
  if (boxmuller) vec_synth_boxmuller_kernel<<<grid, 256, 0, st>>>(s->d_X, seed, n, s->dim, s->dim_pad, s->synth_stride, s->synth_offset);
  else
  vec_synth_kernel<<<grid, 256, 0, st>>>(s->d_X, seed, n, s->dim, s->dim_pad, s->synth_stride, s->synth_offset);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// ---------------------------------------------------------------- BM25 image from host arrays
// comp[0..255] = bm25_component_cache (commit.rs:321-325): K * (1 - b + b * dl / avgdl) per SmallFloat length byte.
// The weight of a posting, tf * (K + 1) / (tf + comp[len]) (add_result.rs:1445-1447 without the idf factor), is computed
// from it at image-build time and stored IN the posting as its 19-bit code (bm_wcode, ss_common.h).
static void fill_comp(float avgdl, float* comp) {
  for (int i = 0; i < 256; i++) {
    float q = (float)ss_byte4_to_int((uint32_t)i) / avgdl;
    comp[i] = 1.2f * (1.0f - 0.75f + 0.75f * q);
  }
}
constexpr int SS_COMP_N = 256;
// the device generator's postings have tf <= 25 (1 + a geometric draw from 32 hash bits, lex_geom06): their codes come from a host-computed table
// [33][256] so that no float arithmetic of the weight runs on the device (the host's is the reference's, bit for bit)
constexpr int SS_SYNTH_TF_MAX = 32;
// lists that can meet the all_terms_frequent condition (posting_count / indexed_doc_count >= 0.5 in f32, intersection.rs:
// 198-209; one indexed field) carry (tf < 10) in the last bit of every weight code
static inline bool bm_list_flagged(const ss_shard* s, uint64_t df) {
  return s->bm_n_fields == 1 && (float)df / (float)s->bm_n_docs >= 0.5f;
}
// several indexed fields: the MERGED list of such a term (its df is the term's posting_count) carries the multi-field form of the
// rule in the same bit -- decode_positions_multiterm_multifield returns "counted, not ranked" for an embedded pointer and for a
// record whose FIRST field has fewer than 10 positions (add_result.rs:1595-1607); a posting whose first field holds >= 10 positions
// is never embedded (embedding stops at 4 positions, index_posting.rs:437), so the rule is "tf of the doc's lowest field < 10"
static inline bool bm_merged_list_flagged(const ss_shard* s, uint64_t df) {
  return s->bm_merged && (float)df / (float)s->bm_n_docs >= 0.5f;
}
static inline uint32_t bm_code_of(uint32_t tf, float comp_len, bool flagged) {
  uint32_t c = bm_wcode(bm_weight_exact(tf, comp_len));
  if (flagged) c = (c & ~1u) | (tf < 10u ? 1u : 0u);
  return c;
}

// an image array: from the owner's block pool when the image is an incremental one (ss_block_pool), else a plain allocation
template <class T>
static hipError_t img_malloc(ss_shard* s, T** p, size_t bytes) {
  if (!s->pool) return hipMalloc(p, bytes);
  const int rc = s->pool->alloc((void**)p, bytes);
  return rc == 0 ? hipSuccess : rc == -2 ? hipErrorOutOfMemory : hipErrorUnknown;
}
// segments are padded to 16 bytes; the image ends with 1 KB of NULL postings so that a whole-wave load of the last
// unit never leaves the allocation
static int alloc_post(ss_shard* s, u64 n_units) {
  s->bm_n_post_pad = n_units * 4;
  SS_HIP(img_malloc(s, &s->d_post, (s->bm_n_post_pad + 256) * sizeof(uint32_t)));
  SS_HIP(hipMemset(s->d_post + s->bm_n_post_pad, 0, 256 * sizeof(uint32_t)));
  return SS_OK;
}

// Probe index + per-term weight maxima.  Skipped (d_probe stays null: searches then always scan exhaustively) when it
// would not fit beside the postings.
constexpr int BM_GROUPS = BM_SUB / 64;
static int alloc_probe(ss_shard* s, hipStream_t st) {
  // Everything the builders do not overwrite is cleared ON THE BUILD STREAM: the shard stream is non-blocking, so a
  // null-stream hipMemset is not ordered against the generator kernels that follow and could wipe what they wrote.
  const uint32_t nt = s->bm_n_terms;
  const size_t row_elems = (size_t)s->bm_n_sub * BM_GROUPS;
  SS_HIP(img_malloc(s, &s->d_umax, ((size_t)nt + 1) * sizeof(float)));
  SS_HIP(hipMemsetAsync(s->d_umax, 0, ((size_t)nt + 1) * sizeof(float), st));
  SS_HIP(img_malloc(s, &s->d_submax, ((size_t)nt + 1) * s->bm_n_sub * sizeof(float)));
  SS_HIP(hipMemsetAsync(s->d_submax, 0, ((size_t)nt + 1) * s->bm_n_sub * sizeof(float), st));
  size_t free_b = 0, total_b = 0;
  SS_HIP(hipMemGetInfo(&free_b, &total_b));
  const size_t budget = s->probe_budget ? (size_t)s->probe_budget : free_b / 2;
  const size_t row_bytes = row_elems * (sizeof(uint2) + sizeof(uint32_t));
  size_t max_rows = row_bytes ? budget / row_bytes : 0;
  max_rows = max_rows ? max_rows - 1 : 0;  // the all-zero row
  s->h_probe_row.assign((size_t)nt + 1, BM_NO_PROBE_ROW);
  s->bm_probe_rows = 0;
  if (max_rows == 0) return SS_OK;  // no probe index at all: every search scans
  // rows for the longest lists first (s->h_df is known at this point in both builders)
  std::vector<uint32_t> order(nt);
  for (uint32_t t = 0; t < nt; t++) order[t] = t;
  if (max_rows < nt)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return s->h_df[x] > s->h_df[y]; });
  const uint32_t rows = (uint32_t)std::min<size_t>(max_rows, nt);
  // rationed rows: a quarter of them (at most 8192) stay unassigned as the pool of rows built on demand
  s->probe_pool_rows = (max_rows < nt && rows >= 32) ? std::min<uint32_t>(rows / 4, 8192u) : 0u;
  s->probe_pool_begin = rows - s->probe_pool_rows;
  s->pool_list.assign(s->probe_pool_rows, BM_NO_PROBE_ROW);
  s->pool_tick.assign(s->probe_pool_rows, 0);
  s->pool_clock = 0;
  for (uint32_t i = 0; i < s->probe_pool_begin; i++) s->h_probe_row[order[i]] = i;
  s->h_probe_row[nt] = rows;  // absent terms of a short query: the zero row
  for (uint32_t t = 0; t < nt; t++)  // ... which also serves every empty list
    if (s->h_probe_row[t] == BM_NO_PROBE_ROW && s->h_df[t] == 0) s->h_probe_row[t] = rows;
  s->bm_probe_rows = rows;
  SS_HIP(img_malloc(s, &s->d_probe, ((size_t)rows + 1) * row_elems * sizeof(uint2)));
  SS_HIP(img_malloc(s, &s->d_probe_z, ((size_t)rows + 1) * row_elems * sizeof(uint32_t)));
  SS_HIP(img_malloc(s, &s->d_probe_row, ((size_t)nt + 1) * sizeof(uint32_t)));
  SS_HIP(hipMemsetAsync(s->d_probe + (size_t)rows * row_elems, 0, row_elems * sizeof(uint2), st));
  SS_HIP(hipMemsetAsync(s->d_probe_z + (size_t)rows * row_elems, 0, row_elems * sizeof(uint32_t), st));
  SS_HIP(hipMemcpyAsync(s->d_probe_row, s->h_probe_row.data(), ((size_t)nt + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  SS_HIP(hipStreamSynchronize(st));  // h_probe_row may be re-assigned by a later build while the copy is in flight
  return SS_OK;
}
int ssi_bm25_build_from_host(ss_shard* s, const uint8_t* doclen, const uint64_t* offs, const uint32_t* docs,
                             const uint16_t* tfs, uint64_t positions_sum, const float* merged_boost, float merged_scale) {
  (void)merged_scale;
  if (merged_boost || s->bm_merged) return SS_EINVAL;  // images with merged lists: ssi_bm25_build_from_host_merged
  return ssi_bm25_build_from_host_merged(s, doclen, offs, docs, tfs, positions_sum, nullptr, nullptr);
}

int ssi_bm25_build_from_host_merged(ss_shard* s, const uint8_t* doclen, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                                    uint64_t positions_sum, const float* merged_boost, float* merged_scale) {
  const uint32_t nt = s->bm_n_terms, ns = s->bm_n_sub;
  const uint32_t L = s->bm_n_fields, RF = bm_real_fields(s);  // lists per term, indexed fields
  if (s->bm_merged && (!merged_boost || !merged_scale)) return SS_EINVAL;
  static const bool trace = getenv("SS_LOAD_TRACE") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[load]       %s %.0f ms\n", what, std::chrono::duration<double, std::milli>(n - t_prev).count());
    t_prev = n;
  };
  u64 psum = 0;
  if (positions_sum) psum = positions_sum;
  else
    for (u64 d = 0; d < s->bm_n_docs * RF; d++) psum += ss_byte4_to_int(doclen[d]);  // all fields (index.rs:5848)
  s->bm_avgdl = (float)psum / (float)s->bm_n_docs;  // commit.rs:318-319
  float comp[SS_COMP_N];
  fill_comp(s->bm_avgdl, comp);
  // merged lists: their weights first -- sum_f boost_f * tf (K + 1) / (tf + comp[len_f]) over the fields that hold the doc,
  // fields ascending -- so that the scale can be chosen from what the corpus really holds: the smallest power of two that
  // brings the largest weight under the code's 4.0.  The code spans 2^-14 .. 2^2; a corpus whose merged weights do not fit
  // (boosts very far apart) gets no merged lists rather than clamped scores.
  std::vector<float> mw;
  std::vector<uint8_t> mw_lt10;  // tf of the doc's lowest field < 10 (bm_merged_list_flagged)
  std::vector<u64> mw_base;
  float mscale = 1.0f;
  if (s->bm_merged) {
    mw_base.assign(nt / L + 1, 0);
    for (uint32_t t = L - 1; t < nt; t += L) mw_base[t / L + 1] = mw_base[t / L] + (offs[t + 1] - offs[t]);
    mw.resize(mw_base[nt / L]);
    mw_lt10.assign(mw_base[nt / L], 0);
    float wmin = 3.0e38f, wmax = 0.f;
    {  // (terms in parallel; the extremes per worker, then combined)
      const unsigned nw = ss_loader_threads();
      std::vector<float> wmins(nw + 1, 3.0e38f), wmaxs(nw + 1, 0.f);
      std::atomic<int> bad{0};
      ss_parallel_for(nt / L, 16, [&](size_t ia, size_t ib, unsigned wk) {
        float lo = 3.0e38f, hi = 0.f;
        for (size_t ti = ia; ti < ib; ti++) {
          const uint32_t t = (uint32_t)(ti * L + (L - 1));
          u64 fcur[8];
          for (uint32_t f = 0; f < RF; f++) fcur[f] = offs[t - RF + f];
          for (u64 j = offs[t]; j < offs[t + 1]; j++) {
            if (docs[j] >= s->bm_n_docs) { bad.store(1); return; }
            float w = 0.f;
            bool any = false;
            for (uint32_t f = 0; f < RF; f++) {
              const uint32_t vf = t - RF + f;
              while (fcur[f] < offs[vf + 1] && docs[fcur[f]] < docs[j]) fcur[f]++;
              if (fcur[f] < offs[vf + 1] && docs[fcur[f]] == docs[j]) {
                if (tfs[fcur[f]] == 0) { bad.store(1); return; }
                const volatile float part = merged_boost[f] * bm_weight_exact(tfs[fcur[f]], comp[doclen[(size_t)f * s->bm_n_docs + docs[j]]]);
                w = w + part;
                if (!any) mw_lt10[mw_base[t / L] + (j - offs[t])] = tfs[fcur[f]] < 10u ? 1 : 0;  // the lowest field that holds the doc
                any = true;
              }
            }
            if (!any || !(w > 0.f)) { bad.store(1); return; }  // a doc of the merged list that no field list holds
            mw[mw_base[t / L] + (j - offs[t])] = w;
            lo = std::min(lo, w);
            hi = std::max(hi, w);
          }
        }
        const size_t slot = std::min<size_t>(wk, nw);
        wmins[slot] = std::min(wmins[slot], lo);
        wmaxs[slot] = std::max(wmaxs[slot], hi);
      });
      if (bad.load()) return SS_EINVAL;
      for (float x : wmins) wmin = std::min(wmin, x);
      for (float x : wmaxs) wmax = std::max(wmax, x);
    }
    if (!mw.empty()) {
      int e = 0;
      (void)std::frexp(wmax / 3.99f, &e);  // wmax / 3.99 = m * 2^e, m in [0.5, 1)  ->  2^e > wmax / 3.99
      mscale = std::ldexp(1.0f, e);
      if (wmin / mscale < 6.2e-5f) return SS_MERGED_RANGE;  // below 2^-14 + a margin: the code would clamp it
    }
    *merged_scale = mscale;
  }
  lap("merged weights");
  // pass 1: validate, segment boundaries in 16-byte units (4 postings, zero padded) relative to the term base
  // (the per-term loops below run on the loader's worker threads: a term's rows, postings and probe row are its own)
  std::vector<uint32_t> sub((size_t)nt * (ns + 1));
  std::vector<u64> tbase((size_t)nt + 1);
  s->h_df.assign(nt, 0);
  for (uint32_t t = 0; t < nt; t++)
    if (offs[t + 1] < offs[t]) return SS_EINVAL;
  std::atomic<int> fail{SS_OK};
  ss_parallel_for(nt, 64, [&](size_t ta, size_t tb, unsigned) {
    for (size_t t = ta; t < tb; t++) {
      s->h_df[t] = offs[t + 1] - offs[t];
      uint32_t* row = sub.data() + (size_t)t * (ns + 1);
      u64 i = offs[t], u = 0;
      for (uint32_t sb = 0; sb < ns; sb++) {
        row[sb] = (uint32_t)u;
        const u64 lim = ((u64)sb + 1) << BM_SUB_LOG2;
        const u64 i0 = i;
        while (i < offs[t + 1] && docs[i] < lim) i++;
        u += (i - i0 + 3) >> 2;
        if (u >= (1ull << 28)) { fail.store(SS_ENOTSUP); return; }  // a term's segment offsets must stay below 4 GB
      }
      row[ns] = (uint32_t)u;
      if (i != offs[t + 1]) { fail.store(SS_EINVAL); return; }  // doc id >= n_docs
    }
  });
  if (fail.load()) return fail.load();
  u64 units = 0;
  for (uint32_t t = 0; t < nt; t++) { tbase[t] = units; units += sub[(size_t)t * (ns + 1) + ns]; }
  tbase[nt] = units;
  std::vector<uint32_t> post(units ? units * 4 : 4, 0u);
  std::vector<float> umax((size_t)nt + 1, 0.f);
  std::vector<float> submax((size_t)(nt + 1) * ns, 0.f);
  ss_parallel_for(nt, 16, [&](size_t ta, size_t tb, unsigned) {
   for (size_t t = ta; t < tb; t++) {
    const bool flagged = bm_list_flagged(s, s->h_df[t]);
    const bool merged = s->bm_merged && t % L == L - 1;  // the term's merged list: weights from its field lists t - RF .. t - 1
    const uint8_t* dl = doclen + (size_t)(merged ? 0 : t % L) * s->bm_n_docs;  // the list's field (virtual term = term * L + field)
    const uint32_t* row = sub.data() + (size_t)t * (ns + 1);
    u64 j = offs[t];
    for (uint32_t sb = 0; sb < ns; sb++) {
      const u64 lim = ((u64)sb + 1) << BM_SUB_LOG2;
      u64 w = (tbase[t] + row[sb]) * 4;
      for (; j < offs[t + 1] && docs[j] < lim; j++, w++) {
        if (docs[j] >= s->bm_n_docs || (j > offs[t] && docs[j] <= docs[j - 1]) || tfs[j] == 0) { fail.store(SS_EINVAL); return; }
        uint32_t code;
        if (merged) {
          code = bm_wcode(mw[mw_base[t / L] + (j - offs[t])] / mscale);
          if (bm_merged_list_flagged(s, s->h_df[t])) code = (code & ~1u) | mw_lt10[mw_base[t / L] + (j - offs[t])];
        } else {
          code = bm_code_of(tfs[j], comp[dl[docs[j]]], flagged);
        }
        post[w] = bm_pack(docs[j] & (BM_SUB - 1), code);
        umax[t] = std::max(umax[t], bm_wdecode(code));  // the bound of the pruned kernel: over the weights as the kernels see them
        submax[(size_t)t * ns + sb] = std::max(submax[(size_t)t * ns + sb], bm_wdecode(code));
      }
    }
   }
  });
  if (fail.load()) return fail.load();
  lap("segments + packing");
  s->bm_n_post = offs[nt];
  const size_t rows = (size_t)nt * (ns + 1);
  int rc = alloc_post(s, units);
  if (rc) return rc;
  SS_HIP(hipMalloc(&s->d_term_base, ((size_t)nt + 1) * sizeof(u64)));
  SS_HIP(hipMalloc(&s->d_sub_off, (rows + ns + 1) * sizeof(uint32_t)));  // + one all-zero row (absent terms)
  SS_HIP(hipMemset(s->d_sub_off + rows, 0, ((size_t)ns + 1) * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_comp, SS_COMP_N * sizeof(float)));
  if (units) SS_HIP(hipMemcpy(s->d_post, post.data(), units * 4 * sizeof(uint32_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_term_base, tbase.data(), ((size_t)nt + 1) * sizeof(u64), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_sub_off, sub.data(), sub.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_comp, comp, sizeof(comp), hipMemcpyHostToDevice));
  if (L == 1 || s->bm_merged) {  // the length bytes ([indexed fields][n_docs]) stay on the device for ss_bm25_append_sparse[_fields]
    SS_HIP(hipMalloc(&s->d_doclen, (size_t)RF * s->bm_n_docs));
    SS_HIP(hipMemcpy(s->d_doclen, doclen, (size_t)RF * s->bm_n_docs, hipMemcpyHostToDevice));
  }
  lap("image arrays to the device");
  rc = alloc_probe(s, s->stream);
  if (rc) return rc;
  SS_HIP(hipStreamSynchronize(s->stream));
  // the fixed probe rows: built on the device from the postings just uploaded (ssi_bm25_fill_fixed_probe_rows)
  rc = ssi_bm25_fill_fixed_probe_rows(s, s->stream);
  if (rc) return rc;
  lap("probe rows");
  SS_HIP(hipMemcpy(s->d_umax, umax.data(), umax.size() * sizeof(float), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_submax, submax.data(), submax.size() * sizeof(float), hipMemcpyHostToDevice));
  // Are block maxima worth a pass per search?  Only where they vary over the doc ids: for the longest lists, the mean over
  // 32 equal partitions of (largest weight inside the partition) / (largest weight of the list).  A corpus whose weights are
  // spread evenly (every synthetic one here) gives ~1; docs ordered by source / time / length give clearly less.
  {
    std::vector<uint32_t> order(nt);
    for (uint32_t t = 0; t < nt; t++) order[t] = t;
    const uint32_t top = std::min<uint32_t>(nt, 64);
    std::partial_sort(order.begin(), order.begin() + top, order.end(), [&](uint32_t a, uint32_t b) { return s->h_df[a] > s->h_df[b]; });
    double worst = 1.0;
    const uint32_t parts = std::min<uint32_t>(32, ns);
    for (uint32_t i = 0; i < top; i++) {
      const uint32_t t = order[i];
      if (umax[t] <= 0.f || s->h_df[t] < 64) continue;
      double acc = 0.0;
      for (uint32_t pi = 0; pi < parts; pi++) {
        float m = 0.f;
        for (uint32_t sb = (uint32_t)((u64)ns * pi / parts); sb < (uint32_t)((u64)ns * (pi + 1) / parts); sb++) m = std::max(m, submax[(size_t)t * ns + sb]);
        acc += m / umax[t];
      }
      worst = std::min(worst, acc / parts);
    }
    s->bm_partmax = worst < 0.9;
  }

  return SS_OK;
}

// Are block maxima worth a pass per search?  Only where they vary over the doc ids: for the 64 longest lists, the mean over 32 equal
// partitions of (largest weight inside the partition) / (largest weight of the list) -- ~1 on a corpus whose weights are spread evenly,
// clearly less where docs are ordered by source / time / length.  The host builder applies the rule to its own arrays; the device
// builders (synthetic corpora, incremental images) fetch what it needs: umax and the 64 rows of submax.
static int bm_decide_partmax_dev(ss_shard* img) {
  const uint32_t nt = img->bm_n_terms, ns = img->bm_n_sub;
  std::vector<uint32_t> order(nt);
  for (uint32_t t = 0; t < nt; t++) order[t] = t;
  const uint32_t top = std::min<uint32_t>(nt, 64);
  std::partial_sort(order.begin(), order.begin() + top, order.end(), [&](uint32_t a, uint32_t b) { return img->h_df[a] > img->h_df[b]; });
  std::vector<float> um(nt), row(ns);
  SS_HIP(hipMemcpy(um.data(), img->d_umax, (size_t)nt * sizeof(float), hipMemcpyDeviceToHost));
  double worst = 1.0;
  const uint32_t parts = std::min<uint32_t>(32, ns);
  for (uint32_t i = 0; i < top; i++) {
    const uint32_t t = order[i];
    if (um[t] <= 0.f || img->h_df[t] < 64) continue;
    SS_HIP(hipMemcpy(row.data(), img->d_submax + (size_t)t * ns, (size_t)ns * sizeof(float), hipMemcpyDeviceToHost));
    double acc = 0.0;
    for (uint32_t pi = 0; pi < parts; pi++) {
      float m = 0.f;
      for (uint32_t sb = (uint32_t)((u64)ns * pi / parts); sb < (uint32_t)((u64)ns * (pi + 1) / parts); sb++) m = std::max(m, row[sb]);
      acc += m / um[t];
    }
    worst = std::min(worst, acc / parts);
  }
  img->bm_partmax = worst < 0.9;
  return SS_OK;
}

// ---------------------------------------------------------------- image from the raw levels of an incremental shard, ON THE DEVICE
// What ssi_bm25_build_from_host does on the host, as three launches over (term, sub-block) pairs -- count, scan, fill -- so that a
// commit costs a rebuild at HBM speed instead of a host pass + PCIe: the raw postings are read once, the image written once.
__global__ void lex_scan_rows_kernel(uint32_t* __restrict__ sub, uint32_t n_sub, u64* __restrict__ term_tot);
__global__ void lex_scan_terms_kernel(const u64* __restrict__ tot, u64* __restrict__ base, uint32_t n_terms);
struct RawLevelDev {
  const unsigned long long* off; const uint32_t* doc; const uint16_t* tf; uint32_t n_terms, pad;
  const uint16_t* npos; const uint32_t* prel; const unsigned long long* tpos; const uint16_t* pos;  // positions (or null)
};
__device__ __forceinline__ void raw_segment(const RawLevelDev* __restrict__ levels, uint32_t level_shift, uint32_t t, uint32_t sb, u64* lo_out,
                                            u64* hi_out, const uint32_t** doc_out, const uint16_t** tf_out) {
  const RawLevelDev L = levels[sb >> level_shift];  // incremental images: a level = 65 536 docs = 16 sub-blocks; a one-shot upload: one "level"
  *doc_out = L.doc; *tf_out = L.tf;
  if (t >= L.n_terms) { *lo_out = 0; *hi_out = 0; return; }
  const u64 a = L.off[t], b = L.off[t + 1];
  const uint32_t d0 = sb << BM_SUB_LOG2, d1 = d0 + (uint32_t)BM_SUB;
  u64 lo = a, hi = b;
  while (lo < hi) { const u64 m = (lo + hi) >> 1; if (L.doc[m] < d0) lo = m + 1; else hi = m; }
  const u64 first = lo;
  hi = b;
  while (lo < hi) { const u64 m = (lo + hi) >> 1; if (L.doc[m] < d1) lo = m + 1; else hi = m; }
  *lo_out = first; *hi_out = lo;
}
// one thread per (term, sub-block): the segment's size in 16-byte units (shifted by one for the exclusive scan) and the term's df
__global__ void raw_count_kernel(const RawLevelDev* __restrict__ levels, uint32_t level_shift, uint32_t n_terms, uint32_t n_sub,
                                 uint32_t* __restrict__ sub, u64* __restrict__ df) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (u64)n_terms * n_sub) return;
  const uint32_t t = (uint32_t)(i / n_sub), sb = (uint32_t)(i % n_sub);
  u64 lo, hi;
  const uint32_t* dp; const uint16_t* tp;
  raw_segment(levels, level_shift, t, sb, &lo, &hi, &dp, &tp);
  const uint32_t n = (uint32_t)(hi - lo);
  sub[(size_t)t * (n_sub + 1) + sb + 1] = (n + 3u) >> 2;
  if (n) atomicAdd(&df[t], (u64)n);
}
// one wave per (term, sub-block): packs the segment's postings (weight code from tf and the doc's length byte, computed with the
// host builder's operations: t * (K + 1) / (t + comp), every step rounded to f32), writes the probe row's 64-doc bit records and
// ranks, the segment's and the list's largest weight, the NULL padding
__global__ void raw_fill_kernel(const RawLevelDev* __restrict__ levels, uint32_t level_shift, uint32_t n_terms, uint32_t n_sub,
                                const uint8_t* __restrict__ doclen,
                                const float* __restrict__ comp, float k1, const uint32_t* __restrict__ sub, const u64* __restrict__ term_base,
                                uint32_t* __restrict__ post, uint2* __restrict__ probe, uint32_t* __restrict__ probe_z,
                                const uint32_t* __restrict__ probe_row, uint32_t* __restrict__ umax_bits, float* __restrict__ submax,
                                const uint8_t* __restrict__ flagged) {
  __shared__ unsigned long long masks[4][BM_SUB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u64 gw = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (gw >= (u64)n_terms * n_sub) return;
  const uint32_t t = (uint32_t)(gw / n_sub), sb = (uint32_t)(gw % n_sub);
  u64 lo, hi;
  const uint32_t* dp; const uint16_t* tp;
  raw_segment(levels, level_shift, t, sb, &lo, &hi, &dp, &tp);
  const uint32_t seg0 = sub[(size_t)t * (n_sub + 1) + sb] * 4u;
  const u64 base = term_base[t] * 4ull + seg0;
  const bool flg = flagged[t] != 0;
  const bool have_row = probe && probe_row[t] != BM_NO_PROBE_ROW;
  masks[w][lane] = 0ull;
  __builtin_amdgcn_wave_barrier();
  float wmax = 0.f;
  for (u64 i = lo + (u64)lane; i < hi; i += 64u) {
    const uint32_t d = dp[i], tf = tp[i];
    const float tt = (float)tf;
    const float wgt = __fdiv_rn(ss_fmul(tt, k1), ss_fadd(tt, comp[doclen[d]]));  // bm_weight_exact
    uint32_t code = bm_wcode(wgt);
    if (flg) code = (code & ~1u) | (tf < 10u ? 1u : 0u);
    const uint32_t din = d & (uint32_t)(BM_SUB - 1);
    post[base + (i - lo)] = bm_pack(din, code);
    wmax = fmaxf(wmax, bm_wdecode(code));
    if (have_row) atomicOr(&masks[w][din >> 6], 1ull << (din & 63u));
  }
  const uint32_t n = (uint32_t)(hi - lo);
  if ((uint32_t)lane < ((4u - (n & 3u)) & 3u)) post[base + n + lane] = 0u;  // NULL padding
  for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
  if (lane == 0) {
    submax[(size_t)t * n_sub + sb] = wmax;
    if (wmax > 0.f) atomicMax(&umax_bits[t], __float_as_uint(wmax));
  }
  if (have_row) {
    __builtin_amdgcn_wave_barrier();
    const unsigned long long m = masks[w][lane];
    uint32_t run = (uint32_t)__popcll(m);
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(run, o); if (lane >= o) run += v; }
    run -= (uint32_t)__popcll(m);
    const size_t gi = ((size_t)probe_row[t] * n_sub + sb) * (BM_SUB / 64) + (uint32_t)lane;
    probe[gi] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
    probe_z[gi] = seg0 + run;
  }
}

// ---- positions of an incremental image (phrase queries after commits): what ssi_bm25_upload_positions builds on the host, from the
// levels' own position pools.  d_pos = every term's positions, term after term, level after level (= doc order); d_pos_base = a term's
// first; d_pos_off = per image slot the END of its posting's positions relative to the term (padding slots repeat the running end).
// lstart [n_terms][n_levels]: positions of the term in the levels before level l (the two kernels below look it up per block / wave)
__global__ void raw_pos_total_kernel(const RawLevelDev* __restrict__ levels, uint32_t n_levels, uint32_t n_terms, u64* __restrict__ tot,
                                     u64* __restrict__ lstart) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  u64 c = 0;
  for (uint32_t l = 0; l < n_levels; l++) {
    lstart[(size_t)t * n_levels + l] = c;
    if (t < levels[l].n_terms) c += levels[l].tpos[t + 1] - levels[l].tpos[t];
  }
  tot[t] = c;
}
// one block per (term, level): the level's positions of the term into the term's pool, behind the earlier levels'
__global__ void raw_pos_copy_kernel(const RawLevelDev* __restrict__ levels, uint32_t n_levels, uint32_t n_terms, const u64* __restrict__ pos_base,
                                    const u64* __restrict__ lstart, uint16_t* __restrict__ pos) {
  const uint32_t t = blockIdx.x, l = blockIdx.y;
  if (t >= levels[l].n_terms) return;
  const u64 a = levels[l].tpos[t], b = levels[l].tpos[t + 1];
  if (a == b) return;
  const u64 at = pos_base[t] + lstart[(size_t)t * n_levels + l];
  const uint16_t* __restrict__ src = levels[l].pos + a;
  for (u64 i = threadIdx.x; i < b - a; i += blockDim.x) pos[at + i] = src[i];
}
// one wave per (term, sub-block): the END offsets of the segment's postings
__global__ void raw_pos_off_kernel(const RawLevelDev* __restrict__ levels, uint32_t n_levels, uint32_t level_shift, uint32_t n_terms, uint32_t n_sub,
                                   const uint32_t* __restrict__ sub, const u64* __restrict__ term_base, const u64* __restrict__ lstart,
                                   uint32_t* __restrict__ pos_off) {
  const int lane = threadIdx.x & 63;
  const u64 gw = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (gw >= (u64)n_terms * n_sub) return;
  const uint32_t t = (uint32_t)(gw / n_sub), sb = (uint32_t)(gw % n_sub);
  u64 lo, hi;
  const uint32_t* dp; const uint16_t* tp;
  raw_segment(levels, level_shift, t, sb, &lo, &hi, &dp, &tp);
  const uint32_t li = sb >> level_shift;
  const RawLevelDev L = levels[li];
  // positions of the term before this segment: the earlier levels' + the earlier postings of this level
  u64 start = lstart[(size_t)t * n_levels + li];
  const uint32_t seg0 = sub[(size_t)t * (n_sub + 1) + sb] * 4u, seg1 = sub[(size_t)t * (n_sub + 1) + sb + 1] * 4u;
  const u64 base = term_base[t] * 4ull + seg0;
  const uint32_t n = (uint32_t)(hi - lo);
  if (n) start += L.prel[lo];
  else if (t < L.n_terms) {  // an empty segment: the running end = the positions of the postings before its place
    const u64 a = L.off[t], b = L.off[t + 1];
    start += lo < b ? (u64)L.prel[lo] : (L.tpos[t + 1] - L.tpos[t]);
    (void)a;
  }
  u64 run = start;
  for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
    const uint32_t i = i0 + (uint32_t)lane;
    uint32_t c = i < n ? (uint32_t)L.npos[lo + i] : 0u, inc = c;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (i < n) pos_off[base + i] = (uint32_t)(run + inc);
    run += (u64)__shfl(inc, 63);
  }
  const uint32_t pad = seg1 - seg0 - n;  // NULL padding slots of the segment (0 .. 3)
  if ((uint32_t)lane < pad) pos_off[base + n + lane] = (uint32_t)run;
}

// one_shot: `levels` holds ONE set of arrays covering every doc (a whole-image upload builds through the same kernels: the host then
// only validates and copies), the image arrays are plain allocations and `img` may be the shard itself
int ssi_bm25_rebuild_from_raw(const ss_shard* s, const std::vector<ss_raw_level>& levels, uint32_t n_terms, const uint8_t* doclen, uint64_t n_doclen,
                              ss_shard* img, hipStream_t st, bool one_shot) {
  u64 nd = 0, psum = 0, npost = 0;
  for (const ss_raw_level& L : levels) { nd += L.n_docs; psum += L.psum; npost += L.n_post; }
  if (one_shot) nd = n_doclen;
  if (nd == 0 || n_terms == 0 || n_doclen != nd || (one_shot && levels.size() != 1)) return SS_EINVAL;
  const uint32_t level_shift = one_shot ? 31u : (uint32_t)(16 - BM_SUB_LOG2);
  const uint32_t nt = n_terms, ns = (uint32_t)((nd + BM_SUB - 1) >> BM_SUB_LOG2);
  img->device = s->device;
  img->probe_budget = s->probe_budget;
  img->pool = one_shot ? nullptr : const_cast<ss_block_pool*>(&s->blocks);  // (the caller serialises commits: nobody else touches the pool meanwhile)
  img->bm_n_docs = nd; img->bm_n_terms = nt; img->bm_n_sub = ns; img->bm_n_fields = 1; img->bm_merged = false;
  img->bm_n_post = npost;
  img->bm_avgdl = (float)psum / (float)nd;  // commit.rs:318-319
  static const bool trace = [] { const char* e = getenv("SS_APPEND_TRACE"); return e && atoi(e) != 0; }();
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  std::vector<RawLevelDev> lv(levels.size());
  for (size_t i = 0; i < levels.size(); i++)
    lv[i] = RawLevelDev{(const unsigned long long*)levels[i].d_off, levels[i].d_doc, levels[i].d_tf, levels[i].n_terms, 0u,
                        levels[i].d_npos, levels[i].d_prel, (const unsigned long long*)levels[i].d_tpos, levels[i].d_pos};
  const bool with_pos = !levels.empty() && levels[0].d_tpos != nullptr;
  u64 n_positions = 0;
  for (const ss_raw_level& L : levels) {
    if ((L.d_tpos != nullptr) != with_pos) return SS_EINVAL;  // positions for every level or for none
    n_positions += L.n_pos;
  }
  RawLevelDev* d_lv = nullptr;
  u64 *d_tot = nullptr, *d_df = nullptr;
  uint8_t* d_flg = nullptr;
  auto cleanup = [&]() { for (void* p : {(void*)d_lv, (void*)d_tot, (void*)d_df, (void*)d_flg}) if (p) (void)hipFree(p); };
#define SS_HIP_C(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return e_ == hipErrorOutOfMemory ? SS_ENOMEM : SS_EDEVICE; } } while (0)
  SS_HIP_C(hipMalloc(&d_lv, lv.size() * sizeof(RawLevelDev)));
  SS_HIP_C(hipMemcpyAsync(d_lv, lv.data(), lv.size() * sizeof(RawLevelDev), hipMemcpyHostToDevice, st));
  SS_HIP_C(hipMalloc(&d_tot, (size_t)nt * sizeof(u64)));
  SS_HIP_C(hipMalloc(&d_df, (size_t)nt * sizeof(u64)));
  SS_HIP_C(hipMemsetAsync(d_df, 0, (size_t)nt * sizeof(u64), st));
  SS_HIP_C(img_malloc(img, &img->d_doclen, nd));
  SS_HIP_C(hipMemcpyAsync(img->d_doclen, doclen, nd, hipMemcpyHostToDevice, st));
  const size_t rows = (size_t)nt * (ns + 1);
  SS_HIP_C(img_malloc(img, &img->d_sub_off, (rows + ns + 1) * sizeof(uint32_t)));  // + one all-zero row (absent terms)
  SS_HIP_C(hipMemsetAsync(img->d_sub_off + rows, 0, ((size_t)ns + 1) * sizeof(uint32_t), st));
  SS_HIP_C(img_malloc(img, &img->d_term_base, ((size_t)nt + 1) * sizeof(u64)));
  SS_HIP_C(img_malloc(img, &img->d_comp, SS_COMP_N * sizeof(float)));
  float comp[SS_COMP_N];
  fill_comp(img->bm_avgdl, comp);
  SS_HIP_C(hipMemcpyAsync(img->d_comp, comp, sizeof(comp), hipMemcpyHostToDevice, st));
  const u64 pairs = (u64)nt * ns;
  const auto t1 = now();
  raw_count_kernel<<<(uint32_t)((pairs + 255) / 256), 256, 0, st>>>(d_lv, level_shift, nt, ns, img->d_sub_off, d_df);
  lex_scan_rows_kernel<<<nt, 1024, 0, st>>>(img->d_sub_off, ns, d_tot);
  lex_scan_terms_kernel<<<1, 64, 0, st>>>(d_tot, (u64*)img->d_term_base, nt);
  SS_HIP_C(hipStreamSynchronize(st));
  std::vector<u64> tot(nt);
  img->h_df.resize(nt);
  SS_HIP_C(hipMemcpy(tot.data(), d_tot, (size_t)nt * sizeof(u64), hipMemcpyDeviceToHost));
  SS_HIP_C(hipMemcpy(img->h_df.data(), d_df, (size_t)nt * sizeof(u64), hipMemcpyDeviceToHost));
  u64 units = 0;
  for (uint32_t t = 0; t < nt; t++) {
    if (tot[t] >= (1ull << 28)) { cleanup(); return SS_ENOTSUP; }  // a term's segment offsets must stay below 4 GB
    units += tot[t];
  }
  const auto t2 = now();
  int rc = alloc_post(img, units);
  if (rc) { cleanup(); return rc; }
  rc = alloc_probe(img, st);
  if (rc) { cleanup(); return rc; }
  const auto t3 = now();
  std::vector<uint8_t> flg(nt);
  for (uint32_t t = 0; t < nt; t++) flg[t] = bm_list_flagged(img, img->h_df[t]) ? 1 : 0;
  SS_HIP_C(hipMalloc(&d_flg, nt));
  SS_HIP_C(hipMemcpyAsync(d_flg, flg.data(), nt, hipMemcpyHostToDevice, st));
  const volatile float k1 = 1.2f + 1.0f;  // (K + 1) as bm_weight_exact forms it
  raw_fill_kernel<<<(uint32_t)((pairs + 3) / 4), 256, 0, st>>>(d_lv, level_shift, nt, ns, img->d_doclen, img->d_comp, k1, img->d_sub_off,
                                                              (const u64*)img->d_term_base, img->d_post, img->d_probe, img->d_probe_z,
                                                              img->d_probe_row, (uint32_t*)img->d_umax, img->d_submax, d_flg);
  SS_HIP_C(hipGetLastError());
  if (with_pos) {  // the position arrays, from the levels' pools
    u64 *d_ptot = nullptr, *d_lstart = nullptr;
    SS_HIP_C(hipMalloc(&d_ptot, (size_t)nt * sizeof(u64)));
    if (hipMalloc(&d_lstart, (size_t)nt * lv.size() * sizeof(u64)) != hipSuccess) { (void)hipFree(d_ptot); cleanup(); return SS_ENOMEM; }
    auto drop = [&]() { (void)hipFree(d_ptot); (void)hipFree(d_lstart); };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = img_malloc(img, &img->d_pos_base, ((size_t)nt + 1) * sizeof(u64));
    if (e == hipSuccess) e = img_malloc(img, &img->d_pos, (size_t)std::max<u64>(n_positions, 1) * sizeof(uint16_t));
    if (e == hipSuccess) e = img_malloc(img, &img->d_pos_off, ((size_t)img->bm_n_post_pad + 8) * sizeof(uint32_t));
    if (e != hipSuccess) { drop(); cleanup(); return e == hipErrorOutOfMemory ? SS_ENOMEM : SS_EDEVICE; }
    raw_pos_total_kernel<<<(nt + 255) / 256, 256, 0, st>>>(d_lv, (uint32_t)lv.size(), nt, d_ptot, d_lstart);
    lex_scan_terms_kernel<<<1, 64, 0, st>>>(d_ptot, (u64*)img->d_pos_base, nt);
    raw_pos_copy_kernel<<<dim3(nt, (uint32_t)lv.size()), 256, 0, st>>>(d_lv, (uint32_t)lv.size(), nt, (const u64*)img->d_pos_base, d_lstart, img->d_pos);
    raw_pos_off_kernel<<<(uint32_t)((pairs + 3) / 4), 256, 0, st>>>(d_lv, (uint32_t)lv.size(), level_shift, nt, ns, img->d_sub_off,
                                                                    (const u64*)img->d_term_base, d_lstart, img->d_pos_off);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    drop();
    if (e != hipSuccess) { cleanup(); return SS_EDEVICE; }
  }
  SS_HIP_C(hipStreamSynchronize(st));
  const auto t4 = now();
  // block maxima worth a pass per search?  (the host builder's rule, on the 64 longest lists)
  {
    const int rcp = bm_decide_partmax_dev(img);
    if (rcp) { cleanup(); return rcp; }
  }
  if (trace)
    fprintf(stderr, "[append] docs %llu postings %llu: setup+copies %.2f ms, count+scan %.2f, alloc post+probe %.2f, fill %.2f, maxima rule %.2f\n",
            (unsigned long long)nd, (unsigned long long)npost, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, now()));
#undef SS_HIP_C
  cleanup();
  return SS_OK;
}

// ---------------------------------------------------------------- sparse tier: append lists of rare terms to the image
// (bm25_sparse.hip).  The weights are computed here, on the host, by the same routine and component cache as every dense
// posting's (bm_code_of): the length bytes come back from the device once per call.
// the packed postings of n_lists new sparse lists (list i: base[i] .. base[i + 1] of `packed`) behind the ones the tier holds
// counts / pos (optional): positions per new posting and their concatenation (elem = 2: u16, one indexed field; 4: u32 field-tagged)
static int sparse_install(ss_shard* s, uint32_t n_lists, const std::vector<u64>& lbase, const NoInitVec<u64>& packed,
                          const NoInitVec<uint32_t>* counts = nullptr, const void* pos = nullptr, u64 n_pos = 0, size_t elem = 0);

int ssi_bm25_append_sparse(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                           const uint16_t* positions, uint64_t n_positions, const uint16_t* npos) {
  if (!s->d_post || !s->d_doclen) return SS_ESTATE;
  if (s->bm_n_fields != 1 || s->bm_merged) return SS_ENOTSUP;  // several indexed fields: ssi_bm25_append_sparse_fields
  if ((uint64_t)s->sp_n + n_lists > 0x7FFFFFFFull - s->bm_n_terms) return SS_ENOTSUP;
  const u64 n_new = offs[n_lists] - offs[0];
  std::vector<uint8_t> dl(s->bm_n_docs);
  SS_HIP(hipMemcpy(dl.data(), s->d_doclen, s->bm_n_docs, hipMemcpyDeviceToHost));
  float comp[SS_COMP_N];
  fill_comp(s->bm_avgdl, comp);
  NoInitVec<u64> packed(n_new ? n_new : 1);  // (not value-initialised: the workers below touch their own ranges)
  for (uint32_t i = 0; i < n_lists; i++)
    if (offs[i + 1] < offs[i]) return SS_EINVAL;
  std::atomic<int> fail{SS_OK};
  ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++)
      for (u64 j = offs[i]; j < offs[i + 1]; j++) {
        if (docs[j] >= s->bm_n_docs || tfs[j] == 0 || (j > offs[i] && docs[j] <= docs[j - 1])) { fail.store(SS_EINVAL); return; }
        packed[j - offs[0]] = ((u64)bm_code_of(tfs[j], comp[dl[docs[j]]], false) << 32) | docs[j];
      }
  });
  if (fail.load()) return fail.load();
  std::vector<u64> lbase((size_t)n_lists + 1);
  for (uint32_t i = 0; i <= n_lists; i++) lbase[i] = offs[i] - offs[0];
  packed.resize(n_new);
  if (!positions) return sparse_install(s, n_lists, lbase, packed);
  // positions (phrase queries): per posting its tf positions -- or npos of them: the component terms of an n-gram key, whose own
  // positions stand behind the FIRST component's postings (ssi_bm25_upload_positions)
  NoInitVec<uint32_t> counts(n_new);
  ss_parallel_for(n_new, 1u << 20, [&](size_t a, size_t b, unsigned) {
    for (size_t j = a; j < b; j++) counts[j] = npos ? npos[offs[0] + j] : tfs[offs[0] + j];
  });
  return sparse_install(s, n_lists, lbase, packed, &counts, positions, n_positions, sizeof(uint16_t));
}

// Several indexed fields (an image with MERGED lists): the entries (doc, field, tf) of every rare term, sorted by (doc, field) like
// ssi_bm25_upload_fields takes them.  The sparse tier keeps a term's MERGED list only -- every doc once, its weight the sum over the
// doc's fields of boost_f * tf (K + 1) / (tf + comp[len_f]), fields ascending, coded against the scale the dense merged lists were
// built with (ssi_bm25_build_from_host_merged): what a query without a field filter reads of a dense term, too.
int ssi_bm25_append_sparse_fields(ss_shard* s, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields, const uint16_t* tfs,
                                  const uint16_t* positions, uint64_t n_positions, const uint16_t* npos) {
  if (!s->d_post) return SS_ESTATE;
  const uint32_t L = s->bm_n_fields, RF = bm_real_fields(s);
  if (L == 1) return SS_EINVAL;
  if (!s->bm_merged || !s->d_doclen || s->h_boost.size() != L) return SS_ENOTSUP;  // boosts too far apart for merged lists: no sparse tier either
  if ((uint64_t)s->sp_n + n_lists > 0x7FFFFFFFull - s->bm_n_terms / L) return SS_ENOTSUP;
  std::vector<uint8_t> dl((size_t)RF * s->bm_n_docs);
  SS_HIP(hipMemcpy(dl.data(), s->d_doclen, dl.size(), hipMemcpyDeviceToHost));
  float comp[SS_COMP_N];
  fill_comp(s->bm_avgdl, comp);
  const float mscale = s->h_boost[L - 1];
  for (uint32_t i = 0; i < n_lists; i++)
    if (offs[i + 1] < offs[i]) return SS_EINVAL;
  // docs per list first (an entry opens a doc when its doc id differs from the entry before)
  std::vector<u64> lbase((size_t)n_lists + 1, 0);
  std::atomic<int> fail{SS_OK};
  ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      u64 n = 0;
      for (u64 j = offs[i]; j < offs[i + 1]; j++) {
        if (docs[j] >= s->bm_n_docs || tfs[j] == 0 || fields[j] >= RF) { fail.store(SS_EINVAL); return; }
        if (j > offs[i] && (docs[j] < docs[j - 1] || (docs[j] == docs[j - 1] && fields[j] <= fields[j - 1]))) { fail.store(SS_EINVAL); return; }
        n += (j == offs[i] || docs[j] != docs[j - 1]) ? 1u : 0u;
      }
      lbase[i + 1] = n;
    }
  });
  if (fail.load()) return fail.load();
  for (uint32_t i = 0; i < n_lists; i++) lbase[i + 1] += lbase[i];
  NoInitVec<u64> packed(lbase[n_lists]);
  ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      u64 w_at = lbase[i];
      for (u64 j = offs[i]; j < offs[i + 1];) {
        float w = 0.f;
        u64 e = j;
        uint32_t fmask = 0u;  // the fields that hold the term in this doc: what a field filter asks of a sparse posting
        for (; e < offs[i + 1] && docs[e] == docs[j]; e++) {
          const volatile float part = s->h_boost[fields[e]] * bm_weight_exact(tfs[e], comp[dl[(size_t)fields[e] * s->bm_n_docs + docs[e]]]);
          w = w + part;
          fmask |= 1u << fields[e];
        }
        // the dense lists chose the scale from their own weights: a sparse weight outside the code's range is refused, not clamped
        if (!(w > 0.f) || w / mscale < 6.2e-5f || w / mscale >= 4.0f) { fail.store(SS_ENOTSUP); return; }
        packed[w_at++] = ((u64)(bm_wcode(w / mscale) | (fmask << BM_SP_FIELD_SHIFT)) << 32) | docs[j];
        j = e;
      }
    }
  });
  if (fail.load()) return fail.load();
  if (!positions) return sparse_install(s, n_lists, lbase, packed);
  // positions: per ENTRY (doc, field) in order; a merged posting owns those of all the doc's entries, each tagged with its field
  // (ssi_bm25_upload_positions_fields)
  if (!npos) npos = tfs;
  NoInitVec<uint32_t> counts(packed.size()), pool(n_positions ? n_positions : 1);
  std::vector<u64> lstart((size_t)n_lists + 1, 0);  // first position of every list (lists in parallel below)
  ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      u64 c = 0;
      for (u64 j = offs[i]; j < offs[i + 1]; j++) c += npos[j];
      lstart[i + 1] = c;
    }
  });
  for (uint32_t i = 0; i < n_lists; i++) lstart[i + 1] += lstart[i];
  if (lstart[n_lists] != n_positions) return SS_EINVAL;
  ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++) {
      u64 at = lstart[i], w = lbase[i];
      for (u64 j = offs[i]; j < offs[i + 1];) {
        uint32_t c = 0;
        u64 e = j;
        for (; e < offs[i + 1] && docs[e] == docs[j]; e++) {
          for (uint32_t x = 0; x < npos[e]; x++, at++) {
            if (x && positions[at] <= positions[at - 1]) { fail.store(SS_EINVAL); return; }  // ascending inside a field
            pool[at] = ((uint32_t)fields[e] << BM_POS_FIELD_SHIFT) | positions[at];
          }
          c += npos[e];
        }
        counts[w++] = c;
        j = e;
      }
    }
  });
  if (fail.load()) return fail.load();
  return sparse_install(s, n_lists, lbase, packed, &counts, pool.data(), n_positions, sizeof(uint32_t));
}

static int sparse_install(ss_shard* s, uint32_t n_lists, const std::vector<u64>& lbase, const NoInitVec<u64>& packed,
                          const NoInitVec<uint32_t>* counts, const void* pos, u64 n_pos, size_t elem) {
  if (ssi_bm25_sparse_levels_has(s)) return SS_ESTATE;  // a tier that grows level by level takes levels only (ss_bm25_append_sparse_level)
  const u64 n_new = packed.size();
  const u64* offs = lbase.data();
  const u64 old_n = s->h_sp_base.empty() ? 0 : s->h_sp_base.back();
  if (counts && s->d_sp_pos_end && s->sp_pos_elem != elem) return SS_EINVAL;
  uint64_t *nb = nullptr, *np = nullptr, *ne = nullptr;
  void* npp = nullptr;
  const bool with_pos = counts != nullptr || s->d_sp_pos_end != nullptr;  // the tier carries positions from the first append that brings some
  const size_t el = counts ? elem : s->sp_pos_elem;
  SS_HIP(hipMalloc(&nb, ((size_t)s->sp_n + n_lists + 1) * sizeof(u64)));
  if (hipMalloc(&np, (size_t)std::max<u64>(old_n + n_new, 1) * sizeof(u64)) != hipSuccess) { (void)hipFree(nb); return SS_ENOMEM; }
  std::vector<uint64_t> base = s->h_sp_base;
  if (base.empty()) base.push_back(0);
  for (uint32_t i = 0; i < n_lists; i++) base.push_back(old_n + (offs[i + 1] - offs[0]));
  bool ok = hipMemcpy(nb, base.data(), base.size() * sizeof(u64), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && old_n) ok = hipMemcpy(np, s->d_sp_post, old_n * sizeof(u64), hipMemcpyDeviceToDevice) == hipSuccess;
  if (ok && n_new) ok = hipMemcpy(np + old_n, packed.data(), n_new * sizeof(u64), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && with_pos) {
    // END of every posting's positions in the pool (a posting appended without positions has none: its end = its start)
    NoInitVec<u64> ends(std::max<u64>(old_n + n_new, 1));
    ends[0] = 0;
    if (old_n && s->d_sp_pos_end) ok = hipMemcpy(ends.data(), s->d_sp_pos_end, old_n * sizeof(u64), hipMemcpyDeviceToHost) == hipSuccess;
    else if (old_n) std::fill(ends.begin(), ends.begin() + old_n, (u64)0);  // (appended without positions: none)
    // running ends: per list its positions (lists in parallel), the lists' starts, then every list fills its own postings
    std::vector<u64> lsum((size_t)n_lists + 1, 0);
    if (counts)
      ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
        for (size_t i = a; i < b; i++) {
          u64 c = 0;
          for (u64 x = offs[i] - offs[0]; x < offs[i + 1] - offs[0]; x++) c += (*counts)[x];
          lsum[i + 1] = c;
        }
      });
    lsum[0] = s->sp_pos_n;
    for (uint32_t i = 0; i < n_lists; i++) lsum[i + 1] += lsum[i];
    const u64 run = lsum[n_lists];
    ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
      for (size_t i = a; i < b; i++) {
        u64 r = lsum[i];
        for (u64 x = offs[i] - offs[0]; x < offs[i + 1] - offs[0]; x++) { r += counts ? (*counts)[x] : 0u; ends[old_n + x] = r; }
      }
    });
    if (ok && counts && run - s->sp_pos_n != n_pos) { (void)hipFree(nb); (void)hipFree(np); return SS_EINVAL; }
    ok = ok && hipMalloc(&ne, ends.size() * sizeof(u64)) == hipSuccess &&
         hipMemcpy(ne, ends.data(), ends.size() * sizeof(u64), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMalloc(&npp, (size_t)std::max<u64>(run, 1) * el) == hipSuccess;
    if (ok && s->sp_pos_n) ok = hipMemcpy(npp, s->d_sp_pos, s->sp_pos_n * el, hipMemcpyDeviceToDevice) == hipSuccess;
    if (ok && counts && n_pos) ok = hipMemcpy((char*)npp + s->sp_pos_n * el, pos, n_pos * el, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
      if (s->d_sp_pos_end) (void)hipFree(s->d_sp_pos_end);
      if (s->d_sp_pos) (void)hipFree(s->d_sp_pos);
      s->d_sp_pos_end = ne; s->d_sp_pos = npp; s->sp_pos_n = run; s->sp_pos_elem = (uint32_t)el;
      ne = nullptr; npp = nullptr;
    }
  }
  if (!ok) { (void)hipFree(nb); (void)hipFree(np); if (ne) (void)hipFree(ne); if (npp) (void)hipFree(npp); return SS_EDEVICE; }
  if (s->d_sp_base) (void)hipFree(s->d_sp_base);
  if (s->d_sp_post) (void)hipFree(s->d_sp_post);
  s->d_sp_base = nb;
  s->d_sp_post = np;
  s->h_sp_base = std::move(base);
  s->sp_n += n_lists;
  return SS_OK;
}

// ---------------------------------------------------------------- the sparse tier level by level (sparse_levels.h)
namespace {
struct SparseLevels {
  uint16_t* d_tf = nullptr;    // [sparse postings] the tf behind every posting's code
  std::vector<uint64_t> h_pbase;    // [sp_n + 1] first position of every list in d_sp_pos (a tier with positions)
  int64_t last_level = -1;          // the level the last call brought (a re-commit of it replaces what it brought) ...
  std::vector<uint32_t> h_last_n, h_last_pos;  // ... per list: its postings / positions of that level, the lists' tails
};
std::mutex g_spl_mu;
std::unordered_map<const ss_shard*, SparseLevels> g_spl;

struct SpExtend {
  const uint64_t* old_base; uint32_t n_old; const uint64_t* new_base; uint32_t n_lists;
  const uint64_t* old_post; const uint16_t* old_tf; uint64_t* new_post; uint16_t* new_tf;
  const uint64_t* lvl_off; const uint32_t* lvl_doc; const uint16_t* lvl_tf;
  const uint32_t* drop_n; const uint32_t* drop_p;  // a replaced level: the postings / positions at every old list's tail that go (or nullptr)
  const uint64_t* old_pbase; const uint64_t* new_pbase; const uint64_t* old_pend; uint64_t* new_pend;  // positions (new_pend == nullptr: a tier without)
  const uint16_t* old_pool; uint16_t* new_pool; const uint32_t* lvl_pend; const uint64_t* lvl_pbase; const uint16_t* lvl_pool;
  uint32_t n_docs; uint32_t* bad;
};
}  // namespace
// one wave per list: the list's old postings (and their positions) to their new places, the level's behind them
__global__ void __launch_bounds__(256) sp_extend_kernel(SpExtend A) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t i64 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6;
  if (i64 >= A.n_lists) return;
  const uint32_t i = (uint32_t)i64;
  const bool was = i < A.n_old;
  const uint64_t ob = was ? A.old_base[i] : 0ull, no = was ? A.old_base[i + 1] - ob - (A.drop_n ? A.drop_n[i] : 0u) : 0ull;
  const uint64_t nb = A.new_base[i], lo = A.lvl_off[i], nn = A.lvl_off[i + 1] - lo;
  const bool pos = A.new_pend != nullptr;
  const uint64_t opb = (pos && was) ? A.old_pbase[i] : 0ull, opn = (pos && was) ? A.old_pbase[i + 1] - opb - (A.drop_p ? A.drop_p[i] : 0u) : 0ull;
  const uint64_t npb = pos ? A.new_pbase[i] : 0ull;
  for (uint64_t j = lane; j < no; j += 64u) {
    A.new_post[nb + j] = A.old_post[ob + j];
    A.new_tf[nb + j] = A.old_tf[ob + j];
    if (pos) A.new_pend[nb + j] = npb + (A.old_pend[ob + j] - opb);
  }
  for (uint64_t j = lane; j < nn; j += 64u) {
    const uint32_t d = A.lvl_doc[lo + j];
    bool ok = d < A.n_docs && A.lvl_tf[lo + j] != 0;
    if (j > 0) ok = ok && d > A.lvl_doc[lo + j - 1];
    else if (no) ok = ok && d > (uint32_t)A.old_post[ob + no - 1];  // behind everything the list holds
    if (!ok) atomicOr(A.bad, 1u);
    A.new_post[nb + no + j] = (uint64_t)d;  // (its code: sp_recode_kernel)
    A.new_tf[nb + no + j] = A.lvl_tf[lo + j];
    if (pos) A.new_pend[nb + no + j] = npb + opn + A.lvl_pend[lo + j];
  }
  if (pos) {
    for (uint64_t j = lane; j < opn; j += 64u) A.new_pool[npb + j] = A.old_pool[opb + j];
    const uint64_t lpb = A.lvl_pbase[i], lpn = A.lvl_pbase[i + 1] - lpb;
    for (uint64_t j = lane; j < lpn; j += 64u) A.new_pool[npb + opn + j] = A.lvl_pool[lpb + j];
  }
}
// a sparse posting's code from its tf and its doc's length byte -- the operations of bm_weight_exact / raw_fill_kernel
__global__ void __launch_bounds__(256) sp_recode_kernel(uint64_t* __restrict__ post, const uint16_t* __restrict__ tf, uint64_t n,
                                                        const uint8_t* __restrict__ doclen, uint32_t n_docs, const float* __restrict__ comp, float k1) {
  for (uint64_t p = (uint64_t)blockIdx.x * 256u + threadIdx.x; p < n; p += (uint64_t)gridDim.x * 256u) {
    const uint32_t d = (uint32_t)post[p];
    if (d >= n_docs) continue;  // (a re-committed level that shrank: its sparse postings go with the sparse re-commit)
    const float tt = (float)tf[p];
    const float wgt = __fdiv_rn(ss_fmul(tt, k1), ss_fadd(tt, comp[doclen[d]]));
    post[p] = ((uint64_t)bm_wcode(wgt) << 32) | d;
  }
}

bool ssi_bm25_sparse_levels_has(const ss_shard* s) {
  std::lock_guard<std::mutex> g(g_spl_mu);
  return g_spl.find(s) != g_spl.end();
}
void ssi_bm25_sparse_levels_drop(const ss_shard* s) {
  std::lock_guard<std::mutex> g(g_spl_mu);
  auto it = g_spl.find(s);
  if (it == g_spl.end()) return;
  if (it->second.d_tf) (void)hipFree(it->second.d_tf);
  g_spl.erase(it);
}
int ssi_bm25_sparse_levels_recode(ss_shard* s, hipStream_t st) {
  uint16_t* d_tf = nullptr;
  {
    std::lock_guard<std::mutex> g(g_spl_mu);
    auto it = g_spl.find(s);
    if (it == g_spl.end()) return SS_OK;
    d_tf = it->second.d_tf;
  }
  const uint64_t n = s->h_sp_base.empty() ? 0 : s->h_sp_base.back();
  if (!n) return SS_OK;
  if (!s->d_sp_post || !d_tf || !s->d_doclen || !s->d_comp) return SS_ESTATE;
  const volatile float k1 = 1.2f + 1.0f;  // (K + 1) as bm_weight_exact forms it
  sp_recode_kernel<<<(uint32_t)std::min<uint64_t>((n + 255) / 256, 65536), 256, 0, st>>>(s->d_sp_post, d_tf, n, s->d_doclen, s->bm_n_docs, s->d_comp, k1);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ssi_bm25_append_sparse_level(ss_shard* s, uint32_t level, uint32_t n_lists, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs,
                                 const uint16_t* npos, const uint16_t* positions, uint64_t n_positions) {
  if (!s->d_post || !s->d_doclen || !s->d_comp) return SS_ESTATE;
  if (s->bm_n_fields != 1 || s->bm_merged) return SS_ENOTSUP;  // several indexed fields: whole-image uploads
  const bool fresh = s->sp_n == 0;
  if (!fresh && !ssi_bm25_sparse_levels_has(s)) return SS_ESTATE;  // a tier of whole lists (ss_bm25_append_sparse) keeps no tfs
  if (n_lists < s->sp_n || (uint64_t)n_lists > 0x7FFFFFFFull - s->bm_n_terms) return n_lists < s->sp_n ? SS_EINVAL : SS_ENOTSUP;
  const bool with_pos = fresh ? positions != nullptr : s->d_sp_pos_end != nullptr;
  if (!with_pos && (positions || n_positions)) return SS_EINVAL;  // positions for every level of a tier, or for none
  if (with_pos && n_positions && !positions) return SS_EINVAL;
  if (with_pos && !fresh && s->sp_pos_elem != sizeof(uint16_t)) return SS_ESTATE;
  for (uint32_t i = 0; i < n_lists; i++)
    if (offs[i + 1] < offs[i]) return SS_EINVAL;
  const uint64_t n_new = offs[n_lists] - offs[0], o0 = offs[0];
  const uint32_t n_old = s->sp_n;
  const uint64_t old_n = s->h_sp_base.empty() ? 0 : s->h_sp_base.back();
  // the level: the one the dense image committed last; a level this tier has seen already is replaced (the re-commit of a level
  // that was incomplete, commit.rs:204-206 merge_incomplete_index_level_to_level0)
  if ((uint64_t)level + 1 != s->raw.size()) return SS_EINVAL;
  std::vector<uint32_t> drop_n, drop_p;
  std::vector<uint64_t> old_pbase;
  {
    std::lock_guard<std::mutex> g(g_spl_mu);
    auto it = g_spl.find(s);
    if (it != g_spl.end()) {
      if ((int64_t)level < it->second.last_level) return SS_EINVAL;
      if ((int64_t)level == it->second.last_level) { drop_n = it->second.h_last_n; drop_p = it->second.h_last_pos; }
      old_pbase = it->second.h_pbase;
    }
  }
  const bool replace = !drop_n.empty();
  if (replace && (drop_n.size() != n_old || (with_pos && drop_p.size() != n_old))) return SS_ESTATE;
  const uint64_t d_lo = (uint64_t)level << 16;
  std::atomic<int> fail{SS_OK};
  ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
    for (size_t i = a; i < b; i++)
      for (uint64_t j = offs[i]; j < offs[i + 1]; j++)
        if (docs[j] < d_lo || docs[j] >= s->bm_n_docs || tfs[j] == 0 || (j > offs[i] && docs[j] <= docs[j - 1])) { fail.store(SS_EINVAL); return; }
  });
  if (fail.load()) return fail.load();
  // the new starts of the lists (postings; positions), the level's own offsets
  std::vector<uint64_t> base((size_t)n_lists + 1, 0), lvl_off((size_t)n_lists + 1), pbase, lvl_pbase;
  uint64_t dropped_p = 0;
  for (uint32_t i = 0; i < n_lists; i++) {
    uint64_t had = i < n_old ? s->h_sp_base[i + 1] - s->h_sp_base[i] : 0ull;
    if (replace && i < n_old) { if (drop_n[i] > had) return SS_ESTATE; had -= drop_n[i]; }
    base[i + 1] = base[i] + had + (offs[i + 1] - offs[i]);
  }
  for (uint32_t i = 0; i <= n_lists; i++) lvl_off[i] = offs[i] - o0;
  std::vector<uint32_t> lvl_pend;
  if (with_pos) {
    if (old_pbase.size() != (size_t)n_old + 1) { if (n_old) return SS_ESTATE; old_pbase.assign(1, 0); }
    lvl_pend.resize(n_new ? n_new : 1);
    lvl_pbase.assign((size_t)n_lists + 1, 0);
    ss_parallel_for(n_lists, 4096, [&](size_t a, size_t b, unsigned) {
      for (size_t i = a; i < b; i++) {
        uint64_t run = 0;
        for (uint64_t j = offs[i]; j < offs[i + 1]; j++) {
          run += npos ? npos[j] : tfs[j];
          if (run >= (1ull << 32)) { fail.store(SS_ENOTSUP); return; }
          lvl_pend[j - o0] = (uint32_t)run;
        }
        lvl_pbase[i + 1] = run;
      }
    });
    if (fail.load()) return fail.load();
    for (uint32_t i = 0; i < n_lists; i++) lvl_pbase[i + 1] += lvl_pbase[i];
    if (lvl_pbase[n_lists] != n_positions) return SS_EINVAL;
    pbase.assign((size_t)n_lists + 1, 0);
    for (uint32_t i = 0; i < n_lists; i++) {
      uint64_t had = i < n_old ? old_pbase[i + 1] - old_pbase[i] : 0ull;
      if (replace && i < n_old) { if (drop_p[i] > had) return SS_ESTATE; had -= drop_p[i]; dropped_p += drop_p[i]; }
      pbase[i + 1] = pbase[i] + had + (lvl_pbase[i + 1] - lvl_pbase[i]);
    }
    if (pbase[n_lists] != s->sp_pos_n - dropped_p + n_positions) return SS_ESTATE;
  }
  // device: the new arrays, the level's staging
  std::vector<void*> owned;
  auto dalloc = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(bytes, 8)) != hipSuccess) return nullptr;
    owned.push_back(p);
    return p;
  };
  auto drop_all = [&]() { for (void* p : owned) (void)hipFree(p); owned.clear(); };
  auto up = [&](void* d, const void* h, size_t bytes) { return bytes == 0 || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) == hipSuccess; };
  const uint64_t tot = base[n_lists];
  (void)old_n;
  uint64_t* nbase = (uint64_t*)dalloc(base.size() * 8);
  uint64_t* npost = (uint64_t*)dalloc((size_t)tot * 8);
  uint16_t* ntf = (uint16_t*)dalloc((size_t)tot * 2);
  uint64_t* d_lvl_off = (uint64_t*)dalloc(lvl_off.size() * 8);
  uint32_t* d_lvl_doc = (uint32_t*)dalloc((size_t)n_new * 4);
  uint16_t* d_lvl_tf = (uint16_t*)dalloc((size_t)n_new * 2);
  uint32_t* d_bad = (uint32_t*)dalloc(4);
  uint32_t *d_drop_n = nullptr, *d_drop_p = nullptr;
  uint64_t *d_old_pbase = nullptr, *d_new_pbase = nullptr, *npend = nullptr, *d_lvl_pbase = nullptr;
  uint16_t *npool = nullptr, *d_lvl_pool = nullptr;
  uint32_t* d_lvl_pend = nullptr;
  bool ok = nbase && npost && ntf && d_lvl_off && d_lvl_doc && d_lvl_tf && d_bad;
  if (ok && with_pos) {
    d_old_pbase = (uint64_t*)dalloc(old_pbase.size() * 8);
    d_new_pbase = (uint64_t*)dalloc(pbase.size() * 8);
    npend = (uint64_t*)dalloc((size_t)tot * 8);
    d_lvl_pbase = (uint64_t*)dalloc(lvl_pbase.size() * 8);
    npool = (uint16_t*)dalloc((size_t)pbase[n_lists] * 2);
    d_lvl_pool = (uint16_t*)dalloc((size_t)n_positions * 2);
    d_lvl_pend = (uint32_t*)dalloc((size_t)n_new * 4);
    ok = d_old_pbase && d_new_pbase && npend && d_lvl_pbase && npool && d_lvl_pool && d_lvl_pend;
  }
  if (ok && replace && n_old) {
    d_drop_n = (uint32_t*)dalloc((size_t)n_old * 4);
    if (with_pos) d_drop_p = (uint32_t*)dalloc((size_t)n_old * 4);
    ok = d_drop_n && (!with_pos || d_drop_p);
  }
  if (!ok) { drop_all(); return SS_ENOMEM; }
  const uint32_t zero = 0;
  ok = up(nbase, base.data(), base.size() * 8) && up(d_lvl_off, lvl_off.data(), lvl_off.size() * 8) && up(d_lvl_doc, docs + o0, (size_t)n_new * 4) &&
       up(d_lvl_tf, tfs + o0, (size_t)n_new * 2) && up(d_bad, &zero, 4);
  if (ok && with_pos)
    ok = up(d_old_pbase, old_pbase.data(), old_pbase.size() * 8) && up(d_new_pbase, pbase.data(), pbase.size() * 8) &&
         up(d_lvl_pbase, lvl_pbase.data(), lvl_pbase.size() * 8) && up(d_lvl_pool, positions, (size_t)n_positions * 2) &&
         up(d_lvl_pend, lvl_pend.data(), (size_t)n_new * 4);
  if (ok && d_drop_n) ok = up(d_drop_n, drop_n.data(), (size_t)n_old * 4) && (!d_drop_p || up(d_drop_p, drop_p.data(), (size_t)n_old * 4));
  if (!ok) { drop_all(); return SS_EDEVICE; }
  uint16_t* old_tf = nullptr;
  {
    std::lock_guard<std::mutex> g(g_spl_mu);
    auto it = g_spl.find(s);
    if (it != g_spl.end()) old_tf = it->second.d_tf;
  }
  SpExtend A;
  A.old_base = s->d_sp_base; A.n_old = n_old; A.new_base = nbase; A.n_lists = n_lists;
  A.old_post = s->d_sp_post; A.old_tf = old_tf; A.new_post = npost; A.new_tf = ntf;
  A.lvl_off = d_lvl_off; A.lvl_doc = d_lvl_doc; A.lvl_tf = d_lvl_tf;
  A.drop_n = d_drop_n; A.drop_p = d_drop_p;
  A.old_pbase = d_old_pbase; A.new_pbase = d_new_pbase; A.old_pend = s->d_sp_pos_end; A.new_pend = with_pos ? npend : nullptr;
  A.old_pool = (const uint16_t*)s->d_sp_pos; A.new_pool = npool; A.lvl_pend = d_lvl_pend; A.lvl_pbase = d_lvl_pbase; A.lvl_pool = d_lvl_pool;
  A.n_docs = s->bm_n_docs; A.bad = d_bad;
  sp_extend_kernel<<<(n_lists + 3u) / 4u, 256, 0, s->stream>>>(A);
  uint32_t bad = 0;
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess ||
      hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost) != hipSuccess) { drop_all(); return SS_EDEVICE; }
  if (bad) { drop_all(); return SS_EINVAL; }  // a level posting not behind its list's last doc: the tier stays as it was
  // the swap (the caller holds s->mu and found the device idle)
  auto keep = [&](void* p) { owned.erase(std::find(owned.begin(), owned.end(), p)); };
  keep(nbase); keep(npost); keep(ntf);
  if (with_pos) { keep(npend); keep(npool); }
  drop_all();  // the staging
  if (s->d_sp_base) (void)hipFree(s->d_sp_base);
  if (s->d_sp_post) (void)hipFree(s->d_sp_post);
  s->d_sp_base = nbase; s->d_sp_post = npost;
  if (with_pos) {
    if (s->d_sp_pos_end) (void)hipFree(s->d_sp_pos_end);
    if (s->d_sp_pos) (void)hipFree(s->d_sp_pos);
    s->d_sp_pos_end = npend; s->d_sp_pos = npool; s->sp_pos_n = pbase[n_lists]; s->sp_pos_elem = (uint32_t)sizeof(uint16_t);
  }
  s->h_sp_base = std::move(base);
  s->sp_n = n_lists;
  {
    std::lock_guard<std::mutex> g(g_spl_mu);
    SparseLevels& E = g_spl[s];
    if (E.d_tf) (void)hipFree(E.d_tf);
    E.d_tf = ntf;
    E.h_pbase = std::move(pbase);
    E.last_level = level;
    E.h_last_n.resize(n_lists);
    for (uint32_t i = 0; i < n_lists; i++) E.h_last_n[i] = (uint32_t)(offs[i + 1] - offs[i]);
    E.h_last_pos.clear();
    if (with_pos) {
      E.h_last_pos.resize(n_lists);
      for (uint32_t i = 0; i < n_lists; i++) E.h_last_pos[i] = (uint32_t)(lvl_pbase[i + 1] - lvl_pbase[i]);
    }
  }
  const int rc = ssi_bm25_sparse_levels_recode(s, s->stream);
  if (rc != SS_OK) return rc;
  SS_HIP(hipStreamSynchronize(s->stream));
  return SS_OK;
}

// ---------------------------------------------------------------- positions (phrase queries)
// After ssi_bm25_upload of the same CSR (one indexed field): d_pos = every posting's positions in image order, d_pos_off =
// END offset of the positions of the posting at each (padded) image index relative to its term's first position (the start
// is the previous slot's end, 0 at the term's first slot; NULL padding slots repeat the running end), d_pos_base = first
// position of every term.  The image order of the postings equals the CSR order, so the pool is the caller's array as is.
// npos (optional): the number of positions of every posting where that is not its tf -- the component terms of an n-gram key
// (ref_format.hip): the key's own positions stand behind its FIRST component's postings, the other components have none.
int ssi_bm25_upload_positions(ss_shard* s, const uint64_t* offs, const uint32_t* docs, const uint16_t* tfs, const uint16_t* positions,
                              uint64_t n_positions, const uint16_t* npos) {
  if (!npos) npos = tfs;
  const uint32_t nt = s->bm_n_terms, ns = s->bm_n_sub;
  if (s->bm_n_fields != 1) return SS_ENOTSUP;
  std::vector<u64> pbase((size_t)nt + 1), tbase((size_t)nt + 1);
  SS_HIP(hipMemcpy(tbase.data(), s->d_term_base, tbase.size() * sizeof(u64), hipMemcpyDeviceToHost));
  // (+ 4: the padding loop of a segment may run to the next multiple of 4 before the walk is checked; not value-initialised -- every
  // slot of a term is written by the term's worker below, the slots behind the last term here)
  NoInitVec<uint32_t> poff((size_t)s->bm_n_post_pad + 5);
  for (size_t w = (size_t)std::min<u64>(tbase[nt] * 4ull, poff.size()); w < poff.size(); w++) poff[w] = 0u;
  // a term's positions: the sum of its postings' counts (terms in parallel), then every term fills its own slots
  pbase[0] = 0;
  {
    std::vector<u64> tsum(nt, 0);
    ss_parallel_for(nt, 64, [&](size_t a, size_t b, unsigned) {
      for (size_t t = a; t < b; t++) { u64 c = 0; for (u64 j = offs[t]; j < offs[t + 1]; j++) c += npos[j]; tsum[t] = c; }
    });
    for (uint32_t t = 0; t < nt; t++) pbase[t + 1] = pbase[t] + tsum[t];
  }
  const u64 total = pbase[nt];
  if (total != n_positions) return SS_EINVAL;  // never read past the caller's array (index.bin path: counts come from the file)
  std::atomic<int> fail{SS_OK};
  ss_parallel_for(nt, 16, [&](size_t ta, size_t tb, unsigned) {
    for (size_t t = ta; t < tb; t++) {
      u64 rel = 0, j = offs[t], w = tbase[t] * 4ull;
      for (uint32_t sb = 0; sb < ns; sb++) {
        const u64 lim = ((u64)sb + 1) << BM_SUB_LOG2;
        u64 n = 0;
        for (; j < offs[t + 1] && docs[j] < lim; j++, n++) {
          for (uint32_t x = 1; x < npos[j]; x++)
            if (positions[pbase[t] + rel + x] <= positions[pbase[t] + rel + x - 1]) { fail.store(SS_EINVAL); return; }  // ascending inside a posting
          rel += npos[j];
          if (rel >= (1ull << 32)) { fail.store(SS_ENOTSUP); return; }
          if (w >= tbase[t + 1] * 4ull) { fail.store(SS_EINVAL); return; }  // more postings than the image holds: not the CSR it was built from
          poff[w++] = (uint32_t)rel;
        }
        for (u64 pad = (4 - (n & 3)) & 3; pad > 0; pad--) poff[w++] = (uint32_t)rel;  // NULL padding of the segment
      }
      if (w != tbase[t + 1] * 4ull || j != offs[t + 1]) { fail.store(SS_EINVAL); return; }
    }
  });
  if (fail.load()) return fail.load();
  SS_HIP(hipMalloc(&s->d_pos, (total ? total : 1) * sizeof(uint16_t)));
  SS_HIP(hipMalloc(&s->d_pos_off, poff.size() * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_pos_base, pbase.size() * sizeof(u64)));
  if (total) SS_HIP(hipMemcpy(s->d_pos, positions, total * sizeof(uint16_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_pos_off, poff.data(), poff.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_pos_base, pbase.data(), pbase.size() * sizeof(u64), hipMemcpyHostToDevice));
  return SS_OK;
}

// Several indexed fields, after ssi_bm25_upload_fields of the same entries (term, doc, field, tf sorted by (doc, field) inside a
// term): the positions of the MERGED lists' postings.  A merged posting (term, doc) owns the positions of all the doc's entries
// of that term, fields ascending -- exactly the order of the caller's array -- each tagged with its field above bit 20, so the
// pool is the caller's array with tags, d_pos_off / d_pos_base as for one field but indexed by the merged list's image slots.
int ssi_bm25_upload_positions_fields(ss_shard* s, uint32_t n_terms, const uint64_t* offs, const uint32_t* docs, const uint8_t* fields,
                                     const uint16_t* tfs, const uint16_t* positions, uint64_t n_positions, const uint16_t* npos) {
  if (!s->bm_merged) return SS_ENOTSUP;  // boosts too far apart for merged lists: no phrase path over this corpus
  if (!npos) npos = tfs;  // (npos: positions per entry where that is not its tf -- the component terms of an n-gram key)
  const uint32_t L = s->bm_n_fields, nv = s->bm_n_terms, ns = s->bm_n_sub;
  if ((uint64_t)n_terms * L != nv) return SS_EINVAL;
  std::vector<u64> tbase((size_t)nv + 1), pbase((size_t)nv + 1, 0);
  SS_HIP(hipMemcpy(tbase.data(), s->d_term_base, tbase.size() * sizeof(u64), hipMemcpyDeviceToHost));
  std::vector<uint32_t> poff((size_t)s->bm_n_post_pad + 5, 0u);  // + 4: the padding loop of a segment may run to the next multiple of 4 before the walk is checked
  std::vector<uint32_t> pool(n_positions ? n_positions : 1);
  // a term's positions: the sum of its entries' counts (terms in parallel), then every term fills its own slots
  std::vector<u64> tstart((size_t)n_terms + 1, 0);
  ss_parallel_for(n_terms, 64, [&](size_t ta, size_t tb, unsigned) {
    for (size_t t = ta; t < tb; t++) {
      u64 c = 0;
      for (u64 j = offs[t]; j < offs[t + 1]; j++) c += npos[j];
      tstart[t + 1] = c;
    }
  });
  for (uint32_t t = 0; t < n_terms; t++) tstart[t + 1] += tstart[t];
  if (tstart[n_terms] != n_positions) return SS_EINVAL;
  std::atomic<int> fail{SS_OK};
  ss_parallel_for(n_terms, 16, [&](size_t ta, size_t tb, unsigned) {
    for (size_t t = ta; t < tb; t++) {
      const uint32_t v = (uint32_t)t * L + (L - 1u);  // the term's merged list
      const u64 total = tstart[t];
      for (uint32_t f = 0; f < L; f++) pbase[t * L + f] = total;
      u64 w = tbase[v] * 4ull, rel = 0, j = offs[t];
      for (uint32_t sb = 0; sb < ns; sb++) {
        const u64 lim = ((u64)sb + 1) << BM_SUB_LOG2;
        u64 n = 0;
        while (j < offs[t + 1] && docs[j] < lim) {
          const uint32_t d = docs[j];
          for (; j < offs[t + 1] && docs[j] == d; j++) {  // the doc's entries, fields ascending
            for (uint32_t x = 0; x < npos[j]; x++) {
              const u64 at = total + rel + x;
              if (x && positions[at] <= positions[at - 1]) { fail.store(SS_EINVAL); return; }  // ascending inside a field
              pool[at] = ((uint32_t)fields[j] << BM_POS_FIELD_SHIFT) | positions[at];
            }
            rel += npos[j];
          }
          if (rel >= (1ull << 32)) { fail.store(SS_ENOTSUP); return; }
          if (w >= s->bm_n_post_pad) { fail.store(SS_EINVAL); return; }
          poff[w++] = (uint32_t)rel;
          n++;
        }
        for (u64 pad = (4 - (n & 3)) & 3; pad > 0; pad--) poff[w++] = (uint32_t)rel;
      }
      if (w != tbase[v + 1] * 4ull) { fail.store(SS_EINVAL); return; }  // the walk must land on the merged list's end: same entries as the image's
    }
  });
  if (fail.load()) return fail.load();
  const u64 total = tstart[n_terms];
  pbase[nv] = total;
  SS_HIP(hipMalloc(&s->d_pos32, pool.size() * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_pos_off, poff.size() * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&s->d_pos_base, pbase.size() * sizeof(u64)));
  SS_HIP(hipMemcpy(s->d_pos32, pool.data(), pool.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_pos_off, poff.data(), poff.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  SS_HIP(hipMemcpy(s->d_pos_base, pbase.data(), pbase.size() * sizeof(u64), hipMemcpyHostToDevice));
  return SS_OK;
}

// ---------------------------------------------------------------- BM25 synthetic image, generated on device
__global__ void lex_doclen_kernel(uint8_t* __restrict__ doclen, u64 seed, u64 n_docs, const uint8_t* __restrict__ tab,
                                  u64* __restrict__ psum, u64 gs, u64 go) {
  u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 v = 0;
  if (d < n_docs) {
    uint8_t b = tab[ss_h(seed, 0, d * gs + go) >> 54];
    doclen[d] = b;
    v = ss_byte4_to_int(b);
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(psum, v);
}

// tf - 1 of a synthetic posting: geometric with p = 0.6 (SURVEY 8d's C2 corpus: tf = 1 + min(geom(0.6), 254)) from 32 hash
// bits, integer-exact (the host-side generator of the tests draws the same value): P(j >= m) = 0.4^m, j = how many of the
// thresholds floor(0.4^m * 2^32), m = 1 .. 24, lie above u.  tf <= 25.
__device__ __forceinline__ uint32_t lex_geom06(uint32_t u) {
  constexpr uint32_t T[24] = {1717986918u, 687194767u, 274877906u, 109951162u, 43980465u, 17592186u, 7036874u, 2814749u,
                              1125899u,    450359u,    180143u,    72057u,     28823u,    11529u,    4611u,    1844u,
                              737u,        295u,       118u,       47u,        18u,       7u,        3u,       1u};
  uint32_t j = 0;
#pragma unroll
  for (int m = 0; m < 24; m++) j += u < T[m] ? 1u : 0u;
  return j;
}

// CLUSTERED corpora (seeds with bit 63 set; oracle so_lex_cluster_thresh, bit for bit): a term's density varies with the doc's cluster
// -- runs of 1024 (odd terms) or 8192 (even terms) consecutive GLOBAL doc ids: 70 % of a term's clusters hold it 8 times more
// rarely than its threshold says, 25 % as the threshold says, 5 % four times more densely.  Doc ids then come in bursts and the
// per-block maxima are uneven, as in a corpus ordered by source or time.
__device__ __forceinline__ uint32_t lex_cluster_thresh(u64 seed, uint32_t t, u64 d, uint32_t th) {
  const u64 c = (t & 1u) ? (d >> 10) : (d >> 13);
  const uint32_t r = (uint32_t)(ss_h(seed ^ 0xC1ull, (u64)t + 1, c) >> 40) & 0xFFFFu;
  const u64 m = r < 45875u ? 1u : r < 62259u ? 8u : 32u;
  const u64 v = ((u64)th * m) >> 3;
  return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
}

// one wave per (term, sub-block): count / fill postings in ascending doc order.
// !FILL: sub[t][sb+1] = padded units (for the exclusive scan), cnt[t] += postings.  FILL: postings + zero padding.
template <bool FILL>
__global__ void lex_gen_kernel(u64 seed, u64 n_docs, uint32_t n_terms, uint32_t n_sub, const uint32_t* __restrict__ thresh,
                               const uint8_t* __restrict__ doclen, uint32_t* __restrict__ sub /*[nt][ns+1]*/,
                               const u64* __restrict__ term_base, uint32_t* __restrict__ post, u64* __restrict__ df,
                               uint2* __restrict__ probe, uint32_t* __restrict__ probe_z, const uint32_t* __restrict__ probe_row,
                               uint32_t* __restrict__ umax_bits, float* __restrict__ submax /*[nt][n_sub]*/,
                               const uint32_t* __restrict__ wcode /*[33][256] weight codes by (tf, len)*/,
                               const uint8_t* __restrict__ flagged /*[nt] list carries (tf < 10) in its codes*/, u64 gs, u64 go) {
  const int lane = threadIdx.x & 63;
  const u64 gw = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u64 total = (u64)n_terms * n_sub;
  if (gw >= total) return;
  const uint32_t t = (uint32_t)(gw / n_sub), sb = (uint32_t)(gw % n_sub);
  const uint32_t th = thresh[t];
  const u64 d0 = (u64)sb << BM_SUB_LOG2;
  uint32_t run = 0;
  u64 base = 0;
  float wmax = 0.f;
  if (FILL) base = (term_base[t] + sub[(size_t)t * (n_sub + 1) + sb]) * 4;
  for (int i = 0; i < BM_SUB / 64; i++) {
    u64 d = d0 + (u64)i * 64 + lane;
    u64 hv = ss_h(seed, (u64)t + 1, d * gs + go);
    bool present = d < n_docs && (uint32_t)(hv >> 32) < ((seed >> 63) ? lex_cluster_thresh(seed, t, d * gs + go, th) : th);
    u64 m = __ballot(present);
    if (FILL && present) {
      uint32_t pos = run + __popcll(m & ((1ull << lane) - 1ull));
      uint32_t tf = 1u + lex_geom06((uint32_t)hv);
      uint32_t code = wcode[(tf << 8) + doclen[d]];
      if (flagged[t]) code = (code & ~1u) | (tf < 10u ? 1u : 0u);
      post[base + pos] = bm_pack((uint32_t)(d & (BM_SUB - 1)), code);
      wmax = fmaxf(wmax, bm_wdecode(code));
    }
    if (FILL && probe && lane == 0 && probe_row[t] != BM_NO_PROBE_ROW) {
      const size_t gi = ((size_t)probe_row[t] * n_sub + sb) * (BM_SUB / 64) + i;
      probe[gi] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
      probe_z[gi] = sub[(size_t)t * (n_sub + 1) + sb] * 4u + run;
    }
    run += __popcll(m);
  }
  if (FILL) {
    for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    if (lane == 0 && wmax > 0.f) atomicMax(&umax_bits[t], __float_as_uint(wmax));  // positive floats order like their bits
    if (lane == 0) submax[(size_t)t * n_sub + sb] = wmax;
    if ((uint32_t)lane < ((4u - (run & 3u)) & 3u)) post[base + run + lane] = 0u;  // NULL padding
  } else if (lane == 0) {
    sub[(size_t)t * (n_sub + 1) + sb + 1] = (run + 3u) >> 2;  // shifted by one for the exclusive scan
    if (run) atomicAdd(&df[t], (u64)run);
  }
}

// per term: in-place inclusive scan of row[1..ns] (row[0] = 0) -> exclusive offsets; writes the term total
__global__ void lex_scan_rows_kernel(uint32_t* __restrict__ sub, uint32_t n_sub, u64* __restrict__ term_tot) {
  const uint32_t t = blockIdx.x;
  uint32_t* row = sub + (size_t)t * (n_sub + 1);
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) { carry = 0; row[0] = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t b0 = 1; b0 <= n_sub; b0 += blockDim.x) {
    uint32_t i = b0 + threadIdx.x;
    uint32_t v = i <= n_sub ? row[i] : 0;
    uint32_t x = v;
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint32_t pre = carry;
    for (int j = 0; j < w; j++) pre += wsum[j];
    if (i <= n_sub) row[i] = pre + x;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry = pre + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) term_tot[t] = carry;
}

__global__ void lex_scan_terms_kernel(const u64* __restrict__ tot, u64* __restrict__ base, uint32_t n_terms) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    u64 a = 0;
    for (uint32_t t = 0; t < n_terms; t++) { base[t] = a; a += tot[t]; }
    base[n_terms] = a;
  }
}

int ssi_bm25_synth(ss_shard* s, uint64_t seed, const uint32_t* d_thresh, const uint8_t* d_lentab, hipStream_t st) {
  const uint32_t nt = s->bm_n_terms, ns = s->bm_n_sub;
  const u64 nd = s->bm_n_docs;
  uint8_t* d_doclen = nullptr;
  u64* d_psum = nullptr;
  u64* d_tot = nullptr;
  u64* d_df = nullptr;
  SS_HIP(hipMalloc(&d_doclen, nd));
  SS_HIP(hipMalloc(&d_psum, sizeof(u64)));
  SS_HIP(hipMalloc(&d_tot, (size_t)nt * sizeof(u64)));
  SS_HIP(hipMalloc(&d_df, (size_t)nt * sizeof(u64)));
  SS_HIP(hipMemsetAsync(d_psum, 0, sizeof(u64), st));
  SS_HIP(hipMemsetAsync(d_df, 0, (size_t)nt * sizeof(u64), st));
  lex_doclen_kernel<<<(uint32_t)((nd + 255) / 256), 256, 0, st>>>(d_doclen, seed, nd, d_lentab, d_psum, s->synth_stride, s->synth_offset);
  const size_t rows = (size_t)nt * (ns + 1);
  SS_HIP(hipMalloc(&s->d_sub_off, (rows + ns + 1) * sizeof(uint32_t)));  // + one all-zero row (absent terms)
  SS_HIP(hipMemsetAsync(s->d_sub_off + rows, 0, ((size_t)ns + 1) * sizeof(uint32_t), st));
  SS_HIP(hipMalloc(&s->d_term_base, ((size_t)nt + 1) * sizeof(u64)));
  SS_HIP(hipMalloc(&s->d_comp, SS_COMP_N * sizeof(float)));
  const u64 waves = (u64)nt * ns;
  const uint32_t grid = (uint32_t)((waves + 3) / 4);
  lex_gen_kernel<false><<<grid, 256, 0, st>>>(seed, nd, nt, ns, d_thresh, d_doclen, s->d_sub_off, nullptr, nullptr, d_df,
                                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s->synth_stride, s->synth_offset);
  lex_scan_rows_kernel<<<nt, 1024, 0, st>>>(s->d_sub_off, ns, d_tot);
  lex_scan_terms_kernel<<<1, 64, 0, st>>>(d_tot, (u64*)s->d_term_base, nt);
  SS_HIP(hipStreamSynchronize(st));
  std::vector<u64> tot(nt);
  s->h_df.resize(nt);
  SS_HIP(hipMemcpy(tot.data(), d_tot, (size_t)nt * sizeof(u64), hipMemcpyDeviceToHost));
  SS_HIP(hipMemcpy(s->h_df.data(), d_df, (size_t)nt * sizeof(u64), hipMemcpyDeviceToHost));
  u64 units = 0, psum = 0, npost = 0;
  for (uint32_t t = 0; t < nt; t++) {
    if (tot[t] >= (1ull << 28)) return SS_ENOTSUP;  // a term's segment offsets must stay below 4 GB
    units += tot[t];
    npost += s->h_df[t];
  }
  SS_HIP(hipMemcpy(&psum, d_psum, sizeof(u64), hipMemcpyDeviceToHost));
  s->bm_n_post = npost;
  s->bm_avgdl = (float)psum / (float)nd;
  float comp[SS_COMP_N];
  fill_comp(s->bm_avgdl, comp);
  SS_HIP(hipMemcpy(s->d_comp, comp, sizeof(comp), hipMemcpyHostToDevice));
  int rc = alloc_post(s, units);
  if (rc) return rc;
  rc = alloc_probe(s, st);
  if (rc) return rc;
  // weight codes of every (tf, len) the generator can produce, computed on the host like every other image's
  std::vector<uint32_t> wtab((size_t)(SS_SYNTH_TF_MAX + 1) * 256, 0u);
  for (uint32_t tf = 1; tf <= (uint32_t)SS_SYNTH_TF_MAX; tf++)
    for (uint32_t l = 0; l < 256; l++) wtab[(tf << 8) + l] = bm_code_of(tf, comp[l], false);
  std::vector<uint8_t> flg(nt);
  for (uint32_t t = 0; t < nt; t++) flg[t] = bm_list_flagged(s, s->h_df[t]) ? 1 : 0;
  uint32_t* d_wtab = nullptr;
  uint8_t* d_flg = nullptr;
  SS_HIP(hipMalloc(&d_wtab, wtab.size() * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&d_flg, nt));
  SS_HIP(hipMemcpyAsync(d_wtab, wtab.data(), wtab.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  SS_HIP(hipMemcpyAsync(d_flg, flg.data(), nt, hipMemcpyHostToDevice, st));
  lex_gen_kernel<true><<<grid, 256, 0, st>>>(seed, nd, nt, ns, d_thresh, d_doclen, s->d_sub_off,
                                             (const u64*)s->d_term_base, s->d_post, nullptr, s->d_probe, s->d_probe_z, s->d_probe_row,
                                             (uint32_t*)s->d_umax, s->d_submax, d_wtab, d_flg, s->synth_stride, s->synth_offset);
  SS_HIP(hipStreamSynchronize(st));
  (void)hipFree(d_wtab);
  (void)hipFree(d_flg);
  rc = bm_decide_partmax_dev(s);  // (uniform corpora: no; clustered ones -- seed bit 63 -- : yes)
  if (rc) return rc;
  s->d_doclen = d_doclen;  // kept: ss_bm25_append_sparse weighs its postings with the docs' length bytes
  (void)hipFree(d_psum);
  (void)hipFree(d_tot);
  (void)hipFree(d_df);
  SS_HIP(hipGetLastError());
  return SS_OK;
}
