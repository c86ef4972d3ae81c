// NT-specialised BM25 scan: the same algorithm as bm25_scan_kernel (bm25.hip) with the per-term state of a query
// (posting base, sub-block boundaries, idf) held in SCALAR registers for a compile-time term count NT <= 4, so the
// per-item bookkeeping is a handful of SALU instructions instead of readlane traffic, and with a trigger-based
// phase 2: while accumulating, every update compares the new partial score with the current k-th best; only when
// some doc could enter the list (or exact counts are requested) is the 4096-entry tile scanned, otherwise it is just
// cleared.  Reference path replaced: union_docid_2/3, intersection_blockid/docid, single_blockid + add_result +
// get_bm25f + add_topk (see bm25.hip header).
#include "bm25_dev.h"

template <int NT> struct FastCfg { static constexpr int CPT = NT <= 2 ? 4 : 3; static constexpr int RC = NT * CPT; };

constexpr int BM_WAVE_ACC = BM_SUB * 4 + 256;  // accumulator tile + 64 per-lane dump slots

template <int NT, bool HAS_AND, int KPL>
__global__ void __launch_bounds__((HAS_AND ? BM_WAVES_AND : BM_WAVES_OR) * 64) bm25_scan_fast_kernel(BmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WAVES = HAS_AND ? BM_WAVES_AND : BM_WAVES_OR;
  constexpr int CPT = FastCfg<NT>::CPT;
  constexpr int RC = FastCfg<NT>::RC;
  constexpr int WAVE_LDS = BM_WAVE_ACC + (HAS_AND ? BM_SUB : 0);
  float* comp = (float*)smem;
  float* wlut = comp + 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* wbase = smem + BM_LUT_BYTES + w * WAVE_LDS;
  float* acc = (float*)wbase;                  // [4096] + dump[64]
  uint8_t* cnt8 = (uint8_t*)(wbase + BM_WAVE_ACC);
  uint32_t* cntw = (uint32_t*)cnt8;
  const uint32_t* __restrict__ post = p.post;
  const uint32_t* __restrict__ sub_off = p.sub_off;

  for (int i = tid; i < 256 + 4096; i += WAVES * 64) comp[i] = p.comp[i];
  for (int i = lane; i < BM_SUB + 64; i += 64) acc[i] = 0.f;
  if (HAS_AND)
    for (int i = lane; i < BM_SUB / 4; i += 64) cntw[i] = 0u;
  __syncthreads();

  const uint32_t total_waves = gridDim.x * WAVES;
  const uint32_t A = p.nq * p.P;
  const uint32_t row_len = p.n_sub + 1;
  const uint32_t k = p.k;
  const bool count_mode = p.count != 0;
  const uint32_t lane4 = lane * 4;

  for (uint32_t a = blockIdx.x * WAVES + w; a < A; a += total_waves) {
    const uint32_t qi = a % p.nq, part = a / p.nq;
    const ss_bm25_query* __restrict__ Q = p.q + qi;
    const uint32_t nt = __builtin_amdgcn_readfirstlane(Q->n_terms);
    const bool is_and = HAS_AND && (__builtin_amdgcn_readfirstlane(Q->op) == SS_OP_INTERSECTION) && nt > 1;
    // per-term scalars (terms >= nt are empty)
    uint32_t row[NT];
    u64 pbase[NT];
    float idf[NT];
    bool have[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      have[t] = (uint32_t)t < nt;
      const uint32_t term = have[t] ? __builtin_amdgcn_readfirstlane(Q->term[t]) : 0u;
      idf[t] = have[t] ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(Q->idf[t]))) : 0.f;
      row[t] = term * row_len;
      pbase[t] = p.term_base[term];
    }
    const uint32_t s_begin = (uint32_t)(((u64)p.n_sub * part) / p.P);
    const uint32_t s_end = (uint32_t)(((u64)p.n_sub * (part + 1)) / p.P);

    u64 topk[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) topk[r] = 0ull;
    u64 worst = 0ull;
    float wsc = -1.0f;  // score of the current k-th best (-1 while the list is not full): trigger threshold
    u64 matched = 0;

    auto bnd = [&](int t, uint32_t j) -> uint32_t {  // uniform address -> scalar load
      return have[t] ? sub_off[row[t] + (j < s_end ? j : s_end)] : 0u;
    };

    // exactly RC loads per item (inactive chunks / lanes read post[0], one cached line) => counted vmcnt waits
    auto issue_loads = [&](uint4(&v)[RC], const uint32_t (&b0)[NT], const uint32_t (&b1)[NT], uint32_t c0) {
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const u64 abs0 = pbase[t] + b0[t];
        const uint32_t lead = (uint32_t)abs0 & 3u;
        const uint32_t span = lead + (b1[t] - b0[t]);
#pragma unroll
        for (int c = 0; c < CPT; c++) {
          const uint32_t vs = (c0 + c) << 8;
          const bool chunk_on = vs < span;
          const uint32_t* base = chunk_on ? (post + (abs0 - lead) + vs) : post;
          const uint32_t off = (chunk_on && (lane4 < span - vs)) ? lane4 : 0u;
          v[t * CPT + c] = *(const uint4*)(base + off);
        }
      }
    };

    bool trig = false;
    // phase 1: acc[doc] += idf_t * wlut[tf,len]   (add_result.rs:1445-1447); plain gather/scatter, docs of one chunk
    // are distinct; out-of-span lanes are redirected to a private dump slot so nothing is predicated
    auto phase1 = [&](uint4(&v)[RC], const uint32_t (&b0)[NT], const uint32_t (&b1)[NT], uint32_t c0) {
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const uint32_t lead = (uint32_t)(pbase[t] + b0[t]) & 3u;
        const uint32_t len = b1[t] - b0[t];
        const uint32_t span = lead + len;
#pragma unroll
        for (int c = 0; c < CPT; c++) {
          const uint32_t vs = (c0 + c) << 8;
          if (vs < span) {
            const uint32_t i0 = vs + lane4 - lead;
            const uint4 q = v[t * CPT + c];
            const uint32_t pv[4] = {q.x, q.y, q.z, q.w};
            uint32_t doc[4];
            float old[4], wp[4];
            uint32_t cold[4];
            bool anybig = false;
#pragma unroll
            for (int x = 0; x < 4; x++) {
              const bool valid = (i0 + x) < len;
              doc[x] = valid ? (pv[x] & 0xFFFu) : (uint32_t)(BM_SUB + lane);
              old[x] = acc[doc[x]];
              wp[x] = wlut[(pv[x] >> 13) & 0xFFFu];
              anybig |= valid && (pv[x] >> 25) != 0;
              if (HAS_AND && is_and) cold[x] = valid ? cnt8[doc[x]] : 0u;
            }
            if (__ballot(anybig)) {  // tf >= 16: outside the table (rare)
#pragma unroll
              for (int x = 0; x < 4; x++) {
                if (pv[x] >> 25) {
                  float tf = (float)(pv[x] >> 21);
                  wp[x] = tf * BM_K1P * __builtin_amdgcn_rcpf(tf + comp[(pv[x] >> 13) & 0xFFu]);
                }
              }
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
              const float nw = old[x] + idf[t] * wp[x];
              acc[doc[x]] = nw;
              trig |= (doc[x] < BM_SUB) && (nw >= wsc);
              if (HAS_AND && is_and) {
                if (doc[x] < BM_SUB) cnt8[doc[x]] = (uint8_t)(cold[x] + 1u);
              }
            }
          }
        }
      }
    };

    auto clear_tile = [&]() {
#pragma unroll
      for (int i = 0; i < BM_SUB / 256; i++) {
        *(float4*)(acc + (i * 64 + lane) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (HAS_AND && is_and) cntw[i * 64 + lane] = 0u;
      }
    };

    // dense scan (only when some doc may enter the list, or exact counts are wanted)
    auto scan_tile = [&](uint32_t doc_base) {
#pragma unroll 2
      for (int i = 0; i < BM_SUB / 256; i++) {
        const int slot = i * 64 + lane;
        float4 x = *(const float4*)(acc + slot * 4);
        *(float4*)(acc + slot * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        bool h0 = x.x != 0.f, h1 = x.y != 0.f, h2 = x.z != 0.f, h3 = x.w != 0.f;
        if (HAS_AND && is_and) {
          uint32_t cw = cntw[slot];
          cntw[slot] = 0u;
          h0 = (cw & 0xFFu) == nt;
          h1 = ((cw >> 8) & 0xFFu) == nt;
          h2 = ((cw >> 16) & 0xFFu) == nt;
          h3 = (cw >> 24) == nt;
          if (!h0) x.x = 0.f;
          if (!h1) x.y = 0.f;
          if (!h2) x.z = 0.f;
          if (!h3) x.w = 0.f;
        }
        if (count_mode)
          matched += __popcll(__ballot(h0)) + __popcll(__ballot(h1)) + __popcll(__ballot(h2)) + __popcll(__ballot(h3));
        if (k) {
          const float m = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
          if (__ballot(m > 0.f && m >= wsc)) {
            const uint32_t d0 = doc_base + slot * 4;
            u64 k0 = ((u64)__float_as_uint(x.x) << 32) | (u64)(0xFFFFFFFFu - d0);
            u64 k1 = ((u64)__float_as_uint(x.y) << 32) | (u64)(0xFFFFFFFFu - (d0 + 1));
            u64 k2 = ((u64)__float_as_uint(x.z) << 32) | (u64)(0xFFFFFFFFu - (d0 + 2));
            u64 k3 = ((u64)__float_as_uint(x.w) << 32) | (u64)(0xFFFFFFFFu - (d0 + 3));
            k0 = (x.x > 0.f && k0 > worst) ? k0 : 0ull;
            k1 = (x.y > 0.f && k1 > worst) ? k1 : 0ull;
            k2 = (x.z > 0.f && k2 > worst) ? k2 : 0ull;
            k3 = (x.w > 0.f && k3 > worst) ? k3 : 0ull;
            if (__ballot((k0 | k1 | k2 | k3) != 0ull)) {
              worst = topk_offer<KPL>(topk, k0, k1, k2, k3, worst, k);
              if (worst) wsc = __uint_as_float((uint32_t)(worst >> 32));
            }
          }
        }
      }
    };

    uint4 vA[RC], vB[RC];
    uint32_t B0[NT], B1[NT], B2[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { B0[t] = bnd(t, s_begin); B1[t] = bnd(t, s_begin + 1); B2[t] = bnd(t, s_begin + 2); }
    issue_loads(vA, B0, B1, 0);

    auto body = [&](uint4(&cur)[RC], uint4(&nxt)[RC], uint32_t s) {
      uint32_t B3[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) B3[t] = bnd(t, s + 3);
      issue_loads(nxt, B1, B2, 0);
      uint32_t maxspan = 0;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const uint32_t len = B1[t] - B0[t];
        const uint32_t span = len ? (((uint32_t)(pbase[t] + B0[t]) & 3u) + len) : 0u;
        maxspan = span > maxspan ? span : maxspan;
      }
      if (maxspan) {
        trig = false;
        const uint32_t maxc = (maxspan + 255) >> 8;
        for (uint32_t c0 = 0; c0 < maxc; c0 += CPT) {
          if (c0) issue_loads(cur, B0, B1, c0);  // oversized item: later rounds load synchronously
          phase1(cur, B0, B1, c0);
        }
        if (count_mode || (k && __ballot(trig))) scan_tile(s << BM_SUB_LOG2);
        else clear_tile();
      }
#pragma unroll
      for (int t = 0; t < NT; t++) { B0[t] = B1[t]; B1[t] = B2[t]; B2[t] = B3[t]; }
    };
    for (uint32_t s = s_begin; s < s_end; s += 2) {
      body(vA, vB, s);
      if (s + 1 < s_end) body(vB, vA, s + 1);
      else break;
    }

    u64* out = p.part_keys + ((size_t)qi * p.P + part) * (64 * KPL);
#pragma unroll
    for (int r = 0; r < KPL; r++) out[r * 64 + lane] = topk[r];
    if (lane == 0 && matched) atomicAdd(&p.total[qi], matched);
  }
}

template <int NT, bool HAS_AND, int KPL>
static int launch_fast(const BmParams& p, hipStream_t st) {
  constexpr int WAVES = HAS_AND ? BM_WAVES_AND : BM_WAVES_OR;
  constexpr int lds = BM_LUT_BYTES + WAVES * (BM_WAVE_ACC + (HAS_AND ? BM_SUB : 0));
  static bool done = false;
  if (!done) {
    SS_HIP(hipFuncSetAttribute((const void*)bm25_scan_fast_kernel<NT, HAS_AND, KPL>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done = true;
  }
  const uint32_t A = p.nq * p.P;
  const uint32_t grid = std::min<uint32_t>((A + WAVES - 1) / WAVES, 256);
  bm25_scan_fast_kernel<NT, HAS_AND, KPL><<<grid, WAVES * 64, lds, st>>>(p);
  return SS_OK;
}

int ssi_bm25_launch_fast(const BmParams& p, uint32_t nt_max, bool has_and, int KPL, hipStream_t st) {
  if (nt_max == 0 || nt_max > 4 || (KPL != 1 && KPL != 2)) return SS_ENOTSUP;
  const int NT = nt_max <= 2 ? 2 : (int)nt_max;
#define SS_F(NT_, AND_, KPL_) \
  if (NT == NT_ && has_and == AND_ && KPL == KPL_) return launch_fast<NT_, AND_, KPL_>(p, st);
  SS_F(2, false, 1) SS_F(3, false, 1) SS_F(4, false, 1) SS_F(2, true, 1) SS_F(3, true, 1) SS_F(4, true, 1)
  SS_F(2, false, 2) SS_F(3, false, 2) SS_F(4, false, 2) SS_F(2, true, 2) SS_F(3, true, 2) SS_F(4, true, 2)
#undef SS_F
  return SS_ENOTSUP;
}
