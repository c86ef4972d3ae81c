// The multi-shard exchange behind the C ABI: ONE RCCL all-gather of the per-shard top-k of a query batch over xGMI, then
// the same merge on every rank.  Replaces the reference's await-all-JoinHandles + Vec::append + sort/offset/length of
// <IndexArc as Search>::search (search.rs:1669-1673 global ids, 1875-1940 gather, 2098-2119 sort / truncate) for shards
// that live on different GPUs.  The payload is tiny ((2 k + 1) words per query and rank), so the step is latency bound:
// doc ids, score bits and counts travel in a single packed buffer -- one collective, not three.
// Two ways to form the communicator, as the two host shapes need:
//   ss_comm_create      one PROCESS per GPU (torchrun-style): every rank calls it with the same 128-byte id
//   ss_comm_create_all  one process holding all S shards (the reference's shape: one process, S shard tasks)
#include "ss_common.h"

#include <rccl/rccl.h>

#include <cstring>
#include <new>

struct ss_comm {
  int device = 0, rank = 0, n_ranks = 1;
  ncclComm_t comm = nullptr;
  uint32_t* d_send = nullptr;  // [(2 k + 1) nq] packed lists of this rank
  uint32_t* d_recv = nullptr;  // [n_ranks][(2 k + 1) nq]
  size_t cap_words = 0;        // capacity of d_send
  // ss_*_search_sharded: one arena for the packed send / receive buffers, the merged lists and the fused answer (grow-only)
  char* d_ws = nullptr;
  size_t ws_cap = 0;
  // optional timing of the collective itself (ss_comm_profile): HIP events around every all-gather of the sharded searches
  bool prof_on = false;
  uint64_t prof_calls = 0;
  double prof_us = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending;
  std::mutex mu;
};

#define SS_TRY(x)          \
  do {                     \
    int _rc = (x);         \
    if (_rc) return _rc;   \
  } while (0)
#define SS_NCCL(x)                              \
  do {                                          \
    ncclResult_t _r = (x);                      \
    if (_r != ncclSuccess) return SS_EDEVICE;   \
  } while (0)

__global__ void comm_pack_kernel(const uint32_t* __restrict__ doc, const float* __restrict__ score, const uint32_t* __restrict__ cnt,
                                 uint32_t* __restrict__ out, size_t nk, uint32_t nq) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nk) {
    out[i] = doc[i];
    out[nk + i] = __float_as_uint(score[i]);
  }
  if (i < nq) out[2 * nk + i] = cnt[i];
}

static int comm_reserve(ss_comm* c, size_t words) {
  if (words <= c->cap_words) return SS_OK;
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  c->d_send = c->d_recv = nullptr;
  c->cap_words = 0;
  SS_HIP(hipMalloc(&c->d_send, words * sizeof(uint32_t)));
  SS_HIP(hipMalloc(&c->d_recv, words * sizeof(uint32_t) * (size_t)c->n_ranks));
  c->cap_words = words;
  return SS_OK;
}

extern "C" {

int ss_comm_unique_id(uint8_t id_out[SS_COMM_ID_BYTES]) {
  if (!id_out) return SS_EINVAL;
  static_assert(SS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId id;
  SS_NCCL(ncclGetUniqueId(&id));
  memcpy(id_out, id.internal, SS_COMM_ID_BYTES);
  return SS_OK;
}

int ss_comm_create(int device, int rank, int n_ranks, const uint8_t id[SS_COMM_ID_BYTES], ss_comm** out) {
  if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return SS_EINVAL;
  SS_HIP(hipSetDevice(device));
  ss_comm* c = new (std::nothrow) ss_comm;
  if (!c) return SS_ENOMEM;
  c->device = device; c->rank = rank; c->n_ranks = n_ranks;
  ncclUniqueId uid;
  memcpy(uid.internal, id, SS_COMM_ID_BYTES);
  if (ncclCommInitRank(&c->comm, n_ranks, uid, rank) != ncclSuccess) { delete c; return SS_EDEVICE; }
  *out = c;
  return SS_OK;
}

int ss_comm_create_all(int n_devices, const int* devices, ss_comm** out) {
  if (!out || n_devices < 1 || !devices) return SS_EINVAL;
  std::vector<ncclComm_t> comms((size_t)n_devices);
  SS_NCCL(ncclCommInitAll(comms.data(), n_devices, devices));
  for (int i = 0; i < n_devices; i++) out[i] = nullptr;
  for (int i = 0; i < n_devices; i++) {
    ss_comm* c = new (std::nothrow) ss_comm;
    if (!c) {  // nothing half-made is left behind: the handles created so far and every RCCL communicator go
      for (int j = 0; j < i; j++) { (void)ss_comm_destroy(out[j]); out[j] = nullptr; }
      for (int j = i; j < n_devices; j++) (void)ncclCommDestroy(comms[(size_t)j]);
      return SS_ENOMEM;
    }
    c->device = devices[i]; c->rank = i; c->n_ranks = n_devices; c->comm = comms[(size_t)i];
    out[i] = c;
  }
  return SS_OK;
}

int ss_comm_destroy(ss_comm* c) {
  if (!c) return SS_EINVAL;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->d_ws) (void)hipFree(c->d_ws);
  for (auto& e : c->prof_pending) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete c;
  return SS_OK;
}

int ss_comm_info(const ss_comm* c, int* rank, int* n_ranks, int* device) {
  if (!c) return SS_EINVAL;
  if (rank) *rank = c->rank;
  if (n_ranks) *n_ranks = c->n_ranks;
  if (device) *device = c->device;
  return SS_OK;
}

int ss_topk_allgather_merge(ss_comm* c, uint32_t n_queries, uint32_t k, const uint32_t* d_doc, const float* d_score,
                            const uint32_t* d_count, uint64_t* d_out_doc, float* d_out_score, uint32_t* d_out_count, void* stream) {
  if (!c || !d_doc || !d_score || !d_count || !d_out_doc || !d_out_score || !d_out_count) return SS_EINVAL;
  if (k == 0 || (uint64_t)c->n_ranks * k > 0xFFFFFFFFull) return SS_EINVAL;
  if (n_queries == 0) return SS_OK;
  std::lock_guard<std::mutex> g(c->mu);
  SS_HIP(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const size_t nk = (size_t)n_queries * k, words = 2 * nk + n_queries;
  if (words > c->cap_words) {
    SS_HIP(hipStreamSynchronize(st));  // earlier calls on this stream may still read the old buffers
    int rc = comm_reserve(c, words);
    if (rc) return rc;
  }
  comm_pack_kernel<<<(uint32_t)((nk + 255) / 256), 256, 0, st>>>(d_doc, d_score, d_count, c->d_send, nk, n_queries);
  SS_HIP(hipGetLastError());
  SS_NCCL(ncclAllGather(c->d_send, c->d_recv, words, ncclInt32, c->comm, st));
  return ss_topk_merge_dev_packed(c->device, n_queries, (uint32_t)c->n_ranks, k, c->d_recv, d_out_doc, d_out_score, d_out_count, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The exchange of the ss_*_search_sharded entry points: ONE all-gather per call.  Every rank contributes, as 32-bit words,
//   per list (lexical and / or vector):  [nq * k doc ids | nq * k score bits | nq counts]      (+ 1 pad word when odd)
//   then                                  [nq totals (u64) | status (u64)]
// -- the totals travel with the lists (summed by a small kernel after the gather: result_count_total is summed over the
// shards, search.rs:1884-1921; Hybrid sums max(lexical, vector) per shard) instead of through an all-reduce of their own: the
// step is latency bound, one collective costs one latency.  The status word is how the ranks AGREE on failure: a rank whose
// local search failed (image missing, a query its shard cannot serve, an allocation) still enters the collective, with
// empty lists and status 1, and every rank returns an error after the exchange -- nobody is left waiting in a collective
// the failing rank never entered.
__global__ void comm_pack_lists_kernel(uint32_t nq, uint32_t k, const uint32_t* __restrict__ doc, const float* __restrict__ score,
                                       const uint32_t* __restrict__ cnt, uint32_t* __restrict__ out) {
  const size_t nk = (size_t)nq * k, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nk) {
    out[i] = doc[i];
    out[nk + i] = __float_as_uint(score[i]);
  }
  if (i < nq) out[2 * nk + i] = cnt[i];
}
// totals: a alone, or max(a, b) per query (Hybrid, search.rs:1919-1921); word nq = the rank's status
__global__ void comm_pack_totals_kernel(uint32_t nq, const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b,
                                        unsigned long long status, unsigned long long* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) {
    unsigned long long v = a ? a[i] : 0ull;
    if (b) v = v > b[i] ? v : b[i];
    out[i] = v;
  }
  if (i == nq) out[nq] = status;
}
__global__ void comm_sum_totals_kernel(uint32_t nq, uint32_t S, const uint32_t* __restrict__ recv, size_t words_per_rank, size_t tot_off,
                                       unsigned long long* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nq) return;
  unsigned long long sum = 0ull;
  for (uint32_t r = 0; r < S; r++) sum += ((const unsigned long long*)(recv + (size_t)r * words_per_rank + tot_off))[i];
  out[i] = sum;
}

static inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

int ssi_comm_exchange_hf(ss_comm* c, uint32_t nq, int n_lists, const ss_dev_list* L, const uint64_t* d_tot_a, const uint64_t* d_tot_b,
                         int local_rc, bool hybrid, uint32_t offset, uint32_t length, uint64_t* out_doc, float* out_score,
                         uint8_t* out_source, uint32_t* out_count, uint64_t* out_total, hipStream_t st, bool host_fuse);
int ssi_comm_exchange(ss_comm* c, uint32_t nq, int n_lists, const ss_dev_list* L, const uint64_t* d_tot_a, const uint64_t* d_tot_b,
                      int local_rc, bool hybrid, uint32_t offset, uint32_t length, uint64_t* out_doc, float* out_score,
                      uint8_t* out_source, uint32_t* out_count, uint64_t* out_total, hipStream_t st) {
  return ssi_comm_exchange_hf(c, nq, n_lists, L, d_tot_a, d_tot_b, local_rc, hybrid, offset, length, out_doc, out_score, out_source, out_count, out_total, st, false);
}
int ssi_comm_exchange_hf(ss_comm* c, uint32_t nq, int n_lists, const ss_dev_list* L, const uint64_t* d_tot_a, const uint64_t* d_tot_b,
                         int local_rc, bool hybrid, uint32_t offset, uint32_t length, uint64_t* out_doc, float* out_score,
                         uint8_t* out_source, uint32_t* out_count, uint64_t* out_total, hipStream_t st, bool host_fuse) {
  // host_fuse (hybrid): the two sorted concatenations come home and ss_merge_results fuses them there -- a page deeper than SS_MAX_K, or
  // concatenations beyond the fusion kernel's 8192 LDS entries; the same f32 operations in the same order as rrf_merge_kernel's
  if (!c || n_lists < 0 || n_lists > 2 || (hybrid && n_lists != 2)) return SS_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  if (hipSetDevice(c->device) != hipSuccess) { (void)ncclCommAbort(c->comm); c->comm = nullptr; return SS_EDEVICE; }
  const uint32_t S = (uint32_t)c->n_ranks;
  // layout of one rank's contribution (words) and of the arena (bytes)
  size_t list_off[2] = {0, 0}, w = 0;
  for (int l = 0; l < n_lists; l++) {
    list_off[l] = w;
    w += 2 * (size_t)nq * L[l].k + nq;
    w += w & 1;  // the totals are u64
  }
  const size_t tot_off = w;
  w += 2 * ((size_t)nq + 1);
  const size_t W = w;
  size_t a = 0;
  const size_t o_send = a; a += al16(W * 4);
  const size_t o_recv = a; a += al16(W * 4 * S);
  size_t o_mdoc[2], o_msc[2], o_mcnt[2];
  uint32_t mlen[2] = {0, 0};
  for (int l = 0; l < n_lists; l++) {
    mlen[l] = hybrid ? S * L[l].k : L[l].k;  // RRF ranks run over the WHOLE concatenation (search.rs:1962-2035), a plain search keeps k
    o_mdoc[l] = a; a += al16((size_t)nq * mlen[l] * 8);
    o_msc[l] = a; a += al16((size_t)nq * mlen[l] * 4);
    o_mcnt[l] = a; a += al16((size_t)nq * 4);
  }
  const size_t o_tot = a; a += al16(((size_t)nq + 1) * 8);
  const size_t o_fdoc = a; a += hybrid ? al16((size_t)nq * length * 8) : 0;
  const size_t o_fsc = a; a += hybrid ? al16((size_t)nq * length * 4) : 0;
  const size_t o_fsrc = a; a += hybrid ? al16((size_t)nq * length) : 0;
  const size_t o_fcnt = a; a += hybrid ? al16((size_t)nq * 4) : 0;
  bool fatal = false;  // this rank cannot even enter the collective: abort the communicator so that the peers error out
  if (a > c->ws_cap) {
    if (hipStreamSynchronize(st) != hipSuccess) fatal = true;
    if (c->d_ws) (void)hipFree(c->d_ws);
    c->d_ws = nullptr; c->ws_cap = 0;
    if (!fatal && hipMalloc(&c->d_ws, a) != hipSuccess) fatal = true;
    if (!fatal) c->ws_cap = a;
  }
  if (fatal) { (void)ncclCommAbort(c->comm); c->comm = nullptr; return SS_ENOMEM; }
  uint32_t* d_send = (uint32_t*)(c->d_ws + o_send);
  uint32_t* d_recv = (uint32_t*)(c->d_ws + o_recv);
  if (local_rc != SS_OK) {
    if (hipMemsetAsync(d_send, 0, W * 4, st) != hipSuccess) fatal = true;
  } else {
    for (int l = 0; l < n_lists && !fatal; l++) {
      const size_t nk = (size_t)nq * L[l].k;
      comm_pack_lists_kernel<<<(uint32_t)((std::max<size_t>(nk, nq) + 255) / 256), 256, 0, st>>>(nq, L[l].k, L[l].doc, L[l].score, L[l].count,
                                                                                                d_send + list_off[l]);
    }
  }
  comm_pack_totals_kernel<<<(nq + 256) / 256, 256, 0, st>>>(nq, local_rc == SS_OK ? (const unsigned long long*)d_tot_a : nullptr,
                                                            local_rc == SS_OK ? (const unsigned long long*)d_tot_b : nullptr,
                                                            local_rc == SS_OK ? 0ull : 1ull, (unsigned long long*)(d_send + tot_off));
  if (hipGetLastError() != hipSuccess) fatal = true;
  if (fatal) { (void)ncclCommAbort(c->comm); c->comm = nullptr; return SS_EDEVICE; }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->prof_on && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, st);
  if (ncclAllGather(d_send, d_recv, W, ncclInt32, c->comm, st) != ncclSuccess) return SS_EDEVICE;
  if (e0 && e1) { (void)hipEventRecord(e1, st); c->prof_pending.emplace_back(e0, e1); }
  unsigned long long* d_tot = (unsigned long long*)(c->d_ws + o_tot);
  comm_sum_totals_kernel<<<(nq + 256) / 256, 256, 0, st>>>(nq, S, d_recv, W, tot_off, d_tot);
  SS_HIP(hipGetLastError());
  for (int l = 0; l < n_lists; l++) {
    const uint32_t* base = d_recv + list_off[l];
    const size_t nk = (size_t)nq * L[l].k;
    SS_TRY(ssi_topk_merge_launch(c->device, nq, S, L[l].k, base, (const float*)(base + nk), base + 2 * nk, W, W, mlen[l],
                                 (uint64_t*)(c->d_ws + o_mdoc[l]), (float*)(c->d_ws + o_msc[l]), (uint32_t*)(c->d_ws + o_mcnt[l]), st));
  }
  unsigned long long h_status = 0ull;
  if (hybrid && host_fuse) {
    std::vector<uint64_t> hd[2];
    std::vector<float> hsc[2];
    std::vector<uint32_t> hc[2];
    for (int l = 0; l < 2; l++) {
      hd[l].resize((size_t)nq * mlen[l]); hsc[l].resize((size_t)nq * mlen[l]); hc[l].resize(nq);
      SS_HIP(hipMemcpyAsync(hd[l].data(), c->d_ws + o_mdoc[l], hd[l].size() * 8, hipMemcpyDeviceToHost, st));
      SS_HIP(hipMemcpyAsync(hsc[l].data(), c->d_ws + o_msc[l], hsc[l].size() * 4, hipMemcpyDeviceToHost, st));
      SS_HIP(hipMemcpyAsync(hc[l].data(), c->d_ws + o_mcnt[l], (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    }
    SS_HIP(hipMemcpyAsync(out_total, d_tot, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
    SS_HIP(hipMemcpyAsync(&h_status, d_tot + nq, 8, hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
    if (local_rc != SS_OK) return local_rc;
    if (h_status) return SS_EPEER;
    for (uint32_t i = 0; i < nq; i++) {
      for (uint32_t r = 0; r < length; r++) { out_doc[(size_t)i * length + r] = ~0ull; out_score[(size_t)i * length + r] = 0.f; if (out_source) out_source[(size_t)i * length + r] = 0; }
      const int w = ss_merge_results(SS_MODE_HYBRID, hd[0].data() + (size_t)i * mlen[0], hsc[0].data() + (size_t)i * mlen[0], std::min(hc[0][i], mlen[0]),
                                     hd[1].data() + (size_t)i * mlen[1], hsc[1].data() + (size_t)i * mlen[1], std::min(hc[1][i], mlen[1]), offset, length,
                                     out_doc + (size_t)i * length, out_score + (size_t)i * length, out_source ? out_source + (size_t)i * length : nullptr);
      if (w < 0) return w;
      out_count[i] = (uint32_t)w;
    }
    return SS_OK;
  }
  if (hybrid) {
    SS_TRY(ss_rrf_merge_dev(c->device, nq, mlen[0], c->d_ws + o_mdoc[0], (const uint32_t*)(c->d_ws + o_mcnt[0]), mlen[1], c->d_ws + o_mdoc[1],
                            (const uint32_t*)(c->d_ws + o_mcnt[1]), 1, offset, length, (uint64_t*)(c->d_ws + o_fdoc), (float*)(c->d_ws + o_fsc),
                            (uint8_t*)(c->d_ws + o_fsrc), (uint32_t*)(c->d_ws + o_fcnt), (void*)st));
    SS_HIP(hipMemcpyAsync(out_doc, c->d_ws + o_fdoc, (size_t)nq * length * 8, hipMemcpyDeviceToHost, st));
    SS_HIP(hipMemcpyAsync(out_score, c->d_ws + o_fsc, (size_t)nq * length * 4, hipMemcpyDeviceToHost, st));
    if (out_source) SS_HIP(hipMemcpyAsync(out_source, c->d_ws + o_fsrc, (size_t)nq * length, hipMemcpyDeviceToHost, st));
    SS_HIP(hipMemcpyAsync(out_count, c->d_ws + o_fcnt, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
  } else if (n_lists == 1) {
    SS_HIP(hipMemcpyAsync(out_doc, c->d_ws + o_mdoc[0], (size_t)nq * mlen[0] * 8, hipMemcpyDeviceToHost, st));
    SS_HIP(hipMemcpyAsync(out_score, c->d_ws + o_msc[0], (size_t)nq * mlen[0] * 4, hipMemcpyDeviceToHost, st));
    SS_HIP(hipMemcpyAsync(out_count, c->d_ws + o_mcnt[0], (size_t)nq * 4, hipMemcpyDeviceToHost, st));
  }
  SS_HIP(hipMemcpyAsync(out_total, d_tot, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
  SS_HIP(hipMemcpyAsync(&h_status, d_tot + nq, 8, hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  if (local_rc != SS_OK) return local_rc;
  return h_status ? SS_EPEER : SS_OK;
}

extern "C" {

int ss_comm_profile(ss_comm* c, int on) {
  if (!c) return SS_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  c->prof_on = on != 0;
  return SS_OK;
}
int ss_comm_profile_read(ss_comm* c, uint64_t* collectives, double* total_us, int reset) {
  if (!c) return SS_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  SS_HIP(hipSetDevice(c->device));
  for (auto& e : c->prof_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
      c->prof_calls++;
      c->prof_us += (double)ms * 1e3;
    }
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  c->prof_pending.clear();
  if (collectives) *collectives = c->prof_calls;
  if (total_us) *total_us = c->prof_us;
  if (reset) { c->prof_calls = 0; c->prof_us = 0.0; }
  return SS_OK;
}

}  // extern "C"
