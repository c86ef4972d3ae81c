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
  uint64_t* d_mdoc = nullptr;  // merged lists of ss_*_search_sharded: [nq][k] global ids | scores | counts | summed totals
  float* d_mscore = nullptr;
  uint32_t* d_mcount = nullptr;
  uint64_t* d_mtotal = nullptr;
  size_t cap_m = 0, cap_mq = 0;
  std::mutex mu;
};

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
  for (int i = 0; i < n_devices; i++) {
    ss_comm* c = new (std::nothrow) ss_comm;
    if (!c) return SS_ENOMEM;
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
  if (c->d_mdoc) (void)hipFree(c->d_mdoc);
  if (c->d_mscore) (void)hipFree(c->d_mscore);
  if (c->d_mcount) (void)hipFree(c->d_mcount);
  if (c->d_mtotal) (void)hipFree(c->d_mtotal);
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
  if (k == 0 || (uint64_t)c->n_ranks * k > 8192) return SS_EINVAL;
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

// ss_bm25_search_sharded's exchange: the shard's device lists -> merged lists + summed totals on the host of every rank
// (result_count_total is summed over the shards, search.rs:1884-1921).  k = 0: totals only (ResultType::Count).
int ssi_comm_exchange_to_host(ss_comm* c, uint32_t nq, uint32_t k, const uint32_t* d_doc, const float* d_score, const uint32_t* d_count,
                              const uint64_t* d_total, uint64_t* out_doc, float* out_score, uint32_t* out_count, uint64_t* out_total,
                              hipStream_t st) {
  if (!c) return SS_EINVAL;
  SS_HIP(hipSetDevice(c->device));
  {
    std::lock_guard<std::mutex> g(c->mu);
    const size_t nk = (size_t)nq * std::max<uint32_t>(k, 1);
    if (nk > c->cap_m || nq > c->cap_mq) {
      SS_HIP(hipStreamSynchronize(st));
      if (c->d_mdoc) (void)hipFree(c->d_mdoc);
      if (c->d_mscore) (void)hipFree(c->d_mscore);
      if (c->d_mcount) (void)hipFree(c->d_mcount);
      if (c->d_mtotal) (void)hipFree(c->d_mtotal);
      c->d_mdoc = nullptr; c->d_mscore = nullptr; c->d_mcount = nullptr; c->d_mtotal = nullptr;
      c->cap_m = c->cap_mq = 0;
      SS_HIP(hipMalloc(&c->d_mdoc, nk * sizeof(uint64_t)));
      SS_HIP(hipMalloc(&c->d_mscore, nk * sizeof(float)));
      SS_HIP(hipMalloc(&c->d_mcount, (size_t)nq * sizeof(uint32_t)));
      SS_HIP(hipMalloc(&c->d_mtotal, (size_t)nq * sizeof(uint64_t)));
      c->cap_m = nk; c->cap_mq = nq;
    }
  }
  SS_NCCL(ncclAllReduce(d_total, c->d_mtotal, nq, ncclUint64, ncclSum, c->comm, st));
  if (k) {
    int rc = ss_topk_allgather_merge(c, nq, k, d_doc, d_score, d_count, c->d_mdoc, c->d_mscore, c->d_mcount, (void*)st);
    if (rc) return rc;
    SS_HIP(hipMemcpyAsync(out_doc, c->d_mdoc, (size_t)nq * k * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(hipMemcpyAsync(out_score, c->d_mscore, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
    SS_HIP(hipMemcpyAsync(out_count, c->d_mcount, (size_t)nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  SS_HIP(hipMemcpyAsync(out_total, c->d_mtotal, (size_t)nq * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}
