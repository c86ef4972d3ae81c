// PMC calibration microkernels: known byte counts in the access patterns of the BM25 probe kernel,
// so that FETCH_SIZE / TCC_EA0_RDREQ[_32B] can be turned into real bytes for THAT pattern instead of assuming the
// guide's x2 rule (calibrated only for 16 B/lane coalesced streams).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/pmc_calib.bin tools/probes/pmc_calib.hip
//   run  : rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -o x -- tools/probes/pmc_calib.bin
// Every kernel reads a region far larger than L2 + Infinity Cache (4 GiB) exactly once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// (a) 16 B per lane, fully coalesced stream: the guide's calibrated case (FETCH_SIZE reports half)
__global__ void calib_stream16(const uint4* __restrict__ p, u64 n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
    uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// (b) 4 B per lane, coalesced: one 256-byte run per wave load (the probe kernel's driver posting streams)
__global__ void calib_stream4(const uint32_t* __restrict__ p, u64 n4, uint32_t* sink) {
  uint32_t acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (u64)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) *sink = acc;
}
// (c) 8-byte gathers, every access in its own 128-byte line, lines visited in a scrambled order (the bit-record probes)
__global__ void calib_gather8(const uint2* __restrict__ p, u64 n_lines, u64 mul, uint32_t* sink) {
  uint32_t acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (u64)gridDim.x * blockDim.x) {
    const u64 line = (i * mul) % n_lines;  // mul coprime to n_lines: a permutation
    uint2 v = p[line * 16 + (i & 15)];
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// (d) 4-byte gathers, one per 128-byte line (rank table / posting fetches on hits)
__global__ void calib_gather4(const uint32_t* __restrict__ p, u64 n_lines, u64 mul, uint32_t* sink) {
  uint32_t acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (u64)gridDim.x * blockDim.x) {
    const u64 line = (i * mul) % n_lines;
    acc ^= p[line * 32 + (i & 31)];
  }
  if (acc == 0x12345678u) *sink = acc;
}
// (e) 8-byte gathers, one per 64-byte half line (two gathers share a 128-byte line but arrive far apart in time)
__global__ void calib_gather8_64(const uint2* __restrict__ p, u64 n_half, u64 mul, uint32_t* sink) {
  uint32_t acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_half; i += (u64)gridDim.x * blockDim.x) {
    const u64 h = (i * mul) % n_half;
    uint2 v = p[h * 8 + (i & 7)];
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// (f) 16-byte stores, coalesced (WRITE_SIZE calibration)
__global__ void calib_store16(uint4* __restrict__ p, u64 n16) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
  const u64 bytes = 4ull << 30;
  void* d = nullptr; uint32_t* sink = nullptr;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(d, 1, bytes));
  const u64 n_lines = bytes / 128, mul = 0x9E3779B1ull | 1ull;  // odd multiplier, n_lines a power of two -> permutation
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double real_bytes, auto launch) {
    launch();  // warm (instruction cache, page tables)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-18s useful/known bytes %.3f GB  %.3f ms  %.1f GB/s\n", name, real_bytes / 1e9, ms, real_bytes / ms / 1e6);
  };
  const int grid = 256 * 16, block = 256;
  timeit("stream16", (double)bytes, [&] { calib_stream16<<<grid, block>>>((const uint4*)d, bytes / 16, sink); });
  timeit("stream4", (double)bytes, [&] { calib_stream4<<<grid, block>>>((const uint32_t*)d, bytes / 4, sink); });
  timeit("gather8_line128", (double)n_lines * 8, [&] { calib_gather8<<<grid, block>>>((const uint2*)d, n_lines, mul, sink); });
  timeit("gather4_line128", (double)n_lines * 4, [&] { calib_gather4<<<grid, block>>>((const uint32_t*)d, n_lines, mul, sink); });
  timeit("gather8_half64", (double)n_lines * 2 * 8, [&] { calib_gather8_64<<<grid, block>>>((const uint2*)d, n_lines * 2, mul, sink); });
  timeit("store16", (double)bytes, [&] { calib_store16<<<grid, block>>>((uint4*)d, bytes / 16); });
  printf("lines128 %llu  region %llu bytes\n", n_lines, bytes);
  CK(hipFree(d)); CK(hipFree(sink));
  return 0;
}
