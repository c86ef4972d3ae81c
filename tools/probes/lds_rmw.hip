// Probe: cost of scattered LDS read-modify-write on gfx950 -- (a) ds_read_u16 + ds_write_b16, (b) ds_add_rtn_u32 on the
// dword holding the u16, (c) ds_add_u32 without return.  8 KB tile per wave, 16 waves per CU resident, random docs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) uint16_t l16;
typedef __attribute__((address_space(3))) uint32_t l32;
template <int MODE>
__global__ void __launch_bounds__(512) k(const uint32_t* __restrict__ docs, uint32_t iters, uint32_t* out) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t wb = w * 9216;
  for (int i = lane; i < 9216 / 4; i += 64) *(l32*)(uintptr_t)(wb + i * 4) = 0;
  uint32_t d[8];
  for (int j = 0; j < 8; j++) d[j] = docs[(blockIdx.x * 512 + threadIdx.x) * 8 + j] & 4095u;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t doc = (d[j] + it * 37u) & 4095u;
      if (MODE == 0) {
        const uint32_t a = wb + doc * 2;
        const uint32_t o = *(l16*)(uintptr_t)a;
        *(l16*)(uintptr_t)a = (uint16_t)(o + 3);
        acc = max(acc, o);
      } else if (MODE == 1) {
        const uint32_t a = wb + (doc >> 1) * 4;
        const uint32_t o = __hip_atomic_fetch_add((l32*)(uintptr_t)a, 3u << ((doc & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        acc = max(acc, (o >> ((doc & 1) * 16)) & 0xFFFFu);
      } else {
        const uint32_t a = wb + (doc >> 1) * 4;
        __hip_atomic_fetch_add((l32*)(uintptr_t)a, 3u << ((doc & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}
int main() {
  const int blocks = 512, iters = 2000;
  uint32_t* h = (uint32_t*)malloc(blocks * 512 * 8 * 4);
  uint32_t x = 12345;
  for (int i = 0; i < blocks * 512 * 8; i++) { x = x * 1664525u + 1013904223u; h[i] = x >> 8; }
  uint32_t *d, *o;
  hipMalloc(&d, blocks * 512 * 8 * 4); hipMalloc(&o, 4);
  hipMemcpy(d, h, blocks * 512 * 8 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, blocks, 512, 8 * 9216, 0, d, iters, o);
      if (mode == 1) hipLaunchKernelGGL(k<1>, blocks, 512, 8 * 9216, 0, d, iters, o);
      if (mode == 2) hipLaunchKernelGGL(k<2>, blocks, 512, 8 * 9216, 0, d, iters, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // wave-ops: blocks * 8 waves * iters * 8; per CU: / 256
      const double ops_per_cu = (double)blocks * 8 * iters * 8 / 256.0;
      if (rep) printf("mode %d (%s): %.3f ms  -> %.1f ns per wave-level RMW per CU (%.1f cycles at 2.1 GHz)\n", mode,
                      mode == 0 ? "ds_read_u16 + ds_write_b16" : mode == 1 ? "ds_add_rtn_u32" : "ds_add_u32", ms, ms * 1e6 / ops_per_cu, ms * 1e6 / ops_per_cu * 2.1);
    }
  }
  return 0;
}
