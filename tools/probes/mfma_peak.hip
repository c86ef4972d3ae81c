// What the matrix pipe sustains on this box, without memory: v_mfma_f32_32x32x2_f32 in a register-only loop (NACC independent
// accumulators per wave, W waves per SIMD), timed with HIP events; the shader clock from s_memtime against the 100 MHz s_memrealtime.
// The f32 vector scan's roofline denominator (157.3 TFLOP/s) assumes 2.4 GHz and a new MFMA every 64 cycles.
// KIND 2: the same loop over 16 A and 32 B registers of random per-lane data (the scan's operand pattern: every MFMA sees new
// operands) -- constant operands toggle nothing in the multipliers, and the clock the chip holds depends on that.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_peak.bin tools/probes/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  f32x4 acc4[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
    s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

__device__ __forceinline__ float rnd(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(int32_t)x * (1.0f / 2147483648.0f);
}
__global__ void __launch_bounds__(256) krand(float* out, unsigned long long* clk, int iters) {
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float xa[16], q0[16], q1[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    xa[i] = rnd(threadIdx.x * 64 + i + blockIdx.x * 7919);
    q0[i] = rnd(threadIdx.x * 64 + 16 + i + blockIdx.x * 104729);
    q1[i] = rnd(threadIdx.x * 64 + 32 + i + blockIdx.x * 1299709);
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u], q0[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u], q1[u], acc1, 0, 0, 0);
    }
    if ((it & 31) == 31) {  // keep the sums finite (a tile of the scan ends every 24 chunks)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc0[r] *= 0.001f; acc1[r] *= 0.001f; }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
void run_rand(int wgs_per_cu, int iters) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMallocManaged(&clk, 16);
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    krand<<<grid, 256>>>(out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 32 * (32.0 * 32 * 2 * 2);
    printf("%-28s waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s  frac of 157.3: %.3f  shader clock %.0f MHz\n", "32x32x2 f32, random operands", wgs_per_cu, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3, (double)clk[0] / ((double)clk[1] / 100.0));
  }
  hipFree(out); hipFree(clk);
}

template <int NACC, int KIND>
void run(const char* name, int wgs_per_cu, int iters) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMallocManaged(&clk, 16);
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k<NACC, KIND><<<grid, 256>>>(out, clk, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_mfma = KIND == 0 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2;
    const double flops = (double)grid * 4 * iters * 16 * NACC * flop_per_mfma;
    const double mhz = (double)clk[0] / ((double)clk[1] / 100.0);  // shader cycles per microsecond
    printf("%-28s waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s  frac of 157.3: %.3f  shader clock %.0f MHz  cycles per MFMA and SIMD %.1f\n", name, wgs_per_cu, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3, mhz, (double)clk[0] / ((double)iters * 16 * NACC * wgs_per_cu));
  }
  hipFree(out); hipFree(clk);
}
int main() {
  const int it = 4000;
  run_rand(2, it);
  run_rand(2, it * 4);
  run<2, 0>("32x32x2 f32, 2 accumulators", 1, it);
  run<2, 0>("32x32x2 f32, 2 accumulators", 2, it);
  run<4, 0>("32x32x2 f32, 4 accumulators", 1, it);
  run<4, 0>("32x32x2 f32, 4 accumulators", 2, it);
  run<4, 1>("16x16x4 f32, 4 accumulators", 2, it * 2);
  run<8, 1>("16x16x4 f32, 8 accumulators", 2, it);
  return 0;
}
