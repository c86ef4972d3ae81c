// Does a short kernel on another stream get onto the device while a long grid of LDS-heavy workgroups (the shape of the f32 vector scan:
// 4 waves, 72 KB of LDS, ~80 K workgroups) is running -- and does the stream's priority matter?  (round 6, VERDICT r5 weak 8: a hybrid
// caller's lexical half waited 8 ms behind a coalesced vector pass.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_prio tools/probes/stream_prio.hip && /tmp/stream_prio
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) long_kernel(float* out, int iters) {
  extern __shared__ float sm[];
  float a = threadIdx.x;
  for (int i = 0; i < iters; i++) { sm[(threadIdx.x + i) & 1023] = a; a = a * 1.0001f + sm[(threadIdx.x * 7 + i) & 1023]; }
  if (a == 12345.f) out[blockIdx.x] = a;
}
__global__ void __launch_bounds__(512) short_kernel(float* out, volatile unsigned* flag, unsigned seq) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = sm[5]; __threadfence_system(); *flag = seq; }
}

int main() {
  float* d = nullptr;
  CK(hipMalloc(&d, 1 << 24));
  unsigned* flag = nullptr;
  CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("priority range: least %d greatest %d\n", lo, hi);
  hipStream_t sa, sb_norm, sb_hi, sa_lo;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb_norm, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&sb_hi, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&sa_lo, hipStreamNonBlocking, lo));
  CK(hipFuncSetAttribute((const void*)long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
  // calibrate: 80 K workgroups of ~25 us
  for (int rep = 0; rep < 2; rep++) { long_kernel<<<80000, 256, 73728, sa>>>(d, 2000); CK(hipStreamSynchronize(sa)); }
  auto t0 = std::chrono::steady_clock::now();
  long_kernel<<<80000, 256, 73728, sa>>>(d, 2000);
  CK(hipStreamSynchronize(sa));
  const double long_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  printf("long kernel alone: %.2f ms\n", long_ms);
  struct Case { const char* name; hipStream_t la, sh; } cases[] = {{"same stream", sa, sa}, {"other stream, both default priority", sa, sb_norm},
                                                                   {"short on HIGH priority stream", sa, sb_hi}, {"long on LOW, short on HIGH", sa_lo, sb_hi}};
  unsigned seq = 0;
  for (auto& c : cases) {
    double worst = 0, sum = 0;
    for (int rep = 0; rep < 5; rep++) {
      long_kernel<<<80000, 256, 73728, c.la>>>(d, 2000);
      std::this_thread::sleep_for(std::chrono::microseconds(500));  // the long grid is well under way
      *flag = 0; seq++;
      auto a = std::chrono::steady_clock::now();
      short_kernel<<<512, 512, 55296, c.sh>>>(d, flag, seq);
      while (*(volatile unsigned*)flag != seq) {}
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
      worst = us > worst ? us : worst; sum += us;
      CK(hipDeviceSynchronize());
    }
    printf("%-42s short kernel (512 WGs x 512 threads, 55 KB LDS) done after %8.1f us mean, %8.1f us worst\n", c.name, sum / 5, worst);
  }
  return 0;
}
