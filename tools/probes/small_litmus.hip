// Litmus probe for the hand-over protocol of bm25_small_kernel (seekstorm_amd/csrc/bm25_small.hip; VERDICT r5 weak 12 / "next" 3).
//
// What the product kernel does: the workgroups of a query each write a partition list (u64 keys) and raise the query's shared threshold
// (atomicMax without return), then ONE lane per workgroup adds 1 to the query's arrival counter; whoever reads PB + CB - 1 there is the
// last arriver, reads every list of the query, answers, and zeroes the query's state for the next launch.  The lists travel in RELAXED
// agent-scope atomic stores / loads (global_store / global_load with sc1 on gfx950: performed at the point the 8 XCDs' L2s agree on),
// ordered by `s_waitcnt vmcnt(0)` (a store / no-return atomic is counted until it is performed) before the relaxed arrival atomic --
// not by release / acquire fences, which at agent scope write back and invalidate the XCD's whole L2.
//
// This probe runs that protocol bare, millions of hand-overs, and checks every word the last arriver reads:
//   mode 0  the product's form: sc1 stores, s_waitcnt(0), relaxed arrival, sc1 loads
//   mode 1  the memory model's form: plain stores, arrival with ACQ_REL at agent scope, plain loads
//   mode 2  the BROKEN form, as a sensitivity check: plain stores, relaxed arrival, plain loads (nothing orders or publishes the lists --
//           if this form never fails either, the probe cannot tell the forms apart on this part and says so)
// The (query, list) -> workgroup mapping rotates with the launch number so that writers and the reader sit on different XCDs
// (blockIdx % 8) launch after launch, and every list line was last read through ANOTHER XCD's L2 one launch earlier.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/small_litmus tools/probes/small_litmus.hip && /tmp/small_litmus [launches] [queries] [lists]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WAVES = 8, KS = 64;

template <int MODE>
__global__ void __launch_bounds__(WAVES * 64) litmus_kernel(u64* lists, uint32_t* tau, uint32_t* arrive, u64* total, unsigned long long* errors,
                                                           uint32_t nq, uint32_t G, uint32_t seq) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // rotate: workgroup b serves (query, list) of b + seq * 13 -- another XCD every launch
  const uint32_t v = (blockIdx.x + seq * 13u) % (nq * G);
  const uint32_t qi = v % nq, g = v / nq;
  u64* mine = lists + ((size_t)qi * G + g) * KS;
  const u64 stamp = ((u64)seq << 32) | ((u64)g << 8);
  // every wave raises the query's threshold (no-return atomic), as pb_wave does while it probes
  if (lane == 0) atomicMax(&tau[qi * 16u], seq * 64u + (uint32_t)((g * WAVES + w) & 63u));
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (w != 0) return;
  if (MODE == 0) __hip_atomic_store(mine + lane, stamp | (u64)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else mine[lane] = stamp | (u64)lane;
  if (lane == 0) atomicAdd(&total[qi], (u64)(g + 1u));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0);
  uint32_t prev = 0;
  if (lane == 0) prev = MODE == 1 ? __hip_atomic_fetch_add(&arrive[qi], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) : atomicAdd(&arrive[qi], 1u);
  prev = __builtin_amdgcn_readfirstlane(prev);
  if (prev + 1u != G) return;
  // last arriver: every list of the query must carry this launch's stamp
  uint32_t bad = 0;
  for (uint32_t p = 0; p < G; p++) {
    const u64* l = lists + ((size_t)qi * G + p) * KS;
    const u64 x = MODE == 0 ? __hip_atomic_load(l + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : l[lane];
    bad += x != ((((u64)seq << 32) | ((u64)p << 8)) | (u64)lane);
  }
  const uint32_t t = __hip_atomic_load(&tau[qi * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 tot = __hip_atomic_load(&total[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane == 0) {
    bad += (t >> 6) != seq;                              // a threshold left over from an earlier launch, or one that has not landed
    bad += tot != (u64)G * (G + 1u) / 2u;                // a count that has not landed (or landed on top of the reset)
    tau[qi * 16u] = 0u; total[qi] = 0ull; arrive[qi] = 0u;  // the state the next launch expects
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
  if (lane == 0 && bad) atomicAdd(errors, (unsigned long long)bad);
}

template <int MODE>
static void run(const char* name, uint32_t launches, uint32_t nq, uint32_t G) {
  u64 *lists, *total;
  uint32_t *tau, *arrive;
  unsigned long long* errors;
  CK(hipMalloc(&lists, (size_t)nq * G * KS * 8)); CK(hipMemset(lists, 0, (size_t)nq * G * KS * 8));
  CK(hipMalloc(&total, nq * 8)); CK(hipMemset(total, 0, nq * 8));
  CK(hipMalloc(&tau, nq * 64)); CK(hipMemset(tau, 0, nq * 64));
  CK(hipMalloc(&arrive, nq * 4)); CK(hipMemset(arrive, 0, nq * 4));
  CK(hipMalloc(&errors, 8)); CK(hipMemset(errors, 0, 8));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t s = 1; s <= launches; s++) {
    litmus_kernel<MODE><<<nq * G, WAVES * 64, 0, st>>>(lists, tau, arrive, total, errors, nq, G, s);
    if ((s & 1023u) == 0) CK(hipStreamSynchronize(st));  // (bounded queue depth)
  }
  CK(hipStreamSynchronize(st));
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  unsigned long long h = 0;
  CK(hipMemcpy(&h, errors, 8, hipMemcpyDeviceToHost));
  printf("%-58s %8u launches x %3u queries x %2u lists: %12llu hand-overs, %llu bad words, %.2f us per launch\n", name, launches, nq, G,
         (unsigned long long)launches * nq, h, us / launches);
  fflush(stdout);
  CK(hipFree(lists)); CK(hipFree(total)); CK(hipFree(tau)); CK(hipFree(arrive)); CK(hipFree(errors));
  CK(hipStreamDestroy(st));
}

int main(int argc, char** argv) {
  const uint32_t launches = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000u;
  const uint32_t nq = argc > 2 ? (uint32_t)atoi(argv[2]) : 64u, G = argc > 3 ? (uint32_t)atoi(argv[3]) : 8u;
  for (uint32_t g : {G, 33u, 2u}) {  // (33 lists: not a multiple of the 8 XCDs -- the rotation then walks every pairing)
    run<0>("mode 0 (product: sc1 stores, s_waitcnt, relaxed arrival)", launches, nq, g);
    run<1>("mode 1 (plain stores, ACQ_REL arrival at agent scope)", launches, nq, g);
    run<2>("mode 2 (BROKEN on purpose: plain stores, relaxed arrival)", launches, nq, g);
  }
  return 0;
}
