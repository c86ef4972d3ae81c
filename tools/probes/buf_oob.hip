// Probe: raw buffer load range checking on gfx950 -- is soffset excluded from the bounds check, are OOB lanes zero,
// is the check per dword or per access?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* post, uint32_t b0_bytes, uint32_t nrec_bytes, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)post, 0, nrec_bytes, 0x00020000);
  u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, b0_bytes, 0);
  out[lane * 4 + 0] = v0.x; out[lane * 4 + 1] = v0.y; out[lane * 4 + 2] = v0.z; out[lane * 4 + 3] = v0.w;
}
int main() {
  const int N = 4096;
  uint32_t h[N];
  for (int i = 0; i < N; i++) h[i] = 1000 + i;
  uint32_t *d, *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  uint32_t res[256];
  struct { uint32_t b0, nrec; } cases[] = {{0, 160}, {4096, 4096 + 160}, {4096, 4096 + 168}, {4096, 4096}, {8192, 8192 + 16 * 64}};
  for (auto c : cases) {
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, c.b0, c.nrec, o);
    hipMemcpy(res, o, sizeof(res), hipMemcpyDeviceToHost);
    int last_nz = -1; for (int i = 0; i < 256; i++) if (res[i]) last_nz = i;
    printf("soffset=%u num_records=%u: first=%u last_nonzero_dword=%d (value %u)\n", c.b0, c.nrec, res[0], last_nz, last_nz >= 0 ? res[last_nz] : 0);
  }
  return 0;
}
