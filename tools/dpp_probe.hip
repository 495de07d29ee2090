// tools/dpp_probe.hip -- issue rate of DPP lane shifts on gfx950: row_shr:1 vs wave_shr:1 vs v_permlane32_swap vs plain v_add.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < iters; ++i) {
#define STEP(a) \
    if (MODE == 0) a = a + 1.0f; \
    else if (MODE == 1) a = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x111, 0xf, 0xf, false)); \
    else if (MODE == 2) a = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138, 0xf, 0xf, false)); \
    else if (MODE == 3) a = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x130, 0xf, 0xf, false));
    STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run(const char* name, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;
  k<MODE><<<256 * 4, 256>>>(d, iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<256 * 4, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 blocks/CU * 4 waves / 4 SIMDs = 4 waves per SIMD, each 8*iters ops
  double ops_per_simd = 4.0 * 8 * iters;
  printf("%-12s %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / ops_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
  run<0>("v_add", d); run<1>("row_shr:1", d); run<2>("wave_shr:1", d); run<3>("wave_shl:1", d);
  return 0;
}
