// VALU issue-rate probe for gfx950: decides the MAC formulation of the depthwise-conv kernels.
// Measures lane-MACs/s for v_fma_f32 (SGPR weight), v_pk_fma_f32, v_dot2c_f32_bf16, v_dot2c_f32_f16.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
#define NACC 16
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template<int MODE>
__global__ __launch_bounds__(256) void probe(float* out, const float* wts, int iters) {
  float acc[NACC]; f32x2 acc2[NACC];
  float xv = threadIdx.x * 1e-3f;
  unsigned xp = __float_as_uint(xv) | 0x3f803f80u;
#pragma unroll
  for (int i = 0; i < NACC; ++i) { acc[i] = i; acc2[i] = f32x2{(float)i, (float)i}; }
  for (int it = 0; it < iters; ++it) {
    // 8 wave-uniform weights per iteration (scalar loads -> SGPR operands)
    const float* w = wts + (it & 7) * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float wt = w[t];
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(wt), "v"(xv));
        else if constexpr (MODE == 1) acc2[i] = __builtin_elementwise_fma(f32x2{wt, wt}, f32x2{xv, xv}, acc2[i]);
        else if constexpr (MODE == 2) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, __float_as_uint(wt)), __builtin_bit_cast(bf16x2, xp), acc[i], false);
        else if constexpr (MODE == 3) acc[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, __float_as_uint(wt)), __builtin_bit_cast(f16x2, xp), acc[i], false);
      }
      xv += 1e-6f; xp ^= (t + 1);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i] + acc2[i].x + acc2[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<int MODE> double run(const char* name, int macs_per_instr, float* out, float* wts) {
  const int iters = 4096, blocks = 256 * 8, threads = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, threads>>>(out, wts, 64);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a); probe<MODE><<<blocks, threads>>>(out, wts, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  double instr = (double)blocks * threads * iters * 8.0 * NACC;       // lane-instructions
  double tmacs = instr * macs_per_instr / (best * 1e-3) / 1e12;
  printf("%-22s %8.3f ms  %7.2f T lane-instr/s  %7.2f TMAC/s  (%6.1f TFLOP/s)\n", name, best, instr / (best*1e-3) / 1e12, tmacs, 2 * tmacs);
  return tmacs;
}
int main() {
  float *out, *wts; CHECK(hipMalloc(&out, 256*8*256*4)); CHECK(hipMalloc(&wts, 64*4));
  std::vector<float> h(64); for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 1e-3f;
  CHECK(hipMemcpy(wts, h.data(), 256, hipMemcpyHostToDevice));
  run<0>("v_fma_f32 (sgpr w)", 1, out, wts);
  run<1>("v_pk_fma_f32", 2, out, wts);
  run<2>("v_dot2c_f32_bf16", 2, out, wts);
  run<3>("v_dot2c_f32_f16", 2, out, wts);
  return 0;
}
