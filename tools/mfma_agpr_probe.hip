// Dev probe (round 6): does a dependent v_mfma_f32_32x32x16_bf16 chain cost more when its A operand comes from the ACCUMULATOR file?
// dwconv_mfma_wide_tri.hip keeps 320 fragment registers of which hipcc puts ~170 into a[...] and feeds them to the MFMA directly.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_agpr_probe tools/mfma_agpr_probe.hip && /tmp/mfma_agpr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// MODE 0: A, B in VGPRs, C/D in AGPRs.  1: A in AGPRs (a[32:35] ..), B in VGPRs.  2: A and B in AGPRs.  3: everything in VGPRs (C/D v[..]).
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, int reps) {
    u32x4 va = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, vb = va;
    asm volatile("v_accvgpr_write_b32 a32, %0\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_write_b32 a35, %0\n\t"
                 "v_accvgpr_write_b32 a36, %0\n\tv_accvgpr_write_b32 a37, %0\n\tv_accvgpr_write_b32 a38, %0\n\tv_accvgpr_write_b32 a39, %0" :: "v"(va[0])
                 : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            if constexpr (MODE == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(va), "v"(vb) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
            else if constexpr (MODE == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[0:15], a[32:35], %0, a[0:15]" :: "v"(vb) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
            else if constexpr (MODE == 2) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[0:15], a[32:35], a[36:39], a[0:15]" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
            else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 v[100:115], %0, %1, v[100:115]" :: "v"(va), "v"(vb) : "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const int reps = 2000;
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, d, reps);
        hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, d, reps);
        hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, d, reps);
        hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, d, reps);
        hipDeviceSynchronize();
    }
    unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    const char* names[4] = {"A,B in VGPRs, C/D in AGPRs", "A in AGPRs, B in VGPRs", "A and B in AGPRs", "everything in VGPRs"};
    for (int m = 0; m < 4; ++m) printf("%-30s %6.1f cycles (s_memtime ticks x ?) per MFMA  [raw %llu for %d]\n", names[m], (double)h[m] / (reps * 20.0), h[m], reps * 20);
    return 0;
}
