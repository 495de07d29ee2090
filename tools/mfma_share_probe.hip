// Dev probe: how two waves of one SIMD share the MFMA pipe.  v_mfma_f32_32x32x16_bf16 dependent chains, B operand from registers (mode 0) or
// one ds_read_b128 behind each MFMA (mode 1: the conv kernels' stream), with 4 waves per CU (one per SIMD) and 8 (two per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_share_probe tools/mfma_share_probe.hip && tools/mfma_share_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define SB __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }

template <int MODE, int NT>
__global__ __launch_bounds__(NT, 1) void probe(unsigned long long* out, float* sink, int reps, int slot) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int o = tid * 16; o < 65536; o += NT * 16) *(u32x4*)(lds + o) = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    u32x4 a[4];
    for (int i = 0; i < 4; ++i) a[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    const char* base = lds + wave * 8192 + (lane & 31) * 112 + (lane >> 5) * 16;
    f32x16 c0;
    for (int i = 0; i < 16; ++i) c0[i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 20; ++j) { c0 = mf(a[j & 3], a[(j + 1) & 3], c0); SB; }
        } else {
            u32x4 b[6];
#pragma unroll
            for (int j = 0; j < 5; ++j) b[j] = *(const u32x4*)(base + j * 112);
            SB;
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                c0 = mf(a[j & 3], b[j % 6], c0);
                if (j + 5 < 20) { b[(j + 5) % 6] = *(const u32x4*)(base + ((j + 5) % 5) * 112 + ((j + 5) / 5) * 32); asm volatile("" :: "v"(b[j % 6])); }
                SB;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i];
    sink[blockIdx.x * NT + tid] = s;
    if (lane == 0 && blockIdx.x == 0) out[slot * 8 + wave] = t1 - t0;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(out, 0, 64 * 8);
    const int reps = 200;
    hipFuncSetAttribute((const void*)probe<0, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<1, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<0, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<1, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL((probe<0, 256>), dim3(256), dim3(256), 65536, 0, out, sink, reps, 0);
        hipLaunchKernelGGL((probe<1, 256>), dim3(256), dim3(256), 65536, 0, out, sink, reps, 1);
        hipLaunchKernelGGL((probe<0, 512>), dim3(256), dim3(512), 65536, 0, out, sink, reps, 2);
        hipLaunchKernelGGL((probe<1, 512>), dim3(256), dim3(512), 65536, 0, out, sink, reps, 3);
        hipDeviceSynchronize();
    }
    unsigned long long h[64]; hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"registers only, 4 waves / CU", "ds_read_b128 behind each MFMA, 4 waves / CU", "registers only, 8 waves / CU", "ds_read_b128 behind each MFMA, 8 waves / CU"};
    for (int m = 0; m < 4; ++m) {
        printf("%-46s cycles per MFMA per wave:", names[m]);
        for (int w = 0; w < (m < 2 ? 4 : 8); ++w) printf(" %6.1f", (double)h[m * 8 + w] / (reps * 20.0));
        printf("\n");
    }
    return 0;
}
