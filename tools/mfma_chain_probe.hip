// Dev probe: cycles per v_mfma_f32_32x32x16_bf16 for the instruction patterns the conv kernels could use -- one wave per SIMD,
// B operand from LDS (ds_read_b128, row-per-lane, conflict-free pitch), A operand in registers.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain_probe tools/mfma_chain_probe.hip && tools/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define SB __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float* sink, int reps) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int o = tid * 16; o < 32768; o += 256 * 16) *(u32x4*)(lds + o) = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    u32x4 a[4];
    for (int i = 0; i < 4; ++i) a[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    const char* base = lds + wave * 8192 + (lane & 31) * 112 + (lane >> 5) * 16;
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (MODE == 0) {                     // 20 dependent MFMAs, operands in registers
#pragma unroll
            for (int j = 0; j < 20; ++j) { c0 = mf(a[j & 3], a[(j + 1) & 3], c0); SB; }
        } else if constexpr (MODE == 1) {              // dependent chain, one ds_read_b128 after each MFMA (5 in flight): the kernels' pattern
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
        } else if constexpr (MODE == 2) {              // two accumulators alternating, one load after each MFMA
            u32x4 b[6];
#pragma unroll
            for (int j = 0; j < 5; ++j) b[j] = *(const u32x4*)(base + j * 112);
            SB;
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                if (j & 1) c1 = mf(a[j & 3], b[j % 6], c1); else c0 = mf(a[j & 3], b[j % 6], c0);
                if (j + 5 < 20) { b[(j + 5) % 6] = *(const u32x4*)(base + ((j + 5) % 5) * 112 + ((j + 5) / 5) * 32); asm volatile("" :: "v"(b[j % 6])); }
                SB;
            }
        } else if constexpr (MODE == 3) {              // four accumulators round-robin, one load after each MFMA
            u32x4 b[6];
#pragma unroll
            for (int j = 0; j < 5; ++j) b[j] = *(const u32x4*)(base + j * 112);
            SB;
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                if ((j & 3) == 0) c0 = mf(a[j & 3], b[j % 6], c0); else if ((j & 3) == 1) c1 = mf(a[j & 3], b[j % 6], c1);
                else if ((j & 3) == 2) c2 = mf(a[j & 3], b[j % 6], c2); else c3 = mf(a[j & 3], b[j % 6], c3);
                if (j + 5 < 20) { b[(j + 5) % 6] = *(const u32x4*)(base + ((j + 5) % 5) * 112 + ((j + 5) / 5) * 32); asm volatile("" :: "v"(b[j % 6])); }
                SB;
            }
        } else if constexpr (MODE == 4) {              // batched: 5 loads (next k-step), then 5 back-to-back dependent MFMAs, double-buffered
            u32x4 b[2][5];
#pragma unroll
            for (int j = 0; j < 5; ++j) b[0][j] = *(const u32x4*)(base + j * 112);
            SB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) b[(ks + 1) & 1][j] = *(const u32x4*)(base + j * 112 + (ks + 1) * 32);
                }
                SB;
#pragma unroll
                for (int j = 0; j < 5; ++j) c0 = mf(a[j & 3], b[ks & 1][j], c0);
                SB;
            }
        } else if constexpr (MODE == 5) {              // batched as 4, two accumulators
            u32x4 b[2][5];
#pragma unroll
            for (int j = 0; j < 5; ++j) b[0][j] = *(const u32x4*)(base + j * 112);
            SB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) b[(ks + 1) & 1][j] = *(const u32x4*)(base + j * 112 + (ks + 1) * 32);
                }
                SB;
#pragma unroll
                for (int j = 0; j < 5; ++j) { if (j & 1) c1 = mf(a[j & 3], b[ks & 1][j], c1); else c0 = mf(a[j & 3], b[ks & 1][j], c0); }
                SB;
            }
        } else if constexpr (MODE == 6) {              // no pinning at all: let hipcc schedule loads + dependent MFMAs
            u32x4 b[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) b[j] = *(const u32x4*)(base + (j % 5) * 112 + (j / 5) * 32);
#pragma unroll
            for (int j = 0; j < 20; ++j) c0 = mf(a[j & 3], b[j], c0);
        } else if constexpr (MODE == 7) {              // 20 independent-pair MFMAs from registers (two accumulators), no loads
#pragma unroll
            for (int j = 0; j < 20; ++j) { if (j & 1) c1 = mf(a[j & 3], a[(j + 1) & 3], c1); else c0 = mf(a[j & 3], a[(j + 1) & 3], c0); SB; }
        } else if constexpr (MODE == 8) {              // dependent chain; the load sits BEFORE the MFMA that frees nothing: load j+5, then MFMA j (load first)
            u32x4 b[6];
#pragma unroll
            for (int j = 0; j < 5; ++j) b[j] = *(const u32x4*)(base + j * 112);
            SB;
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                if (j + 5 < 20) b[(j + 5) % 6] = *(const u32x4*)(base + ((j + 5) % 5) * 112 + ((j + 5) / 5) * 32);
                c0 = mf(a[j & 3], b[j % 6], c0);
                SB;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 256 * 256 * 4);
    hipMemset(out, 0, 64 * 8);
    const int reps = 200;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, sink, reps); hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, sink, reps);
        hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, out, sink, reps); hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, out, sink, reps);
        hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 0, 0, out, sink, reps); hipLaunchKernelGGL(probe<5>, dim3(256), dim3(256), 0, 0, out, sink, reps);
        hipLaunchKernelGGL(probe<6>, dim3(256), dim3(256), 0, 0, out, sink, reps); hipLaunchKernelGGL(probe<7>, dim3(256), dim3(256), 0, 0, out, sink, reps);
        hipLaunchKernelGGL(probe<8>, dim3(256), dim3(256), 0, 0, out, sink, reps);
        hipDeviceSynchronize();
    }
    unsigned long long h[64]; hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"0 dependent chain, registers only", "1 dependent chain + load after each MFMA (kernels today)", "2 two accumulators + load after each",
                           "3 four accumulators + load after each", "4 batched 5 loads / 5 dependent MFMAs", "5 batched, two accumulators",
                           "6 unpinned (hipcc schedules)", "7 two accumulators, registers only", "8 dependent chain, load BEFORE each MFMA"};
    for (int m = 0; m < 9; ++m) printf("mode %-60s %7.1f cycles per MFMA (s_memtime/readcyclecounter units)\n", names[m], (double)h[m] / (reps * 20.0));
    return 0;
}
