// slak_amd/csrc/stem_conv.hip -- forward of the stem convolution (Conv2d(3, C, kernel_size=4, stride=4), models/SLaK.py:189-193) under bf16
// autocast, straight from the fp32 image: patch extraction, the product with the weight and the bias in ONE pass, the patch matrix written
// on the way for the weight gradient (slak_stem_wgrad).
//     a[n][p][k]  = bf16(x[n][ci][4 ph + kh][4 pw + kw]),  k = ci*16 + kh*4 + kw, p = ph*Wo + pw          (what slak_stem_patchify writes)
//     y[n][co][p] = bf16(bf16(b[co]) + sum_k bf16(w[co][k]) * a[n][p][k])                                   (fp32 accumulate, one rounding)
// Before: slak_stem_patchify (32 us) + a batched library GEMM with K = 48 (macro tile 64x96x32: 132 us at N = 128, 224 px) + two cast kernels
// for the weight and the bias.  The work is 192 MB of traffic (x 77 read, a 38 + y 77 written) against 3.7 GFLOP: HBM bound.
//
// A unit = 64 consecutive pixels of one image, owned by one wavefront (no workgroup barriers):
//   * lane (l31, lhi) loads, for the pixels l31 and 32 + l31 of the unit, the k-half [24 lhi, 24 lhi + 24): six (ci, kh) rows of four floats,
//     16 bytes each, consecutive lanes on consecutive patches (512 contiguous bytes per image row and half wave);
//   * rounded to bf16 these 24 values are 48 contiguous bytes of a (three 16-byte stores per pixel) AND the lane's MFMA operand: the MFMA
//     reduction index (step ks, lane half lhi, element j) stands for k = 24 lhi + 8 ks + j -- the weight fragments are loaded once per wave in
//     the same order, so no value crosses lanes;
//   * D[co][pixel] = W (M = co, three or four 32-row tiles, resident in registers) x patches (N = 32 pixels), accumulators start at the bias;
//   * the results go through a wave-private LDS tile [co][64 pixels] so that y is written with 16-byte stores along the pixels (NCHW);
//   * the loads of unit i+1 are issued before the MFMAs of unit i.
#include "mfma_common.h"

namespace slak {

constexpr int SC_WAVES = 4;
constexpr int SC_RS = 144;              // LDS row stride in bytes (64 pixels x 2 + 16: rows shift by 36 banks)

__device__ __forceinline__ uint32_t sc_pack2(float a, float b) { return pack2<bf16_t>(a, b); }

template <int MT>
__global__ __launch_bounds__(SC_WAVES * 64) void stem_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                     uint16_t* __restrict__ a, uint16_t* __restrict__ y,
                                                                     int H, int W, int units, int upw) {
    constexpr int Co = MT * 32, K = 48, CI = 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    float* bl = (float*)smem;                                               // bias rounded to bf16, as fp32 [Co]
    unsigned char* tile = smem + Co * 4 + (size_t)wave * Co * SC_RS;
    for (int c = threadIdx.x; c < Co; c += SC_WAVES * 64) bl[c] = bias ? __uint_as_float((uint32_t)f32_to_bf16_bits(bias[c]) << 16) : 0.f;
    __syncthreads();
    const int Wo = W / 4, P = (H / 4) * Wo, upi = P / 64;
    // weight fragments: lane (co = 32 m + l31, lhi), step ks: w[co][24 lhi + 8 ks + j], j = 0..7
    s16x8 wf[MT][3];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const float4* wr = (const float4*)(w + (size_t)(32 * m + l31) * K + 24 * lhi + 8 * ks);
            const float4 v0 = wr[0], v1 = wr[1];
            u32x4 u; u[0] = sc_pack2(v0.x, v0.y); u[1] = sc_pack2(v0.z, v0.w); u[2] = sc_pack2(v1.x, v1.y); u[3] = sc_pack2(v1.z, v1.w);
            wf[m][ks] = __builtin_bit_cast(s16x8, u);
        }
    const int u0 = (blockIdx.x * SC_WAVES + wave) * upw;
    int u1 = u0 + upw; if (u1 > units) u1 = units;
    float4 nx[2][6];                                                        // next unit: [pixel l31 | pixel 32 + l31][row 6 lhi + i]
    auto issue = [&](int u) {
        const int n = u / upi, p0 = (u - n * upi) * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = p0 + 32 * t + l31, ph = p / Wo, pw = p - ph * Wo;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int r = 6 * lhi + i, ci = r >> 2, kh = r & 3;
                nx[t][i] = *(const float4*)(x + (((size_t)n * CI + ci) * H + 4 * ph + kh) * W + 4 * pw);
            }
        }
    };
    if (u0 < u1) issue(u0);
    for (int u = u0; u < u1; ++u) {
        const int n = u / upi, p0 = (u - n * upi) * 64;
        // round to bf16: the lane's 24 values per pixel = 3 MFMA fragments = 48 contiguous bytes of a
        s16x8 bf[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint32_t d[12];
#pragma unroll
            for (int i = 0; i < 6; ++i) { d[2 * i] = sc_pack2(nx[t][i].x, nx[t][i].y); d[2 * i + 1] = sc_pack2(nx[t][i].z, nx[t][i].w); }
            u32x4* ar = (u32x4*)(a + ((size_t)n * P + p0 + 32 * t + l31) * K + 24 * lhi);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                u32x4 v; v[0] = d[4 * ks]; v[1] = d[4 * ks + 1]; v[2] = d[4 * ks + 2]; v[3] = d[4 * ks + 3];
                ar[ks] = v;
                bf[t][ks] = __builtin_bit_cast(s16x8, v);
            }
        }
        if (u + 1 < u1) issue(u + 1);
        // D[co][pixel]: acc[m][4q + e] = D[co = 32 m + 8 q + 4 lhi + e][pixel 32 t + l31], starting at the bias; one 32-pixel tile at a time
        // (48 accumulator registers instead of 96), through the wave's LDS tile: rows = co, 64 pixels of bf16 each
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *(const f32x4*)(bl + 32 * m + 8 * q + 4 * lhi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][4 * q + e] = b4[e];
                }
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma32<bf16_t>(wf[m][ks], bf[t][ks], acc[m]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = 32 * m + 8 * (r >> 2) + 4 * lhi + (r & 3);
                    *(uint16_t*)(tile + co * SC_RS + (32 * t + l31) * 2) = f32_to_bf16_bits(acc[m][r]);
                }
            __builtin_amdgcn_sched_barrier(0);                     // (keeps the scheduler from interleaving the two tiles: both accumulator sets live = 276 VGPRs)
        }
        asm volatile("" ::: "memory");                             // wave-private tile: the LDS operations of a wave complete in order
        uint16_t* yr = y + (size_t)n * Co * P + p0;
#pragma unroll
        for (int i = 0; i < Co / 8; ++i) {
            const int co = i * 8 + (lane >> 3), ch = lane & 7;
            const u32x4 v = *(const u32x4*)(tile + co * SC_RS + ch * 16);
            *(u32x4*)(yr + (size_t)co * P + ch * 8) = v;
        }
        asm volatile("" ::: "memory");
    }
}

struct ScPlan { int units, upw, wgs; size_t lds; };
static bool sc_plan(int N, int Cin, int H, int W, int Co, ScPlan* pl) {
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Co <= 0) return false;
    if (Cin != 3 || (Co & 31) || Co > 128 || (H & 3) || (W & 3)) return false;
    const long long P = (long long)(H / 4) * (W / 4);
    if (P % 64 || (long long)N * 3 * H * W >= (1LL << 40) || (long long)N * Co * P >= (1LL << 40)) return false;
    const long long units = (long long)N * (P / 64);
    if (units > (1LL << 30)) return false;
    const int target = (Co > 96 ? 4 : 8) * mfma_cu_count();                 // waves resident at once: two per SIMD (228 VGPRs at Co = 96), one with four row tiles
    int upw = (int)((units + target - 1) / target); if (upw < 1) upw = 1;
    const long long waves = (units + upw - 1) / upw;
    pl->units = (int)units; pl->upw = upw; pl->wgs = (int)((waves + SC_WAVES - 1) / SC_WAVES);
    pl->lds = (size_t)Co * 4 + (size_t)SC_WAVES * Co * SC_RS;
    return true;
}

}  // namespace slak

using namespace slak;

int slak_stem_conv_forward_supported(int N, int Cin, int H, int W, int Co) { ScPlan pl; return sc_plan(N, Cin, H, W, Co, &pl) ? 1 : 0; }

int slak_stem_conv_forward(const float* x, const float* weight, const float* bias, void* a_bf16, void* y_bf16, int N, int Cin, int H, int W, int Co,
                           void* stream) {
    if (!x || !weight || !a_bf16 || !y_bf16) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Co <= 0) return SLAK_ERR_INVALID_ARG;
    ScPlan pl;
    if (!sc_plan(N, Cin, H, W, Co, &pl)) return SLAK_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define SC_LAUNCH(MT)                                                                                                                    \
    do { if (!slak_set_max_lds((const void*)stem_conv_fwd_kernel<MT>, pl.lds)) return SLAK_ERR_LAUNCH;                                   \
         hipLaunchKernelGGL(stem_conv_fwd_kernel<MT>, dim3((unsigned)pl.wgs), dim3(SC_WAVES * 64), pl.lds, st, x, weight, bias,          \
                            (uint16_t*)a_bf16, (uint16_t*)y_bf16, H, W, pl.units, pl.upw); } while (0)
    switch (Co / 32) { case 1: SC_LAUNCH(1); break; case 2: SC_LAUNCH(2); break; case 3: SC_LAUNCH(3); break; default: SC_LAUNCH(4); break; }
#undef SC_LAUNCH
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}
