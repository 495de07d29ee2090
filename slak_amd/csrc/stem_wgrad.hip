// slak_amd/csrc/stem_wgrad.hip -- weight and bias gradient of the stem convolution (Conv2d(in_chans, C, 4, stride 4), models/SLaK.py:189-193)
// from the NCHW output gradient and the forward's patch matrix, in one pass over both:
//     dw[co][k] = sum_{n,p} dy[n][co][p] * a[n][p][k]          (k = ci*16 + kh*4 + kw, the layout of slak_stem_patchify)
//     db[co]    = sum_{n,p} dy[n][co][p]
// torch ran this as a batched GEMM per image (library macro tile 64x96x32: 132 us at N = 128, 224 px), a sum over the images and a separate
// reduction of dy for the bias: three reads of dy + a, per-image products rounded to bf16.  Here: 115 MB read once, fp32 accumulation throughout.
//
// Shape of the work: M = Co (96 / 128), N = K + 1 (48 patch columns + a column of ones: the bias gradient falls out of the same MFMAs),
// reduction over N*P pixels (401 408).  3.7 GFLOP against 115 MB: HBM bound by a wide margin, so the kernel is organised around the loads:
//   * a unit = 64 consecutive pixels of one image; a wavefront owns `upw` consecutive units, no workgroup barriers in the loop;
//   * dy is the MFMA A operand straight from global memory: lane (co = lane % 32, g = lane / 32) holds pixels p0 + 32 g + 8 ks + {0..7} for
//     MFMA step ks -- the reduction index may be permuted freely as long as B uses the same permutation, and this one gives every lane 64
//     contiguous bytes per unit (four 16-byte loads) and every row one full 128-byte line;
//   * the a-tile of a unit is ONE contiguous block of 64*K*2 bytes: loaded with 16-byte loads, laid into a wave-private LDS tile with rows of
//     K*2 + 4 bytes (odd dword stride: the two lane halves read disjoint banks), read back column-wise as the B operand;
//   * the loads of unit i+1 are issued while unit i is multiplied: the a-tile into a second register set, each dy fragment into its own
//     registers right after the MFMA step that used it.
// The waves' partial sums are added in a fixed order (LDS within the workgroup, block_tail_reduce1 across workgroups): same bits every run.
#include "mfma_common.h"
#include "gelu_grad.h"          // tail_reduce_split

namespace slak {

constexpr int SW_WAVES = 4;
constexpr int SW_UNIT = 64;             // pixels per unit
constexpr int SW_KMAX = 56;             // K + 1 <= 64 (two 32-column tiles), K % 8 == 0

template <int MT>
__global__ __launch_bounds__(SW_WAVES * 64) void stem_wgrad_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ a,
                                                                  float* __restrict__ part, int Co, int P, int K, int units, int upw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int rsd = K / 2 + 1;                                    // a-tile row stride in dwords
    uint32_t* tile = (uint32_t*)smem + (size_t)wave * SW_UNIT * rsd;
    const int upi = P / SW_UNIT;                                   // units per image
    const int kq = K / 8;                                          // 16-byte loads per lane for an a-tile (K/2 dwords per lane)
    const int u0 = (blockIdx.x * SW_WAVES + wave) * upw;
    int u1 = u0 + upw; if (u1 > units) u1 = units;

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    s16x8 nd[MT][4];                                               // dy fragments: reloaded for the next unit as soon as an MFMA step has used them
    u32x4 na[SW_KMAX / 8];                                         // next unit's a-tile, 16 bytes per lane and load
    auto dy_row = [&](int u) {
        const int n = u / upi, p0 = (u - n * upi) * SW_UNIT;
        return dy + ((size_t)n * Co + l31) * P + p0 + 32 * lhi;
    };
    auto issue_a = [&](int u) {
        const int n = u / upi, p0 = (u - n * upi) * SW_UNIT;
        const u32x4* ar = (const u32x4*)(a + ((size_t)n * P + p0) * K);
#pragma unroll
        for (int j = 0; j < SW_KMAX / 8; ++j) if (j < kq) na[j] = ar[j * 64 + lane];
    };
    if (u0 < u1) {
        const uint16_t* dr = dy_row(u0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) nd[m][ks] = *(const s16x8*)(dr + (size_t)m * 32 * P + 8 * ks);
        issue_a(u0);
    }
    // B columns of the two tiles: k = l31 (always a patch column: K >= 32) and k = 32 + l31 (patch column, the ones column at k == K, or zero)
    const int k1 = 32 + l31;
    const bool k1_real = k1 < K, k1_one = k1 == K;
    const int k1c = k1_real ? k1 : 0;
    const int sh0 = (l31 & 1) * 16, sh1 = (k1c & 1) * 16;
    for (int u = u0; u < u1; ++u) {
        // lay the a-tile down: dword d = (j*64 + lane)*4 + e of the block -> row d / (K/2), dword column d % (K/2)  (K % 8 == 0: no straddling)
#pragma unroll
        for (int j = 0; j < SW_KMAX / 8; ++j) if (j < kq) {
            const int d = (j * 64 + lane) * 4, row = d / (K / 2), cd = d - row * (K / 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[row * rsd + cd + e] = na[j][e];
        }
        asm volatile("" ::: "memory");                             // the tile is wave-private: the LDS operations of a wave complete in order -- the compiler must keep them so
        const bool more = u + 1 < u1;
        if (more) issue_a(u + 1);
        const uint16_t* dr = dy_row(more ? u + 1 : u);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s16x8 b0, b1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {                          // (dword reads of the tile the dword stores above wrote: one type, no aliasing games)
                const int row = 32 * lhi + 8 * ks + j;
                b0[j] = (short)(tile[row * rsd + (l31 >> 1)] >> sh0);
                const uint32_t v = tile[row * rsd + (k1c >> 1)] >> sh1;
                b1[j] = (short)(k1_real ? v : (k1_one ? 0x3F80u : 0u));
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m][0] = mfma32<bf16_t>(nd[m][ks], b0, acc[m][0]);
                acc[m][1] = mfma32<bf16_t>(nd[m][ks], b1, acc[m][1]);
            }
            if (more) {
#pragma unroll
                for (int m = 0; m < MT; ++m) nd[m][ks] = *(const s16x8*)(dr + (size_t)m * 32 * P + 8 * ks);
            }
        }
        asm volatile("" ::: "memory");
    }
    // the four waves' sums, added in wave order through LDS (red[co][65]); acc[m][t][4q+e] = D[co = 32m + 8q + 4 lhi + e][k = 32t + l31]
    __syncthreads();
    float* red = (float*)smem + (size_t)SW_WAVES * SW_UNIT * rsd;
    for (int w = 0; w < SW_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* o = red + (32 * m + 8 * (r >> 2) + 4 * lhi + (r & 3)) * 65 + 32 * t + l31;
                        *o = w == 0 ? acc[m][t][r] : *o + acc[m][t][r];
                    }
        }
        __syncthreads();
    }
    float* pt = part + (size_t)blockIdx.x * ((size_t)Co * K + Co);            // [dw Co*K | db Co]
    for (int i = threadIdx.x; i < Co * K; i += SW_WAVES * 64) { const int co = i / K, k = i - co * K; pt[i] = red[co * 65 + k]; }
    for (int co = threadIdx.x; co < Co; co += SW_WAVES * 64) pt[(size_t)Co * K + co] = red[co * 65 + K];
}

struct SwPlan { int units, upw, wgs; size_t lds; };
static bool sw_plan(int N, int Co, int P, int K, SwPlan* pl) {
    if (N <= 0 || Co <= 0 || P <= 0 || K <= 0) return false;
    if ((Co & 31) || Co > 128 || (P % SW_UNIT) || (K & 7) || K < 32 || K > SW_KMAX) return false;
    if ((long long)N * Co * P >= (1LL << 40)) return false;
    const long long units = (long long)N * (P / SW_UNIT);
    if (units > (1LL << 30)) return false;
    const int target = (Co > 96 ? 4 : 8) * mfma_cu_count();        // waves resident at once: two per SIMD (256 VGPRs at Co = 96), one with four row tiles
    int upw = (int)((units + target - 1) / target); if (upw < 1) upw = 1;
    const long long waves = (units + upw - 1) / upw;
    pl->units = (int)units; pl->upw = upw; pl->wgs = (int)((waves + SW_WAVES - 1) / SW_WAVES);
    pl->lds = (size_t)SW_WAVES * SW_UNIT * (K / 2 + 1) * 4 + (size_t)Co * 65 * 4;
    return true;
}

}  // namespace slak

using namespace slak;

int slak_stem_wgrad_supported(int N, int Co, int P, int K) { SwPlan pl; return sw_plan(N, Co, P, K, &pl) ? 1 : 0; }

size_t slak_stem_wgrad_workspace_bytes(int N, int Co, int P, int K) {
    SwPlan pl;
    if (!sw_plan(N, Co, P, K, &pl)) return 0;
    return ((size_t)pl.wgs * ((size_t)Co * K + Co) + Co) * sizeof(float);
}

int slak_stem_wgrad(const void* dy_bf16, const void* a_bf16, float* dw, float* db, int N, int Co, int P, int K,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy_bf16 || !a_bf16 || !dw) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || Co <= 0 || P <= 0 || K <= 0) return SLAK_ERR_INVALID_ARG;
    SwPlan pl;
    if (!sw_plan(N, Co, P, K, &pl)) return SLAK_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < slak_stem_wgrad_workspace_bytes(N, Co, P, K)) return SLAK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    const size_t width = (size_t)Co * K + Co;
    float* db_out = db ? db : part + (size_t)pl.wgs * width;       // no bias: the column of ones still runs, its sums land in the workspace
    const void* fn = nullptr;
#define SW_LAUNCH(MT)                                                                                                                   \
    do { fn = (const void*)stem_wgrad_kernel<MT>;                                                                                       \
         if (!slak_set_max_lds(fn, pl.lds)) return SLAK_ERR_LAUNCH;                                                                      \
         hipLaunchKernelGGL(stem_wgrad_kernel<MT>, dim3((unsigned)pl.wgs), dim3(SW_WAVES * 64), pl.lds, st, (const uint16_t*)dy_bf16,   \
                            (const uint16_t*)a_bf16, part, Co, P, K, pl.units, pl.upw); } while (0)
    switch (Co / 32) { case 1: SW_LAUNCH(1); break; case 2: SW_LAUNCH(2); break; case 3: SW_LAUNCH(3); break; default: SW_LAUNCH(4); break; }
#undef SW_LAUNCH
    SLAK_LAUNCH_CHECK();
    return tail_reduce_split(part, dw, db_out, Co * K, pl.wgs, (int)width, st);
}
