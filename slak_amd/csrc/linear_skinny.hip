// slak_amd/csrc/linear_skinny.hip -- the pointwise (1x1) convolutions of a SLaK block on the LARGE maps, where they are not GEMMs in
// any compute sense: pwconv1 / pwconv2 (models/SLaK.py:158-160) and their data gradients at stage 1-2 multiply a 401,408 x 96
// (100,352 x 192) activation matrix by a 74 KB (295 KB) weight: 2-12 flop per byte moved, pure HBM streaming.  hipBLASLt's tiles run
// them at 1.8-2.9 TB/s (tools/time_gemm.py); these kernels stream at the rate of a copy.
//
//   Y[M x N] = X[M x K] . Wt[N x K]^T (+ bias[N]),  optionally  G = gelu(Y)  written alongside (nn.GELU(), exact erf form, evaluated
//   on the ROUNDED y as autocast does: F.gelu of a bf16 tensor)
//
// X, Wt, bias, Y, G bf16, fp32 accumulate.  "NT": both operands K-contiguous, so an MFMA fragment (one row, 8 consecutive k) is one
// 16-byte load straight from global memory into the operand registers -- no LDS, no barrier; a wave owns 32 rows of X and nothing
// else, the weight comes from L2 (every wave reads the same few hundred KB).  The operands are swapped (D^T = Wt-tile x X-tile^T) so
// that a lane holds 4 consecutive output columns of ONE row; two v_permlane32_swap per register pair turn that into 8 consecutive
// columns = one 16-byte store (row-per-lane 8-byte stores are issue-bound at ~7 B/clk/CU: MI355X_MICROARCH.md, store tail).
//   * linear_nt_smallk<KS>: K = 16 KS <= 192.  The wave's X fragments stay in registers; it walks the N/32 column tiles.
//   * linear_nt_smalln<NT>: N = 32 NT <= 192, K a multiple of 16.  NT accumulators; the wave walks K.
#include "mfma_common.h"

namespace slak {

__device__ __forceinline__ float gelu_erf(float y) { return 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f)); }

__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// One 32 x 32 output tile in registers (lane = row l31, acc[4q + j] = column 8q + 4 lhi + j) -> bias, rounding, optional GELU,
// 16-byte stores.  `bq[q]` = the lane's four bias values of quad q as two packed dwords are fetched by the caller.
template <bool GELU>
__device__ __forceinline__ void store_tile(const f32x16& acc, const uint16_t* __restrict__ bias, uint16_t* __restrict__ Y,
                                           uint16_t* __restrict__ G, size_t row_off, int col0, int lhi, bool row_ok) {
    unsigned py[8], pg[8];                                     // packed pairs: quad q -> py[2q], py[2q+1]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = col0 + 8 * q + 4 * lhi;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (bias) {
            const u32x2 bb = *(const u32x2*)(bias + c);
            b0 = bf16_lo(bb[0]); b1 = bf16_hi(bb[0]); b2 = bf16_lo(bb[1]); b3 = bf16_hi(bb[1]);
        }
        const unsigned y01 = pack2<bf16_t>(acc[4 * q + 0] + b0, acc[4 * q + 1] + b1);
        const unsigned y23 = pack2<bf16_t>(acc[4 * q + 2] + b2, acc[4 * q + 3] + b3);
        py[2 * q] = y01; py[2 * q + 1] = y23;
        if constexpr (GELU) {
            pg[2 * q] = pack2<bf16_t>(gelu_erf(bf16_lo(y01)), gelu_erf(bf16_hi(y01)));
            pg[2 * q + 1] = pack2<bf16_t>(gelu_erf(bf16_lo(y23)), gelu_erf(bf16_hi(y23)));
        }
    }
    // lanes l31 (lhi 0) and l31 + 32 (lhi 1) hold the same row: quads (0,1) -> lhi 0 keeps columns 0..7, lhi 1 gets 8..15; quads (2,3)
    // likewise 16..23 / 24..31.  v_permlane32_swap(a, b): a of lanes 32..63 <-> b of lanes 0..31.
    auto swap = [](unsigned& a, unsigned& b) {
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        swap(py[4 * h + 0], py[4 * h + 2]); swap(py[4 * h + 1], py[4 * h + 3]);
        if constexpr (GELU) { swap(pg[4 * h + 0], pg[4 * h + 2]); swap(pg[4 * h + 1], pg[4 * h + 3]); }
    }
    // after the swaps: lanes lhi 0: {py[4h], py[4h+1], py[4h+2], py[4h+3]} = columns 16h + 0..7; lanes lhi 1: columns 16h + 8..15
    if (row_ok) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t o = row_off + (size_t)(col0 + 16 * h + 8 * lhi);
            *(u32x4*)(Y + o) = u32x4{py[4 * h + 0], py[4 * h + 1], py[4 * h + 2], py[4 * h + 3]};
            if constexpr (GELU) *(u32x4*)(G + o) = u32x4{pg[4 * h + 0], pg[4 * h + 1], pg[4 * h + 2], pg[4 * h + 3]};
        }
    }
}

template <int KS, bool GELU>
__global__ __launch_bounds__(256) void linear_nt_smallk_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                               const uint16_t* __restrict__ bias, uint16_t* __restrict__ Y,
                                                               uint16_t* __restrict__ G, int M, int N) {
    constexpr int K = 16 * KS;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    if (r0 >= M) return;
    const int row = r0 + l31;
    const bool row_ok = row < M;
    const uint16_t* xr = X + (size_t)(row_ok ? row : M - 1) * K + lhi * 8;
    s16x8 xf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = __builtin_bit_cast(s16x8, *(const u32x4*)(xr + ks * 16));
    const uint16_t* wl = Wt + (size_t)l31 * K + lhi * 8;
    const size_t row_off = (size_t)row * N;
    const int ntiles = N >> 5;
    s16x8 wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = __builtin_bit_cast(s16x8, *(const u32x4*)(wl + ks * 16));
    for (int nt = 0; nt < ntiles; ++nt) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const uint16_t* wn = wl + (size_t)(nt + 1 < ntiles ? nt + 1 : nt) * 32 * K;     // next tile's fragments stream in behind the MFMAs
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            acc = mfma32<bf16_t>(wf[ks], xf[ks], acc);
            wf[ks] = __builtin_bit_cast(s16x8, *(const u32x4*)(wn + ks * 16));
        }
        store_tile<GELU>(acc, bias, Y, G, row_off, nt * 32, lhi, row_ok);
    }
}

template <int NT>
__global__ __launch_bounds__(256) void linear_nt_smalln_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                               const uint16_t* __restrict__ bias, uint16_t* __restrict__ Y,
                                                               int M, int K) {
    constexpr int N = 32 * NT;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    if (r0 >= M) return;
    const int row = r0 + l31;
    const bool row_ok = row < M;
    const uint16_t* xr = X + (size_t)(row_ok ? row : M - 1) * K + lhi * 8;
    const uint16_t* wl = Wt + (size_t)l31 * K + lhi * 8;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int nks = K >> 4;
    s16x8 xf = __builtin_bit_cast(s16x8, *(const u32x4*)xr);
    s16x8 wf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wf[t] = __builtin_bit_cast(s16x8, *(const u32x4*)(wl + (size_t)t * 32 * K));
    for (int ks = 0; ks < nks; ++ks) {
        const int kn = (ks + 1 < nks ? ks + 1 : ks) * 16;                               // next k-step's fragments stream in behind the MFMAs
        const s16x8 xn = __builtin_bit_cast(s16x8, *(const u32x4*)(xr + kn));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = mfma32<bf16_t>(wf[t], xf, acc[t]);
            wf[t] = __builtin_bit_cast(s16x8, *(const u32x4*)(wl + (size_t)t * 32 * K + kn));
        }
        xf = xn;
    }
    const size_t row_off = (size_t)row * N;
#pragma unroll
    for (int t = 0; t < NT; ++t) store_tile<false>(acc[t], bias, Y, nullptr, row_off, t * 32, lhi, row_ok);
}

}  // namespace slak

using namespace slak;

extern "C" {

int slak_linear_nt_supported(int M, int N, int K, int gelu) {
    if (M <= 0 || N <= 0 || K <= 0 || (long long)M * (N > K ? N : K) >= (1LL << 31)) return 0;
    if ((K == 96 || K == 192) && N % 32 == 0) return 1;                                   // small K: any N
    if (!gelu && K % 16 == 0 && (N == 96 || N == 192)) return 1;                          // small N: any K
    return 0;
}

int slak_linear_nt(const void* x, const void* wt, const void* bias, void* y, void* gelu_out, int M, int N, int K, void* stream) {
    if (!x || !wt || !y) return SLAK_ERR_INVALID_ARG;
    if (!slak_linear_nt_supported(M, N, K, gelu_out != nullptr)) return SLAK_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const uint16_t *X = (const uint16_t*)x, *W = (const uint16_t*)wt, *B = (const uint16_t*)bias;
    uint16_t *Y = (uint16_t*)y, *G = (uint16_t*)gelu_out;
    if (K == 96 || K == 192) {
        if (K == 96) {
            if (G) hipLaunchKernelGGL((linear_nt_smallk_kernel<6, true>), grid, block, 0, st, X, W, B, Y, G, M, N);
            else hipLaunchKernelGGL((linear_nt_smallk_kernel<6, false>), grid, block, 0, st, X, W, B, Y, G, M, N);
        } else {
            if (G) hipLaunchKernelGGL((linear_nt_smallk_kernel<12, true>), grid, block, 0, st, X, W, B, Y, G, M, N);
            else hipLaunchKernelGGL((linear_nt_smallk_kernel<12, false>), grid, block, 0, st, X, W, B, Y, G, M, N);
        }
    } else if (N == 96) {
        hipLaunchKernelGGL((linear_nt_smalln_kernel<3>), grid, block, 0, st, X, W, B, Y, M, K);
    } else {
        hipLaunchKernelGGL((linear_nt_smalln_kernel<6>), grid, block, 0, st, X, W, B, Y, M, K);
    }
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

}  // extern "C"
