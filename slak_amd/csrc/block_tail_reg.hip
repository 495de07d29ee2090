// slak_amd/csrc/block_tail_reg.hip -- register-tile versions of the block-tail kernels for C <= 256 (stages 1-2 of SLaK, where most
// of the block-tail time is): a LANE owns a pixel (or a pixel and half / a quarter of the channels) and keeps its channels in
// registers, so the NCHW side is plain coalesced 2- / 4-byte accesses along the pixel axis, the per-pixel statistics are register
// sums (no LDS, no barrier), and only the NHWC side goes through a per-wave LDS tile (rows padded to an odd number of 16-byte chunks:
// the lane's row is written / read with conflict-free ds_*_b128) to be moved as coalesced 16-byte pieces.  The kernels of
// block_tail.hip stage BOTH sides in LDS and walk them with 2-byte LDS accesses in three passes: 0.17-0.34 of the HBM roofline.
// Small planes (P <= 256: the 14 x 14 and 7 x 7 stages) take a second family further down (`*_chan_kernel`, `*_chan1_kernel`): a wave per (image,
// channel group, 64 pixel pairs or pixels) -- the pixel tiles of this geometry shrink to 16 pixels per wave on C = 384, i.e. 64-byte NCHW runs.
#include "slak_common.h"
#include <stdlib.h>
#include "mfma_common.h"

namespace slak {

typedef __attribute__((ext_vector_type(4))) unsigned rt_u32x4;
typedef __attribute__((ext_vector_type(2))) float rt_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 rt_bf16x2;
__device__ __forceinline__ float rt_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float rt_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned rt_pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(rt_f32x2{a, b}, rt_bf16x2)); }

// Lane geometry: lane = pixel_pair * G + group.  A lane owns TWO neighbouring pixels (p, p+1: one 4-byte access per bf16 channel row,
// 8 bytes per fp32 one) and the channels [g*CL, (g+1)*CL) of both; a wave covers PW = 128 / G pixels.  The pixel-pair index sits in the
// HIGH lane bits so that v_permlane32_swap / v_permlane16_swap fold pixel halves (per-channel sums); the group in the low bits.
// Needs P even (pairs never straddle the end of an image).
// All global accesses are raw-buffer operations: descriptor = (image, tile) base with the bytes that remain, lane offset in a VGPR
// (tile-invariant), channel row in the scalar offset.  Lanes and tile rows past the end of the image read zeros and their stores are
// dropped by the range check, so there is no per-access predicate and no 64-bit VALU address arithmetic.
template <int CL, int G> struct RtGeom {
    static constexpr int C = CL * G, PW = 128 / G, PITCH = C * 2 + 16, CPR = C / 8, NCH = PW * CPR, LDS_WAVE = PW * PITCH;
    static_assert(CL % 16 == 0 && (G == 1 || G == 2 || G == 4 || G == 8), "lane geometry");
};
typedef __amdgpu_buffer_rsrc_t rt_rsrc;
typedef __attribute__((ext_vector_type(2))) unsigned rt_u32x2;
__device__ __forceinline__ rt_rsrc rt_buf(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
constexpr unsigned RT_OOB = 0x80000000u;                      // a lane offset no descriptor covers

// sum over the G lanes that share a pixel pair
template <int G> __device__ __forceinline__ float rt_group_sum(float v) {
#pragma unroll
    for (int k = 1; k < G; k <<= 1) v += __shfl_xor(v, k, 64);
    return v;
}
__device__ __forceinline__ void rt_lds_fence() {            // LDS hand-off between the lanes of ONE wave
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// PW pixel rows of an NHWC tensor (one contiguous run of PW * C elements) -> the wave's padded LDS tile by LDS-DMA (no registers, no
// ds_write pass): destinations are lane-linear, so destination chunk q = (row q / CD, chunk q % CD) fetches source chunk row * (CD-1) +
// chunk; the padding chunk of a row and the rows the descriptor does not cover (past the image) are written as zeros by the range check.
// The plan (per-lane source offsets) is tile-invariant.  Completion: s_waitcnt vmcnt(0), then a wave barrier.
template <int C, int PW> struct RtDmaPlan { static constexpr int CD = C / 8 + 1, NI = (PW * CD + 63) / 64; unsigned src[NI]; };
template <int C, int PW> __device__ __forceinline__ void rt_dma_plan(RtDmaPlan<C, PW>& d, int lane) {
    constexpr int CD = RtDmaPlan<C, PW>::CD;
#pragma unroll
    for (int k = 0; k < RtDmaPlan<C, PW>::NI; ++k) {
        const int q = 64 * k + lane, r = q / CD, cc = q - r * CD;
        d.src[k] = r >= PW ? 0xffffffffu : (cc < CD - 1 ? (unsigned)(r * (CD - 1) + cc) * 16u : RT_OOB);
    }
}
__device__ __forceinline__ v4i_t rt_desc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return v4i_t{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
template <int C, int PW> __device__ __forceinline__ void rt_tile_dma(const RtDmaPlan<C, PW>& d, const void* src, unsigned bytes, unsigned lds_dst) {
    const v4i_t r = rt_desc(src, bytes);
#pragma unroll
    for (int k = 0; k < RtDmaPlan<C, PW>::NI; ++k)
        if (d.src[k] != 0xffffffffu) lds_dma16(d.src[k], r, __builtin_amdgcn_readfirstlane(lds_dst + k * 1024));
}
__device__ __forceinline__ void rt_dma_wait() {             // every outstanding memory operation of the wave, then the lanes meet
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
template <int C, int PW> __device__ __forceinline__ void rt_tile_store(const unsigned char* T, rt_rsrc r, int lane) {
    constexpr int CPR = C / 8, NCH = PW * CPR, PITCH = C * 2 + 16, RINC = 64 / CPR, CINC = 64 % CPR, NK = (NCH + 63) / 64;
    int row = lane / CPR, cc = lane - row * CPR;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (NCH % 64 == 0 || k * 64 + lane < NCH)
            __builtin_amdgcn_raw_buffer_store_b128(*(const rt_u32x4*)(T + row * PITCH + cc * 16), r, (unsigned)(k * 64 + lane) * 16u, 0, 0);
        cc += CINC; row += RINC;
        if (cc >= CPR) { cc -= CPR; ++row; }
    }
}
// Per-channel sums over the wave's pixels, folded as they come: a0..a3 = the lane's terms of its channels 4m..4m+3; the result register
// holds, in its 16-lane row r, partial sums of channel 4m + {0,2,1,3}[r] (a quarter of the wave's lanes each).  Two swaps and three adds
// replace four accumulator registers per lane by one.
__device__ __forceinline__ float rt_fold4(float a0, float a1, float a2, float a3) {
    const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);   // [a0.lo a1.lo], [a0.hi a1.hi]
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2), __float_as_uint(a3), false, false);
    const float s01 = __uint_as_float(p[0]) + __uint_as_float(p[1]);     // lanes 0..31: channel 4m, lanes 32..63: channel 4m+1
    const float s23 = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);  // rows [a0 b0 a2 b2], [a1 b1 a3 b3]
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);               // rows: 4m, 4m+2, 4m+1, 4m+3
}
// end of the tile loops: finish the fold over the remaining pixel bits of a 16-lane row, add the workgroup's four waves through LDS (the
// tile area: everyone is done with it) and write ONE partial row per workgroup: part[0..C) (first quantity), part[C..2C) (second); f1
// applies a per-channel factor to the second.  Every wave of the workgroup must call this (two workgroup barriers).
template <int CL, int G, typename F1>
__device__ __forceinline__ void rt_write_partials(float (&acc0)[CL / 4], float (&acc1)[CL / 4], float* __restrict__ part, float* L, int wave, int lane, F1 f1) {
    constexpr int C = CL * G;
    const int g = lane & (G - 1), rowi = lane >> 4;
    const int cofs = rowi == 0 ? 0 : rowi == 1 ? 2 : rowi == 2 ? 1 : 3;
    __syncthreads();
    float* const mine = L + wave * 2 * C;
#pragma unroll
    for (int m = 0; m < CL / 4; ++m) {
        float a = acc0[m], b = acc1[m];
#pragma unroll
        for (int k = G; k < 16; k <<= 1) { a += __shfl_xor(a, k, 64); b += __shfl_xor(b, k, 64); }
        if ((lane & 15) < G) { const int c = g * CL + 4 * m + cofs; mine[c] = a; mine[C + c] = b; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const float v = ((L[i] + L[2 * C + i]) + L[4 * C + i]) + L[6 * C + i];
        part[i] = i < C ? v : f1(v, i - C);
    }
}
// The NCHW operand of the LayerNorm kernels: bf16 (one dword = the lane's two pixels of a channel) or fp32 (two dwords)
template <bool F32> struct RtX;
template <> struct RtX<false> {
    typedef unsigned reg; static constexpr unsigned EB = 2;
    static __device__ __forceinline__ reg load(rt_rsrc r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0); }
    static __device__ __forceinline__ float p0(reg v) { return rt_lo(v); }
    static __device__ __forceinline__ float p1(reg v) { return rt_hi(v); }
    static __device__ __forceinline__ void store(rt_rsrc r, unsigned vo, unsigned so, float a, float b) { __builtin_amdgcn_raw_buffer_store_b32(rt_pack2(a, b), r, vo, so, 0); }
    static __device__ __forceinline__ void opaque(reg& v) { asm volatile("" : "+v"(v)); }
};
template <> struct RtX<true> {
    typedef rt_u32x2 reg; static constexpr unsigned EB = 4;
    static __device__ __forceinline__ reg load(rt_rsrc r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0); }
    static __device__ __forceinline__ float p0(reg v) { return __uint_as_float(v[0]); }
    static __device__ __forceinline__ float p1(reg v) { return __uint_as_float(v[1]); }
    static __device__ __forceinline__ void store(rt_rsrc r, unsigned vo, unsigned so, float a, float b) { __builtin_amdgcn_raw_buffer_store_b64(rt_u32x2{__float_as_uint(a), __float_as_uint(b)}, r, vo, so, 0); }
    static __device__ __forceinline__ void opaque(reg& v) { asm volatile("" : "+v"(v)); }
};
// PATCH mode (the downsample layers, models/SLaK.py:285-311: LayerNorm(channels_first) -> Conv2d(k = 2, stride 2)): the NHWC side is the
// conv's patch matrix A[n][ho * Wo + wo][(kh * 2 + kw) * C + c] -- pixel (h, w)'s C-vector is the segment (h & 1, w & 1) of row (h/2, w/2),
// still one contiguous run of C elements, so only the row address of a tile row changes.  Byte offset inside the image's [P/4][4C] matrix:
template <int C> __device__ __forceinline__ unsigned rt_patch_off(int p, int W) {
    const int h = p / W, w = p - h * W;
    return (unsigned)((((h >> 1) * (W >> 1) + (w >> 1)) * 4 + (h & 1) * 2 + (w & 1)) * (C * 2));
}
template <int C, int PW> __device__ __forceinline__ void rt_dma_plan_patch(RtDmaPlan<C, PW>& d, int lane, int p0, int W, int rows) {
    constexpr int CD = RtDmaPlan<C, PW>::CD;
#pragma unroll
    for (int k = 0; k < RtDmaPlan<C, PW>::NI; ++k) {
        const int q = 64 * k + lane, r = q / CD, cc = q - r * CD;
        d.src[k] = r >= PW ? 0xffffffffu : ((cc < CD - 1 && r < rows) ? rt_patch_off<C>(p0 + r, W) + (unsigned)cc * 16u : RT_OOB);
    }
}
template <int C, int PW> __device__ __forceinline__ void rt_tile_store_patch(const unsigned char* T, rt_rsrc r, int p0, int W, int rows, int lane) {
    constexpr int CPR = C / 8, NCH = PW * CPR, PITCH = C * 2 + 16, RINC = 64 / CPR, CINC = 64 % CPR, NK = (NCH + 63) / 64;
    int row = lane / CPR, cc = lane - row * CPR;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if ((NCH % 64 == 0 || k * 64 + lane < NCH) && row < rows)
            __builtin_amdgcn_raw_buffer_store_b128(*(const rt_u32x4*)(T + row * PITCH + cc * 16), r, rt_patch_off<C>(p0 + row, W) + (unsigned)cc * 16u, 0, 0);
        cc += CINC; row += RINC;
        if (cc >= CPR) { cc -= CPR; ++row; }
    }
}
// channel c of a row chunk array t[] (2 bf16 per register)
#define RT_CH(t, c) (((c) & 1) ? rt_hi((t)[(c) >> 1]) : rt_lo((t)[(c) >> 1]))
// per-channel parameters staged in LDS behind the four wave tiles; the lane's 4 channels 4m..4m+3
template <int CL> __device__ __forceinline__ float4 rt_par4(const float* L, int g, int m) { return *(const float4*)(L + g * CL + 4 * m); }
__device__ __forceinline__ float rt_f4(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

// y[n,p,:] = LN_C(x[n,:,p]) * w + b   (x bf16 or fp32 NCHW, y bf16 NHWC or patch matrix, statistics fp32, two-pass variance); saves mean,
// rstd.  One tile per wave.  Wimg: image width (PATCH mode only).
template <int CL, int G, bool XF32 = false, bool PATCH = false>
__global__ __launch_bounds__(256, 2) void ln_nchw_to_nhwc_fwd_reg_kernel(const void* __restrict__ xin, const float* __restrict__ w,
                                                                        const float* __restrict__ b, uint16_t* __restrict__ y,
                                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                                        int N, int P, float eps, int tiles_per_image, int ntiles, int Wimg) {
    using XR = RtX<XF32>;
    const char* const x = (const char*)xin;
    using GE = RtGeom<CL, G>;
    constexpr int C = GE::C, PW = GE::PW, PITCH = GE::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* const Lw = (float*)(smem + 4 * GE::LDS_WAVE);
    float* const Lb = Lw + C;
    const bool has_tile = blockIdx.x * 4 + wave < ntiles;                                   // a wave without a tile runs on an empty one (rows = 0)
    const int tile = has_tile ? blockIdx.x * 4 + wave : 0;
    unsigned char* const T = smem + wave * GE::LDS_WAVE;
    const int n = tile / tiles_per_image, p0 = (tile - n * tiles_per_image) * PW;
    const int pp = lane / G, g = lane & (G - 1);
    const int rows = has_tile ? min(PW, P - p0) : 0;
    const unsigned rb = (unsigned)P * XR::EB;                                               // bytes per channel row
    const rt_rsrc rx = rt_buf(x + ((size_t)n * C * P + p0) * XR::EB, (unsigned)(C * P - p0) * XR::EB);
    const unsigned vo = 2 * pp < rows ? (unsigned)(g * CL) * rb + (unsigned)pp * 2u * XR::EB : RT_OOB;
    typename XR::reg v[CL];                                                                 // (pixel p, pixel p+1) of channel c
#pragma unroll
    for (int c = 0; c < CL; ++c) v[c] = XR::load(rx, vo, (unsigned)c * rb);
    for (int i = threadIdx.x; i < C; i += 256) { Lw[i] = w[i]; Lb[i] = b[i]; }              // behind the tile's loads: one memory latency, not two
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < CL; ++c) { s0 += XR::p0(v[c]); s1 += XR::p1(v[c]); }
    const float mu0 = rt_group_sum<G>(s0) / (float)C, mu1 = rt_group_sum<G>(s1) / (float)C;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int c = 0; c < CL; ++c) { const float a = XR::p0(v[c]) - mu0, d = XR::p1(v[c]) - mu1; q0 += a * a; q1 += d * d; }
    const float r0 = 1.0f / sqrtf(rt_group_sum<G>(q0) / (float)C + eps), r1 = 1.0f / sqrtf(rt_group_sum<G>(q1) / (float)C + eps);
    {
        const unsigned so = (g == 0 && 2 * pp < rows) ? (unsigned)pp * 8u : RT_OOB;
        __builtin_amdgcn_raw_buffer_store_b64(rt_u32x2{__float_as_uint(mu0), __float_as_uint(mu1)}, rt_buf(mean + (size_t)n * P + p0, (unsigned)(P - p0) * 4u), so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(rt_u32x2{__float_as_uint(r0), __float_as_uint(r1)}, rt_buf(rstd + (size_t)n * P + p0, (unsigned)(P - p0) * 4u), so, 0, 0);
    }
#pragma unroll
    for (int j4 = 0; j4 < CL / 8; ++j4) {
        rt_u32x4 oa, ob;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 w4 = rt_par4<CL>(Lw, g, 2 * j4 + h), b4 = rt_par4<CL>(Lb, g, 2 * j4 + h);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = j4 * 8 + 4 * h + 2 * k;
                const float w0 = rt_f4(w4, 2 * k), w1 = rt_f4(w4, 2 * k + 1), b0 = rt_f4(b4, 2 * k), b1 = rt_f4(b4, 2 * k + 1);
                oa[2 * h + k] = rt_pack2((XR::p0(v[c]) - mu0) * r0 * w0 + b0, (XR::p0(v[c + 1]) - mu0) * r0 * w1 + b1);
                ob[2 * h + k] = rt_pack2((XR::p1(v[c]) - mu1) * r1 * w0 + b0, (XR::p1(v[c + 1]) - mu1) * r1 * w1 + b1);
            }
        }
        *(rt_u32x4*)(T + (2 * pp) * PITCH + g * (CL * 2) + j4 * 16) = oa;
        *(rt_u32x4*)(T + (2 * pp + 1) * PITCH + g * (CL * 2) + j4 * 16) = ob;
    }
    rt_lds_fence();
    if constexpr (PATCH) rt_tile_store_patch<C, PW>(T, rt_buf(y + (size_t)n * P * C, has_tile ? (unsigned)P * C * 2u : 0u), p0, Wimg, rows, lane);
    else rt_tile_store<C, PW>(T, rt_buf(y + ((size_t)n * P + p0) * C, (unsigned)rows * C * 2u), lane);
}

// dx[n,:,p] = rstd * (g*w - mean_C(g*w) - xhat * mean_C(g*w*xhat)) (NCHW, bf16 or fp32 like x) from g (bf16 NHWC or patch matrix), x;
// part[workgroup][0..C) = sum_p g * xhat, [C..2C) = sum_p g over its tiles.  Persistent waves (accumulators live in registers).
template <int CL, int G, bool XF32 = false, bool PATCH = false>
__global__ __launch_bounds__(256, 2) void ln_nchw_to_nhwc_bwd_reg_kernel(const uint16_t* __restrict__ gy, const void* __restrict__ xin,
                                                                        const float* __restrict__ w, const float* __restrict__ mean,
                                                                        const float* __restrict__ rstd, void* __restrict__ dxout,
                                                                        float* __restrict__ part, int N, int P, int tiles_per_image, int ntiles, int Wimg) {
    using XR = RtX<XF32>;
    const char* const x = (const char*)xin; char* const dx = (char*)dxout;
    using GE = RtGeom<CL, G>;
    constexpr int C = GE::C, PW = GE::PW, PITCH = GE::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* const Lw = (float*)(smem + 4 * GE::LDS_WAVE);
    bool staged = false;                                                       // w -> LDS behind the first tile's loads (each wave meets the barrier once)
    unsigned char* const T = smem + wave * GE::LDS_WAVE;
    const unsigned lds_T = (unsigned)(uintptr_t)SLAK_LDS(unsigned char, smem) + (unsigned)wave * GE::LDS_WAVE;
    RtDmaPlan<C, PW> plan; rt_dma_plan(plan, lane);
    const int pp = lane / G, g = lane & (G - 1);
    const unsigned rb = (unsigned)P * XR::EB;
    float accw[CL / 4], accb[CL / 4];
#pragma unroll
    for (int m = 0; m < CL / 4; ++m) { accw[m] = 0.f; accb[m] = 0.f; }
    const int gwave = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    for (int tile = gwave; tile < ntiles; tile += nwaves) {
        const int n = tile / tiles_per_image, p0 = (tile - n * tiles_per_image) * PW;
        const int rows = min(PW, P - p0);
        if constexpr (PATCH) { rt_dma_plan_patch(plan, lane, p0, Wimg, rows); rt_tile_dma<C, PW>(plan, gy + (size_t)n * P * C, (unsigned)P * C * 2u, lds_T); }
        else rt_tile_dma<C, PW>(plan, gy + ((size_t)n * P + p0) * C, (unsigned)rows * C * 2u, lds_T);
        const rt_rsrc rx = rt_buf(x + ((size_t)n * C * P + p0) * XR::EB, (unsigned)(C * P - p0) * XR::EB);
        const bool valid = 2 * pp < rows;
        const unsigned vo = valid ? (unsigned)(g * CL) * rb + (unsigned)pp * 2u * XR::EB : RT_OOB;
        typename XR::reg xv[CL];
#pragma unroll
        for (int c = 0; c < CL; ++c) xv[c] = XR::load(rx, vo, (unsigned)c * rb);
        const unsigned so = valid ? (unsigned)pp * 8u : RT_OOB;
        const rt_u32x2 mu2 = __builtin_amdgcn_raw_buffer_load_b64(rt_buf(mean + (size_t)n * P + p0, (unsigned)(P - p0) * 4u), so, 0, 0);
        const rt_u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rt_buf(rstd + (size_t)n * P + p0, (unsigned)(P - p0) * 4u), so, 0, 0);
        const float mu0 = __uint_as_float(mu2[0]), mu1 = __uint_as_float(mu2[1]), r0 = __uint_as_float(r2[0]), r1 = __uint_as_float(r2[1]);
        if (!staged) { for (int i = threadIdx.x; i < C; i += 256) Lw[i] = w[i]; __syncthreads(); staged = true; }
        rt_dma_wait();
        // g is read from the LDS tile in both passes (rows past the image were written as zeros: g = 0 there); x stays packed in registers
        const unsigned char* const ra = T + (2 * pp) * PITCH + g * (CL * 2);
        float s10 = 0.f, s11 = 0.f, s20 = 0.f, s21 = 0.f;
#pragma unroll
        for (int j4 = 0; j4 < CL / 8; ++j4) {
            const rt_u32x4 ta = *(const rt_u32x4*)(ra + j4 * 16), tb = *(const rt_u32x4*)(ra + PITCH + j4 * 16);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 w4 = rt_par4<CL>(Lw, g, 2 * j4 + h);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int k = 4 * h + k4, c = 8 * j4 + k;
                    const float gw0 = RT_CH(ta, k) * rt_f4(w4, k4), gw1 = RT_CH(tb, k) * rt_f4(w4, k4);
                    s10 += gw0; s11 += gw1;
                    s20 += gw0 * ((XR::p0(xv[c]) - mu0) * r0); s21 += gw1 * ((XR::p1(xv[c]) - mu1) * r1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                                  // one octet's LDS reads and unpacked values live at a time
        }
        const float m10 = rt_group_sum<G>(s10) / (float)C, m11 = rt_group_sum<G>(s11) / (float)C;
        const float m20 = rt_group_sum<G>(s20) / (float)C, m21 = rt_group_sum<G>(s21) / (float)C;
        // second pass from the PACKED registers again: without this the compiler keeps the unpacked fp32 values of the first pass alive
#pragma unroll
        for (int c = 0; c < CL; ++c) XR::opaque(xv[c]);
        const rt_rsrc rdx = rt_buf(dx + ((size_t)n * C * P + p0) * XR::EB, (unsigned)(C * P - p0) * XR::EB);
#pragma unroll
        for (int j4 = 0; j4 < CL / 8; ++j4) {
            const rt_u32x4 ta = *(const rt_u32x4*)(ra + j4 * 16), tb = *(const rt_u32x4*)(ra + PITCH + j4 * 16);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 w4 = rt_par4<CL>(Lw, g, 2 * j4 + h);
                float tw[4], ts[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int k = 4 * h + k4, c = 8 * j4 + k;
                    const float wc = rt_f4(w4, k4);
                    const float g0 = RT_CH(ta, k), g1 = RT_CH(tb, k);
                    const float xh0 = (XR::p0(xv[c]) - mu0) * r0, xh1 = (XR::p1(xv[c]) - mu1) * r1;
                    XR::store(rdx, vo, (unsigned)c * rb, r0 * (g0 * wc - m10 - xh0 * m20), r1 * (g1 * wc - m11 - xh1 * m21));
                    tw[k4] = g0 * xh0 + g1 * xh1; ts[k4] = g0 + g1;
                }
                accw[2 * j4 + h] += rt_fold4(tw[0], tw[1], tw[2], tw[3]);
                accb[2 * j4 + h] += rt_fold4(ts[0], ts[1], ts[2], ts[3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        rt_lds_fence();                                                        // tile consumed: the next iteration's DMA may overwrite it
    }
    if (!staged) { for (int i = threadIdx.x; i < C; i += 256) Lw[i] = w[i]; __syncthreads(); }       // a wave without tiles still stages its share
    rt_write_partials<CL, G>(accw, accb, part + (size_t)blockIdx.x * 2 * C, (float*)smem, wave, lane, [](float a, int) { return a; });
}

// out[n,c,p] (fp32 NCHW) = shortcut[n,c,p] + scale[n] * gamma[c] * z[n,p,c] (bf16 NHWC); optional bf16 copy of out.  One tile per wave.
template <int CL, int G, typename Tsc>
__global__ __launch_bounds__(256, 2) void scale_residual_fwd_reg_kernel(const Tsc* __restrict__ sc, const uint16_t* __restrict__ z,
                                                                       const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                       float* __restrict__ out, uint16_t* __restrict__ out16,
                                                                       int N, int P, int tiles_per_image, int ntiles) {
    using GE = RtGeom<CL, G>;
    constexpr int C = GE::C, PW = GE::PW, PITCH = GE::PITCH;
    constexpr bool SC32 = sizeof(Tsc) == 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* const Lg = (float*)(smem + 4 * GE::LDS_WAVE);
    const bool has_tile = blockIdx.x * 4 + wave < ntiles;                                   // a wave without a tile runs on an empty one (rows = 0)
    const int tile = has_tile ? blockIdx.x * 4 + wave : 0;
    unsigned char* const T = smem + wave * GE::LDS_WAVE;
    const int n = tile / tiles_per_image, p0 = (tile - n * tiles_per_image) * PW;
    const int pp = lane / G, g = lane & (G - 1);
    const int rows = has_tile ? min(PW, P - p0) : 0;
    {
        RtDmaPlan<C, PW> plan; rt_dma_plan(plan, lane);
        rt_tile_dma<C, PW>(plan, z + ((size_t)n * P + p0) * C, (unsigned)rows * C * 2u,
                           (unsigned)(uintptr_t)SLAK_LDS(unsigned char, smem) + (unsigned)wave * GE::LDS_WAVE);
    }
    const float sn = scale ? scale[n] : 1.0f;
    const unsigned P1 = (unsigned)P;
    const bool valid = 2 * pp < rows;
    const unsigned e0 = (unsigned)(g * CL) * P1 + 2u * (unsigned)pp;                     // element offset of the lane's first pixel
    const unsigned vo4 = valid ? e0 * 4u : RT_OOB, vo2 = valid ? e0 * 2u : RT_OOB;
    const rt_rsrc rsc = rt_buf(sc + (size_t)n * C * P + p0, (unsigned)(C * P - p0) * (unsigned)sizeof(Tsc));
    const rt_rsrc ro = rt_buf(out + (size_t)n * C * P + p0, (unsigned)(C * P - p0) * 4u);
    const rt_rsrc ro16 = rt_buf(out16 ? out16 + (size_t)n * C * P + p0 : nullptr, out16 ? (unsigned)(C * P - p0) * 2u : 0u);
    float sx[CL], sy[CL];                        // every channel row of the lane in flight at once (one memory latency for the whole tile)
#pragma unroll
    for (int c = 0; c < CL; ++c) {
        if constexpr (SC32) { const rt_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsc, vo4, (unsigned)c * P1 * 4u, 0); sx[c] = __uint_as_float(v[0]); sy[c] = __uint_as_float(v[1]); }
        else { const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rsc, vo2, (unsigned)c * P1 * 2u, 0); sx[c] = rt_lo(v); sy[c] = rt_hi(v); }
    }
    for (int i = threadIdx.x; i < C; i += 256) Lg[i] = gamma[i];
    __syncthreads();
    rt_dma_wait();
#pragma unroll
    for (int j4 = 0; j4 < CL / 8; ++j4) {
        const rt_u32x4 ta = *(const rt_u32x4*)(T + (2 * pp) * PITCH + g * (CL * 2) + j4 * 16);
        const rt_u32x4 tb = *(const rt_u32x4*)(T + (2 * pp + 1) * PITCH + g * (CL * 2) + j4 * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 g4 = rt_par4<CL>(Lg, g, 2 * j4 + h);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int k = 4 * h + k4, c = j4 * 8 + k;
                const float gm = rt_f4(g4, k4) * sn;
                const float ox = sx[c] + gm * RT_CH(ta, k), oy = sy[c] + gm * RT_CH(tb, k);
                __builtin_amdgcn_raw_buffer_store_b64(rt_u32x2{__float_as_uint(ox), __float_as_uint(oy)}, ro, vo4, (unsigned)c * P1 * 4u, 0);
                if (out16) __builtin_amdgcn_raw_buffer_store_b32(rt_pack2(ox, oy), ro16, vo2, (unsigned)c * P1 * 2u, 0);
            }
        }
    }
}

// The same for SMALL planes (P <= 256 by default: the 14 x 14 stage).  In the lane geometry above a wave covers 128 / G pixels -- 16 on C = 384 --, so
// each of its channel-row accesses is a 64-byte run of the fp32 NCHW tensors that carry 10 of the kernel's 12 bytes per element (measured:
// 3.2 TB/s against 5.7 on the 56 x 56 stage).  Here a wave takes CW channels of ONE image over 64 pixel pairs: a channel row is read and
// written as one 512-byte run (256 for the bf16 copy), and the lane fetches the CW channels of its two pixels of z as 16-byte pieces
// straight into registers (no LDS tile, no barrier).  Same arithmetic, same rounding.
template <int CW>
__global__ __launch_bounds__(256) void scale_residual_fwd_chan_kernel(const float* __restrict__ sc, const uint16_t* __restrict__ z,
                                                                      const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                      float* __restrict__ out, uint16_t* __restrict__ out16,
                                                                      int C, int P, int rounds, int nunits) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int unit = blockIdx.x * 4 + wave;                       // (image, channel group, round of 64 pixel pairs)
    if (unit >= nunits) return;
    const int groups = C / CW;
    const int rd = unit % rounds, t = unit / rounds, cg = t % groups, n = t / groups;
    const int pair = rd * 64 + lane;
    if (2 * pair >= P) return;                                    // (no barrier in the kernel: lanes may leave)
    const float sn = scale ? scale[n] : 1.0f;
    const uint16_t* zp = z + ((size_t)n * P + 2 * pair) * C + cg * CW;
    rt_u32x4 za[CW / 8], zb[CW / 8];
#pragma unroll
    for (int j = 0; j < CW / 8; ++j) { za[j] = *(const rt_u32x4*)(zp + j * 8); zb[j] = *(const rt_u32x4*)(zp + C + j * 8); }
    const size_t row0 = ((size_t)n * C + cg * CW) * P + 2 * pair;
    float sx[CW], sy[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) { const float2 v = *(const float2*)(sc + row0 + (size_t)c * P); sx[c] = v.x; sy[c] = v.y; }
    const float* gp = gamma + cg * CW;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const float gm = gp[c] * sn;                              // (wave-uniform: a scalar load)
        const float ox = sx[c] + gm * RT_CH(za[c >> 3], c & 7), oy = sy[c] + gm * RT_CH(zb[c >> 3], c & 7);
        *(float2*)(out + row0 + (size_t)c * P) = float2{ox, oy};
        if (out16) *(unsigned*)(out16 + row0 + (size_t)c * P) = rt_pack2(ox, oy);
    }
}

// dz[n,p,c] (bf16 NHWC) = scale[n] * gamma[c] * d[n,c,p], d = dout (fp32 NCHW) [+ dout16 (bf16 NCHW), the sum written to dsum];
// part[wave][0..C) = sum scale * d * z, [C..2C) = gamma * sum scale * d.  Persistent waves; z and dz share the wave's LDS tile.
template <int CL, int G>
__global__ __launch_bounds__(256, 2) void scale_residual_bwd_reg_kernel(const float* __restrict__ dout, const uint16_t* __restrict__ dout16,
                                                                       float* __restrict__ dsum, const uint16_t* __restrict__ z,
                                                                       const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                       uint16_t* __restrict__ dz, float* __restrict__ part,
                                                                       int N, int P, int tiles_per_image, int ntiles) {
    using GE = RtGeom<CL, G>;
    constexpr int C = GE::C, PW = GE::PW, PITCH = GE::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* const Lg = (float*)(smem + 4 * GE::LDS_WAVE);
    bool staged = false;                                                       // gamma -> LDS behind the first tile's loads
    unsigned char* const T = smem + wave * GE::LDS_WAVE;
    const unsigned lds_T = (unsigned)(uintptr_t)SLAK_LDS(unsigned char, smem) + (unsigned)wave * GE::LDS_WAVE;
    RtDmaPlan<C, PW> plan; rt_dma_plan(plan, lane);
    const int pp = lane / G, g = lane & (G - 1);
    const unsigned P1 = (unsigned)P;
    float accg[CL / 4], accs[CL / 4];
#pragma unroll
    for (int m = 0; m < CL / 4; ++m) { accg[m] = 0.f; accs[m] = 0.f; }
    const int gwave = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    for (int tile = gwave; tile < ntiles; tile += nwaves) {
        const int n = tile / tiles_per_image, p0 = (tile - n * tiles_per_image) * PW;
        const int rows = min(PW, P - p0);
        rt_tile_dma<C, PW>(plan, z + ((size_t)n * P + p0) * C, (unsigned)rows * C * 2u, lds_T);
        const float sn = scale ? scale[n] : 1.0f;
        const bool valid = 2 * pp < rows;
        const unsigned e0 = (unsigned)(g * CL) * P1 + 2u * (unsigned)pp;
        const unsigned vo4 = valid ? e0 * 4u : RT_OOB, vo2 = valid ? e0 * 2u : RT_OOB;
        const rt_rsrc rd = rt_buf(dout + (size_t)n * C * P + p0, (unsigned)(C * P - p0) * 4u);
        const rt_rsrc rd16 = rt_buf(dout16 ? dout16 + (size_t)n * C * P + p0 : nullptr, dout16 ? (unsigned)(C * P - p0) * 2u : 0u);
        const rt_rsrc rds = rt_buf(dout16 ? dsum + (size_t)n * C * P + p0 : nullptr, dout16 ? (unsigned)(C * P - p0) * 4u : 0u);
        rt_u32x2 dv[CL]; unsigned dh[CL];        // every channel row of the lane in flight at once
#pragma unroll
        for (int c = 0; c < CL; ++c) {
            dv[c] = __builtin_amdgcn_raw_buffer_load_b64(rd, vo4, (unsigned)c * P1 * 4u, 0);                       // lanes past the image: zeros
            dh[c] = dout16 ? __builtin_amdgcn_raw_buffer_load_b32(rd16, vo2, (unsigned)c * P1 * 2u, 0) : 0u;
        }
        if (!staged) { for (int i = threadIdx.x; i < C; i += 256) Lg[i] = gamma[i]; __syncthreads(); staged = true; }
        rt_dma_wait();
#pragma unroll
        for (int j4 = 0; j4 < CL / 8; ++j4) {
            unsigned char* const tpa = T + (2 * pp) * PITCH + g * (CL * 2) + j4 * 16;
            unsigned char* const tpb = tpa + PITCH;
            const rt_u32x4 ta = *(const rt_u32x4*)tpa, tb = *(const rt_u32x4*)tpb;       // rows past the image: zeros
            float d0[8], d1[8], tg[8], ts[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned c = (unsigned)(j4 * 8 + k);
                float vx = __uint_as_float(dv[c][0]), vy = __uint_as_float(dv[c][1]);
                if (dout16) {
                    vx += rt_lo(dh[c]); vy += rt_hi(dh[c]);
                    __builtin_amdgcn_raw_buffer_store_b64(rt_u32x2{__float_as_uint(vx), __float_as_uint(vy)}, rds, vo4, c * P1 * 4u, 0);
                }
                d0[k] = vx * sn; d1[k] = vy * sn;                                       // the per-image scale goes into the terms: a wave's tiles span images
                tg[k] = d0[k] * RT_CH(ta, k) + d1[k] * RT_CH(tb, k);
                ts[k] = d0[k] + d1[k];
            }
            rt_u32x4 oa, ob;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 g4 = rt_par4<CL>(Lg, g, 2 * j4 + h);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    oa[2 * h + k] = rt_pack2(rt_f4(g4, 2 * k) * d0[4 * h + 2 * k], rt_f4(g4, 2 * k + 1) * d0[4 * h + 2 * k + 1]);
                    ob[2 * h + k] = rt_pack2(rt_f4(g4, 2 * k) * d1[4 * h + 2 * k], rt_f4(g4, 2 * k + 1) * d1[4 * h + 2 * k + 1]);
                }
            }
            *(rt_u32x4*)tpa = oa; *(rt_u32x4*)tpb = ob;
            accg[2 * j4] += rt_fold4(tg[0], tg[1], tg[2], tg[3]);
            accg[2 * j4 + 1] += rt_fold4(tg[4], tg[5], tg[6], tg[7]);
            accs[2 * j4] += rt_fold4(ts[0], ts[1], ts[2], ts[3]);
            accs[2 * j4 + 1] += rt_fold4(ts[4], ts[5], ts[6], ts[7]);
        }
        rt_lds_fence();
        rt_tile_store<C, PW>(T, rt_buf(dz + ((size_t)n * P + p0) * C, (unsigned)rows * C * 2u), lane);
        rt_lds_fence();
    }
    if (!staged) { for (int i = threadIdx.x; i < C; i += 256) Lg[i] = gamma[i]; __syncthreads(); }   // a wave without tiles
    rt_write_partials<CL, G>(accg, accs, part + (size_t)blockIdx.x * 2 * C, (float*)smem, wave, lane, [gamma](float a, int c) { return a * gamma[c]; });
}

// The backward residual step for SMALL planes, a wave per (image, CW channels, 64 pixel pairs) like scale_residual_fwd_chan_kernel: the
// fp32 gradient rows (10 of the 14 bytes per element with the bf16 second stream and the summed output) move as 512-byte runs, z and dz
// as 16-byte pieces from / to registers.  The wave holds ALL of its channels' pixels of the round, so the per-channel sums are one fold
// over the wave (rt_fold4 + four shuffles) and land in row (image, round) of the partial matrix: columns of the wave's channels only.
template <int CW>
__global__ __launch_bounds__(256) void scale_residual_bwd_chan_kernel(const float* __restrict__ dout, const uint16_t* __restrict__ dout16,
                                                                      float* __restrict__ dsum, const uint16_t* __restrict__ z,
                                                                      const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                      uint16_t* __restrict__ dz, float* __restrict__ part,
                                                                      int C, int P, int rounds, int nunits) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int unit = blockIdx.x * 4 + wave;                       // (image, channel group, round of 64 pixel pairs)
    if (unit >= nunits) return;                                   // (no workgroup barrier in the kernel)
    const int groups = C / CW;
    const int rd = unit % rounds, t = unit / rounds, cg = t % groups, n = t / groups;
    const int pair = rd * 64 + lane;
    const bool valid = 2 * pair < P;
    const float sn = scale ? scale[n] : 1.0f;
    const size_t zoff = ((size_t)n * P + 2 * pair) * C + cg * CW;
    const size_t row0 = ((size_t)n * C + cg * CW) * P + 2 * pair;
    rt_u32x4 za[CW / 8], zb[CW / 8];
    float vx[CW], vy[CW];
    if (valid) {
#pragma unroll
        for (int j = 0; j < CW / 8; ++j) { za[j] = *(const rt_u32x4*)(z + zoff + j * 8); zb[j] = *(const rt_u32x4*)(z + zoff + C + j * 8); }
#pragma unroll
        for (int c = 0; c < CW; ++c) { const float2 v = *(const float2*)(dout + row0 + (size_t)c * P); vx[c] = v.x; vy[c] = v.y; }
        if (dout16) {
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const unsigned h = *(const unsigned*)(dout16 + row0 + (size_t)c * P);
                vx[c] += rt_lo(h); vy[c] += rt_hi(h);
                *(float2*)(dsum + row0 + (size_t)c * P) = float2{vx[c], vy[c]};
            }
        }
    } else {                                                       // lanes past the image contribute zeros to the sums
#pragma unroll
        for (int j = 0; j < CW / 8; ++j) { za[j] = rt_u32x4{0u, 0u, 0u, 0u}; zb[j] = za[j]; }
#pragma unroll
        for (int c = 0; c < CW; ++c) { vx[c] = 0.f; vy[c] = 0.f; }
    }
    const float* gp = gamma + cg * CW;
    float accg[CW / 4], accs[CW / 4];
#pragma unroll
    for (int j = 0; j < CW / 8; ++j) {
        float d0[8], d1[8], tg[8], ts[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = j * 8 + k;
            d0[k] = vx[c] * sn; d1[k] = vy[c] * sn;
            tg[k] = d0[k] * RT_CH(za[j], k) + d1[k] * RT_CH(zb[j], k);
            ts[k] = d0[k] + d1[k];
        }
        rt_u32x4 oa, ob;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g0 = gp[j * 8 + 2 * k], g1 = gp[j * 8 + 2 * k + 1];       // (wave-uniform: scalar loads)
            oa[k] = rt_pack2(g0 * d0[2 * k], g1 * d0[2 * k + 1]);
            ob[k] = rt_pack2(g0 * d1[2 * k], g1 * d1[2 * k + 1]);
        }
        if (valid) { *(rt_u32x4*)(dz + zoff + j * 8) = oa; *(rt_u32x4*)(dz + zoff + C + j * 8) = ob; }
        accg[2 * j] = rt_fold4(tg[0], tg[1], tg[2], tg[3]); accg[2 * j + 1] = rt_fold4(tg[4], tg[5], tg[6], tg[7]);
        accs[2 * j] = rt_fold4(ts[0], ts[1], ts[2], ts[3]); accs[2 * j + 1] = rt_fold4(ts[4], ts[5], ts[6], ts[7]);
    }
    // 16-lane row r of the wave holds channel 4m + {0,2,1,3}[r]: finish over the row's lanes, lane 0 of the row writes
    const int rowi = lane >> 4, cofs = rowi == 0 ? 0 : rowi == 1 ? 2 : rowi == 2 ? 1 : 3;
    float* const prow = part + (size_t)(n * rounds + rd) * 2 * C + cg * CW;
#pragma unroll
    for (int m = 0; m < CW / 4; ++m) {
        float a = accg[m], b = accs[m];
#pragma unroll
        for (int k = 1; k < 16; k <<= 1) { a += __shfl_xor(a, k, 64); b += __shfl_xor(b, k, 64); }
        if ((lane & 15) == 0) { const int c = 4 * m + cofs; prow[c] = a; prow[C + c] = b * gp[c]; }
    }
}

// LayerNorm backward for SMALL planes in the same mapping: a WORKGROUP per (image, 64 pixel pairs) with one wave per group of CW channels
// (<= 8 waves), so the bf16 NCHW rows move as 256-byte runs; the per-pixel sums over C are the sum of the waves' partials through LDS (one
// workgroup barrier), the per-channel sums one fold over the wave.  (The forward in this mapping measured slower inside a training step than
// the pixel-tile kernel, 16.7 vs 14.9 us on 384 x 14 x 14 -- its input was just written and is cache resident -- and was dropped.)
template <int CW>
__global__ __launch_bounds__(512) void ln_nchw_to_nhwc_bwd_chan_kernel(const uint16_t* __restrict__ gy, const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                       uint16_t* __restrict__ dx, float* __restrict__ part, int C, int P, int rounds) {
    __shared__ float4 red[8][64];
    const int lane = threadIdx.x & 63, cg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (int)(blockDim.x >> 6);
    const int rd = blockIdx.x % rounds, n = blockIdx.x / rounds;
    const int pair = rd * 64 + lane;
    const bool valid = 2 * pair < P;
    const size_t row0 = ((size_t)n * C + cg * CW) * P + 2 * pair, goff = ((size_t)n * P + 2 * pair) * C + cg * CW;
    unsigned xv[CW];
    rt_u32x4 ga[CW / 8], gb[CW / 8];
    float mu0 = 0.f, mu1 = 0.f, r0 = 0.f, r1 = 0.f;
    if (valid) {
#pragma unroll
        for (int j = 0; j < CW / 8; ++j) { ga[j] = *(const rt_u32x4*)(gy + goff + j * 8); gb[j] = *(const rt_u32x4*)(gy + goff + C + j * 8); }
#pragma unroll
        for (int c = 0; c < CW; ++c) xv[c] = *(const unsigned*)(x + row0 + (size_t)c * P);
        const float2 m2 = *(const float2*)(mean + (size_t)n * P + 2 * pair), s2 = *(const float2*)(rstd + (size_t)n * P + 2 * pair);
        mu0 = m2.x; mu1 = m2.y; r0 = s2.x; r1 = s2.y;
    } else {                                                       // lanes past the image contribute zeros everywhere
#pragma unroll
        for (int j = 0; j < CW / 8; ++j) { ga[j] = rt_u32x4{0u, 0u, 0u, 0u}; gb[j] = ga[j]; }
#pragma unroll
        for (int c = 0; c < CW; ++c) xv[c] = 0u;
    }
    const float* wp = w + cg * CW;                                 // (wave-uniform: scalar loads)
    float s10 = 0.f, s11 = 0.f, s20 = 0.f, s21 = 0.f;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const float gw0 = RT_CH(ga[c >> 3], c & 7) * wp[c], gw1 = RT_CH(gb[c >> 3], c & 7) * wp[c];
        s10 += gw0; s11 += gw1;
        s20 += gw0 * ((rt_lo(xv[c]) - mu0) * r0); s21 += gw1 * ((rt_hi(xv[c]) - mu1) * r1);
    }
#pragma unroll
    for (int c = 0; c < CW; ++c) asm volatile("" : "+v"(xv[c]));
#pragma unroll
    for (int j = 0; j < CW / 8; ++j) { asm volatile("" : "+v"(ga[j])); asm volatile("" : "+v"(gb[j])); }
    red[cg][lane] = float4{s10, s11, s20, s21};
    __syncthreads();
    float4 t = float4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nw; ++k) { const float4 q = red[k][lane]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
    const float m10 = t.x / (float)C, m11 = t.y / (float)C, m20 = t.z / (float)C, m21 = t.w / (float)C;
    float accw[CW / 4], accb[CW / 4];
#pragma unroll
    for (int j = 0; j < CW / 8; ++j) {
        float tw[8], ts[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = j * 8 + k;
            const float wc = wp[c];
            const float g0 = RT_CH(ga[j], k), g1 = RT_CH(gb[j], k);
            const float xh0 = (rt_lo(xv[c]) - mu0) * r0, xh1 = (rt_hi(xv[c]) - mu1) * r1;
            if (valid) *(unsigned*)(dx + row0 + (size_t)c * P) = rt_pack2(r0 * (g0 * wc - m10 - xh0 * m20), r1 * (g1 * wc - m11 - xh1 * m21));
            tw[k] = g0 * xh0 + g1 * xh1; ts[k] = g0 + g1;
        }
        accw[2 * j] = rt_fold4(tw[0], tw[1], tw[2], tw[3]); accw[2 * j + 1] = rt_fold4(tw[4], tw[5], tw[6], tw[7]);
        accb[2 * j] = rt_fold4(ts[0], ts[1], ts[2], ts[3]); accb[2 * j + 1] = rt_fold4(ts[4], ts[5], ts[6], ts[7]);
    }
    const int rowi = lane >> 4, cofs = rowi == 0 ? 0 : rowi == 1 ? 2 : rowi == 2 ? 1 : 3;
    float* const prow = part + (size_t)(n * rounds + rd) * 2 * C + cg * CW;
#pragma unroll
    for (int m = 0; m < CW / 4; ++m) {
        float a = accw[m], bb = accb[m];
#pragma unroll
        for (int k = 1; k < 16; k <<= 1) { a += __shfl_xor(a, k, 64); bb += __shfl_xor(bb, k, 64); }
        if ((lane & 15) == 0) { const int c = 4 * m + cofs; prow[c] = a; prow[C + c] = bb; }
    }
}

// ---- ODD plane sizes (P = 49: the 7 x 7 stage) in the same mapping, ONE pixel per lane (pixel pairs would start channel rows of the
// bf16 tensors at 2-byte aligned dwords).  Channel rows are 196-byte (fp32) / 98-byte (bf16) runs instead of the 4 / 2 bytes per lane and
// row the LDS-tile kernels of block_tail.hip end up with on this stage (0.1-0.2 of the HBM roofline).
__device__ __forceinline__ float rt_ld16(const uint16_t* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ uint16_t rt_bf(float v) { return (uint16_t)(rt_pack2(v, 0.f) & 0xffffu); }

template <int CW>
__global__ __launch_bounds__(256) void scale_residual_fwd_chan1_kernel(const float* __restrict__ sc, const uint16_t* __restrict__ z,
                                                                       const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                       float* __restrict__ out, uint16_t* __restrict__ out16,
                                                                       int C, int P, int rounds, int nunits) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int unit = blockIdx.x * 4 + wave;                       // (image, channel group, round of 64 pixels)
    if (unit >= nunits) return;
    const int groups = C / CW;
    const int rd = unit % rounds, t = unit / rounds, cg = t % groups, n = t / groups;
    const int px = rd * 64 + lane;
    if (px >= P) return;
    const float sn = scale ? scale[n] : 1.0f;
    const uint16_t* zp = z + ((size_t)n * P + px) * C + cg * CW;
    rt_u32x4 za[CW / 8];
#pragma unroll
    for (int j = 0; j < CW / 8; ++j) za[j] = *(const rt_u32x4*)(zp + j * 8);
    const size_t row0 = ((size_t)n * C + cg * CW) * P + px;
    float sx[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) sx[c] = sc[row0 + (size_t)c * P];
    const float* gp = gamma + cg * CW;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const float o = sx[c] + (gp[c] * sn) * RT_CH(za[c >> 3], c & 7);
        out[row0 + (size_t)c * P] = o;
        if (out16) out16[row0 + (size_t)c * P] = rt_bf(o);
    }
}

template <int CW>
__global__ __launch_bounds__(256) void scale_residual_bwd_chan1_kernel(const float* __restrict__ dout, const uint16_t* __restrict__ dout16,
                                                                       float* __restrict__ dsum, const uint16_t* __restrict__ z,
                                                                       const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                       uint16_t* __restrict__ dz, float* __restrict__ part,
                                                                       int C, int P, int rounds, int nunits) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= nunits) return;
    const int groups = C / CW;
    const int rd = unit % rounds, t = unit / rounds, cg = t % groups, n = t / groups;
    const int px = rd * 64 + lane;
    const bool valid = px < P;
    const float sn = scale ? scale[n] : 1.0f;
    const size_t zoff = ((size_t)n * P + px) * C + cg * CW;
    const size_t row0 = ((size_t)n * C + cg * CW) * P + px;
    rt_u32x4 za[CW / 8];
    float vx[CW];
    if (valid) {
#pragma unroll
        for (int j = 0; j < CW / 8; ++j) za[j] = *(const rt_u32x4*)(z + zoff + j * 8);
#pragma unroll
        for (int c = 0; c < CW; ++c) vx[c] = dout[row0 + (size_t)c * P];
        if (dout16) {
#pragma unroll
            for (int c = 0; c < CW; ++c) { vx[c] += rt_ld16(dout16 + row0 + (size_t)c * P); dsum[row0 + (size_t)c * P] = vx[c]; }
        }
    } else {
#pragma unroll
        for (int j = 0; j < CW / 8; ++j) za[j] = rt_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < CW; ++c) vx[c] = 0.f;
    }
    const float* gp = gamma + cg * CW;
    float accg[CW / 4], accs[CW / 4];
#pragma unroll
    for (int j = 0; j < CW / 8; ++j) {
        float d0[8], tg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { d0[k] = vx[j * 8 + k] * sn; tg[k] = d0[k] * RT_CH(za[j], k); }
        rt_u32x4 oa;
#pragma unroll
        for (int k = 0; k < 4; ++k) oa[k] = rt_pack2(gp[j * 8 + 2 * k] * d0[2 * k], gp[j * 8 + 2 * k + 1] * d0[2 * k + 1]);
        if (valid) *(rt_u32x4*)(dz + zoff + j * 8) = oa;
        accg[2 * j] = rt_fold4(tg[0], tg[1], tg[2], tg[3]); accg[2 * j + 1] = rt_fold4(tg[4], tg[5], tg[6], tg[7]);
        accs[2 * j] = rt_fold4(d0[0], d0[1], d0[2], d0[3]); accs[2 * j + 1] = rt_fold4(d0[4], d0[5], d0[6], d0[7]);
    }
    const int rowi = lane >> 4, cofs = rowi == 0 ? 0 : rowi == 1 ? 2 : rowi == 2 ? 1 : 3;
    float* const prow = part + (size_t)(n * rounds + rd) * 2 * C + cg * CW;
#pragma unroll
    for (int m = 0; m < CW / 4; ++m) {
        float a = accg[m], b = accs[m];
#pragma unroll
        for (int k = 1; k < 16; k <<= 1) { a += __shfl_xor(a, k, 64); b += __shfl_xor(b, k, 64); }
        if ((lane & 15) == 0) { const int c = 4 * m + cofs; prow[c] = a; prow[C + c] = b * gp[c]; }
    }
}

template <int CW>
__global__ __launch_bounds__(1024) void ln_nchw_to_nhwc_fwd_chan1_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                                         uint16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                                         int C, int P, int rounds, float eps) {
    __shared__ float red[2][16][64];
    const int lane = threadIdx.x & 63, cg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (int)(blockDim.x >> 6);
    const int rd = blockIdx.x % rounds, n = blockIdx.x / rounds;
    const int px = rd * 64 + lane;
    const bool valid = px < P;
    const size_t row0 = ((size_t)n * C + cg * CW) * P + px, yoff = ((size_t)n * P + px) * C + cg * CW;
    float v[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) v[c] = valid ? rt_ld16(x + row0 + (size_t)c * P) : 0.f;
    float s0 = 0.f;
#pragma unroll
    for (int c = 0; c < CW; ++c) s0 += v[c];
    red[0][cg][lane] = s0;
    __syncthreads();
    float t0 = 0.f;
    for (int k = 0; k < nw; ++k) t0 += red[0][k][lane];
    const float mu0 = t0 / (float)C;
    float q0 = 0.f;
#pragma unroll
    for (int c = 0; c < CW; ++c) { const float a = v[c] - mu0; q0 += a * a; }
    red[1][cg][lane] = q0;
    __syncthreads();
    t0 = 0.f;
    for (int k = 0; k < nw; ++k) t0 += red[1][k][lane];
    const float r0 = 1.0f / sqrtf(t0 / (float)C + eps);
    if (!valid) return;
    if (cg == 0) { mean[(size_t)n * P + px] = mu0; rstd[(size_t)n * P + px] = r0; }
    const float* wp = w + cg * CW; const float* bp = b + cg * CW;
#pragma unroll
    for (int j = 0; j < CW / 8; ++j) {
        rt_u32x4 oa;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = j * 8 + 2 * k;
            oa[k] = rt_pack2((v[c] - mu0) * r0 * wp[c] + bp[c], (v[c + 1] - mu0) * r0 * wp[c + 1] + bp[c + 1]);
        }
        *(rt_u32x4*)(y + yoff + j * 8) = oa;
    }
}

template <int CW>
__global__ __launch_bounds__(1024) void ln_nchw_to_nhwc_bwd_chan1_kernel(const uint16_t* __restrict__ gy, const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                         uint16_t* __restrict__ dx, float* __restrict__ part, int C, int P, int rounds) {
    __shared__ float2 red[16][64];
    const int lane = threadIdx.x & 63, cg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (int)(blockDim.x >> 6);
    const int rd = blockIdx.x % rounds, n = blockIdx.x / rounds;
    const int px = rd * 64 + lane;
    const bool valid = px < P;
    const size_t row0 = ((size_t)n * C + cg * CW) * P + px, goff = ((size_t)n * P + px) * C + cg * CW;
    float xh[CW];                                                  // normalised x
    rt_u32x4 ga[CW / 8];
    float r0 = 0.f;
    if (valid) {
#pragma unroll
        for (int j = 0; j < CW / 8; ++j) ga[j] = *(const rt_u32x4*)(gy + goff + j * 8);
        const float mu0 = mean[(size_t)n * P + px]; r0 = rstd[(size_t)n * P + px];
#pragma unroll
        for (int c = 0; c < CW; ++c) xh[c] = (rt_ld16(x + row0 + (size_t)c * P) - mu0) * r0;
    } else {
#pragma unroll
        for (int j = 0; j < CW / 8; ++j) ga[j] = rt_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < CW; ++c) xh[c] = 0.f;
    }
    const float* wp = w + cg * CW;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CW; ++c) { const float gw = RT_CH(ga[c >> 3], c & 7) * wp[c]; s1 += gw; s2 += gw * xh[c]; }
    red[cg][lane] = float2{s1, s2};
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < nw; ++k) { const float2 q = red[k][lane]; t1 += q.x; t2 += q.y; }
    const float m1 = t1 / (float)C, m2 = t2 / (float)C;
    float accw[CW / 4], accb[CW / 4];
#pragma unroll
    for (int j = 0; j < CW / 8; ++j) {
        float tw[8], ts[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = j * 8 + k;
            const float g0 = RT_CH(ga[j], k);
            if (valid) dx[row0 + (size_t)c * P] = rt_bf(r0 * (g0 * wp[c] - m1 - xh[c] * m2));
            tw[k] = g0 * xh[c]; ts[k] = g0;
        }
        accw[2 * j] = rt_fold4(tw[0], tw[1], tw[2], tw[3]); accw[2 * j + 1] = rt_fold4(tw[4], tw[5], tw[6], tw[7]);
        accb[2 * j] = rt_fold4(ts[0], ts[1], ts[2], ts[3]); accb[2 * j + 1] = rt_fold4(ts[4], ts[5], ts[6], ts[7]);
    }
    const int rowi = lane >> 4, cofs = rowi == 0 ? 0 : rowi == 1 ? 2 : rowi == 2 ? 1 : 3;
    float* const prow = part + (size_t)(n * rounds + rd) * 2 * C + cg * CW;
#pragma unroll
    for (int m = 0; m < CW / 4; ++m) {
        float a = accw[m], bb = accb[m];
#pragma unroll
        for (int k = 1; k < 16; k <<= 1) { a += __shfl_xor(a, k, 64); bb += __shfl_xor(bb, k, 64); }
        if ((lane & 15) == 0) { const int c = 4 * m + cofs; prow[c] = a; prow[C + c] = bb; }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
template <typename K> static int rt_persistent_grid(K k, size_t lds, int ntiles) {
    static thread_local int per_cu = 0, cus = 0;                       // one (kernel, lds) pair per instantiation of this template
    if (per_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb;
    }
    int grid = per_cu * cus; if (grid > 2048) grid = 2048;             // <= 8192 partial rows (slak_block_tail_workspace_bytes)
    const int need = (ntiles + 3) / 4;
    return need < grid ? need : grid;
}
template <typename K> static int rt_set_lds(K k, size_t lds) {
    return slak_set_max_lds((const void*)k, lds) ? 0 : 1;
}

template <int CL, int G, bool XF32 = false, bool PATCH = false>
static int launch_ln_fwd_reg(const void* x, const float* w, const float* b, uint16_t* y, float* mean, float* rstd, int N, int P, float eps, hipStream_t st, int Wimg = 0) {
    using GE = RtGeom<CL, G>;
    const int tpi = (P + GE::PW - 1) / GE::PW, ntiles = N * tpi;
    const size_t lds = (size_t)4 * GE::LDS_WAVE + (size_t)2 * GE::C * sizeof(float);
    auto k = ln_nchw_to_nhwc_fwd_reg_kernel<CL, G, XF32, PATCH>;
    if (rt_set_lds(k, lds)) return SLAK_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), lds, st, x, w, b, y, mean, rstd, N, P, eps, tpi, ntiles, Wimg);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}
template <int CL, int G, bool XF32 = false, bool PATCH = false>
static int launch_ln_bwd_reg(const uint16_t* g, const void* x, const float* w, const float* mean, const float* rstd, void* dx, float* part,
                             int* rows, int N, int P, hipStream_t st, int Wimg = 0) {
    using GE = RtGeom<CL, G>;
    const int tpi = (P + GE::PW - 1) / GE::PW, ntiles = N * tpi;
    const size_t lds = (size_t)4 * GE::LDS_WAVE + (size_t)2 * GE::C * sizeof(float);
    auto k = ln_nchw_to_nhwc_bwd_reg_kernel<CL, G, XF32, PATCH>;
    if (rt_set_lds(k, lds)) return SLAK_ERR_LAUNCH;
    const int grid = rt_persistent_grid(k, lds, ntiles);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, g, x, w, mean, rstd, dx, part, N, P, tpi, ntiles, Wimg);
    SLAK_LAUNCH_CHECK();
    *rows = grid;
    return SLAK_OK;
}
template <int CL, int G, typename Tsc>
static int launch_sr_fwd_reg(const Tsc* sc, const uint16_t* z, const float* gamma, const float* scale, float* out, uint16_t* out16, int N, int P, hipStream_t st) {
    using GE = RtGeom<CL, G>;
    const int tpi = (P + GE::PW - 1) / GE::PW, ntiles = N * tpi;
    const size_t lds = (size_t)4 * GE::LDS_WAVE + (size_t)2 * GE::C * sizeof(float);
    auto k = scale_residual_fwd_reg_kernel<CL, G, Tsc>;
    if (rt_set_lds(k, lds)) return SLAK_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), lds, st, sc, z, gamma, scale, out, out16, N, P, tpi, ntiles);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}
template <int CL, int G>
static int launch_sr_bwd_reg(const float* dout, const uint16_t* dout16, float* dsum, const uint16_t* z, const float* gamma, const float* scale,
                             uint16_t* dz, float* part, int* rows, int N, int P, hipStream_t st) {
    using GE = RtGeom<CL, G>;
    const int tpi = (P + GE::PW - 1) / GE::PW, ntiles = N * tpi;
    const size_t lds = (size_t)4 * GE::LDS_WAVE + (size_t)2 * GE::C * sizeof(float);
    auto k = scale_residual_bwd_reg_kernel<CL, G>;
    if (rt_set_lds(k, lds)) return SLAK_ERR_LAUNCH;
    const int grid = rt_persistent_grid(k, lds, ntiles);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, dout, dout16, dsum, z, gamma, scale, dz, part, N, P, tpi, ntiles);
    SLAK_LAUNCH_CHECK();
    *rows = grid;
    return SLAK_OK;
}

static int rt_chan_max_p() {                 // largest plane (pixels) the channel-group kernels take (dev: SLAK_RT_CHAN_MAXP)
    static const int v = [] { const char* e = slak_dev_getenv("SLAK_RT_CHAN_MAXP"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 256; }();
    return v;
}
static bool rt_chan_waves() {                // SLAK_RT_CHAN=0: small planes keep the pixel-tile residual kernel (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_RT_CHAN"); return !(e && e[0] == '0'); }();
    return v;
}
template <int CW>
static int launch_sr_fwd_chan(const float* sc, const uint16_t* z, const float* gamma, const float* scale, float* out, uint16_t* out16, int N, int C, int P, hipStream_t st) {
    const int rounds = (P / 2 + 63) / 64, nunits = N * (C / CW) * rounds;
    hipLaunchKernelGGL(scale_residual_fwd_chan_kernel<CW>, dim3((unsigned)((nunits + 3) / 4)), dim3(256), 0, st, sc, z, gamma, scale, out, out16, C, P, rounds, nunits);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}
template <int CW>
static int launch_sr_bwd_chan(const float* dout, const uint16_t* dout16, float* dsum, const uint16_t* z, const float* gamma, const float* scale,
                              uint16_t* dz, float* part, int* rows, int N, int C, int P, hipStream_t st) {
    const int rounds = (P / 2 + 63) / 64, nunits = N * (C / CW) * rounds;
    hipLaunchKernelGGL(scale_residual_bwd_chan_kernel<CW>, dim3((unsigned)((nunits + 3) / 4)), dim3(256), 0, st, dout, dout16, dsum, z, gamma, scale, dz, part,
                       C, P, rounds, nunits);
    SLAK_LAUNCH_CHECK();
    *rows = N * rounds;
    return SLAK_OK;
}
template <int CW>
static int launch_ln_bwd_chan(const uint16_t* g, const uint16_t* x, const float* w, const float* mean, const float* rstd, uint16_t* dx, float* part, int* rows,
                              int N, int C, int P, hipStream_t st) {
    const int rounds = (P / 2 + 63) / 64;
    hipLaunchKernelGGL(ln_nchw_to_nhwc_bwd_chan_kernel<CW>, dim3((unsigned)(N * rounds)), dim3((unsigned)(C / CW * 64)), 0, st, g, x, w, mean, rstd, dx, part, C, P, rounds);
    SLAK_LAUNCH_CHECK();
    *rows = N * rounds;
    return SLAK_OK;
}
// odd plane sizes: one pixel per lane; channels per wave (the LayerNorm kernels: at most 16 waves); 0 = not covered
static int rt_chan1_cw(int C, int P, int N, int max_waves) {
    if (!(P & 1) || P > rt_chan_max_p() || (long long)N * ((P + 63) / 64) > 8192 || !rt_chan_waves()) return 0;
    if (C % 48 == 0 && C / 48 <= max_waves) return 48;
    if (C % 32 == 0 && C / 32 <= max_waves) return 32;
    return 0;
}
// small planes: channels per wave of the workgroup-per-(image, round) LayerNorm kernels (at most 8 waves); 0 = not covered
static int rt_ln_chan_cw(int C, int P, int N) {
    if ((P & 1) || P > rt_chan_max_p() || (long long)N * ((P / 2 + 63) / 64) > 8192 || !rt_chan_waves()) return 0;
    if (C % 48 == 0 && C / 48 <= 8) return 48;
    if (C % 64 == 0 && C / 64 <= 8) return 64;
    return 0;
}
static bool rt_wide_lanes() {                // SLAK_RT_WIDE=0: C = 384 keeps 48 channels per lane on 8 lanes per pixel pair (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_RT_WIDE"); return !(e && e[0] == '0'); }();
    return v;
}
// The instantiations: (channels per lane, lanes per pixel) per channel count.  SLAK_ERR_UNSUPPORTED = none for this C (the caller
// runs the LDS-tile kernels of block_tail.hip).
#define SLAK_RT_DISPATCH(C, CALL)                 \
    if (P & 1) return SLAK_ERR_UNSUPPORTED;       \
    switch (C) {                                  \
        case 64: { CALL(32, 2); }                 \
        case 96: { CALL(48, 2); }                 \
        case 128: { CALL(64, 2); }                \
        case 192: { CALL(48, 4); }                \
        case 256: { CALL(64, 4); }                \
        case 384: { CALL(48, 8); }                \
        case 512: { CALL(64, 8); }                \
        default: return SLAK_ERR_UNSUPPORTED;     \
    }

int launch_ln_nchw_to_nhwc_fwd_reg(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int N, int C, int P, float eps, hipStream_t st) {
    { const int cw1 = rt_chan1_cw(C, P, N, 16), rounds = (P + 63) / 64;
      if (cw1 == 48) { hipLaunchKernelGGL(ln_nchw_to_nhwc_fwd_chan1_kernel<48>, dim3((unsigned)(N * rounds)), dim3((unsigned)(C / 48 * 64)), 0, st, (const uint16_t*)x, w, b, (uint16_t*)y, mean, rstd, C, P, rounds, eps); SLAK_LAUNCH_CHECK(); return SLAK_OK; }
      if (cw1 == 32) { hipLaunchKernelGGL(ln_nchw_to_nhwc_fwd_chan1_kernel<32>, dim3((unsigned)(N * rounds)), dim3((unsigned)(C / 32 * 64)), 0, st, (const uint16_t*)x, w, b, (uint16_t*)y, mean, rstd, C, P, rounds, eps); SLAK_LAUNCH_CHECK(); return SLAK_OK; } }
#define CALL(CL, G) return launch_ln_fwd_reg<CL, G>((const uint16_t*)x, w, b, (uint16_t*)y, mean, rstd, N, P, eps, st)
    SLAK_RT_DISPATCH(C, CALL)
#undef CALL
}
int launch_ln_nchw_to_nhwc_bwd_reg(const void* g, const void* x, const float* w, const float* mean, const float* rstd, void* dx, float* part, int* rows,
                                   int N, int C, int P, hipStream_t st) {
    { const int cw1 = rt_chan1_cw(C, P, N, 16), rounds = (P + 63) / 64;
      if (cw1 == 48) { hipLaunchKernelGGL(ln_nchw_to_nhwc_bwd_chan1_kernel<48>, dim3((unsigned)(N * rounds)), dim3((unsigned)(C / 48 * 64)), 0, st, (const uint16_t*)g, (const uint16_t*)x, w, mean, rstd, (uint16_t*)dx, part, C, P, rounds); SLAK_LAUNCH_CHECK(); *rows = N * rounds; return SLAK_OK; }
      if (cw1 == 32) { hipLaunchKernelGGL(ln_nchw_to_nhwc_bwd_chan1_kernel<32>, dim3((unsigned)(N * rounds)), dim3((unsigned)(C / 32 * 64)), 0, st, (const uint16_t*)g, (const uint16_t*)x, w, mean, rstd, (uint16_t*)dx, part, C, P, rounds); SLAK_LAUNCH_CHECK(); *rows = N * rounds; return SLAK_OK; } }
    { const int cw = rt_ln_chan_cw(C, P, N);
      if (cw == 48) return launch_ln_bwd_chan<48>((const uint16_t*)g, (const uint16_t*)x, w, mean, rstd, (uint16_t*)dx, part, rows, N, C, P, st);
      if (cw == 64) return launch_ln_bwd_chan<64>((const uint16_t*)g, (const uint16_t*)x, w, mean, rstd, (uint16_t*)dx, part, rows, N, C, P, st); }
#define CALL(CL, G) return launch_ln_bwd_reg<CL, G>((const uint16_t*)g, (const uint16_t*)x, w, mean, rstd, (uint16_t*)dx, part, rows, N, P, st)
    SLAK_RT_DISPATCH(C, CALL)
#undef CALL
}
// LayerNorm(channels_first) of an fp32 NCHW tensor written as the bf16 patch matrix of a 2x2 / stride-2 convolution, and its backward
int launch_ln_patch_fwd_reg(const float* x, const float* w, const float* b, void* a, float* mean, float* rstd, int N, int C, int H, int W, float eps, hipStream_t st) {
    const int P = H * W;
    if ((H | W) & 1) return SLAK_ERR_UNSUPPORTED;
#define CALL(CL, G) return (launch_ln_fwd_reg<CL, G, true, true>(x, w, b, (uint16_t*)a, mean, rstd, N, P, eps, st, W))
    SLAK_RT_DISPATCH(C, CALL)
#undef CALL
}
int launch_ln_patch_bwd_reg(const void* g, const float* x, const float* w, const float* mean, const float* rstd, float* dx, float* part, int* rows,
                            int N, int C, int H, int W, hipStream_t st) {
    const int P = H * W;
    if ((H | W) & 1) return SLAK_ERR_UNSUPPORTED;
#define CALL(CL, G) return (launch_ln_bwd_reg<CL, G, true, true>((const uint16_t*)g, x, w, mean, rstd, dx, part, rows, N, P, st, W))
    SLAK_RT_DISPATCH(C, CALL)
#undef CALL
}
int launch_scale_residual_fwd_reg(const void* sc, int sc_dtype, const void* z, const float* gamma, const float* scale, float* out, void* out16,
                                  int N, int C, int P, hipStream_t st) {
    if (sc_dtype == SLAK_F32) {
#define CALL(CL, G) return launch_sr_fwd_reg<CL, G, float>((const float*)sc, (const uint16_t*)z, gamma, scale, out, (uint16_t*)out16, N, P, st)
        // C = 384: 96 channels per lane on 4 lanes per pixel pair double the length of a wave's NCHW row segments (64 -> 128 bytes); the
        // forward residual kernel is the one of the four whose registers allow it without spilling
        { const int cw1 = rt_chan1_cw(C, P, N, 1 << 20), rounds = (P + 63) / 64;      // odd planes: one pixel per lane
          if (cw1 == 48) { const int nu = N * (C / 48) * rounds; hipLaunchKernelGGL(scale_residual_fwd_chan1_kernel<48>, dim3((unsigned)((nu + 3) / 4)), dim3(256), 0, st, (const float*)sc, (const uint16_t*)z, gamma, scale, out, (uint16_t*)out16, C, P, rounds, nu); SLAK_LAUNCH_CHECK(); return SLAK_OK; }
          if (cw1 == 32) { const int nu = N * (C / 32) * rounds; hipLaunchKernelGGL(scale_residual_fwd_chan1_kernel<32>, dim3((unsigned)((nu + 3) / 4)), dim3(256), 0, st, (const float*)sc, (const uint16_t*)z, gamma, scale, out, (uint16_t*)out16, C, P, rounds, nu); SLAK_LAUNCH_CHECK(); return SLAK_OK; } }
        if (!(P & 1) && P <= rt_chan_max_p() && rt_chan_waves()) {              // small planes: a wave per (image, channel group)
            if (C % 48 == 0) return launch_sr_fwd_chan<48>((const float*)sc, (const uint16_t*)z, gamma, scale, out, (uint16_t*)out16, N, C, P, st);
            if (C % 64 == 0) return launch_sr_fwd_chan<64>((const float*)sc, (const uint16_t*)z, gamma, scale, out, (uint16_t*)out16, N, C, P, st);
        }
        if (C == 384 && !(P & 1) && rt_wide_lanes()) { CALL(96, 4); }
        SLAK_RT_DISPATCH(C, CALL)
#undef CALL
    } else if (sc_dtype == SLAK_BF16) {
#define CALL(CL, G) return launch_sr_fwd_reg<CL, G, bf16_t>((const bf16_t*)sc, (const uint16_t*)z, gamma, scale, out, (uint16_t*)out16, N, P, st)
        SLAK_RT_DISPATCH(C, CALL)
#undef CALL
    }
    return SLAK_ERR_UNSUPPORTED;
}
int launch_scale_residual_bwd_reg(const float* dout, const void* dout16, float* dsum, const void* z, const float* gamma, const float* scale, void* dz,
                                  float* part, int* rows, int N, int C, int P, hipStream_t st) {
    { const int cw1 = rt_chan1_cw(C, P, N, 1 << 20), rounds = (P + 63) / 64;          // odd planes: one pixel per lane
      if (cw1 == 48) { const int nu = N * (C / 48) * rounds; hipLaunchKernelGGL(scale_residual_bwd_chan1_kernel<48>, dim3((unsigned)((nu + 3) / 4)), dim3(256), 0, st, dout, (const uint16_t*)dout16, dsum, (const uint16_t*)z, gamma, scale, (uint16_t*)dz, part, C, P, rounds, nu); SLAK_LAUNCH_CHECK(); *rows = N * rounds; return SLAK_OK; }
      if (cw1 == 32) { const int nu = N * (C / 32) * rounds; hipLaunchKernelGGL(scale_residual_bwd_chan1_kernel<32>, dim3((unsigned)((nu + 3) / 4)), dim3(256), 0, st, dout, (const uint16_t*)dout16, dsum, (const uint16_t*)z, gamma, scale, (uint16_t*)dz, part, C, P, rounds, nu); SLAK_LAUNCH_CHECK(); *rows = N * rounds; return SLAK_OK; } }
    if (!(P & 1) && P <= rt_chan_max_p() && (long long)N * ((P / 2 + 63) / 64) <= 8192 && rt_chan_waves()) {     // small planes: a wave per (image, channel group); rows <= the workspace's 8192
        if (C % 24 == 0) return launch_sr_bwd_chan<24>(dout, (const uint16_t*)dout16, dsum, (const uint16_t*)z, gamma, scale, (uint16_t*)dz, part, rows, N, C, P, st);
        if (C % 32 == 0) return launch_sr_bwd_chan<32>(dout, (const uint16_t*)dout16, dsum, (const uint16_t*)z, gamma, scale, (uint16_t*)dz, part, rows, N, C, P, st);
    }
#define CALL(CL, G) return launch_sr_bwd_reg<CL, G>(dout, (const uint16_t*)dout16, dsum, (const uint16_t*)z, gamma, scale, (uint16_t*)dz, part, rows, N, P, st)
    SLAK_RT_DISPATCH(C, CALL)
#undef CALL
}

}  // namespace slak
