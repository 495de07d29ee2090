// slak_amd/csrc/gelu_grad.h -- nn.GELU()'s backward on stored bf16 tensors (models/SLaK.py:159 under autocast): dy1 = dact * gelu'(y1), evaluated
// by table.  Shared by the stand-alone kernel (block_tail.hip: gelu_bwd_bias_kernel) and by the stage-1 data-gradient GEMM that applies it in
// its epilogue (linear_skinny.hip: linear_nt_k96_gbwd_kernel): the SAME evaluation, so the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slak {

typedef __attribute__((ext_vector_type(2))) float gg_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 gg_bf16x2;
__device__ __forceinline__ unsigned gg_pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(gg_f32x2{a, b}, gg_bf16x2)); }
__device__ __forceinline__ float gg_bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// gelu'(x) = Phi(x) + x phi(x) of a bf16 VALUE by table (the pre-activation y1 is a stored bf16 tensor): 2^-18 <= |x| < 16 is
// 22 exponents x 128 mantissas x 2 signs fp32 entries (22 KB of LDS, correctly rounded on the host in double precision); below, the
// linear term 0.5 + x phi(0) is exact to fp32; above, 1 (x > 0) or 0.  One LDS gather + a handful of integer instructions instead of
// exp + rcp + a degree-5 polynomial: the evaluation was 0.4 ms of the kernel's 1.74 ms per SLaK-T step.
constexpr unsigned GD_LO = 109u << 7, GD_N = 22u << 7;
constexpr int GD_BYTES = 2 * (int)GD_N * 4;
__device__ __forceinline__ float gelu_grad_lut(const float* __restrict__ T, unsigned b) {       // b: bf16 bits
    const unsigned mag = b & 0x7fffu, neg = b >> 15;
    const unsigned idx = mag - GD_LO;
    const bool in = idx < GD_N;
    const float t = T[(in ? idx : 0u) + neg * GD_N];
    const float x = __uint_as_float(b << 16);
    const float lo = 0.5f + 0.79788456080286536f * x;              // |x| < 2^-18 (also +-0, subnormals)
    const float hi = mag > 0x7f80u ? x : (neg ? 0.0f : 1.0f);      // |x| >= 16, +-inf; NaN propagates
    return in ? t : (mag < GD_LO ? lo : hi);
}
__device__ __forceinline__ void gelu_bwd8(const float* __restrict__ T, const uint4& gv, const uint4& yv, uint4& ov, float (&acc)[8]) {
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
    unsigned ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float g = gg_bf2f((uint16_t)(h ? gw[k] >> 16 : gw[k] & 0xffff));
            o[h] = g * gelu_grad_lut(T, h ? yw[k] >> 16 : yw[k] & 0xffffu);
        }
        ow[k] = gg_pack2(o[0], o[1]);
        // the bias gradient sums the ROUNDED values (what dy1.sum(0) over the stored tensor gives)
        acc[2 * k] += gg_bf2f((uint16_t)(ow[k] & 0xffff)); acc[2 * k + 1] += gg_bf2f((uint16_t)(ow[k] >> 16));
    }
    ov = uint4{ow[0], ow[1], ow[2], ow[3]};
}
// The same on the table's range only (2^-18 <= |x| < 16: all but ~3e-6 of N(0,1) pre-activations): nine integer / packed-fp32
// instructions per element instead of twenty.  Returns false (wave-uniform callers then redo the eight elements with gelu_bwd8) when an
// element lies outside the table; `ov` / `acc` are only valid on true.
__device__ __forceinline__ bool gelu_bwd8_fast(const float* __restrict__ T, const uint4& gv, const uint4& yv, uint4& ov, float (&acc)[8]) {
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
    unsigned ow[4];
    bool ok = true;
    float t[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i0 = (yw[k] & 0x7fffu) - GD_LO, i1 = ((yw[k] >> 16) & 0x7fffu) - GD_LO;
        ok = ok && i0 < GD_N && i1 < GD_N;
        const unsigned a0 = (i0 < GD_N ? i0 : 0u) + ((yw[k] >> 15) & 1u) * GD_N, a1 = (i1 < GD_N ? i1 : 0u) + (yw[k] >> 31) * GD_N;
        t[2 * k] = T[a0]; t[2 * k + 1] = T[a1];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float o0 = __uint_as_float(gw[k] << 16) * t[2 * k], o1 = __uint_as_float(gw[k] & 0xffff0000u) * t[2 * k + 1];
        ow[k] = gg_pack2(o0, o1);
        acc[2 * k] += __uint_as_float(ow[k] << 16); acc[2 * k + 1] += __uint_as_float(ow[k] & 0xffff0000u);       // the ROUNDED values
    }
    ov = uint4{ow[0], ow[1], ow[2], ow[3]};
    return ok;
}

// gelu_bwd8_fast in two halves, so that a caller can put the gathers of several chunks in flight before it tests and uses any of them:
// the eight table values of yv (false: an element outside the table; the values are then meaningless) ...
__device__ __forceinline__ bool gelu_grad_gather8(const float* __restrict__ T, const uint4& yv, float (&t)[8]) {
    const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w};
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i0 = (yw[k] & 0x7fffu) - GD_LO, i1 = ((yw[k] >> 16) & 0x7fffu) - GD_LO;
        ok = ok && i0 < GD_N && i1 < GD_N;
        const unsigned a0 = (i0 < GD_N ? i0 : 0u) + ((yw[k] >> 15) & 1u) * GD_N, a1 = (i1 < GD_N ? i1 : 0u) + (yw[k] >> 31) * GD_N;
        t[2 * k] = T[a0]; t[2 * k + 1] = T[a1];
    }
    return ok;
}
// ... and the products, their rounding and the column sums of the ROUNDED values (the same operations in the same order as gelu_bwd8_fast)
__device__ __forceinline__ void gelu_bwd8_apply(const uint4& gv, const float (&t)[8], uint4& ov, float (&acc)[8]) {
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w};
    unsigned ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float o0 = __uint_as_float(gw[k] << 16) * t[2 * k], o1 = __uint_as_float(gw[k] & 0xffff0000u) * t[2 * k + 1];
        ow[k] = gg_pack2(o0, o1);
        acc[2 * k] += __uint_as_float(ow[k] << 16); acc[2 * k + 1] += __uint_as_float(ow[k] & 0xffff0000u);
    }
    ov = uint4{ow[0], ow[1], ow[2], ow[3]};
}

// the table of gelu_grad_lut in device memory (one copy per device, built on first use; block_tail.hip)
const float* gelu_grad_table_device();
// column sums of part[ntiles][width] -> out[width] in one launch, fixed order (block_tail.hip: block_tail_reduce1)
int tail_reduce_columns(const float* part, float* out, int ntiles, int width, hipStream_t st);
// the same with the columns [0, split) going to out0 and [split, width) to out1
int tail_reduce_split(const float* part, float* out0, float* out1, int split, int ntiles, int width, hipStream_t st);

}  // namespace slak
