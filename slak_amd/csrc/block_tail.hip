// slak_amd/csrc/block_tail.hip -- the layout / normalisation / residual glue around the two pointwise GEMMs of a SLaK block
// (reference models/SLaK.py:153-166, :253-261):
//
//     x = large_kernel(x)                       (N,C,H,W)
//     x = x.permute(0,2,3,1); x = LayerNorm_C(x)            -> ln_nchw_to_nhwc   (one pass: read NCHW, write NHWC)
//     x = pwconv2(gelu(pwconv1(x)))                            (hipBLASLt GEMMs, untouched)
//     x = gamma * x; x = x.permute(0,3,1,2); x = shortcut + drop_path(x)   -> scale_residual (one pass: read NHWC, write NCHW)
//
// In PyTorch these lines are ~14 kernels per block and direction (permute copies, dtype casts, layer_norm, broadcast
// multiplies, adds, reductions for dgamma): 25 ms of a 61 ms SLaK-T step on MI355X.  Here each arrow is ONE HBM-bound kernel
// per direction: a workgroup owns (image n, TP consecutive pixels, all C channels), stages the tile in LDS and transposes
// through it, so that both the NCHW side (lanes along pixels) and the NHWC side (lanes along channels) are coalesced.
// LDS pitches are odd in dwords, so row and column walks are both conflict-free.  Per-channel gradient sums (dweight, dbias,
// dgamma) are written as per-tile partials and added in a fixed order by block_tail_reduce1 (deterministic, no atomics).
// Activations bf16 (autocast), statistics / residual stream / parameters fp32.
#include "slak_common.h"
#include "gelu_grad.h"
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace slak {

constexpr int BT_THREADS = 256;

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __attribute__((ext_vector_type(2))) float bt_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bt_bf16x2;
__device__ __forceinline__ unsigned bt_pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(bt_f32x2{a, b}, bt_bf16x2)); }
__device__ __forceinline__ uint16_t bt_f2bf(float a) { return (uint16_t)(bt_pack2(a, 0.f) & 0xffffu); }

struct TailDims { int N, C, P, TP, tiles_per_image, ntiles; };

__device__ __forceinline__ float bt_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the 16 lanes of a DPP row, valid in the row's LAST lane (i & 15 == 15): four v_add_f32 with a row_shr operand, no LDS
// crossbar (the wave-wide __shfl_xor reduction is a ds_bpermute + address arithmetic per step)
__device__ __forceinline__ float bt_row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));   // row_shr:8
    return v;
}

// stage x[n, :, p0:p0+TP] (NCHW, bf16) into xs[C][TP+2]; pixels beyond P are zero
__device__ __forceinline__ void load_nchw_tile(uint16_t* xs, const uint16_t* __restrict__ xn, int C, int P, int p0, int TP, int tid) {
    const int pitch = TP + 2;
    if ((P & 3) == 0) {                                   // 8-byte chunks of 4 pixels (rows are 8-byte aligned)
        const int cpr = TP / 4;
        for (int idx = tid; idx < C * cpr; idx += BT_THREADS) {
            const int c = idx / cpr, q = idx - c * cpr, p = p0 + q * 4;
            unsigned lo = 0u, hi = 0u;
            if (p < P) { const unsigned* src = (const unsigned*)(xn + (size_t)c * P + p); lo = src[0]; hi = src[1]; }
            unsigned* dst = (unsigned*)(xs + c * pitch + q * 4);
            dst[0] = lo; dst[1] = hi;
        }
    } else {
        for (int idx = tid; idx < C * TP; idx += BT_THREADS) {
            const int c = idx / TP, q = idx - c * TP, p = p0 + q;
            xs[c * pitch + q] = p < P ? xn[(size_t)c * P + p] : (uint16_t)0;
        }
    }
}
// write xs[C][TP+2] (bf16) back to an NCHW tensor
__device__ __forceinline__ void store_nchw_tile(const uint16_t* xs, uint16_t* __restrict__ xn, int C, int P, int p0, int TP, int tid) {
    const int pitch = TP + 2;
    if ((P & 3) == 0) {
        const int cpr = TP / 4;
        for (int idx = tid; idx < C * cpr; idx += BT_THREADS) {
            const int c = idx / cpr, q = idx - c * cpr, p = p0 + q * 4;
            if (p < P) {
                const unsigned* src = (const unsigned*)(xs + c * pitch + q * 4);
                unsigned* dst = (unsigned*)(xn + (size_t)c * P + p);
                dst[0] = src[0]; dst[1] = src[1];
            }
        }
    } else {
        for (int idx = tid; idx < C * TP; idx += BT_THREADS) {
            const int c = idx / TP, q = idx - c * TP, p = p0 + q;
            if (p < P) xn[(size_t)c * P + p] = xs[c * pitch + q];
        }
    }
}
// stage z[n, p0:p0+TP, :] (NHWC, bf16, C % 2 == 0) into zs[TP][C+2]; pixels beyond P are zero
__device__ __forceinline__ void load_nhwc_tile(uint16_t* zs, const uint16_t* __restrict__ zn, int C, int P, int p0, int TP, int tid) {
    const int pitch = C + 2, cp = C / 2;
    for (int idx = tid; idx < TP * cp; idx += BT_THREADS) {
        const int q = idx / cp, k = idx - q * cp, p = p0 + q;
        unsigned v = 0u;
        if (p < P) v = ((const unsigned*)(zn + (size_t)p * C))[k];
        ((unsigned*)(zs + q * pitch))[k] = v;
    }
}

// ===== variant A (small C: stages 1-2): one workgroup per tile, thread <-> (pixel, channel group); both tiles staged in LDS =====
// y[n,p,:] = LN_C(x[n,:,p]) * w + b   (two-pass statistics in fp32 on the staged tile), y bf16 NHWC; saves mean, rstd
__global__ __launch_bounds__(BT_THREADS) void ln_nchw_to_nhwc_fwd_pix_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                                       const float* __restrict__ b, uint16_t* __restrict__ y,
                                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                                       const TailDims d, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = d.C, P = d.P, TP = d.TP, pitch = TP + 2;
    uint16_t* xs = (uint16_t*)smem;                                        // [C][TP+2]
    float* red = (float*)(smem + (((size_t)C * pitch * 2 + 15) & ~(size_t)15));   // [NG][TP]
    float* st = red + (BT_THREADS / TP) * TP;                              // [2][TP]: mean, rstd
    const int tid = threadIdx.x;
    const int n = blockIdx.x / d.tiles_per_image, p0 = (blockIdx.x % d.tiles_per_image) * TP;
    load_nchw_tile(xs, x + (size_t)n * C * P, C, P, p0, TP, tid);
    __syncthreads();
    const int q = tid % TP, g = tid / TP, NG = BT_THREADS / TP;
    float s = 0.f;
    for (int c = g; c < C; c += NG) s += bf2f(xs[c * pitch + q]);
    red[g * TP + q] = s;
    __syncthreads();
    if (g == 0) { float t = 0.f; for (int k = 0; k < NG; ++k) t += red[k * TP + q]; st[q] = t / (float)C; }
    __syncthreads();
    const float mu = st[q];
    float ss = 0.f;
    for (int c = g; c < C; c += NG) { const float v = bf2f(xs[c * pitch + q]) - mu; ss += v * v; }
    __syncthreads();
    red[g * TP + q] = ss;
    __syncthreads();
    if (g == 0) {
        float t = 0.f; for (int k = 0; k < NG; ++k) t += red[k * TP + q];
        const float r = 1.0f / sqrtf(t / (float)C + eps);
        st[TP + q] = r;
        if (p0 + q < P) { mean[(size_t)n * P + p0 + q] = mu; rstd[(size_t)n * P + p0 + q] = r; }
    }
    __syncthreads();
    // write NHWC: lanes along channel pairs
    const int cp = C / 2;
    uint16_t* yn = y + ((size_t)n * P + p0) * C;
    for (int idx = tid; idx < TP * cp; idx += BT_THREADS) {
        const int qq = idx / cp, k = idx - qq * cp;
        if (p0 + qq < P) {
            const float m = st[qq], r = st[TP + qq];
            const float v0 = (bf2f(xs[(2 * k) * pitch + qq]) - m) * r * w[2 * k] + b[2 * k];
            const float v1 = (bf2f(xs[(2 * k + 1) * pitch + qq]) - m) * r * w[2 * k + 1] + b[2 * k + 1];
            ((unsigned*)(yn + (size_t)qq * C))[k] = bt_pack2(v0, v1);
        }
    }
}

// dx[n,:,p] (bf16 NCHW) from g[n,p,:] (bf16 NHWC); per-tile partial sums part[tile][0][c] = sum_p g*xhat, part[tile][1][c] = sum_p g
__global__ __launch_bounds__(BT_THREADS) void ln_nchw_to_nhwc_bwd_pix_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ x,
                                                                       const float* __restrict__ w, const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd, uint16_t* __restrict__ dx,
                                                                       float* __restrict__ part, const TailDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = d.C, P = d.P, TP = d.TP, xp = TP + 2, gp = C + 2;
    uint16_t* xs = (uint16_t*)smem;                                                  // [C][TP+2]
    uint16_t* gs = (uint16_t*)(smem + (((size_t)C * xp * 2 + 15) & ~(size_t)15));  // [TP][C+2]
    float* red = (float*)((unsigned char*)gs + (((size_t)TP * gp * 2 + 15) & ~(size_t)15));   // [2][NG][TP]
    float* st = red + 2 * (BT_THREADS / TP) * TP;                                    // [4][TP]: mean, rstd, m1, m2
    const int tid = threadIdx.x;
    const int n = blockIdx.x / d.tiles_per_image, p0 = (blockIdx.x % d.tiles_per_image) * TP;
    load_nchw_tile(xs, x + (size_t)n * C * P, C, P, p0, TP, tid);
    load_nhwc_tile(gs, g + (size_t)n * P * C, C, P, p0, TP, tid);
    if (tid < TP) {
        const bool ok = p0 + tid < P;
        st[tid] = ok ? mean[(size_t)n * P + p0 + tid] : 0.f;
        st[TP + tid] = ok ? rstd[(size_t)n * P + p0 + tid] : 0.f;
    }
    __syncthreads();
    const int q = tid % TP, grp = tid / TP, NG = BT_THREADS / TP;
    {
        const float mu = st[q], r = st[TP + q];
        float s1 = 0.f, s2 = 0.f;
        for (int c = grp; c < C; c += NG) {
            const float gw = bf2f(gs[q * gp + c]) * w[c];
            s1 += gw; s2 += gw * ((bf2f(xs[c * xp + q]) - mu) * r);
        }
        red[grp * TP + q] = s1; red[(NG + grp) * TP + q] = s2;
    }
    __syncthreads();
    if (grp == 0) {
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < NG; ++k) { t1 += red[k * TP + q]; t2 += red[(NG + k) * TP + q]; }
        st[2 * TP + q] = t1 / (float)C; st[3 * TP + q] = t2 / (float)C;
    }
    __syncthreads();
    // dx: lanes along pixels (4 pixels per thread when rows are 8-byte aligned)
    uint16_t* dxn = dx + (size_t)n * C * P;
    float* pt = part + (size_t)blockIdx.x * 2 * C;
    if ((P & 3) == 0 && TP == 64) {
        // thread <-> (4 pixels qq..qq+3, channels c0, c0+16, ...): its pixel statistics are loop constants, and the per-channel
        // partials (sum_p g*xhat, sum_p g) are folded from the values it has in hand, reduced over the 16 lanes that share a channel
        const int qq = (tid & 15) * 4, c0 = tid >> 4;
        float mu[4], r[4], m1[4], m2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu[e] = st[qq + e]; r[e] = st[TP + qq + e]; m1[e] = st[2 * TP + qq + e]; m2[e] = st[3 * TP + qq + e]; }
        const bool live = p0 + qq < P;
        for (int c = c0; c < C; c += BT_THREADS / 16) {
            const float wc = w[c];
            float o[4], a = 0.f, b = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gv = bf2f(gs[(qq + e) * gp + c]);                    // padding pixels hold g == 0
                const float xh = (bf2f(xs[c * xp + qq + e]) - mu[e]) * r[e];
                o[e] = r[e] * (gv * wc - m1[e] - xh * m2[e]);
                a += gv * xh; b += gv;
            }
            if (live) {
                unsigned* dst = (unsigned*)(dxn + (size_t)c * P + p0 + qq);
                dst[0] = bt_pack2(o[0], o[1]); dst[1] = bt_pack2(o[2], o[3]);
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 16); b += __shfl_xor(b, m, 16); }
            if ((tid & 15) == 0) { pt[c] = a; pt[C + c] = b; }
        }
        return;
    }
    if ((P & 3) == 0) {
        const int cpr = TP / 4;
        for (int idx = tid; idx < C * cpr; idx += BT_THREADS) {
            const int c = idx / cpr, qq = (idx - c * cpr) * 4;
            if (p0 + qq < P) {
                const float wc = w[c];
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float mu = st[qq + e], r = st[TP + qq + e];
                    const float xh = (bf2f(xs[c * xp + qq + e]) - mu) * r;
                    o[e] = r * (bf2f(gs[(qq + e) * gp + c]) * wc - st[2 * TP + qq + e] - xh * st[3 * TP + qq + e]);
                }
                unsigned* dst = (unsigned*)(dxn + (size_t)c * P + p0 + qq);
                dst[0] = bt_pack2(o[0], o[1]); dst[1] = bt_pack2(o[2], o[3]);
            }
        }
    } else {
        for (int idx = tid; idx < C * TP; idx += BT_THREADS) {
            const int c = idx / TP, qq = idx - c * TP;
            if (p0 + qq < P) {
                const float mu = st[qq], r = st[TP + qq];
                const float xh = (bf2f(xs[c * xp + qq]) - mu) * r;
                dxn[(size_t)c * P + p0 + qq] = bt_f2bf(r * (bf2f(gs[qq * gp + c]) * w[c] - st[2 * TP + qq] - xh * st[3 * TP + qq]));
            }
        }
    }
    // per-channel partials over the tile's pixels (padding pixels have g == 0)
    for (int c = tid; c < C; c += BT_THREADS) {
        float a = 0.f, bsum = 0.f;
        for (int qq = 0; qq < TP; ++qq) {
            const float gv = bf2f(gs[qq * gp + c]);
            a += gv * ((bf2f(xs[c * xp + qq]) - st[qq]) * st[TP + qq]);
            bsum += gv;
        }
        pt[c] = a; pt[C + c] = bsum;
    }
}

// out[n,c,p] (fp32 NCHW) = shortcut[n,c,p] + scale[n] * gamma[c] * z[n,p,c] (bf16 NHWC)
template <typename Tsc>
__global__ __launch_bounds__(BT_THREADS) void scale_residual_fwd_pix_kernel(const Tsc* __restrict__ sc, const uint16_t* __restrict__ z,
                                                                      const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                      float* __restrict__ out, uint16_t* __restrict__ out16, const TailDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = d.C, P = d.P, TP = d.TP, zp = C + 2;
    uint16_t* zs = (uint16_t*)smem;
    const int tid = threadIdx.x;
    const int n = blockIdx.x / d.tiles_per_image, p0 = (blockIdx.x % d.tiles_per_image) * TP;
    load_nhwc_tile(zs, z + (size_t)n * P * C, C, P, p0, TP, tid);
    __syncthreads();
    const float sn = scale ? scale[n] : 1.0f;
    const Tsc* scn = sc + (size_t)n * C * P;
    float* on = out + (size_t)n * C * P;
    if ((P & 3) == 0) {
        const int cpr = TP / 4;
        for (int idx = tid; idx < C * cpr; idx += BT_THREADS) {
            const int c = idx / cpr, qq = (idx - c * cpr) * 4;
            if (p0 + qq < P) {
                const float gm = gamma[c] * sn;
                const size_t off = (size_t)c * P + p0 + qq;
                float4 o;
                o.x = to_f32(scn[off + 0]) + gm * bf2f(zs[(qq + 0) * zp + c]);
                o.y = to_f32(scn[off + 1]) + gm * bf2f(zs[(qq + 1) * zp + c]);
                o.z = to_f32(scn[off + 2]) + gm * bf2f(zs[(qq + 2) * zp + c]);
                o.w = to_f32(scn[off + 3]) + gm * bf2f(zs[(qq + 3) * zp + c]);
                *(float4*)(on + off) = o;
                if (out16) *(uint2*)(out16 + (size_t)n * C * P + off) = uint2{bt_pack2(o.x, o.y), bt_pack2(o.z, o.w)};
            }
        }
    } else {
        for (int idx = tid; idx < C * TP; idx += BT_THREADS) {
            const int c = idx / TP, qq = idx - c * TP;
            if (p0 + qq < P) {
                const size_t off = (size_t)c * P + p0 + qq;
                const float ov = to_f32(scn[off]) + gamma[c] * sn * bf2f(zs[qq * zp + c]);
                on[off] = ov;
                if (out16) out16[(size_t)n * C * P + off] = bt_f2bf(ov);
            }
        }
    }
}

// dz[n,p,c] (bf16 NHWC) = scale[n] * gamma[c] * dout[n,c,p] (fp32 NCHW); part[tile][c] = scale[n] * sum_p dout * z
__global__ __launch_bounds__(BT_THREADS) void scale_residual_bwd_pix_kernel(const float* __restrict__ dout, const uint16_t* __restrict__ dout16, float* __restrict__ dsum,
                                                                      const uint16_t* __restrict__ z,
                                                                      const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                      uint16_t* __restrict__ dz, float* __restrict__ part, const TailDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = d.C, P = d.P, TP = d.TP, dp = TP + 1, zp = C + 2;
    float* ds = (float*)smem;                                                        // [C][TP+1] fp32
    uint16_t* zs = (uint16_t*)(smem + (size_t)C * dp * 4);                          // [TP][C+2]
    const int tid = threadIdx.x;
    const int n = blockIdx.x / d.tiles_per_image, p0 = (blockIdx.x % d.tiles_per_image) * TP;
    const float* dn = dout + (size_t)n * C * P;
    if ((P & 3) == 0) {
        const int cpr = TP / 4;
        for (int idx = tid; idx < C * cpr; idx += BT_THREADS) {
            const int c = idx / cpr, qq = (idx - c * cpr) * 4;
            float4 v = float4{0.f, 0.f, 0.f, 0.f};
            if (p0 + qq < P) {
                const size_t off = (size_t)c * P + p0 + qq;
                v = *(const float4*)(dn + off);
                if (dout16) {                                          // second gradient stream (bf16 copy of `out`): add, hand the sum on
                    const uint2 h = *(const uint2*)(dout16 + (size_t)n * C * P + off);
                    v.x += bf2f((uint16_t)(h.x & 0xffff)); v.y += bf2f((uint16_t)(h.x >> 16));
                    v.z += bf2f((uint16_t)(h.y & 0xffff)); v.w += bf2f((uint16_t)(h.y >> 16));
                    *(float4*)(dsum + (size_t)n * C * P + off) = v;
                }
            }
            float* dst = ds + c * dp + qq;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    } else {
        for (int idx = tid; idx < C * TP; idx += BT_THREADS) {
            const int c = idx / TP, qq = idx - c * TP;
            float v = 0.f;
            if (p0 + qq < P) {
                const size_t off = (size_t)c * P + p0 + qq;
                v = dn[off];
                if (dout16) { v += bf2f(dout16[(size_t)n * C * P + off]); dsum[(size_t)n * C * P + off] = v; }
            }
            ds[c * dp + qq] = v;
        }
    }
    load_nhwc_tile(zs, z + (size_t)n * P * C, C, P, p0, TP, tid);
    __syncthreads();
    const float sn = scale ? scale[n] : 1.0f;
    const int cp = C / 2;
    uint16_t* dzn = dz + ((size_t)n * P + p0) * C;
    for (int idx = tid; idx < TP * cp; idx += BT_THREADS) {
        const int qq = idx / cp, k = idx - qq * cp;
        if (p0 + qq < P)
            ((unsigned*)(dzn + (size_t)qq * C))[k] = bt_pack2(sn * gamma[2 * k] * ds[(2 * k) * dp + qq], sn * gamma[2 * k + 1] * ds[(2 * k + 1) * dp + qq]);
    }
    float* pt = part + (size_t)blockIdx.x * 2 * C;               // [dgamma partial | column sums of dz (= the next Linear's bias gradient)]
    for (int c = tid; c < C; c += BT_THREADS) {
        float a = 0.f, b = 0.f;
        for (int qq = 0; qq < TP; ++qq) { a += ds[c * dp + qq] * bf2f(zs[qq * zp + c]); b += ds[c * dp + qq]; }
        pt[c] = a * sn; pt[C + c] = b * sn * gamma[c];
    }
}

// ===== variant B (large C: stages 3-4): persistent workgroups, wavefront <-> pixel, lane <-> channel pair; one tile staged =====
// All four kernels are persistent: gridDim.x workgroups walk the (image, pixel-tile) list.  Inside a tile, wavefront w owns
// pixels q = w, w+4, ...; on the NHWC side a lane owns the channel pair (2*lane + 128*k, +1).
// ---------------------------------------------------------------------------------------------------------------
// y[n,p,:] = LN_C(x[n,:,p]) * w + b   (two-pass statistics in fp32 on the staged tile), y bf16 NHWC; saves mean, rstd
__global__ __launch_bounds__(BT_THREADS) void ln_nchw_to_nhwc_fwd_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                                       const float* __restrict__ b, uint16_t* __restrict__ y,
                                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                                       const TailDims d, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = d.C, P = d.P, TP = d.TP, pitch = TP + 2, cp = C / 2;
    uint16_t* xs = (uint16_t*)smem;                                        // [C][TP+2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int tile = blockIdx.x; tile < d.ntiles; tile += gridDim.x) {
        const int n = tile / d.tiles_per_image, p0 = (tile - n * d.tiles_per_image) * TP;
        load_nchw_tile(xs, x + (size_t)n * C * P, C, P, p0, TP, tid);
        __syncthreads();
        uint16_t* yn = y + ((size_t)n * P + p0) * C;
        for (int q = wave; q < TP && p0 + q < P; q += BT_THREADS / 64) {
            float s = 0.f;
            for (int k = lane; k < cp; k += 64) s += bf2f(xs[(2 * k) * pitch + q]) + bf2f(xs[(2 * k + 1) * pitch + q]);
            const float mu = bt_wave_sum(s) / (float)C;
            float ss = 0.f;
            for (int k = lane; k < cp; k += 64) {
                const float v0 = bf2f(xs[(2 * k) * pitch + q]) - mu, v1 = bf2f(xs[(2 * k + 1) * pitch + q]) - mu;
                ss += v0 * v0 + v1 * v1;
            }
            const float r = 1.0f / sqrtf(bt_wave_sum(ss) / (float)C + eps);
            if (lane == 0) { mean[(size_t)n * P + p0 + q] = mu; rstd[(size_t)n * P + p0 + q] = r; }
            for (int k = lane; k < cp; k += 64) {
                const float v0 = (bf2f(xs[(2 * k) * pitch + q]) - mu) * r * w[2 * k] + b[2 * k];
                const float v1 = (bf2f(xs[(2 * k + 1) * pitch + q]) - mu) * r * w[2 * k + 1] + b[2 * k + 1];
                ((unsigned*)(yn + (size_t)q * C))[k] = bt_pack2(v0, v1);
            }
        }
        __syncthreads();
    }
}

// dx[n,:,p] (bf16 NCHW) from g[n,p,:] (bf16 NHWC, read straight from global on its coalesced side; second read hits L2);
// x is staged, dx overwrites it in LDS and leaves on the NCHW side.  part[wg][0][c] = sum g*xhat, part[wg][1][c] = sum g over
// every pixel the workgroup processed.
__global__ __launch_bounds__(BT_THREADS) void ln_nchw_to_nhwc_bwd_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ x,
                                                                       const float* __restrict__ w, const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd, uint16_t* __restrict__ dx,
                                                                       float* __restrict__ part, const TailDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = d.C, P = d.P, TP = d.TP, xp = TP + 2, cp = C / 2;
    uint16_t* xs = (uint16_t*)smem;                                                  // [C][TP+2]
    float* red = (float*)(smem + (((size_t)C * xp * 2 + 15) & ~(size_t)15));       // [4 waves][2][C] at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int KMAX = 8;                                                          // channel pairs per lane: C <= 1024
    float aw[2 * KMAX], ab[2 * KMAX];
#pragma unroll
    for (int i = 0; i < 2 * KMAX; ++i) { aw[i] = 0.f; ab[i] = 0.f; }
    for (int tile = blockIdx.x; tile < d.ntiles; tile += gridDim.x) {
        const int n = tile / d.tiles_per_image, p0 = (tile - n * d.tiles_per_image) * TP;
        load_nchw_tile(xs, x + (size_t)n * C * P, C, P, p0, TP, tid);
        __syncthreads();
        const uint16_t* gn = g + ((size_t)n * P + p0) * C;
        for (int q = wave; q < TP && p0 + q < P; q += BT_THREADS / 64) {
            const float mu = mean[(size_t)n * P + p0 + q], r = rstd[(size_t)n * P + p0 + q];
            const unsigned* gq = (const unsigned*)(gn + (size_t)q * C);
            float s1 = 0.f, s2 = 0.f;
            for (int k = lane; k < cp; k += 64) {
                const unsigned gv = gq[k];
                const float g0 = bf2f((uint16_t)(gv & 0xffff)) * w[2 * k], g1 = bf2f((uint16_t)(gv >> 16)) * w[2 * k + 1];
                s1 += g0 + g1;
                s2 += g0 * ((bf2f(xs[(2 * k) * xp + q]) - mu) * r) + g1 * ((bf2f(xs[(2 * k + 1) * xp + q]) - mu) * r);
            }
            const float m1 = bt_wave_sum(s1) / (float)C, m2 = bt_wave_sum(s2) / (float)C;
#pragma unroll
            for (int kk = 0; kk < KMAX; ++kk) {
                const int k = lane + 64 * kk;
                if (k < cp) {
                    const unsigned gv = gq[k];
                    const float gr0 = bf2f((uint16_t)(gv & 0xffff)), gr1 = bf2f((uint16_t)(gv >> 16));
                    const float xh0 = (bf2f(xs[(2 * k) * xp + q]) - mu) * r, xh1 = (bf2f(xs[(2 * k + 1) * xp + q]) - mu) * r;
                    aw[2 * kk] += gr0 * xh0; aw[2 * kk + 1] += gr1 * xh1; ab[2 * kk] += gr0; ab[2 * kk + 1] += gr1;
                    const unsigned pk = bt_pack2(r * (gr0 * w[2 * k] - m1 - xh0 * m2), r * (gr1 * w[2 * k + 1] - m1 - xh1 * m2));
                    xs[(2 * k) * xp + q] = (uint16_t)(pk & 0xffffu);
                    xs[(2 * k + 1) * xp + q] = (uint16_t)(pk >> 16);
                }
            }
        }
        __syncthreads();
        store_nchw_tile(xs, dx + (size_t)n * C * P, C, P, p0, TP, tid);
        __syncthreads();
    }
    // per-channel partials of this workgroup: the four wavefronts' sums added in wave order
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) {
        const int k = lane + 64 * kk;
        if (k < cp) {
            red[(wave * 2 + 0) * C + 2 * k] = aw[2 * kk]; red[(wave * 2 + 0) * C + 2 * k + 1] = aw[2 * kk + 1];
            red[(wave * 2 + 1) * C + 2 * k] = ab[2 * kk]; red[(wave * 2 + 1) * C + 2 * k + 1] = ab[2 * kk + 1];
        }
    }
    __syncthreads();
    float* pt = part + (size_t)blockIdx.x * 2 * C;
    for (int i = tid; i < 2 * C; i += BT_THREADS) {
        const int which = i / C, c = i - which * C;
        pt[i] = ((red[(0 * 2 + which) * C + c] + red[(1 * 2 + which) * C + c]) + red[(2 * 2 + which) * C + c]) + red[(3 * 2 + which) * C + c];
    }
}

// out[n,c,p] (fp32 NCHW) = shortcut[n,c,p] + scale[n] * gamma[c] * z[n,p,c] (bf16 NHWC)
template <typename Tsc>
__global__ __launch_bounds__(BT_THREADS) void scale_residual_fwd_kernel(const Tsc* __restrict__ sc, const uint16_t* __restrict__ z,
                                                                      const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                      float* __restrict__ out, uint16_t* __restrict__ out16, const TailDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = d.C, P = d.P, TP = d.TP, zp = C + 2;
    uint16_t* zs = (uint16_t*)smem;
    const int tid = threadIdx.x;
    for (int tile = blockIdx.x; tile < d.ntiles; tile += gridDim.x) {
        const int n = tile / d.tiles_per_image, p0 = (tile - n * d.tiles_per_image) * TP;
        load_nhwc_tile(zs, z + (size_t)n * P * C, C, P, p0, TP, tid);
        __syncthreads();
        const float sn = scale ? scale[n] : 1.0f;
        const Tsc* scn = sc + (size_t)n * C * P;
        float* on = out + (size_t)n * C * P;
        if ((P & 3) == 0) {
            const int cpr = TP / 4;
            for (int idx = tid; idx < C * cpr; idx += BT_THREADS) {
                const int c = idx / cpr, qq = (idx - c * cpr) * 4;
                if (p0 + qq < P) {
                    const float gm = gamma[c] * sn;
                    const size_t off = (size_t)c * P + p0 + qq;
                    float4 o;
                    o.x = to_f32(scn[off + 0]) + gm * bf2f(zs[(qq + 0) * zp + c]);
                    o.y = to_f32(scn[off + 1]) + gm * bf2f(zs[(qq + 1) * zp + c]);
                    o.z = to_f32(scn[off + 2]) + gm * bf2f(zs[(qq + 2) * zp + c]);
                    o.w = to_f32(scn[off + 3]) + gm * bf2f(zs[(qq + 3) * zp + c]);
                    *(float4*)(on + off) = o;
                    if (out16) *(uint2*)(out16 + (size_t)n * C * P + off) = uint2{bt_pack2(o.x, o.y), bt_pack2(o.z, o.w)};
                }
            }
        } else {
            for (int idx = tid; idx < C * TP; idx += BT_THREADS) {
                const int c = idx / TP, qq = idx - c * TP;
                if (p0 + qq < P) {
                    const size_t off = (size_t)c * P + p0 + qq;
                    const float ov = to_f32(scn[off]) + gamma[c] * sn * bf2f(zs[qq * zp + c]);
                    on[off] = ov;
                    if (out16) out16[(size_t)n * C * P + off] = bt_f2bf(ov);
                }
            }
        }
        __syncthreads();
    }
}

// dz[n,p,c] (bf16 NHWC) = scale[n] * gamma[c] * dout[n,c,p] (fp32 NCHW, staged);  part[wg][c] = sum scale[n] * dout * z
// (z read straight from global on its coalesced side)
__global__ __launch_bounds__(BT_THREADS) void scale_residual_bwd_kernel(const float* __restrict__ dout, const uint16_t* __restrict__ dout16, float* __restrict__ dsum,
                                                                      const uint16_t* __restrict__ z,
                                                                      const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                      uint16_t* __restrict__ dz, float* __restrict__ part, const TailDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = d.C, P = d.P, TP = d.TP, dp = TP + 1, cp = C / 2;
    float* ds = (float*)smem;                                                        // [C][TP+1] fp32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int KMAX = 8;
    float ag[2 * KMAX], bg[2 * KMAX];
#pragma unroll
    for (int i = 0; i < 2 * KMAX; ++i) { ag[i] = 0.f; bg[i] = 0.f; }
    for (int tile = blockIdx.x; tile < d.ntiles; tile += gridDim.x) {
        const int n = tile / d.tiles_per_image, p0 = (tile - n * d.tiles_per_image) * TP;
        const float* dn = dout + (size_t)n * C * P;
        if ((P & 3) == 0) {
            const int cpr = TP / 4;
            for (int idx = tid; idx < C * cpr; idx += BT_THREADS) {
                const int c = idx / cpr, qq = (idx - c * cpr) * 4;
                float4 v = float4{0.f, 0.f, 0.f, 0.f};
                if (p0 + qq < P) {
                    const size_t off = (size_t)c * P + p0 + qq;
                    v = *(const float4*)(dn + off);
                    if (dout16) {
                        const uint2 h = *(const uint2*)(dout16 + (size_t)n * C * P + off);
                        v.x += bf2f((uint16_t)(h.x & 0xffff)); v.y += bf2f((uint16_t)(h.x >> 16));
                        v.z += bf2f((uint16_t)(h.y & 0xffff)); v.w += bf2f((uint16_t)(h.y >> 16));
                        *(float4*)(dsum + (size_t)n * C * P + off) = v;
                    }
                }
                float* dst = ds + c * dp + qq;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        } else {
            for (int idx = tid; idx < C * TP; idx += BT_THREADS) {
                const int c = idx / TP, qq = idx - c * TP;
                float v = 0.f;
                if (p0 + qq < P) {
                    const size_t off = (size_t)c * P + p0 + qq;
                    v = dn[off];
                    if (dout16) { v += bf2f(dout16[(size_t)n * C * P + off]); dsum[(size_t)n * C * P + off] = v; }
                }
                ds[c * dp + qq] = v;
            }
        }
        __syncthreads();
        const float sn = scale ? scale[n] : 1.0f;
        const uint16_t* zn = z + ((size_t)n * P + p0) * C;
        uint16_t* dzn = dz + ((size_t)n * P + p0) * C;
        for (int q = wave; q < TP && p0 + q < P; q += BT_THREADS / 64) {
            const unsigned* zq = (const unsigned*)(zn + (size_t)q * C);
#pragma unroll
            for (int kk = 0; kk < KMAX; ++kk) {
                const int k = lane + 64 * kk;
                if (k < cp) {
                    const unsigned zv = zq[k];
                    const float d0 = ds[(2 * k) * dp + q], d1 = ds[(2 * k + 1) * dp + q];
                    ag[2 * kk] += sn * d0 * bf2f((uint16_t)(zv & 0xffff)); ag[2 * kk + 1] += sn * d1 * bf2f((uint16_t)(zv >> 16));
                    const float z0 = sn * gamma[2 * k] * d0, z1 = sn * gamma[2 * k + 1] * d1;
                    bg[2 * kk] += z0; bg[2 * kk + 1] += z1;
                    ((unsigned*)(dzn + (size_t)q * C))[k] = bt_pack2(z0, z1);
                }
            }
        }
        __syncthreads();
    }
    float* red = ds;                                                                 // 2 x [4][C], the tile is dead now
    float* red2 = ds + 4 * C;
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) {
        const int k = lane + 64 * kk;
        if (k < cp) {
            red[wave * C + 2 * k] = ag[2 * kk]; red[wave * C + 2 * k + 1] = ag[2 * kk + 1];
            red2[wave * C + 2 * k] = bg[2 * kk]; red2[wave * C + 2 * k + 1] = bg[2 * kk + 1];
        }
    }
    __syncthreads();
    float* pt = part + (size_t)blockIdx.x * 2 * C;               // [dgamma partial | column sums of dz]
    for (int c = tid; c < C; c += BT_THREADS) {
        pt[c] = ((red[c] + red[C + c]) + red[2 * C + c]) + red[3 * C + c];
        pt[C + c] = ((red2[c] + red2[C + c]) + red2[2 * C + c]) + red2[3 * C + c];
    }
}

// ===== channels_first LayerNorm (stem and downsample layers, models/SLaK.py:192-203, :256-261): y[n,c,p] = LN_C(x[n,:,p])*w+b, NCHW in
// and out.  PyTorch runs it as ~10 elementwise/reduction kernels forward and ~25 backward on fp32 tensors. =====
template <typename Tin> __device__ __forceinline__ float cf_load(const Tin* p) { return to_f32(*p); }
template <typename Tout> __device__ __forceinline__ void cf_store(Tout* p, float v);
template <> __device__ __forceinline__ void cf_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void cf_store<bf16_t>(bf16_t* p, float v) { p->v = bt_f2bf(v); }

// No transposition is involved here (NCHW in, NCHW out), so nothing is staged: a lane owns ONE pixel and walks the channels with
// coalesced 4-byte loads (256 B per wave and channel), statistics in registers, and the second pass re-reads the tile (L2 / MALL).
// The earlier LDS-tile version spent its time in per-element index arithmetic and a serial per-channel tail (3-5x the HBM time).
// pixels per workgroup = blockDim.x: 256, or 64 when that leaves too few workgroups (P = 196: one partial workgroup per image)

template <typename Tin, typename Tout>
__global__ __launch_bounds__(BT_THREADS) void ln_cf_fwd_kernel(const Tin* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                             Tout* __restrict__ y, uint16_t* __restrict__ y2, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int C, int P, int tiles_per_image, float eps) {
    const int n = blockIdx.x / tiles_per_image, p = (blockIdx.x - n * tiles_per_image) * (int)blockDim.x + threadIdx.x;
    if (p >= P) return;
    const Tin* xp = x + (size_t)n * C * P + p;
    // one pass, Welford's update (as accurate as the reference's two-pass (x - u)^2 form; 1/(c+1) is wave-uniform)
    float mu = 0.f, m2 = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float v = cf_load(xp + (size_t)c * P), d0 = v - mu;
        mu += d0 * (1.0f / (float)(c + 1));
        m2 += d0 * (v - mu);
    }
    const float r = 1.0f / sqrtf(m2 / (float)C + eps);
    mean[(size_t)n * P + p] = mu; rstd[(size_t)n * P + p] = r;
    Tout* yp = y + (size_t)n * C * P + p;
    if (y2) {                                                      // + the bf16 copy of the (fp32) result: what the first block's convs read
        uint16_t* y2p = y2 + (size_t)n * C * P + p;
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            const float v = (cf_load(xp + (size_t)c * P) - mu) * r * w[c] + b[c];
            cf_store(yp + (size_t)c * P, v); y2p[(size_t)c * P] = bt_f2bf(v);
        }
        return;
    }
#pragma unroll 8
    for (int c = 0; c < C; ++c) cf_store(yp + (size_t)c * P, (cf_load(xp + (size_t)c * P) - mu) * r * w[c] + b[c]);
}

// dx from g (same NCHW layout); per-workgroup partials part[wg][0][c] = sum g*xhat, part[wg][1][c] = sum g.
// CSPLIT: the workgroup's four waves share ONE 64-pixel tile and split the channels (few pixels, many channels: P = 196, C = 384
// would otherwise leave two long-running waves per CU); their per-pixel sums meet in LDS.
template <typename Tin, typename Tg, bool CSPLIT>
__global__ __launch_bounds__(BT_THREADS) void ln_cf_bwd_kernel(const Tg* __restrict__ g, const Tin* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             Tin* __restrict__ dx, float* __restrict__ part, int C, int P, int tiles_per_image) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = (float*)smem;                                             // [16 rows of 16 lanes][2][C] (CSPLIT: [4 rows][2][C])
    float* psum = red + (size_t)(BT_THREADS / 16) * 2 * C;                 // CSPLIT: [4 waves][2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pix = CSPLIT ? 64 : BT_THREADS;
    const int n = blockIdx.x / tiles_per_image, p = (blockIdx.x - n * tiles_per_image) * pix + (CSPLIT ? lane : tid);
    const bool ok = p < P;
    const size_t base = (size_t)n * C * P + (ok ? p : 0);
    const Tin* xp = x + base; const Tg* gp = g + base;
    const float mu = ok ? mean[(size_t)n * P + p] : 0.f, r = ok ? rstd[(size_t)n * P + p] : 0.f;
    const int cq = (C + 3) / 4;
    const int c_lo = CSPLIT ? wave * cq : 0;
    int c_hi = CSPLIT ? c_lo + cq : C; if (c_hi > C) c_hi = C;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int c = c_lo; c < c_hi; ++c) {
        const float gw = cf_load(gp + (size_t)c * P) * w[c];
        s1 += gw; s2 += gw * ((cf_load(xp + (size_t)c * P) - mu) * r);
    }
    if constexpr (CSPLIT) {
        psum[(wave * 2 + 0) * 64 + lane] = s1; psum[(wave * 2 + 1) * 64 + lane] = s2;
        __syncthreads();
        s1 = (psum[0 * 64 + lane] + psum[2 * 64 + lane]) + (psum[4 * 64 + lane] + psum[6 * 64 + lane]);
        s2 = (psum[1 * 64 + lane] + psum[3 * 64 + lane]) + (psum[5 * 64 + lane] + psum[7 * 64 + lane]);
    }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
    Tin* dxp = dx + base;
    const int row = CSPLIT ? (lane >> 4) : (tid >> 4);                     // CSPLIT: a channel belongs to one wave: 4 rows per channel
#pragma unroll 4
    for (int c = c_lo; c < c_hi; ++c) {
        const float gv = ok ? cf_load(gp + (size_t)c * P) : 0.f;          // lanes beyond the image contribute nothing
        const float xh = (cf_load(xp + (size_t)c * P) - mu) * r;
        if (ok) cf_store(dxp + (size_t)c * P, r * (gv * w[c] - m1 - xh * m2));
        const float a = bt_row16_sum(gv * xh), bs = bt_row16_sum(gv);     // per 16-lane row; the rows are added below
        if ((lane & 15) == 15) { red[(row * 2 + 0) * C + c] = a; red[(row * 2 + 1) * C + c] = bs; }
    }
    __syncthreads();
    float* pt = part + (size_t)blockIdx.x * 2 * C;
    const int nrows = CSPLIT ? 4 : BT_THREADS / 16;
    for (int i = tid; i < 2 * C; i += BT_THREADS) {
        const int h = i / C, c = i - h * C;
        float t = 0.f;
        for (int k = 0; k < nrows; ++k) t += red[(k * 2 + h) * C + c];
        pt[i] = t;
    }
}

// The same in ONE pass over g and x (x bf16, g fp32, C = CT at compile time): the workgroup's four waves share a 64-pixel tile and split the
// channels, each lane keeps its CT/4 channels of g and x in registers (36 VGPRs at C = 96: eight waves per SIMD, every load of the tile in
// flight at once), the per-pixel sums meet in LDS.  The two-pass kernel above moved 539 MB for the stem of SLaK-T (its 256-pixel tile, 147 KB,
// does not survive in L2 between the passes); a first one-pass version with a lane's WHOLE column in registers (256 VGPRs, two waves per SIMD)
// was slower than that (163 us against 141).  Optional second gradient g2 (bf16: the data gradient of the first block's convs, which read the
// bf16 copy the forward wrote), added on load: torch's `dshortcut + dx16` pass and the cast behind it are gone.
template <int CT, bool HAS2>
__global__ __launch_bounds__(BT_THREADS) void ln_cf_bwd_cs_kernel(const float* __restrict__ g, const uint16_t* __restrict__ g2,
                                                                const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                uint16_t* __restrict__ dx, float* __restrict__ part, int P, int tiles_per_image) {
    static_assert(BT_THREADS == 256 && CT % 8 == 0, "four waves, an even number of channels per wave");
    constexpr int CW = CT / 4;
    __shared__ float psum[4][2][64];
    __shared__ float red[4][2][CT];                                        // [row of 16 lanes][sum g*xhat | sum g][channel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / tiles_per_image, p = (blockIdx.x - n * tiles_per_image) * 64 + lane;
    const bool ok = p < P;
    const int c_lo = wave * CW;
    const size_t base = ((size_t)n * CT + c_lo) * P + (ok ? p : 0);
    const float mu = ok ? mean[(size_t)n * P + p] : 0.f, r = ok ? rstd[(size_t)n * P + p] : 0.f;
    float gv[CW];
    uint32_t xp[CW / 2];
#pragma unroll
    for (int c = 0; c < CW; c += 2) {
        float g0 = g[base + (size_t)c * P], g1 = g[base + (size_t)(c + 1) * P];
        if constexpr (HAS2) {
            g0 += __uint_as_float((uint32_t)g2[base + (size_t)c * P] << 16); g1 += __uint_as_float((uint32_t)g2[base + (size_t)(c + 1) * P] << 16);
        }
        gv[c] = ok ? g0 : 0.f; gv[c + 1] = ok ? g1 : 0.f;                   // lanes beyond the image contribute nothing
        xp[c / 2] = (uint32_t)x[base + (size_t)c * P] | ((uint32_t)x[base + (size_t)(c + 1) * P] << 16);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const float xv = __uint_as_float((c & 1) ? (xp[c / 2] & 0xffff0000u) : (xp[c / 2] << 16));
        const float gw = gv[c] * w[c_lo + c];
        s1 += gw; s2 += gw * ((xv - mu) * r);
    }
    psum[wave][0][lane] = s1; psum[wave][1][lane] = s2;
    __syncthreads();
    s1 = (psum[0][0][lane] + psum[1][0][lane]) + (psum[2][0][lane] + psum[3][0][lane]);
    s2 = (psum[0][1][lane] + psum[1][1][lane]) + (psum[2][1][lane] + psum[3][1][lane]);
    const float m1 = s1 / (float)CT, m2 = s2 / (float)CT;
    const int row = lane >> 4;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const float xv = __uint_as_float((c & 1) ? (xp[c / 2] & 0xffff0000u) : (xp[c / 2] << 16));
        const float xh = (xv - mu) * r;
        if (ok) dx[base + (size_t)c * P] = bt_f2bf(r * (gv[c] * w[c_lo + c] - m1 - xh * m2));
        const float a = bt_row16_sum(gv[c] * xh), bs = bt_row16_sum(gv[c]);
        if ((lane & 15) == 15) { red[row][0][c_lo + c] = a; red[row][1][c_lo + c] = bs; }
    }
    __syncthreads();
    float* pt = part + (size_t)blockIdx.x * 2 * CT;
    for (int i = tid; i < 2 * CT; i += BT_THREADS) {
        const int h = i / CT, c = i - h * CT;
        pt[i] = (red[0][h][c] + red[1][h][c]) + (red[2][h][c] + red[3][h][c]);
    }
}

// ===== GELU backward fused with the bias gradient of the Linear in front of it: dy1 = dact * gelu'(y1) (exact erf form, like
// nn.GELU()), part[wg][col] = sum over the workgroup's rows of dy1.  [rows][cols] bf16, cols % 8 == 0.  Thread <-> 8 columns. =====
__global__ __launch_bounds__(BT_THREADS) void gelu_bwd_bias_kernel(const uint16_t* __restrict__ dact, const uint16_t* __restrict__ y1,
                                                                 uint16_t* __restrict__ dy1, float* __restrict__ part,
                                                                 int rows, int cols, int rows_per_wg, const float* __restrict__ table) {
    // threads as (row lane ry, column chunk cx): narrow matrices (cols = 384) still use the whole workgroup; the workgroup size is
    // chosen by the launcher (64..256) so that ccn * nry covers it (cols = 768, 1536, 3072: 192 threads instead of 3/4 of 256)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const T = (float*)smem;                               // gelu' table, then red[nry][cols]
    float* red = (float*)(smem + GD_BYTES);
    const int nthr = blockDim.x;
    for (int i = threadIdx.x; i < GD_BYTES / 16; i += nthr) ((uint4*)T)[i] = ((const uint4*)table)[i];
    __syncthreads();
    const int cchunks = cols / 8;
    const int ccn = cchunks < nthr ? cchunks : nthr, nry = nthr / ccn;
    const int ry = threadIdx.x / ccn, cx = threadIdx.x - ry * ccn;
    const int r0 = blockIdx.x * rows_per_wg;
    int r1 = r0 + rows_per_wg; if (r1 > rows) r1 = rows;
    for (int cc0 = 0; cc0 < cchunks; cc0 += ccn) {
        const int cc = cc0 + cx;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ry < nry && cc < cchunks) {
            // the thread's rows r0 + ry + i*nry in pairs; the loads of pair p+1 are issued BEFORE pair p is evaluated (two register
            // sets), so that the memory pipe stays full while the VALU works (issued after it, the two phases added up: 4.4 TB/s)
            const size_t rs = (size_t)nry * cols;
            const size_t o00 = (size_t)(r0 + ry) * cols + cc * 8;
            const int nrows_t = r1 > r0 + ry ? (r1 - (r0 + ry) + nry - 1) / nry : 0, npairs = nrows_t >> 1;
            struct Pair { uint4 g0, y0, g1, y1v; };
            auto ld = [&](int pi, Pair& P) {
                const size_t o0 = o00 + (size_t)(2 * pi) * rs, o1 = o0 + rs;
                P.g0 = *(const uint4*)(dact + o0); P.y0 = *(const uint4*)(y1 + o0); P.g1 = *(const uint4*)(dact + o1); P.y1v = *(const uint4*)(y1 + o1);
            };
            auto row = [&](const uint4& g, const uint4& y, size_t o) {
                float a2[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) a2[k] = acc[k];
                uint4 v;
                const bool ok = gelu_bwd8_fast(T, g, y, v, a2);
                if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {     // (wave-uniform, rare) an element outside the table: the general evaluation
                    gelu_bwd8(T, g, y, v, acc);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = a2[k];
                }
                *(uint4*)(dy1 + o) = v;
            };
            auto go = [&](int pi, const Pair& P) {
                const size_t o0 = o00 + (size_t)(2 * pi) * rs;
                row(P.g0, P.y0, o0); row(P.g1, P.y1v, o0 + rs);
            };
            Pair A, B;
            int pi = 0;
            if (npairs > 0) ld(0, A);
            for (; pi + 2 < npairs; pi += 2) {
                ld(pi + 1, B); go(pi, A);
                ld(pi + 2, A); go(pi + 1, B);
            }
            if (pi + 1 < npairs) { ld(pi + 1, B); go(pi, A); go(pi + 1, B); }
            else if (pi < npairs) go(pi, A);
            if (nrows_t & 1) {
                const size_t o0 = o00 + (size_t)(nrows_t - 1) * rs;
                row(*(const uint4*)(dact + o0), *(const uint4*)(y1 + o0), o0);
            }
        }
        __syncthreads();                                         // previous pass's partials have been consumed
        if (ry < nry && cc < cchunks) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[(size_t)ry * cols + cc * 8 + k] = acc[k];
        }
        __syncthreads();
        if (ry == 0 && cc < cchunks) {                           // fixed order over the row lanes
            float* pt = part + (size_t)blockIdx.x * cols + cc * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t = 0.f;
                for (int q = 0; q < nry; ++q) t += red[(size_t)q * cols + cc * 8 + k];
                pt[k] = t;
            }
        }
    }
}

constexpr int BT_SLICES = 64;                                   // (workspace head-room kept for older callers' layouts)
// Column sums of part[ntiles][width] in ONE launch, fixed order: a workgroup owns 32 columns, its 32 row groups add rows g, g + 32, ...
// (four independent partial sums per thread so that the loads pipeline), the groups are added in order through LDS.  The round-1
// two-pass version (64 slices, then the slices) cost two ~5 us launches per reduction: 116 launches per SLaK-T step.
__global__ __launch_bounds__(1024) void block_tail_reduce1(const float* __restrict__ part, float* __restrict__ out0, float* __restrict__ out1, int split,
                                                           int ntiles, int width) {
    __shared__ float acc[32][33];
    const int cx = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < width) {
        const float* p = part + j;
        int t = g;
        for (; t + 96 < ntiles; t += 128) {
            s0 += p[(size_t)t * width]; s1 += p[(size_t)(t + 32) * width]; s2 += p[(size_t)(t + 64) * width]; s3 += p[(size_t)(t + 96) * width];
        }
        for (; t < ntiles; t += 32) s0 += p[(size_t)t * width];
    }
    acc[g][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && j < width) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += acc[k][cx];
        if (j < split) out0[j] = t; else out1[j - split] = t;
    }
}
static int reduce_partials(const float* part, float* tmp, float* out0, float* out1, int split, int ntiles, int width, hipStream_t st) {
    (void)tmp;
    if (reduce_defer_push(0, part, out0, out1, split, ntiles, width, st)) return SLAK_OK;      // (slak_defer_reductions_begin: added later, with others, in one launch)
    hipLaunchKernelGGL(block_tail_reduce1, dim3((unsigned)((width + 31) / 32)), dim3(1024), 0, st, part, out0, out1, split, ntiles, width);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_hip_error(e); return SLAK_ERR_LAUNCH; }
    return SLAK_OK;
}

int tail_reduce_columns(const float* part, float* out, int ntiles, int width, hipStream_t st) { return reduce_partials(part, nullptr, out, out, width, ntiles, width, st); }
int tail_reduce_split(const float* part, float* out0, float* out1, int split, int ntiles, int width, hipStream_t st) { return reduce_partials(part, nullptr, out0, out1, split, ntiles, width, st); }

// tile width: the largest of {128,64,32,16} pixels whose tile (elem_bytes per element) fits lds_budget, not (much) wider than an image
static TailDims make_dims(int N, int C, int P, int elem_bytes, size_t lds_budget = 50 * 1024) {
    // tile width: the persistent kernels want (i) little padding in the last tile of an image (P = 196: 64-pixel tiles waste 31 %),
    // (ii) enough tiles to keep ~4 workgroups per CU busy, (iii) then the widest tile that fits the LDS budget
    TailDims d; d.N = N; d.C = C; d.P = P;
    static const int cand[] = {128, 64, 56, 48, 40, 32, 28, 24, 20, 16, 12, 8};
    static const char* ov = slak_dev_getenv("SLAK_TAIL_TP");
    int best = 16; double best_score = -1e30;
    for (int TP : cand) {
        if (TP > 16 && (size_t)C * (TP + 2) * elem_bytes > lds_budget) continue;
        if (TP > 16 && TP / 2 >= P) continue;
        const int tiles = (P + TP - 1) / TP;
        const double waste = (double)tiles * TP / P - 1.0;
        const double fill = (double)N * tiles / 1024.0;                  // tiles per "4 workgroups per CU"
        const double score = -4.0 * waste + (fill < 1.0 ? fill - 1.0 : 0.0) + 0.001 * TP;
        if (score > best_score) { best_score = score; best = TP; }
    }
    if (ov && atoi(ov) > 0) best = atoi(ov);
    d.TP = best;
    d.tiles_per_image = (P + d.TP - 1) / d.TP;
    d.ntiles = N * d.tiles_per_image;
    return d;
}
// variant A: fixed 64-pixel tiles, one workgroup per tile
static TailDims make_dims_pix(int N, int C, int P) {
    TailDims d; d.N = N; d.C = C; d.P = P; d.TP = 64;
    d.tiles_per_image = (P + d.TP - 1) / d.TP; d.ntiles = N * d.tiles_per_image;
    return d;
}
static int g_tail_cus = 0;
static int tail_grid(const TailDims& d, size_t lds) {                  // persistent: as many workgroups as stay resident
    if (g_tail_cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        g_tail_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int per_cu = (int)((size_t)(160 * 1024) / (lds + 1024)); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
    const int resident = per_cu * g_tail_cus;
    return d.ntiles < resident ? d.ntiles : resident;
}
static int set_lds(const void* k, size_t lds) {
    return slak_set_max_lds(k, lds) ? 0 : 1;
}

// Stem (models/SLaK.py:276-279: Conv2d(in_chans, C, kernel_size=4, stride=4)): the non-overlapping 4x4 patches of the fp32 NCHW image as the
// bf16 GEMM operand a[n][ho * Wo + wo][(c * 4 + kh) * 4 + kw] (the conv weight's own (c, kh, kw) order).  Thread <-> one (patch, c, kh):
// a 16-byte read of four pixels, an 8-byte write; writes are fully coalesced (96 bytes per patch, patches contiguous).
__global__ __launch_bounds__(256) void stem_patchify_kernel(const float* __restrict__ x, uint16_t* __restrict__ a, int N, int Cin, int H, int W) {
    const int Ho = H / 4, Wo = W / 4, J = Cin * 4;
    const long long total = (long long)N * Ho * Wo * J;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int j = (int)(t % J); const long long pt = t / J;
        const int wo = (int)(pt % Wo); const long long q = pt / Wo;
        const int ho = (int)(q % Ho), n = (int)(q / Ho);
        const int c = j >> 2, kh = j & 3;
        const float4 v = *(const float4*)(x + (((size_t)n * Cin + c) * H + 4 * ho + kh) * W + 4 * wo);
        *(uint2*)(a + (size_t)t * 4) = uint2{bt_pack2(v.x, v.y), bt_pack2(v.z, v.w)};
    }
}

// the table of gelu_grad_lut in device memory (one copy per device, built on first use): entry [sign * GD_N + (mag - GD_LO)]
const float* gelu_grad_table_device() {
    static std::mutex mu;
    static const float* tab[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (tab[dev]) return tab[dev];
    std::vector<float> h(2 * GD_N);
    for (unsigned sgn = 0; sgn < 2; ++sgn)
        for (unsigned i = 0; i < GD_N; ++i) {
            const uint32_t bits = ((sgn << 15) | (GD_LO + i)) << 16;
            float xf; memcpy(&xf, &bits, 4);
            const double x = xf;
            h[sgn * GD_N + i] = (float)(0.5 * erfc(-x * 0.70710678118654752440) + x * 0.39894228040143267794 * exp(-0.5 * x * x));
        }
    void* d = nullptr;
    if (hipMalloc(&d, h.size() * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    tab[dev] = (const float*)d;
    return tab[dev];
}

}  // namespace slak

using namespace slak;

extern "C" {

size_t slak_block_tail_workspace_bytes(int N, int C, int P) {
    if (N <= 0 || C <= 0 || P <= 0) return 0;
    size_t rows = 8 * 1024;                                                 // variant B: one partial row per persistent workgroup (<= 8 per CU)
    if (C <= 256) { const size_t t = (size_t)N * ((P + 63) / 64); if (t > rows) rows = t; }   // variant A: one per tile
    return align_up((rows + BT_SLICES) * 2 * C * sizeof(float), 256);
}

static bool tail_use_reg() {                  // SLAK_TAIL_REG=0 keeps the LDS-tile kernels (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_TAIL_REG"); return !(e && e[0] == '0'); }();
    return v;
}
static int tail_args_ok(int N, int C, int P) {
    if (N <= 0 || C <= 0 || P <= 0) return SLAK_ERR_INVALID_ARG;
    if ((C & 1) || C > 1024) return SLAK_ERR_UNSUPPORTED;
    if ((long long)N * C * P >= (1LL << 31)) return SLAK_ERR_UNSUPPORTED;
    return SLAK_OK;
}

int slak_ln_nchw_to_nhwc_forward(const void* x, const float* weight, const float* bias, void* y, float* mean, float* rstd,
                                 int N, int C, int P, float eps, void* stream) {
    if (!x || !weight || !bias || !y || !mean || !rstd) return SLAK_ERR_INVALID_ARG;
    int rc = tail_args_ok(N, C, P); if (rc) return rc;
    if (tail_use_reg()) {                               // register-tile kernel (block_tail_reg.hip) where an instantiation exists
        rc = launch_ln_nchw_to_nhwc_fwd_reg(x, weight, bias, y, mean, rstd, N, C, P, eps, (hipStream_t)stream);
        if (rc != SLAK_ERR_UNSUPPORTED) return rc;
    }
    if (C <= 256) {
        const TailDims d = make_dims_pix(N, C, P);
        const size_t lds = (((size_t)C * (d.TP + 2) * 2 + 15) & ~(size_t)15) + (size_t)(BT_THREADS / d.TP) * d.TP * 4 + 2 * d.TP * 4;
        if (set_lds((const void*)ln_nchw_to_nhwc_fwd_pix_kernel, lds)) return SLAK_ERR_LAUNCH;
        hipLaunchKernelGGL(ln_nchw_to_nhwc_fwd_pix_kernel, dim3((unsigned)d.ntiles), dim3(BT_THREADS), lds, (hipStream_t)stream,
                           (const uint16_t*)x, weight, bias, (uint16_t*)y, mean, rstd, d, eps);
        SLAK_LAUNCH_CHECK();
        return SLAK_OK;
    }
    const TailDims d = make_dims(N, C, P, 2);
    const size_t lds = (size_t)C * (d.TP + 2) * 2 + 16;
    if (set_lds((const void*)ln_nchw_to_nhwc_fwd_kernel, lds)) return SLAK_ERR_LAUNCH;
    hipLaunchKernelGGL(ln_nchw_to_nhwc_fwd_kernel, dim3((unsigned)tail_grid(d, lds)), dim3(BT_THREADS), lds, (hipStream_t)stream,
                       (const uint16_t*)x, weight, bias, (uint16_t*)y, mean, rstd, d, eps);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_ln_nchw_to_nhwc_backward(const void* g, const void* x, const float* weight, const float* mean, const float* rstd,
                                  void* dx, float* dweight, float* dbias, int N, int C, int P,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!g || !x || !weight || !mean || !rstd || !dx || !dweight || !dbias) return SLAK_ERR_INVALID_ARG;
    int rc = tail_args_ok(N, C, P); if (rc) return rc;
    if (!workspace || workspace_bytes < slak_block_tail_workspace_bytes(N, C, P)) return SLAK_ERR_WORKSPACE;
    if (tail_use_reg()) {
        int rows = 0; float* part = (float*)workspace;
        rc = launch_ln_nchw_to_nhwc_bwd_reg(g, x, weight, mean, rstd, dx, part, &rows, N, C, P, (hipStream_t)stream);
        if (rc == SLAK_OK) return reduce_partials(part, part + (size_t)rows * 2 * C, dweight, dbias, C, rows, 2 * C, (hipStream_t)stream);
        if (rc != SLAK_ERR_UNSUPPORTED) return rc;
    }
    if (C <= 256) {
        const TailDims d = make_dims_pix(N, C, P);
        const size_t lds = (((size_t)C * (d.TP + 2) * 2 + 15) & ~(size_t)15) + (((size_t)d.TP * (C + 2) * 2 + 15) & ~(size_t)15) +
                           (size_t)2 * (BT_THREADS / d.TP) * d.TP * 4 + 4 * d.TP * 4;
        if (set_lds((const void*)ln_nchw_to_nhwc_bwd_pix_kernel, lds)) return SLAK_ERR_LAUNCH;
        float* part = (float*)workspace;
        hipLaunchKernelGGL(ln_nchw_to_nhwc_bwd_pix_kernel, dim3((unsigned)d.ntiles), dim3(BT_THREADS), lds, (hipStream_t)stream,
                           (const uint16_t*)g, (const uint16_t*)x, weight, mean, rstd, (uint16_t*)dx, part, d);
        SLAK_LAUNCH_CHECK();
        return reduce_partials(part, part + (size_t)d.ntiles * 2 * C, dweight, dbias, C, d.ntiles, 2 * C, (hipStream_t)stream);
    }
    const TailDims d = make_dims(N, C, P, 2);
    const size_t lds = (((size_t)C * (d.TP + 2) * 2 + 15) & ~(size_t)15) + (size_t)4 * 2 * C * 4;
    if (set_lds((const void*)ln_nchw_to_nhwc_bwd_kernel, lds)) return SLAK_ERR_LAUNCH;
    const int grid = tail_grid(d, lds);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(ln_nchw_to_nhwc_bwd_kernel, dim3((unsigned)grid), dim3(BT_THREADS), lds, (hipStream_t)stream,
                       (const uint16_t*)g, (const uint16_t*)x, weight, mean, rstd, (uint16_t*)dx, part, d);
    SLAK_LAUNCH_CHECK();
    return reduce_partials(part, part + (size_t)grid * 2 * C, dweight, dbias, C, grid, 2 * C, (hipStream_t)stream);
}

// LayerNorm(channels_first) of the downsample layers fused with the layout the following Conv2d(k = 2, stride 2) wants as a GEMM operand
int slak_ln_patch_supported(int N, int C, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || ((H | W) & 1) || tail_args_ok(N, C, H * W)) return 0;
    return C == 64 || C == 96 || C == 128 || C == 192 || C == 256 || C == 384 || C == 512;
}
int slak_ln_patch_forward(const float* x, const float* weight, const float* bias, void* a_bf16, float* mean, float* rstd,
                          int N, int C, int H, int W, float eps, void* stream) {
    if (!x || !weight || !bias || !a_bf16 || !mean || !rstd) return SLAK_ERR_INVALID_ARG;
    if (!slak_ln_patch_supported(N, C, H, W)) return SLAK_ERR_UNSUPPORTED;
    return launch_ln_patch_fwd_reg(x, weight, bias, a_bf16, mean, rstd, N, C, H, W, eps, (hipStream_t)stream);
}
int slak_ln_patch_backward(const void* g_bf16, const float* x, const float* weight, const float* mean, const float* rstd,
                           float* dx, float* dweight, float* dbias, int N, int C, int H, int W,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!g_bf16 || !x || !weight || !mean || !rstd || !dx || !dweight || !dbias) return SLAK_ERR_INVALID_ARG;
    if (!slak_ln_patch_supported(N, C, H, W)) return SLAK_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < slak_block_tail_workspace_bytes(N, C, H * W)) return SLAK_ERR_WORKSPACE;
    int rows = 0; float* part = (float*)workspace;
    const int rc = launch_ln_patch_bwd_reg(g_bf16, x, weight, mean, rstd, dx, part, &rows, N, C, H, W, (hipStream_t)stream);
    if (rc != SLAK_OK) return rc;
    return reduce_partials(part, part + (size_t)rows * 2 * C, dweight, dbias, C, rows, 2 * C, (hipStream_t)stream);
}

int slak_stem_patchify(const float* x, void* a_bf16, int N, int Cin, int H, int W, void* stream) {
    if (!x || !a_bf16) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0) return SLAK_ERR_INVALID_ARG;
    if ((H & 3) || (W & 3) || (long long)N * Cin * H * W >= (1LL << 40)) return SLAK_ERR_UNSUPPORTED;
    const long long total = (long long)N * (H / 4) * (W / 4) * Cin * 4;
    long long g = (total + 255) / 256; if (g > 65536) g = 65536;
    hipLaunchKernelGGL(stem_patchify_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)a_bf16, N, Cin, H, W);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

// y[n][c][p] = bf16(bias[c]): the accumulator the downsample convolutions' batched GEMM starts from (beta = 1).  torch materialises the
// broadcast bias with a strided copy kernel at ~1.3 TB/s (30 us for the 38 MB of the first downsample layer); this writes 16 bytes per lane.
static __global__ __launch_bounds__(256) void fill_channel_bias_kernel(const float* __restrict__ bias, uint16_t* __restrict__ y, int C, int P, long long chunks, long long total) {
    for (long long ch = (long long)blockIdx.x * 256 + threadIdx.x; ch < chunks; ch += (long long)gridDim.x * 256) {
        const long long e0 = ch * 8;
        uint16_t v[8];
        const long long row = e0 / P;                              // one 64-bit division per chunk; the elements step through (channel, pixel) from there
        int c = (int)(row % C), rem = (int)(e0 - row * P);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = bt_f2bf(bias[c]);
            if (++rem == P) { rem = 0; if (++c == C) c = 0; }
        }
        if (e0 + 8 <= total) {
            *(uint4*)(y + e0) = uint4{(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                                      (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
        } else {
            for (int j = 0; e0 + j < total; ++j) y[e0 + j] = v[j];
        }
    }
}

int slak_fill_channel_bias_bf16(const float* bias, void* y_bf16, int N, int C, int P, void* stream) {
    if (!bias || !y_bf16) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || C <= 0 || P <= 0) return SLAK_ERR_INVALID_ARG;
    const long long total = (long long)N * C * P;
    if (total >= (1LL << 40)) return SLAK_ERR_UNSUPPORTED;
    const long long chunks = (total + 7) / 8;
    long long g = (chunks + 255) / 256; if (g > 16384) g = 16384;
    hipLaunchKernelGGL(fill_channel_bias_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, bias, (uint16_t*)y_bf16, C, P, chunks, total);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

// Per-channel sums of a bf16 NCHW gradient (the bias gradient of the stem / downsample convolutions): out[c] = sum_{n,p} x[n][c][p], fp32.
// One workgroup per (channel, image slice): a wave reads whole rows of P pixels (8-byte loads when P % 4 == 0), lanes keep fp32 partial sums,
// the slices are added by block_tail_reduce1 in a fixed order -> the same bits on every run.  (torch's sum((0, 2)) on this layout reads at
// ~1 TB/s: 0.19 ms per SLaK-T step for four tensors of 144 MB together.)
constexpr int CS_SLICES = 32;
static __global__ __launch_bounds__(256) void channel_sums_kernel(const uint16_t* __restrict__ x, float* __restrict__ part, int N, int C, int P, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int n = s + S * w; n < N; n += S * 4) {
        const uint16_t* row = x + ((size_t)n * C + c) * P;
        if ((P & 3) == 0) {
            const uint2* r4 = (const uint2*)row;
            for (int i = lane; i < (P >> 2); i += 64) {
                const uint2 v = r4[i];
                a0 += __uint_as_float(v.x << 16); a1 += __uint_as_float(v.x & 0xffff0000u);
                a2 += __uint_as_float(v.y << 16); a3 += __uint_as_float(v.y & 0xffff0000u);
            }
        } else {
            for (int i = lane; i < P; i += 64) a0 += __uint_as_float((uint32_t)row[i] << 16);
        }
    }
    float t = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if (lane == 0) red[w] = t;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)s * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

size_t slak_channel_sums_workspace_bytes(int C) { return C > 0 ? (size_t)CS_SLICES * C * sizeof(float) : 0; }

int slak_channel_sums_bf16(const void* x_bf16, float* out, int N, int C, int P, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x_bf16 || !out) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || C <= 0 || P <= 0) return SLAK_ERR_INVALID_ARG;
    if (C > 65535 || (long long)N * C * P >= (1LL << 40)) return SLAK_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < slak_channel_sums_workspace_bytes(C)) return SLAK_ERR_WORKSPACE;
    const int S = N < CS_SLICES ? N : CS_SLICES;
    hipLaunchKernelGGL(channel_sums_kernel, dim3((unsigned)C, (unsigned)S), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16,
                       (float*)workspace, N, C, P, S);
    SLAK_LAUNCH_CHECK();
    return reduce_partials((const float*)workspace, nullptr, out, out, C, S, C, (hipStream_t)stream);
}

int slak_scale_residual_forward(const void* shortcut, int shortcut_dtype, const void* z, const float* gamma, const float* sample_scale,
                                float* out, void* out_bf16, int N, int C, int P, void* stream) {
    if (!shortcut || !z || !gamma || !out) return SLAK_ERR_INVALID_ARG;
    int rc = tail_args_ok(N, C, P); if (rc) return rc;
    if (tail_use_reg()) {
        rc = launch_scale_residual_fwd_reg(shortcut, shortcut_dtype, z, gamma, sample_scale, out, out_bf16, N, C, P, (hipStream_t)stream);
        if (rc != SLAK_ERR_UNSUPPORTED) return rc;
    }
    if (C <= 256) {
        const TailDims d = make_dims_pix(N, C, P);
        const size_t lds = (size_t)d.TP * (C + 2) * 2 + 16;
        const dim3 grid((unsigned)d.ntiles);
        if (shortcut_dtype == SLAK_F32) {
            if (set_lds((const void*)scale_residual_fwd_pix_kernel<float>, lds)) return SLAK_ERR_LAUNCH;
            hipLaunchKernelGGL(scale_residual_fwd_pix_kernel<float>, grid, dim3(BT_THREADS), lds, (hipStream_t)stream,
                               (const float*)shortcut, (const uint16_t*)z, gamma, sample_scale, out, (uint16_t*)out_bf16, d);
        } else if (shortcut_dtype == SLAK_BF16) {
            if (set_lds((const void*)scale_residual_fwd_pix_kernel<bf16_t>, lds)) return SLAK_ERR_LAUNCH;
            hipLaunchKernelGGL(scale_residual_fwd_pix_kernel<bf16_t>, grid, dim3(BT_THREADS), lds, (hipStream_t)stream,
                               (const bf16_t*)shortcut, (const uint16_t*)z, gamma, sample_scale, out, (uint16_t*)out_bf16, d);
        } else return SLAK_ERR_UNSUPPORTED;
        SLAK_LAUNCH_CHECK();
        return SLAK_OK;
    }
    TailDims d = make_dims(N, C, P, 2);
    // the NHWC tile is [TP][C+2]: same byte count as [C][TP+2] up to the padding
    while (d.TP > 16 && (size_t)d.TP * (C + 2) * 2 > 50 * 1024) { d.TP /= 2; d.tiles_per_image = (P + d.TP - 1) / d.TP; d.ntiles = N * d.tiles_per_image; }
    const size_t lds = (size_t)d.TP * (C + 2) * 2 + 16;
    const dim3 grid((unsigned)tail_grid(d, lds));
    if (shortcut_dtype == SLAK_F32) {
        if (set_lds((const void*)scale_residual_fwd_kernel<float>, lds)) return SLAK_ERR_LAUNCH;
        hipLaunchKernelGGL(scale_residual_fwd_kernel<float>, grid, dim3(BT_THREADS), lds, (hipStream_t)stream,
                           (const float*)shortcut, (const uint16_t*)z, gamma, sample_scale, out, (uint16_t*)out_bf16, d);
    } else if (shortcut_dtype == SLAK_BF16) {
        if (set_lds((const void*)scale_residual_fwd_kernel<bf16_t>, lds)) return SLAK_ERR_LAUNCH;
        hipLaunchKernelGGL(scale_residual_fwd_kernel<bf16_t>, grid, dim3(BT_THREADS), lds, (hipStream_t)stream,
                           (const bf16_t*)shortcut, (const uint16_t*)z, gamma, sample_scale, out, (uint16_t*)out_bf16, d);
    } else return SLAK_ERR_UNSUPPORTED;
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_scale_residual_backward(const float* dout, const void* dout_bf16, float* dout_sum, const void* z, const float* gamma,
                                 const float* sample_scale, void* dz, float* dgamma, float* dz_colsum, int N, int C, int P,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!dout || !z || !gamma || !dz || !dgamma || !dz_colsum || (dout_bf16 && !dout_sum)) return SLAK_ERR_INVALID_ARG;
    int rc = tail_args_ok(N, C, P); if (rc) return rc;
    if (!workspace || workspace_bytes < slak_block_tail_workspace_bytes(N, C, P)) return SLAK_ERR_WORKSPACE;
    if (tail_use_reg()) {
        int rows = 0; float* part = (float*)workspace;
        rc = launch_scale_residual_bwd_reg(dout, dout_bf16, dout_sum, z, gamma, sample_scale, dz, part, &rows, N, C, P, (hipStream_t)stream);
        if (rc == SLAK_OK) return reduce_partials(part, part + (size_t)rows * 2 * C, dgamma, dz_colsum, C, rows, 2 * C, (hipStream_t)stream);
        if (rc != SLAK_ERR_UNSUPPORTED) return rc;
    }
    if (C <= 256) {
        const TailDims d = make_dims_pix(N, C, P);
        const size_t lds = (size_t)C * (d.TP + 1) * 4 + (size_t)d.TP * (C + 2) * 2 + 16;
        if (set_lds((const void*)scale_residual_bwd_pix_kernel, lds)) return SLAK_ERR_LAUNCH;
        float* part = (float*)workspace;
        hipLaunchKernelGGL(scale_residual_bwd_pix_kernel, dim3((unsigned)d.ntiles), dim3(BT_THREADS), lds, (hipStream_t)stream,
                           dout, (const uint16_t*)dout_bf16, dout_sum, (const uint16_t*)z, gamma, sample_scale, (uint16_t*)dz, part, d);
        SLAK_LAUNCH_CHECK();
        return reduce_partials(part, part + (size_t)d.ntiles * 2 * C, dgamma, dz_colsum, C, d.ntiles, 2 * C, (hipStream_t)stream);
    }
    const TailDims d = make_dims(N, C, P, 4);
    size_t lds = (size_t)C * (d.TP + 1) * 4 + 16;
    if (lds < (size_t)4 * C * 4 + 16) lds = (size_t)4 * C * 4 + 16;
    if (set_lds((const void*)scale_residual_bwd_kernel, lds)) return SLAK_ERR_LAUNCH;
    const int grid = tail_grid(d, lds);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(scale_residual_bwd_kernel, dim3((unsigned)grid), dim3(BT_THREADS), lds, (hipStream_t)stream,
                       dout, (const uint16_t*)dout_bf16, dout_sum, (const uint16_t*)z, gamma, sample_scale, (uint16_t*)dz, part, d);
    SLAK_LAUNCH_CHECK();
    return reduce_partials(part, part + (size_t)grid * 2 * C, dgamma, dz_colsum, C, grid, 2 * C, (hipStream_t)stream);
}

static int cf_block(int N, int P) { return (long long)N * ((P + BT_THREADS - 1) / BT_THREADS) >= 1024 ? BT_THREADS : 64; }
static int cf_tiles(int N, int P) { const int b = cf_block(N, P); return (P + b - 1) / b; }

size_t slak_ln_cf_workspace_bytes(int N, int C, int P) {
    if (N <= 0 || C <= 0 || P <= 0) return 0;
    return align_up(((size_t)N * ((P + 63) / 64) + BT_SLICES) * 2 * C * sizeof(float), 256);   // rows: one per 64-pixel tile (the most any of the kernels writes)
}

int slak_ln_channels_first_forward(const void* x, int x_dtype, const float* weight, const float* bias, void* y, int y_dtype,
                                   float* mean, float* rstd, int N, int C, int P, float eps, void* stream) {
    return slak_ln_channels_first_forward_pair(x, x_dtype, weight, bias, y, y_dtype, nullptr, mean, rstd, N, C, P, eps, stream);
}

int slak_ln_channels_first_forward_pair(const void* x, int x_dtype, const float* weight, const float* bias, void* y, int y_dtype, void* y_bf16,
                                        float* mean, float* rstd, int N, int C, int P, float eps, void* stream) {
    if (!x || !weight || !bias || !y || !mean || !rstd) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || C <= 0 || P <= 0) return SLAK_ERR_INVALID_ARG;
    if (C > 1024 || (long long)N * C * P >= (1LL << 31)) return SLAK_ERR_UNSUPPORTED;
    const int tpi = cf_tiles(N, P), blk = cf_block(N, P);
    const dim3 grid((unsigned)(N * tpi));
#define SLAK_CF_FWD(TI, TO)                                                                                                   \
    hipLaunchKernelGGL((ln_cf_fwd_kernel<TI, TO>), grid, dim3(blk), 0, (hipStream_t)stream, (const TI*)x, weight, bias,   \
                       (TO*)y, (uint16_t*)y_bf16, mean, rstd, C, P, tpi, eps)
    if (x_dtype == SLAK_F32 && y_dtype == SLAK_F32) SLAK_CF_FWD(float, float);
    else if (x_dtype == SLAK_F32 && y_dtype == SLAK_BF16) SLAK_CF_FWD(float, bf16_t);
    else if (x_dtype == SLAK_BF16 && y_dtype == SLAK_F32) SLAK_CF_FWD(bf16_t, float);
    else if (x_dtype == SLAK_BF16 && y_dtype == SLAK_BF16) SLAK_CF_FWD(bf16_t, bf16_t);
    else return SLAK_ERR_UNSUPPORTED;
#undef SLAK_CF_FWD
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

// the one-pass kernel (64-pixel tiles, the waves split the channels, a lane's share in registers): bf16 x, fp32 g, C = 96 / 128 / 192 (the stems of
// SLaK-T/S, -B and -L)
static bool cf_reg_covers(int g_dtype, int x_dtype, int N, int C, int P) {
    static const bool off = [] { const char* e = getenv("SLAK_LN_CF_REG"); return e && atoi(e) == 0; }();
    (void)N; (void)P;
    return !off && g_dtype == SLAK_F32 && x_dtype == SLAK_BF16 && (C == 96 || C == 128 || C == 192);
}
int slak_ln_channels_first_backward_pair_supported(int g_dtype, int x_dtype, int N, int C, int P) {
    if (N <= 0 || C <= 0 || P <= 0 || (long long)N * C * P >= (1LL << 31)) return 0;
    return cf_reg_covers(g_dtype, x_dtype, N, C, P) ? 1 : 0;
}

int slak_ln_channels_first_backward(const void* g, int g_dtype, const void* x, int x_dtype, const float* weight, const float* mean,
                                    const float* rstd, void* dx, float* dweight, float* dbias, int N, int C, int P,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return slak_ln_channels_first_backward_pair(g, g_dtype, nullptr, x, x_dtype, weight, mean, rstd, dx, dweight, dbias, N, C, P, workspace, workspace_bytes, stream);
}

int slak_ln_channels_first_backward_pair(const void* g, int g_dtype, const void* g2_bf16, const void* x, int x_dtype, const float* weight,
                                         const float* mean, const float* rstd, void* dx, float* dweight, float* dbias, int N, int C, int P,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!g || !x || !weight || !mean || !rstd || !dx || !dweight || !dbias) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || C <= 0 || P <= 0) return SLAK_ERR_INVALID_ARG;
    if (C > 1024 || (long long)N * C * P >= (1LL << 31)) return SLAK_ERR_UNSUPPORTED;
    if (g2_bf16 && !cf_reg_covers(g_dtype, x_dtype, N, C, P)) return SLAK_ERR_UNSUPPORTED;      // (the caller adds the two gradients itself)
    if (!workspace || workspace_bytes < slak_ln_cf_workspace_bytes(N, C, P)) return SLAK_ERR_WORKSPACE;
    if (cf_reg_covers(g_dtype, x_dtype, N, C, P)) {
        const int tpi = (P + 63) / 64, nwg = N * tpi;
        float* part = (float*)workspace;
#define SLAK_CF_REG(CT, H2)                                                                                                               \
        hipLaunchKernelGGL((ln_cf_bwd_cs_kernel<CT, H2>), dim3((unsigned)nwg), dim3(BT_THREADS), 0, (hipStream_t)stream, (const float*)g, \
                           (const uint16_t*)g2_bf16, (const uint16_t*)x, weight, mean, rstd, (uint16_t*)dx, part, P, tpi)
        if (C == 96) { if (g2_bf16) SLAK_CF_REG(96, true); else SLAK_CF_REG(96, false); }
        else if (C == 128) { if (g2_bf16) SLAK_CF_REG(128, true); else SLAK_CF_REG(128, false); }
        else { if (g2_bf16) SLAK_CF_REG(192, true); else SLAK_CF_REG(192, false); }
#undef SLAK_CF_REG
        SLAK_LAUNCH_CHECK();
        return reduce_partials(part, part + (size_t)nwg * 2 * C, dweight, dbias, C, nwg, 2 * C, (hipStream_t)stream);
    }
    const bool csplit = cf_block(N, P) == 64;                      // few pixels per image: 64-pixel tiles, waves split the channels
    const int tpi = csplit ? (P + 63) / 64 : cf_tiles(N, P), nwg = N * tpi;
    const size_t lds = (size_t)(BT_THREADS / 16) * 2 * C * 4 + 8 * 64 * 4 + 16;
    const dim3 grid((unsigned)nwg);
    float* part = (float*)workspace;
#define SLAK_CF_BWD(TI, TG)                                                                                                   \
    do { if (csplit) { if (set_lds((const void*)ln_cf_bwd_kernel<TI, TG, true>, lds)) return SLAK_ERR_LAUNCH;                 \
                       hipLaunchKernelGGL((ln_cf_bwd_kernel<TI, TG, true>), grid, dim3(BT_THREADS), lds, (hipStream_t)stream, (const TG*)g, \
                                          (const TI*)x, weight, mean, rstd, (TI*)dx, part, C, P, tpi); }                       \
         else { if (set_lds((const void*)ln_cf_bwd_kernel<TI, TG, false>, lds)) return SLAK_ERR_LAUNCH;                        \
                hipLaunchKernelGGL((ln_cf_bwd_kernel<TI, TG, false>), grid, dim3(BT_THREADS), lds, (hipStream_t)stream, (const TG*)g, \
                                   (const TI*)x, weight, mean, rstd, (TI*)dx, part, C, P, tpi); } } while (0)
    if (x_dtype == SLAK_F32 && g_dtype == SLAK_F32) SLAK_CF_BWD(float, float);
    else if (x_dtype == SLAK_F32 && g_dtype == SLAK_BF16) SLAK_CF_BWD(float, bf16_t);
    else if (x_dtype == SLAK_BF16 && g_dtype == SLAK_F32) SLAK_CF_BWD(bf16_t, float);
    else if (x_dtype == SLAK_BF16 && g_dtype == SLAK_BF16) SLAK_CF_BWD(bf16_t, bf16_t);
    else return SLAK_ERR_UNSUPPORTED;
#undef SLAK_CF_BWD
    SLAK_LAUNCH_CHECK();
    return reduce_partials(part, part + (size_t)nwg * 2 * C, dweight, dbias, C, nwg, 2 * C, (hipStream_t)stream);
}

/* dy1 = dact * gelu'(y1), dbias[col] = sum_rows dy1 (fp32).  workspace >= slak_gelu_bwd_workspace_bytes(rows, cols). */
size_t slak_gelu_bwd_workspace_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return align_up(((size_t)2048 + BT_SLICES) * cols * sizeof(float), 256);
}
int slak_gelu_backward_bias(const void* dact, const void* y1, void* dy1, float* dbias, int rows, int cols,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!dact || !y1 || !dy1 || !dbias) return SLAK_ERR_INVALID_ARG;
    if (rows <= 0 || cols <= 0) return SLAK_ERR_INVALID_ARG;
    if (cols % 8) return SLAK_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < slak_gelu_bwd_workspace_bytes(rows, cols)) return SLAK_ERR_WORKSPACE;
    // enough workgroups to fill the chip, each a contiguous block of rows (>= 8 rows to amortise the partial row)
    static const int nwg_target = [] { const char* e = slak_dev_getenv("SLAK_GELU_NWG"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    int nwg = nwg_target;
    int rpw = (rows + nwg - 1) / nwg; if (rpw < 8) rpw = 8;
    nwg = (rows + rpw - 1) / rpw;
    float* part = (float*)workspace;
    const int cchunks = cols / 8;
    int threads = BT_THREADS; double best = -1.0;                  // workgroup size with the fewest idle lanes over the column passes
    for (int t = 64; t <= BT_THREADS; t += 64) {
        const int ccn_t = cchunks < t ? cchunks : t, passes = (cchunks + ccn_t - 1) / ccn_t;
        const double util = (double)cchunks * (t / ccn_t) / ((double)passes * t);   // useful thread slots / all thread slots
        if (util > best + 1e-9 || (util > best - 1e-9 && t > threads)) { best = util; threads = t; }
    }
    const int ccn = cchunks < threads ? cchunks : threads;
    const size_t lds = (size_t)GD_BYTES + (size_t)(threads / ccn) * cols * sizeof(float) + 16;
    const float* table = gelu_grad_table_device();
    if (!table) return SLAK_ERR_LAUNCH;
    if (set_lds((const void*)gelu_bwd_bias_kernel, lds)) return SLAK_ERR_LAUNCH;
    hipLaunchKernelGGL(gelu_bwd_bias_kernel, dim3((unsigned)nwg), dim3((unsigned)threads), lds, (hipStream_t)stream,
                       (const uint16_t*)dact, (const uint16_t*)y1, (uint16_t*)dy1, part, rows, cols, rpw, table);
    SLAK_LAUNCH_CHECK();
    return reduce_partials(part, part + (size_t)nwg * cols, dbias, dbias, cols, nwg, cols, (hipStream_t)stream);
}

}  // extern "C"
