// slak_amd/csrc/transpose_batch.hip -- the transposed bf16 copies of the pointwise weights that the data-gradient GEMMs read (models/SLaK.py:158-160
// backwards: dz . W2 and dy1 . W1 take the nn.Linear weights with the REDUCTION index contiguous, i.e. transposed), for ALL weights of a model in ONE launch.
// Autograd's path makes them per block and step with a strided copy kernel each (25 launches of ~14 us per SLaK-T step, round-5 profile); weights change
// once per optimizer step, so slak_amd/block_ops.py caches the transposes by the parameter's version counter and refreshes the stale ones here.
#include "slak_common.h"

namespace slak {

constexpr int TRB_MAX = 64;
struct TrbJobs { const uint16_t* src[TRB_MAX]; uint16_t* dst[TRB_MAX]; int rows[TRB_MAX], cols[TRB_MAX], tile0[TRB_MAX + 1]; int n; };

// dst[c][r] = src[r][c] on 64 x 64 tiles through LDS: 16-byte accesses on both sides (rows and cols multiples of 8)
__global__ __launch_bounds__(256) void transpose_batch_kernel(const TrbJobs jobs) {
    __shared__ uint16_t tile[64][64 + 2];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.tile0[j + 1]) ++j;
    const int rows = jobs.rows[j], cols = jobs.cols[j];
    const int tcs = (cols + 63) / 64, tl = (int)blockIdx.x - jobs.tile0[j];
    const int r0 = (tl / tcs) * 64, c0 = (tl % tcs) * 64;
    const uint16_t* __restrict__ src = jobs.src[j];
    uint16_t* __restrict__ dst = jobs.dst[j];
    const int ty = threadIdx.x >> 3, tx = threadIdx.x & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = r0 + ty + 32 * h, c = c0 + tx * 8;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (r < rows && c < cols) v = *(const uint4*)(src + (size_t)r * cols + c);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { tile[ty + 32 * h][tx * 8 + 2 * e] = (uint16_t)(w[e] & 0xffffu); tile[ty + 32 * h][tx * 8 + 2 * e + 1] = (uint16_t)(w[e] >> 16); }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = c0 + ty + 32 * h, r = r0 + tx * 8;                      // output row c, eight consecutive source rows
        if (c < cols && r < rows) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (unsigned)tile[tx * 8 + 2 * e][ty + 32 * h] | ((unsigned)tile[tx * 8 + 2 * e + 1][ty + 32 * h] << 16);
            *(uint4*)(dst + (size_t)c * rows + r) = uint4{w[0], w[1], w[2], w[3]};
        }
    }
}

// dst[n][p][c] = src[n][c][p]: the NCHW output gradient of a downsample convolution in pixel-major order (the row operand of its weight-gradient GEMM).
// 64 x 64 tiles through LDS; the loads run along p with the widest vector the row alignment allows (P % 8 == 0: 16 bytes, P % 4 == 0: 8 bytes, else
// 2 bytes -- the 7 x 7 maps), the stores along c with 16 bytes (C % 8 == 0).  torch's strided copy of this permutation runs at ~1.9 TB/s.
template <int VEC>
__global__ __launch_bounds__(256) void nchw_to_pixel_major_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int C, int P) {
    __shared__ uint16_t tile[64][64 + 2];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    src += (size_t)blockIdx.z * C * P; dst += (size_t)blockIdx.z * C * P;
    constexpr int CPR = 64 / VEC;                                  // chunks per tile row
    for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
        const int r = i / CPR, ch = i - r * CPR;
        const int c = c0 + r, p = p0 + ch * VEC;
        uint16_t v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = 0;
        if (c < C && p < P) {                                      // (P % VEC == 0: a chunk is inside the row or outside)
            if constexpr (VEC == 8) { const uint4 u = *(const uint4*)(src + (size_t)c * P + p); const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] = (uint16_t)(w[e] & 0xffffu); v[2 * e + 1] = (uint16_t)(w[e] >> 16); } }
            else if constexpr (VEC == 4) { const uint2 u = *(const uint2*)(src + (size_t)c * P + p);
                v[0] = (uint16_t)(u.x & 0xffffu); v[1] = (uint16_t)(u.x >> 16); v[2] = (uint16_t)(u.y & 0xffffu); v[3] = (uint16_t)(u.y >> 16); }
            else v[0] = src[(size_t)c * P + p];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) tile[r][ch * VEC + e] = v[e];
    }
    __syncthreads();
    const int ty = threadIdx.x >> 3, tx = threadIdx.x & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = p0 + ty + 32 * h, c = c0 + tx * 8;          // output row p, eight consecutive channels
        if (p < P && c < C) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (unsigned)tile[tx * 8 + 2 * e][ty + 32 * h] | ((unsigned)tile[tx * 8 + 2 * e + 1][ty + 32 * h] << 16);
            *(uint4*)(dst + (size_t)p * C + c) = uint4{w[0], w[1], w[2], w[3]};
        }
    }
}

}  // namespace slak

using namespace slak;

extern "C" {

int slak_transpose_bf16_batch(const void* const* src, void* const* dst, const int* rows, const int* cols, int n, void* stream) {
    if (n <= 0) return SLAK_OK;
    if (!src || !dst || !rows || !cols) return SLAK_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < n; b += TRB_MAX) {
        TrbJobs jobs;
        jobs.n = n - b < TRB_MAX ? n - b : TRB_MAX;
        int t = 0;
        for (int i = 0; i < jobs.n; ++i) {
            if (!src[b + i] || !dst[b + i] || rows[b + i] <= 0 || cols[b + i] <= 0 || rows[b + i] % 8 || cols[b + i] % 8) return SLAK_ERR_INVALID_ARG;
            jobs.src[i] = (const uint16_t*)src[b + i]; jobs.dst[i] = (uint16_t*)dst[b + i]; jobs.rows[i] = rows[b + i]; jobs.cols[i] = cols[b + i];
            jobs.tile0[i] = t;
            t += ((rows[b + i] + 63) / 64) * ((cols[b + i] + 63) / 64);
        }
        jobs.tile0[jobs.n] = t;
        hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)t), dim3(256), 0, st, jobs);
        SLAK_LAUNCH_CHECK();
    }
    return SLAK_OK;
}

int slak_nchw_to_pixel_major_bf16(const void* src, void* dst, int N, int C, int P, void* stream) {
    if (!src || !dst) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || C <= 0 || P <= 0) return SLAK_ERR_INVALID_ARG;
    if (C % 8 || N > 65535 || (C + 63) / 64 > 65535 || (long long)N * C * P >= (1LL << 40)) return SLAK_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((P + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)N);
    hipStream_t st = (hipStream_t)stream;
    if (P % 8 == 0) hipLaunchKernelGGL(nchw_to_pixel_major_kernel<8>, grid, dim3(256), 0, st, (const uint16_t*)src, (uint16_t*)dst, C, P);
    else if (P % 4 == 0) hipLaunchKernelGGL(nchw_to_pixel_major_kernel<4>, grid, dim3(256), 0, st, (const uint16_t*)src, (uint16_t*)dst, C, P);
    else hipLaunchKernelGGL(nchw_to_pixel_major_kernel<1>, grid, dim3(256), 0, st, (const uint16_t*)src, (uint16_t*)dst, C, P);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

}  // extern "C"
