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

}  // extern "C"
