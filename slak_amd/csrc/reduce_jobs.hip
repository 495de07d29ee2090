// slak_amd/csrc/reduce_jobs.hip -- the fixed-order column sums that END a block's backward passes, several in ONE launch.
// Every streaming kernel of the block tail leaves its parameter gradients (LayerNorm weight / bias, layer-scale gamma, pwconv biases:
// models/SLaK.py:153-166 backwards) as per-workgroup partial rows, and the pointwise weight gradients leave per-slab partial tiles; a small
// kernel then adds the rows in a fixed order (deterministic, no atomics).  Per block that is five launches of 5-8 us each whose results
// nothing in the block's backward reads: 97 launches and ~0.6 ms of a SLaK-T step, each alone on the GPU.  Between
// slak_defer_reductions_begin() and slak_defer_reductions_end(stream) on a thread those launches are RECORDED instead (the partial rows
// stay where the call put them: the caller hands every call its own workspace until _end), and _end adds all of them in one launch --
// the same additions in the same order as the stand-alone kernels (block_tail.hip: block_tail_reduce1, linear_wgrad.hip:
// linear_wgrad_reduce_kernel / _few_kernel), so the same bits.
#include "slak_common.h"
#include "../../include/slak_hip.h"

namespace slak {

constexpr int RJ_MAX = 8;
struct ReduceJob {
    const float* part; float* out0; float* out1;
    int split, ntiles, width;            // type 0: columns `width`, rows `ntiles`; types 1, 2: width = float4 elements, ntiles = slabs
    int type;                            // 0: block_tail_reduce1, 1: linear_wgrad_reduce_kernel, 2: linear_wgrad_reduce_few_kernel
    unsigned wg_begin;                   // first workgroup of the job in the merged grid
};
struct ReduceJobs { ReduceJob j[RJ_MAX]; int n; };

// One 1024-thread workgroup = what ONE workgroup of block_tail_reduce1 (type 0) or FOUR workgroups of the 256-thread slab kernels do.
__global__ __launch_bounds__(1024) void reduce_jobs_kernel(const ReduceJobs jobs) {
    __shared__ float4 smem[4 * 16 * 16 + 16];
    int ji = 0;
#pragma unroll
    for (int k = 1; k < RJ_MAX; ++k) if (k < jobs.n && blockIdx.x >= jobs.j[k].wg_begin) ji = k;
    const ReduceJob& J = jobs.j[ji];
    const unsigned wg = blockIdx.x - J.wg_begin;
    if (J.type == 0) {
        float (*acc)[33] = (float (*)[33])smem;                   // [32][33]
        const int cx = threadIdx.x & 31, g = threadIdx.x >> 5;
        const int j = (int)wg * 32 + cx;
        const int ntiles = J.ntiles, width = J.width;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (j < width) {
            const float* p = J.part + j;
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
            if (j < J.split) J.out0[j] = t; else J.out1[j - J.split] = t;
        }
    } else if (J.type == 1) {
        float4 (*red)[16][16] = (float4 (*)[16][16])smem;         // [4][16][16]
        const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
        const int e = tid & 15, g = tid >> 4;
        const int n4 = J.width, S = J.ntiles;
        const size_t stride4 = (size_t)n4;
        const int i = ((int)wg * 4 + sub) * 16 + e;
        float4 a = float4{0.f, 0.f, 0.f, 0.f};
        if (i < n4) {
            const float4* p = (const float4*)J.part + i;
            for (int s = g; s < S; s += 16) { const float4 v = p[(size_t)s * stride4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        }
        red[sub][g][e] = a;
        __syncthreads();
        if (g == 0 && i < n4) {
#pragma unroll
            for (int k = 1; k < 16; ++k) { const float4 v = red[sub][k][e]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            ((float4*)J.out0)[i] = a;
        }
    } else {
        const int n4 = J.width, S = J.ntiles;
        const size_t stride4 = (size_t)n4;
        const int i = (int)wg * 1024 + threadIdx.x;
        if (i < n4) {
            const float4* p = (const float4*)J.part + i;
            float4 a = p[0];
            for (int s = 1; s < S; ++s) { const float4 v = p[(size_t)s * stride4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            ((float4*)J.out0)[i] = a;
        }
    }
}

struct DeferState { bool on = false; ReduceJobs jobs{}; unsigned wgs = 0; hipStream_t st = nullptr; bool have_stream = false; int rc = SLAK_OK; };
static thread_local DeferState g_defer;

static int defer_flush() {
    DeferState& d = g_defer;
    if (d.jobs.n == 0) return SLAK_OK;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3(d.wgs), dim3(1024), 0, d.st, d.jobs);
    d.jobs.n = 0; d.wgs = 0;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_hip_error(e); return SLAK_ERR_LAUNCH; }
    return SLAK_OK;
}

// -> true: recorded (the caller launches nothing); false: not deferring (the caller launches its own kernel)
bool reduce_defer_push(int type, const float* part, float* out0, float* out1, int split, int ntiles, int width, hipStream_t st) {
    DeferState& d = g_defer;
    if (!d.on) return false;
    if (d.have_stream && st != d.st) return false;               // one launch = one stream (the first recorded call's): work on another stream reduces at once
    d.st = st; d.have_stream = true;
    if (d.jobs.n == RJ_MAX) { const int rc = defer_flush(); if (rc != SLAK_OK) d.rc = rc; }
    ReduceJob& J = d.jobs.j[d.jobs.n++];
    J.part = part; J.out0 = out0; J.out1 = out1; J.split = split; J.ntiles = ntiles; J.width = width; J.type = type; J.wg_begin = d.wgs;
    d.wgs += type == 0 ? (unsigned)((width + 31) / 32) : type == 1 ? (unsigned)((width + 63) / 64) : (unsigned)((width + 1023) / 1024);
    return true;
}

}  // namespace slak

using namespace slak;

extern "C" {

int slak_defer_reductions_begin(void) {
    DeferState& d = g_defer;
    if (d.on) return SLAK_ERR_INVALID_ARG;                       // not nestable
    d = DeferState{}; d.on = true;
    return SLAK_OK;
}

int slak_defer_reductions_end(void) {
    DeferState& d = g_defer;
    if (!d.on) return SLAK_ERR_INVALID_ARG;
    int rc = defer_flush();
    if (rc == SLAK_OK) rc = d.rc;
    d = DeferState{};
    return rc;
}

}  // extern "C"
