// slak_amd/csrc/optim_ema.hip -- SURVEY.md 8f-3: the per-step work AROUND the mask step, batched over every tensor.
//
//   slak_adamw_step : torch.optim.AdamW (optim_factory.py:149-150) + Masking.apply_mask (sparse_core.py:316-333) + the bf16 copies the
//                     pointwise GEMMs read, as ONE launch over all parameters.  The reference runs the optimizer's multi-tensor
//                     kernels, then one elementwise multiply per masked tensor from a Python loop (~100 launches for SLaK-T).
//   slak_ema_update : ModelEma.update(model, mask) (model_sema.py:67-91) as ONE launch over all state-dict entries.  The
//                     reference loops over ~600 entries in Python with 4 (dense) to 11 (masked) elementwise kernels each.
//
// Both are pure streaming passes (HBM-bound: 28 B/element for AdamW without mask, 12-16 B/element for the EMA); a workgroup owns one
// 4096-element chunk of one tensor (table lookup by blockIdx), 16-byte accesses when every pointer of the tensor is 16-byte
// aligned.  The library is compiled with -ffp-contract=off: every product and sum below is rounded on its own, as in the chain of
// separate torch kernels each formula restates.
#include <vector>

#include "slak_common.h"

#define OE_HIPCHK(call)                                                             \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) { slak::set_last_hip_error(e__); return SLAK_ERR_LAUNCH; } \
    } while (0)

namespace slak {
constexpr int OE_THREADS = 256;
constexpr int OE_VEC = 4;
constexpr int OE_ITERS = 4;
constexpr int OE_BLOCK_ELEMS = OE_THREADS * OE_VEC * OE_ITERS;      // 4096 contiguous elements of one tensor

struct AdamwGroups { slak_adamw_group_t g[SLAK_ADAMW_MAX_GROUPS]; };
struct AdamwScalars { float decay_mul, omb1, beta2, omb2, bc2_sqrt, eps, neg_step_size; };

__device__ __forceinline__ float adamw_elem(float p, float g, float& m, float& v, const AdamwScalars& k) {
    p = p * k.decay_mul;                         // param.mul_(1 - lr * weight_decay)
    m = m + k.omb1 * (g - m);                    // exp_avg.lerp_(grad, 1 - beta1)      (weight < 0.5 branch of lerp)
    v = v * k.beta2;                             // exp_avg_sq.mul_(beta2)
    v = v + (k.omb2 * g) * g;                    //           .addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;      // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    return p + k.neg_step_size * (m / denom);    // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(OE_THREADS) void adamw_kernel(const slak_adamw_segment_t* __restrict__ segs, const int* __restrict__ blk_seg,
                                                           const int* __restrict__ seg_blk0, const float* const* __restrict__ grads,
                                                           const AdamwGroups G) {
    const int s = blk_seg[blockIdx.x];
    const slak_adamw_segment_t sg = segs[s];
    const float* __restrict__ g = grads[s];
    if (!g) return;                                                       // no gradient this step: torch skips the parameter
    __shared__ AdamwScalars sh;
    if (threadIdx.x == 0) {
        const slak_adamw_group_t h = G.g[__builtin_amdgcn_readfirstlane(sg.group)];
        const double step = (double)*sg.step;                            // already incremented by the caller
        // Python evaluates these in double and hands them to the kernels as scalars (torch/optim/adamw.py, _single_tensor_adamw)
        const double bc1 = 1.0 - pow(h.beta1, step), bc2 = 1.0 - pow(h.beta2, step);
        sh.decay_mul = (float)(1.0 - h.lr * h.weight_decay);
        sh.omb1 = (float)(1.0 - h.beta1);
        sh.beta2 = (float)h.beta2;
        sh.omb2 = (float)(1.0 - h.beta2);
        sh.bc2_sqrt = (float)sqrt(bc2);
        sh.eps = (float)h.eps;
        sh.neg_step_size = (float)(-(h.lr / bc1));
    }
    __syncthreads();
    const AdamwScalars k = sh;
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * OE_BLOCK_ELEMS;
    float* __restrict__ P = sg.param; float* __restrict__ M = sg.exp_avg; float* __restrict__ V = sg.exp_avg_sq;
    const float* __restrict__ K = sg.mask; uint16_t* __restrict__ B = (uint16_t*)sg.param_bf16;
    const uintptr_t al = (uintptr_t)P | (uintptr_t)M | (uintptr_t)V | (uintptr_t)g | (uintptr_t)K | ((uintptr_t)B << 1);
    if ((al & 15) == 0 && base + OE_BLOCK_ELEMS <= sg.numel) {
#pragma unroll
        for (int it = 0; it < OE_ITERS; ++it) {
            const long long i = base + (long long)(it * OE_THREADS + threadIdx.x) * OE_VEC;
            float4 p = *(const float4*)(P + i), m = *(const float4*)(M + i), v = *(const float4*)(V + i);
            const float4 gr = *(const float4*)(g + i);
            p.x = adamw_elem(p.x, gr.x, m.x, v.x, k); p.y = adamw_elem(p.y, gr.y, m.y, v.y, k);
            p.z = adamw_elem(p.z, gr.z, m.z, v.z, k); p.w = adamw_elem(p.w, gr.w, m.w, v.w, k);
            if (K) { const float4 mk = *(const float4*)(K + i); p.x = p.x * mk.x; p.y = p.y * mk.y; p.z = p.z * mk.z; p.w = p.w * mk.w; }
            *(float4*)(P + i) = p; *(float4*)(M + i) = m; *(float4*)(V + i) = v;
            if (B) {
                uint2 o;
                o.x = (uint32_t)f32_to_bf16_bits(p.x) | ((uint32_t)f32_to_bf16_bits(p.y) << 16);
                o.y = (uint32_t)f32_to_bf16_bits(p.z) | ((uint32_t)f32_to_bf16_bits(p.w) << 16);
                *(uint2*)(B + i) = o;
            }
        }
    } else {
        for (int j = 0; j < OE_VEC * OE_ITERS; ++j) {
            const long long i = base + j * OE_THREADS + threadIdx.x;
            if (i < sg.numel) {
                float m = M[i], v = V[i];
                float p = adamw_elem(P[i], g[i], m, v, k);
                if (K) p = p * K[i];
                P[i] = p; M[i] = m; V[i] = v;
                if (B) B[i] = f32_to_bf16_bits(p);
            }
        }
    }
}

// ---- EMA ------------------------------------------------------------------------------------------------------------------------
// dense entry   : ema = ema * decay + (1 - decay) * model                                              model_sema.py:81, :91
// masked entry  : diff = ((ema != 0).byte() ^ mask.byte()) & mask.byte()       -- weights the mask has (re)grown since the last update
//                 ema = (ema * decay + model * (1 - decay)) * mask + (diff * decay) * model            model_sema.py:83-89
// int64 entries (BatchNorm num_batches_tracked): the products are float32 (an integer tensor times a Python float), the copy_
// back truncates.
__device__ __forceinline__ float ema_elem(float e, float w, float decay, float omd) { return e * decay + omd * w; }
__device__ __forceinline__ float ema_elem_masked(float e, float w, float mk, float decay, float omd) {
    const unsigned char mb = (unsigned char)mk, nz = e != 0.0f ? 1 : 0;
    const unsigned char diff = (unsigned char)((nz ^ mb) & mb);
    return (e * decay + w * omd) * mk + ((float)diff * decay) * w;
}

__global__ __launch_bounds__(OE_THREADS) void ema_kernel(const slak_ema_segment_t* __restrict__ segs, const int* __restrict__ blk_seg,
                                                         const int* __restrict__ seg_blk0, float decay, float omd) {
    const int s = blk_seg[blockIdx.x];
    const slak_ema_segment_t sg = segs[s];
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * OE_BLOCK_ELEMS;
    if (sg.dtype == SLAK_I64) {
        long long* __restrict__ E = (long long*)sg.ema; const long long* __restrict__ W = (const long long*)sg.model;
        for (int j = 0; j < OE_VEC * OE_ITERS; ++j) {
            const long long i = base + j * OE_THREADS + threadIdx.x;
            if (i < sg.numel) E[i] = (long long)ema_elem((float)E[i], (float)W[i], decay, omd);
        }
        return;
    }
    float* __restrict__ E = (float*)sg.ema; const float* __restrict__ W = (const float*)sg.model; const float* __restrict__ K = sg.mask;
    const uintptr_t al = (uintptr_t)E | (uintptr_t)W | (uintptr_t)K;
    if ((al & 15) == 0 && base + OE_BLOCK_ELEMS <= sg.numel) {
#pragma unroll
        for (int it = 0; it < OE_ITERS; ++it) {
            const long long i = base + (long long)(it * OE_THREADS + threadIdx.x) * OE_VEC;
            float4 e = *(const float4*)(E + i); const float4 w = *(const float4*)(W + i);
            if (K) {
                const float4 mk = *(const float4*)(K + i);
                e.x = ema_elem_masked(e.x, w.x, mk.x, decay, omd); e.y = ema_elem_masked(e.y, w.y, mk.y, decay, omd);
                e.z = ema_elem_masked(e.z, w.z, mk.z, decay, omd); e.w = ema_elem_masked(e.w, w.w, mk.w, decay, omd);
            } else {
                e.x = ema_elem(e.x, w.x, decay, omd); e.y = ema_elem(e.y, w.y, decay, omd);
                e.z = ema_elem(e.z, w.z, decay, omd); e.w = ema_elem(e.w, w.w, decay, omd);
            }
            *(float4*)(E + i) = e;
        }
    } else {
        for (int j = 0; j < OE_VEC * OE_ITERS; ++j) {
            const long long i = base + j * OE_THREADS + threadIdx.x;
            if (i < sg.numel) E[i] = K ? ema_elem_masked(E[i], W[i], K[i], decay, omd) : ema_elem(E[i], W[i], decay, omd);
        }
    }
}

// chunk table shared by both plans: blk_seg[b] = tensor of chunk b, seg_blk0[s] = first chunk of tensor s
struct ChunkTable {
    int nblk = 0;
    int* blk_seg = nullptr;
    int* seg_blk0 = nullptr;
    int build(const std::vector<long long>& numel) {
        std::vector<int> bs, b0(numel.size() + 1);
        for (size_t s = 0; s < numel.size(); ++s) {
            b0[s] = (int)bs.size();
            const long long nb = (numel[s] + OE_BLOCK_ELEMS - 1) / OE_BLOCK_ELEMS;
            if ((long long)bs.size() + nb > 0x7fffffffLL) return SLAK_ERR_UNSUPPORTED;
            for (long long b = 0; b < nb; ++b) bs.push_back((int)s);
        }
        b0[numel.size()] = (int)bs.size();
        nblk = (int)bs.size();
        if (nblk == 0) return SLAK_OK;
        OE_HIPCHK(hipMalloc((void**)&blk_seg, sizeof(int) * nblk));
        OE_HIPCHK(hipMalloc((void**)&seg_blk0, sizeof(int) * b0.size()));
        OE_HIPCHK(hipMemcpy(blk_seg, bs.data(), sizeof(int) * nblk, hipMemcpyHostToDevice));
        OE_HIPCHK(hipMemcpy(seg_blk0, b0.data(), sizeof(int) * b0.size(), hipMemcpyHostToDevice));
        return SLAK_OK;
    }
    void release() { if (blk_seg) (void)hipFree(blk_seg); if (seg_blk0) (void)hipFree(seg_blk0); blk_seg = seg_blk0 = nullptr; }
};
}  // namespace slak

using namespace slak;

struct slak_adamw_plan { int nseg = 0; slak_adamw_segment_t* segs = nullptr; ChunkTable tab; };
struct slak_ema_plan { int nseg = 0; slak_ema_segment_t* segs = nullptr; ChunkTable tab; };

extern "C" {

int slak_adamw_plan_create(const slak_adamw_segment_t* segs_host, int nseg, slak_adamw_plan_t** plan_out) {
    if (!segs_host || nseg <= 0 || !plan_out) return SLAK_ERR_INVALID_ARG;
    std::vector<long long> numel(nseg);
    for (int s = 0; s < nseg; ++s) {
        const slak_adamw_segment_t& g = segs_host[s];
        if (!g.param || !g.exp_avg || !g.exp_avg_sq || !g.step || g.numel < 0 || g.group < 0 || g.group >= SLAK_ADAMW_MAX_GROUPS)
            return SLAK_ERR_INVALID_ARG;
        numel[s] = g.numel;
    }
    slak_adamw_plan* p = new slak_adamw_plan();
    p->nseg = nseg;
    int rc = p->tab.build(numel);
    if (rc != SLAK_OK) { delete p; return rc; }
    if (hipMalloc((void**)&p->segs, sizeof(slak_adamw_segment_t) * nseg) != hipSuccess ||
        hipMemcpy(p->segs, segs_host, sizeof(slak_adamw_segment_t) * nseg, hipMemcpyHostToDevice) != hipSuccess) {
        slak::set_last_hip_error(hipGetLastError()); p->tab.release(); delete p; return SLAK_ERR_LAUNCH;
    }
    *plan_out = p;
    return SLAK_OK;
}

int slak_adamw_step(slak_adamw_plan_t* p, const void* const* grads_dev, const slak_adamw_group_t* groups_host, int ngroups, void* stream) {
    if (!p || !grads_dev || !groups_host || ngroups <= 0 || ngroups > SLAK_ADAMW_MAX_GROUPS) return SLAK_ERR_INVALID_ARG;
    if (p->tab.nblk == 0) return SLAK_OK;
    AdamwGroups G;
    for (int i = 0; i < SLAK_ADAMW_MAX_GROUPS; ++i) G.g[i] = groups_host[i < ngroups ? i : 0];
    hipLaunchKernelGGL(adamw_kernel, dim3(p->tab.nblk), dim3(OE_THREADS), 0, (hipStream_t)stream, p->segs, p->tab.blk_seg, p->tab.seg_blk0,
                       (const float* const*)grads_dev, G);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_adamw_plan_destroy(slak_adamw_plan_t* p) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    if (p->segs) (void)hipFree(p->segs);
    p->tab.release();
    delete p;
    return SLAK_OK;
}

int slak_ema_plan_create(const slak_ema_segment_t* segs_host, int nseg, slak_ema_plan_t** plan_out) {
    if (!segs_host || nseg <= 0 || !plan_out) return SLAK_ERR_INVALID_ARG;
    std::vector<long long> numel(nseg);
    for (int s = 0; s < nseg; ++s) {
        const slak_ema_segment_t& g = segs_host[s];
        if (g.numel < 0 || (g.numel > 0 && (!g.ema || !g.model))) return SLAK_ERR_INVALID_ARG;
        if (g.dtype != SLAK_F32 && g.dtype != SLAK_I64) return SLAK_ERR_UNSUPPORTED;
        if (g.dtype == SLAK_I64 && g.mask) return SLAK_ERR_UNSUPPORTED;
        numel[s] = g.numel;
    }
    slak_ema_plan* p = new slak_ema_plan();
    p->nseg = nseg;
    int rc = p->tab.build(numel);
    if (rc != SLAK_OK) { delete p; return rc; }
    if (hipMalloc((void**)&p->segs, sizeof(slak_ema_segment_t) * nseg) != hipSuccess ||
        hipMemcpy(p->segs, segs_host, sizeof(slak_ema_segment_t) * nseg, hipMemcpyHostToDevice) != hipSuccess) {
        slak::set_last_hip_error(hipGetLastError()); p->tab.release(); delete p; return SLAK_ERR_LAUNCH;
    }
    *plan_out = p;
    return SLAK_OK;
}

int slak_ema_update(slak_ema_plan_t* p, double decay, void* stream) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    if (p->tab.nblk == 0) return SLAK_OK;
    // the scalars reach torch's kernels as float(decay) and float(1. - decay), the latter evaluated in double by Python
    hipLaunchKernelGGL(ema_kernel, dim3(p->tab.nblk), dim3(OE_THREADS), 0, (hipStream_t)stream, p->segs, p->tab.blk_seg, p->tab.seg_blk0,
                       (float)decay, (float)(1.0 - decay));
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_ema_plan_destroy(slak_ema_plan_t* p) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    if (p->segs) (void)hipFree(p->segs);
    p->tab.release();
    delete p;
    return SLAK_OK;
}

}  // extern "C"
