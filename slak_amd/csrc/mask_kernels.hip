// slak_amd/csrc/mask_kernels.hip -- the Masking prune / regrow / apply step, on device, batched over
// every masked tensor (one launch per phase for ALL tensors; the reference loops over tensors in Python
// with >= 3 host syncs and two full torch.sort calls per tensor: sparse_core.py:335-357).
//
//   apply  : w *= mask (+ SGD momentum)                                   sparse_core.py:316-333
//   prune  : nonzeros = sum(mask); zeros = numel - nonzeros;
//            num_remove = ceil(rate*nonzeros); k = ceil(zeros + num_remove);
//            num_remove == 0 ? mask = (w != 0)
//                            : mask[ k smallest |w| ] = 0                 funcs.py:107-114
//            removed = nonzeros - sum(new mask)                           sparse_core.py:345-346
//   regrow : key = |grad * (mask == 0)|; mask[ floor(removed) largest key ] = 1   funcs.py:196-205
//   then apply again                                                      sparse_core.py:357
//
// "k smallest" is an exact radix select on the IEEE bit pattern of the non-negative key (4 passes of 8
// bits, LDS histograms, one global histogram per tensor), followed by an index-ordered ranking of the
// elements EQUAL to the k-th key so that ties are taken lowest-flat-index first -- the behaviour of
// torch.sort(stable=True).  (The reference's plain torch.sort is arbitrary on ties)
// "k largest" is the same select on the bitwise complement of the key.
// The prune rate arrives as the host scheduler's fp64 value; ceil(rate*nonzeros) is evaluated on device in
// fp64 exactly as CPython evaluates math.ceil(prune_rate*name2nonzeros) -- no host round trip, masks
// never leave the device.  All bandwidth-bound: 4 B/elem per pass.
#include <vector>

#include "slak_common.h"

namespace slak {

constexpr int MK_THREADS = 256;
constexpr int MK_PER_THREAD = 8;
constexpr int MK_BLOCK_ELEMS = MK_THREADS * MK_PER_THREAD;      // 2048 contiguous elements of one tensor

enum { MODE_SELECT = 0, MODE_NONZERO = 1, MODE_NONE = 2 };
enum { KEY_ABS_W = 0, KEY_GRAD_DESC = 1 };

struct SegState {
    unsigned long long k;          // elements still to take (rank of the k-th key inside the current prefix)
    unsigned prefix;               // high bits of the k-th key found so far
    int mode;
    unsigned long long cnt_mask;   // sum(mask != 0)
    unsigned long long cnt_changed;// prune: elements that went 1 -> 0; grow: elements that went 0 -> 1
    unsigned long long removed;    // prune result, consumed by grow
    unsigned long long pad;
};

}  // namespace slak

struct slak_mask_plan {
    int nseg = 0;
    int nblk = 0;
    long long total = 0;
    std::vector<slak_mask_segment_t> segs_host;
    std::vector<int> seg_first_blk;       // host copy: first block of each segment (+ sentinel)
    slak_mask_segment_t* segs = nullptr;  // device
    int* blk_seg = nullptr;               // device [nblk]
    int* seg_blk0 = nullptr;              // device [nseg+1]
    unsigned* hist = nullptr;             // device [nseg][256]
    slak::SegState* state = nullptr;      // device [nseg]
    unsigned* blk_eq = nullptr;           // device [nblk] (# keys == threshold per block, then exclusive prefix)
    double* stats = nullptr;              // device [nseg][4]
    unsigned long long* checksum = nullptr;
};

namespace slak {

__device__ __forceinline__ unsigned key_of(int keymode, const slak_mask_segment_t& sg, long long i) {
    if (keymode == KEY_ABS_W) {
        return __float_as_uint(sg.weight[i]) & 0x7fffffffu;                       // |w|
    } else {
        const float gm = sg.grad[i] * ((sg.mask[i] == 0.0f) ? 1.0f : 0.0f);        // grad * (mask == 0)
        return ~(__float_as_uint(gm) & 0x7fffffffu);                               // descending |.|
    }
}

// ---- apply -------------------------------------------------------------------------------------
__global__ __launch_bounds__(MK_THREADS) void mask_apply_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                const int* __restrict__ blk_seg, const int* __restrict__ seg_blk0) {
    const int s = blk_seg[blockIdx.x];
    const slak_mask_segment_t sg = segs[s];
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * MK_BLOCK_ELEMS;
#pragma unroll
    for (int j = 0; j < MK_PER_THREAD; ++j) {
        const long long i = base + j * MK_THREADS + threadIdx.x;                    // coalesced
        if (i < sg.numel) {
            const float m = sg.mask[i];
            sg.weight[i] = sg.weight[i] * m;
            if (sg.momentum) sg.momentum[i] = sg.momentum[i] * m;
        }
    }
}

// ---- counts ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned block_sum(unsigned v, unsigned* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    unsigned t = 0;
    if (threadIdx.x == 0) for (int w = 0; w < MK_THREADS / 64; ++w) t += sh[w];
    return t;                                                                       // valid in thread 0
}

__global__ __launch_bounds__(MK_THREADS) void mask_count_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                const int* __restrict__ blk_seg, const int* __restrict__ seg_blk0,
                                                                SegState* __restrict__ state) {
    __shared__ unsigned sh[MK_THREADS / 64];
    const int s = blk_seg[blockIdx.x];
    const slak_mask_segment_t sg = segs[s];
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * MK_BLOCK_ELEMS;
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < MK_PER_THREAD; ++j) {
        const long long i = base + j * MK_THREADS + threadIdx.x;
        if (i < sg.numel) c += (sg.mask[i] != 0.0f) ? 1u : 0u;
    }
    const unsigned t = block_sum(c, sh);
    if (threadIdx.x == 0 && t) atomicAdd(&state[s].cnt_mask, (unsigned long long)t);
}

__global__ void mask_reset_state_kernel(SegState* state, unsigned* hist, int nseg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nseg) { SegState z; z.k = 0; z.prefix = 0; z.mode = MODE_NONE; z.cnt_mask = 0; z.cnt_changed = 0; z.removed = 0; z.pad = 0; state[i] = z; }
    if (i < nseg * 256) hist[i] = 0;
}

// one thread per tensor: the scalar arithmetic of funcs.py:107-109, in fp64 like CPython
__global__ void mask_setup_prune_kernel(const slak_mask_segment_t* __restrict__ segs, SegState* state, double* stats,
                                        int nseg, double prune_rate) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const double nonzeros = (double)state[s].cnt_mask;
    const double zeros = (double)segs[s].numel - nonzeros;
    const double num_remove = ceil(prune_rate * nonzeros);          // math.ceil(masking.prune_rate*name2nonzeros)
    const double k = ceil(zeros + num_remove);                      // math.ceil(num_zeros + num_remove)
    stats[4 * s + 0] = nonzeros; stats[4 * s + 1] = zeros;
    state[s].prefix = 0; state[s].cnt_changed = 0;
    if (num_remove == 0.0) { state[s].mode = MODE_NONZERO; state[s].k = 0; }
    else {
        double kk = k; if (kk > (double)segs[s].numel) kk = (double)segs[s].numel;   // idx[:k] saturates
        state[s].k = (unsigned long long)kk;
        state[s].mode = kk > 0 ? MODE_SELECT : MODE_NONE;
    }
}

__global__ void mask_setup_grow_kernel(SegState* state, double* stats, int nseg) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    // removed = name2nonzeros - new_mask.sum()  (sparse_core.py:345); for MODE_NONZERO cnt_changed holds
    // nonzeros - count(w != 0) computed by the select kernel.
    const unsigned long long removed = state[s].cnt_changed;
    state[s].removed = removed;
    stats[4 * s + 2] = (double)removed;
    state[s].k = removed;                                            // math.floor(removed) of an integer
    state[s].prefix = 0;
    state[s].cnt_changed = 0;
    state[s].mode = removed > 0 ? MODE_SELECT : MODE_NONE;
}

__global__ void mask_finish_kernel(SegState* state, double* stats, int nseg) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    stats[4 * s + 3] = stats[4 * s + 0] - (double)state[s].removed + (double)state[s].cnt_changed;
}

// ---- radix select ------------------------------------------------------------------------------
__global__ __launch_bounds__(MK_THREADS) void mask_hist_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                               const int* __restrict__ blk_seg, const int* __restrict__ seg_blk0,
                                                               const SegState* __restrict__ state, unsigned* __restrict__ hist,
                                                               int keymode, int shift) {
    __shared__ unsigned lh[256];
    const int s = blk_seg[blockIdx.x];
    if (state[s].mode != MODE_SELECT) return;                        // block-uniform
    lh[threadIdx.x] = 0;
    __syncthreads();
    const slak_mask_segment_t sg = segs[s];
    const unsigned prefix = state[s].prefix;
    const unsigned himask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * MK_BLOCK_ELEMS;
#pragma unroll
    for (int j = 0; j < MK_PER_THREAD; ++j) {
        const long long i = base + j * MK_THREADS + threadIdx.x;
        if (i < sg.numel) {
            const unsigned key = key_of(keymode, sg, i);
            if ((key & himask) == prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    const unsigned v = lh[threadIdx.x];
    if (v) atomicAdd(&hist[s * 256 + threadIdx.x], v);
}

// one wave per tensor: find the digit holding the k-th key, narrow the prefix, clear the histogram
__global__ __launch_bounds__(64) void mask_scan_kernel(SegState* state, unsigned* hist, int shift) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (state[s].mode != MODE_SELECT) return;
    unsigned h[4]; unsigned long long local = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = hist[s * 256 + lane * 4 + j]; local += h[j]; hist[s * 256 + lane * 4 + j] = 0; }
    unsigned long long incl = local;                                  // inclusive scan over lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    const unsigned long long excl = incl - local;
    const unsigned long long k = state[s].k;                           // 1-based rank of the wanted key
    if (k > excl && k <= incl) {                                       // exactly one lane
        unsigned long long run = excl; int d = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (k > run && k <= run + h[j]) { d = j; break; } run += h[j]; }
        state[s].prefix |= ((unsigned)(lane * 4 + d)) << shift;
        state[s].k = k - run;                                          // rank inside the chosen digit
    }
}

// per block: how many keys equal the threshold (for index-ordered tie ranking)
__global__ __launch_bounds__(MK_THREADS) void mask_eq_count_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                   const int* __restrict__ blk_seg, const int* __restrict__ seg_blk0,
                                                                   const SegState* __restrict__ state, unsigned* __restrict__ blk_eq, int keymode) {
    __shared__ unsigned sh[MK_THREADS / 64];
    const int s = blk_seg[blockIdx.x];
    if (state[s].mode != MODE_SELECT) { if (threadIdx.x == 0) blk_eq[blockIdx.x] = 0; return; }
    const slak_mask_segment_t sg = segs[s];
    const unsigned thr = state[s].prefix;
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * MK_BLOCK_ELEMS;
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < MK_PER_THREAD; ++j) {
        const long long i = base + (long long)threadIdx.x * MK_PER_THREAD + j;      // same element order as the select kernel
        if (i < sg.numel) c += (key_of(keymode, sg, i) == thr) ? 1u : 0u;
    }
    const unsigned t = block_sum(c, sh);
    if (threadIdx.x == 0) blk_eq[blockIdx.x] = t;
}

// one wave per tensor: exclusive prefix of blk_eq over the tensor's blocks (in index order)
__global__ __launch_bounds__(64) void mask_eq_scan_kernel(const int* __restrict__ seg_blk0, unsigned* blk_eq) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const int b0 = seg_blk0[s], b1 = seg_blk0[s + 1];
    unsigned carry = 0;
    for (int b = b0; b < b1; b += 64) {
        const int i = b + lane;
        const unsigned v = (i < b1) ? blk_eq[i] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { unsigned t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        if (i < b1) blk_eq[i] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
}

// final pass: decide membership, rewrite the mask, count changes
__global__ __launch_bounds__(MK_THREADS) void mask_select_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                 const int* __restrict__ blk_seg, const int* __restrict__ seg_blk0,
                                                                 SegState* __restrict__ state, const unsigned* __restrict__ blk_eq,
                                                                 int keymode) {
    __shared__ unsigned sh[MK_THREADS / 64];
    __shared__ unsigned wave_off[MK_THREADS / 64];
    const int s = blk_seg[blockIdx.x];
    const int mode = state[s].mode;
    if (mode == MODE_NONE) return;
    const slak_mask_segment_t sg = segs[s];
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * MK_BLOCK_ELEMS + (long long)threadIdx.x * MK_PER_THREAD;
    unsigned changed = 0;
    if (mode == MODE_NONZERO) {                                        // prune only: return weight.data != 0.0
#pragma unroll
        for (int j = 0; j < MK_PER_THREAD; ++j) {
            const long long i = base + j;
            if (i < sg.numel) {
                const float nm = (sg.weight[i] != 0.0f) ? 1.0f : 0.0f;
                const float om = sg.mask[i];
                // removed = nonzeros - sum(new): count (old != 0) - (new != 0) as a signed total via two counters
                changed += (om != 0.0f ? 1u : 0u) - (nm != 0.0f ? 1u : 0u);          // wraps mod 2^32; summed mod 2^64 below
                sg.mask[i] = nm;
            }
        }
        const unsigned t = block_sum(changed, sh);
        if (threadIdx.x == 0 && t) atomicAdd(&state[s].cnt_changed, (unsigned long long)(long long)(int)t);
        return;
    }
    const unsigned thr = state[s].prefix;
    const unsigned long long need = state[s].k;                        // how many of the == thr keys to take
    unsigned keys[MK_PER_THREAD]; unsigned eqc = 0;
#pragma unroll
    for (int j = 0; j < MK_PER_THREAD; ++j) {
        const long long i = base + j;
        keys[j] = (i < sg.numel) ? key_of(keymode, sg, i) : 0xffffffffu;
        if (i < sg.numel && keys[j] == thr) ++eqc;
    }
    // exclusive scan of eqc over the block's threads (thread order == element order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = eqc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { unsigned t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned run = 0; for (int w = 0; w < MK_THREADS / 64; ++w) { wave_off[w] = run; run += sh[w]; } }
    __syncthreads();
    unsigned long long rank = (unsigned long long)blk_eq[blockIdx.x] + wave_off[wave] + (incl - eqc);
    const float newval = (keymode == KEY_ABS_W) ? 0.0f : 1.0f;
#pragma unroll
    for (int j = 0; j < MK_PER_THREAD; ++j) {
        const long long i = base + j;
        if (i < sg.numel) {
            bool sel = keys[j] < thr;
            if (keys[j] == thr) { sel = rank < need; ++rank; }
            if (sel) {
                const float om = sg.mask[i];
                if (keymode == KEY_ABS_W) { if (om != 0.0f) ++changed; } else { if (om == 0.0f) ++changed; }
                sg.mask[i] = newval;
            }
        }
    }
    __syncthreads();
    const unsigned t = block_sum(changed, sh);
    if (threadIdx.x == 0 && t) atomicAdd(&state[s].cnt_changed, (unsigned long long)t);
}

__global__ __launch_bounds__(MK_THREADS) void mask_checksum_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                   const int* __restrict__ blk_seg, const int* __restrict__ seg_blk0,
                                                                   unsigned long long* out) {
    const int s = blk_seg[blockIdx.x];
    const slak_mask_segment_t sg = segs[s];
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * MK_BLOCK_ELEMS;
    unsigned long long acc = 0;
#pragma unroll
    for (int j = 0; j < MK_PER_THREAD; ++j) {
        const long long i = base + j * MK_THREADS + threadIdx.x;
        if (i < sg.numel && sg.mask[i] != 0.0f) {
            unsigned long long h = ((unsigned long long)(s + 1) << 40) ^ (unsigned long long)i;   // splitmix64 of (tensor, index)
            h += 0x9e3779b97f4a7c15ull; h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ull; h = (h ^ (h >> 27)) * 0x94d049bb133111ebull; h ^= h >> 31;
            acc += h;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_last_hip_error(e_); return SLAK_ERR_LAUNCH; } } while (0)

static int run_select(slak_mask_plan* p, int keymode, hipStream_t st) {
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(mask_hist_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, st, p->segs, p->blk_seg, p->seg_blk0, p->state, p->hist, keymode, shift);
        hipLaunchKernelGGL(mask_scan_kernel, dim3(p->nseg), dim3(64), 0, st, p->state, p->hist, shift);
    }
    hipLaunchKernelGGL(mask_eq_count_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, st, p->segs, p->blk_seg, p->seg_blk0, p->state, p->blk_eq, keymode);
    hipLaunchKernelGGL(mask_eq_scan_kernel, dim3(p->nseg), dim3(64), 0, st, p->seg_blk0, p->blk_eq);
    hipLaunchKernelGGL(mask_select_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, st, p->segs, p->blk_seg, p->seg_blk0, p->state, p->blk_eq, keymode);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

}  // namespace slak

using namespace slak;

extern "C" {

int slak_mask_plan_create(const slak_mask_segment_t* segs_host, int nseg, slak_mask_plan_t** plan_out) {
    if (!segs_host || nseg <= 0 || !plan_out) return SLAK_ERR_INVALID_ARG;
    slak_mask_plan* p = new slak_mask_plan();
    p->nseg = nseg;
    p->segs_host.assign(segs_host, segs_host + nseg);
    std::vector<int> blk_seg;
    p->seg_first_blk.resize(nseg + 1);
    for (int s = 0; s < nseg; ++s) {
        if (!segs_host[s].weight || !segs_host[s].mask || segs_host[s].numel <= 0) { delete p; return SLAK_ERR_INVALID_ARG; }
        p->seg_first_blk[s] = (int)blk_seg.size();
        const long long nb = (segs_host[s].numel + MK_BLOCK_ELEMS - 1) / MK_BLOCK_ELEMS;
        for (long long b = 0; b < nb; ++b) blk_seg.push_back(s);
        p->total += segs_host[s].numel;
    }
    p->seg_first_blk[nseg] = (int)blk_seg.size();
    p->nblk = (int)blk_seg.size();
#define ALLOC(ptr, bytes) HIPCHK(hipMalloc((void**)&(ptr), (bytes)))
    ALLOC(p->segs, sizeof(slak_mask_segment_t) * nseg);
    ALLOC(p->blk_seg, sizeof(int) * p->nblk);
    ALLOC(p->seg_blk0, sizeof(int) * (nseg + 1));
    ALLOC(p->hist, sizeof(unsigned) * 256 * nseg);
    ALLOC(p->state, sizeof(SegState) * nseg);
    ALLOC(p->blk_eq, sizeof(unsigned) * p->nblk);
    ALLOC(p->stats, sizeof(double) * 4 * nseg);
    ALLOC(p->checksum, sizeof(unsigned long long));
#undef ALLOC
    HIPCHK(hipMemcpy(p->segs, p->segs_host.data(), sizeof(slak_mask_segment_t) * nseg, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->blk_seg, blk_seg.data(), sizeof(int) * p->nblk, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->seg_blk0, p->seg_first_blk.data(), sizeof(int) * (nseg + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(p->stats, 0, sizeof(double) * 4 * nseg));
    *plan_out = p;
    return SLAK_OK;
}

static int upload_segs(slak_mask_plan* p, hipStream_t st) {
    HIPCHK(hipMemcpyAsync(p->segs, p->segs_host.data(), sizeof(slak_mask_segment_t) * p->nseg, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));      // segs_host is pageable; keep the API simple and safe
    return SLAK_OK;
}

int slak_mask_plan_set_grads(slak_mask_plan_t* p, const void* const* grads_host, void* stream) {
    if (!p || !grads_host) return SLAK_ERR_INVALID_ARG;
    for (int s = 0; s < p->nseg; ++s) p->segs_host[s].grad = (const float*)grads_host[s];
    return upload_segs(p, (hipStream_t)stream);
}

int slak_mask_plan_set_momentum(slak_mask_plan_t* p, void* const* momentum_host, void* stream) {
    if (!p || !momentum_host) return SLAK_ERR_INVALID_ARG;
    for (int s = 0; s < p->nseg; ++s) p->segs_host[s].momentum = (float*)momentum_host[s];
    return upload_segs(p, (hipStream_t)stream);
}

int slak_mask_plan_destroy(slak_mask_plan_t* p) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    hipFree(p->segs); hipFree(p->blk_seg); hipFree(p->seg_blk0); hipFree(p->hist); hipFree(p->state);
    hipFree(p->blk_eq); hipFree(p->stats); hipFree(p->checksum);
    delete p;
    return SLAK_OK;
}

int slak_mask_apply(slak_mask_plan_t* p, void* stream) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(mask_apply_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, (hipStream_t)stream, p->segs, p->blk_seg, p->seg_blk0);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_mask_prune_and_grow(slak_mask_plan_t* p, double prune_rate, void* stream) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    for (int s = 0; s < p->nseg; ++s) if (!p->segs_host[s].grad) return SLAK_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int sg = ceil_div(p->nseg, 64), sgh = ceil_div(p->nseg * 256, 256);
    hipLaunchKernelGGL(mask_reset_state_kernel, dim3(sgh), dim3(256), 0, st, p->state, p->hist, p->nseg);
    hipLaunchKernelGGL(mask_count_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, st, p->segs, p->blk_seg, p->seg_blk0, p->state);
    hipLaunchKernelGGL(mask_setup_prune_kernel, dim3(sg), dim3(64), 0, st, p->segs, p->state, p->stats, p->nseg, prune_rate);
    int rc = run_select(p, KEY_ABS_W, st);                 // prune loop, sparse_core.py:337-347
    if (rc != SLAK_OK) return rc;
    hipLaunchKernelGGL(mask_setup_grow_kernel, dim3(sg), dim3(64), 0, st, p->state, p->stats, p->nseg);
    rc = run_select(p, KEY_GRAD_DESC, st);                 // growth loop, sparse_core.py:349-355
    if (rc != SLAK_OK) return rc;
    hipLaunchKernelGGL(mask_finish_kernel, dim3(sg), dim3(64), 0, st, p->state, p->stats, p->nseg);
    SLAK_LAUNCH_CHECK();
    return slak_mask_apply(p, stream);                      // sparse_core.py:357
}

int slak_mask_prune(slak_mask_plan_t* p, double prune_rate, void* stream) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int sg = ceil_div(p->nseg, 64), sgh = ceil_div(p->nseg * 256, 256);
    hipLaunchKernelGGL(mask_reset_state_kernel, dim3(sgh), dim3(256), 0, st, p->state, p->hist, p->nseg);
    hipLaunchKernelGGL(mask_count_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, st, p->segs, p->blk_seg, p->seg_blk0, p->state);
    hipLaunchKernelGGL(mask_setup_prune_kernel, dim3(sg), dim3(64), 0, st, p->segs, p->state, p->stats, p->nseg, prune_rate);
    int rc = run_select(p, KEY_ABS_W, st);                 // prune loop, sparse_core.py:337-347
    if (rc != SLAK_OK) return rc;
    hipLaunchKernelGGL(mask_setup_grow_kernel, dim3(sg), dim3(64), 0, st, p->state, p->stats, p->nseg);   // records `removed` (stats[2])
    hipLaunchKernelGGL(mask_finish_kernel, dim3(sg), dim3(64), 0, st, p->state, p->stats, p->nseg);       // stats[3] = nonzeros - removed
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_mask_read_stats(slak_mask_plan_t* p, double* out_host, void* stream) {
    if (!p || !out_host) return SLAK_ERR_INVALID_ARG;
    HIPCHK(hipMemcpyAsync(out_host, p->stats, sizeof(double) * 4 * p->nseg, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return SLAK_OK;
}

int slak_mask_checksum(slak_mask_plan_t* p, unsigned long long* out_host, void* stream) {
    if (!p || !out_host) return SLAK_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(p->checksum, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(mask_checksum_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, st, p->segs, p->blk_seg, p->seg_blk0, p->checksum);
    SLAK_LAUNCH_CHECK();
    HIPCHK(hipMemcpyAsync(out_host, p->checksum, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SLAK_OK;
}

}  // extern "C"
