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
// One truncate_weights() is FIVE passes over the tensors (round 2: 14 passes, 25 launches), 1 memset + 12 launches of which 5 are
// bookkeeping on a block per tensor / on the candidate lists:
//
//   P1  read w, mask        histogram of the top 11 bits of |w| (LDS, one global histogram per tensor) + sum(mask)
//   pick                    per tensor: k in fp64 exactly as CPython evaluates it, and the bin d1 that holds the k-th key
//   P2  read w              the keys of bin d1 -- a few per cent of the tensor -- are compacted, with their flat index,
//                           into a candidate list (keys that are exactly 0 are counted per block, never listed)
//   R   (candidates only)   one workgroup per tensor finishes the select on the list: 10 + 10 more key bits, then, when
//                           only some of the keys EQUAL to the k-th one are taken, the index of the last one taken
//                           (ties go lowest flat index first -- the behaviour of torch.sort(stable=True); the
//                           reference's plain torch.sort is arbitrary on ties)
//   P3  read w, mask, g     membership is  key < thr || (key == thr && index <= idx_thr); the pruned mask leaves as ONE BYTE PER FOUR
//                           ELEMENTS (`act`) and, in the same pass, the regrow key |g * (new mask == 0)| is histogrammed (its select
//                           is "k largest", done as "k smallest" of 0x7fffffff - key)
//   pick                    per tensor: removed, the regrow k and its bin
//   P4  read g, act         regrow candidates compacted; elements below the cut bin are marked in `grown` (same byte layout)
//   R   (candidates only)
//   mark (candidates only)  the listed candidates that made the cut are or-ed into `grown`
//   P5  read act, grown, w  final mask = act | grown written as exact 0.0 / 1.0, w *= mask (+ momentum): the apply of sparse_core.py:357
//       write mask, w       (g is read again only for a tensor whose cut fell among the zero keys: those are taken by index)
//   finish                  nonzeros after
//
// = 41 B/element against SURVEY 8d's fused ideal of 20 (a k-th element cannot be known before every key was seen once,
// so prune needs >= 2 reads of w and regrow >= 2 of g: 36 B is the floor of this family).  Masks are 0/1 tensors
// (sparse_core.py only ever writes 0.0 and 1.0 into them); prune-and-grow writes exactly those two values.  The prune rate arrives
// as the host scheduler's fp64 value; masks never leave the device.
#include <cstdint>
#include <cstring>
#include <vector>

#include "slak_common.h"

namespace slak {

constexpr int MK_THREADS = 256;
constexpr int MK_PER_THREAD = 8;
constexpr int MK_BLOCK_ELEMS = MK_THREADS * MK_PER_THREAD;      // 2048 contiguous elements of one tensor
constexpr int SB_ITERS = 16;                                     // histogram passes: 16 x 256 x float4 = 16384 elements per block
constexpr int SB_ELEMS = SB_ITERS * MK_THREADS * 4;              //   (a block flushes its non-empty bins with global atomics: fewer, larger blocks)
constexpr int D1_SHIFT = 20, D1_BINS = 2048;                     // top 11 bits of the 31-bit key (12 bits: longer flush, measured no faster)
constexpr int RF_THREADS = 1024;
constexpr int RF_BATCH = 8;                                      // list entries a refine thread has in flight
constexpr int CP_STAGE = 2048;                                   // candidate pairs a compaction block stages in LDS (16 KB)
constexpr unsigned KEY_MAX = 0x7fffffffu;
// The key that a large share of a tensor can hold exactly: |w| == 0 of the masked weights (prune), and |g * (mask == 0)| == 0 of
// every active weight and of the taps a small plane never reaches (regrow; as a "k smallest" key: KEY_MAX).  These keys are
// counted, never listed: when the cut falls among them the index of the last one taken is found from per-block counts.
__host__ __device__ constexpr unsigned special_key(int phase) { return phase == 0 ? 0u : KEY_MAX; }

enum { MODE_NONE = 0, MODE_SELECT = 1, MODE_NONZERO = 2 };       // MODE_NONE == 0: the per-call memset leaves every phase idle
enum { PH_PRUNE = 0, PH_GROW = 1 };

struct SelState {                  // one select (prune or regrow) of one tensor
    unsigned long long k;          // 1-based rank of the wanted key inside the keys not yet excluded
    unsigned long long cnt_changed;// prune: elements that went 1 -> 0 (MODE_NONZERO: old - new, signed); regrow: 0 -> 1
    unsigned thr;                  // the k-th key
    unsigned idx_thr;              // keys == thr are taken while index <= idx_thr
    unsigned d1;                   // top-11-bit bin of the k-th key
    unsigned cand;                 // number of compacted candidates
    unsigned pad[2];
    int mode;
    unsigned nspecial;             // keys exactly equal to the phase's special key (never compacted, see SPECIAL below)
};
struct SegState {
    SelState sel[2];
    unsigned long long cnt_mask;   // sum(mask != 0)
    unsigned long long removed;    // prune result, consumed by regrow
};

}  // namespace slak

struct slak_mask_plan {
    int nseg = 0;
    int nblk = 0;                         // 2048-element blocks
    int nsblk = 0;                        // 16384-element blocks (histogram passes)
    long long total = 0;
    std::vector<slak_mask_segment_t> segs_host;
    std::vector<int> seg_first_blk;       // host copy: first block of each segment (+ sentinel)
    slak_mask_segment_t* segs = nullptr;  // device
    slak_mask_segment_t* segs_pinned = nullptr;   // host staging of the descriptor table (async upload)
    hipEvent_t upload_done = nullptr;
    int* blk_seg = nullptr;               // device [nblk]
    int* seg_blk0 = nullptr;              // device [nseg+1]
    int* sblk_seg = nullptr;              // device [nsblk]
    int* seg_sblk0 = nullptr;             // device [nseg+1]
    long long* seg_off = nullptr;         // device [nseg]: first candidate slot of each segment
    void* zeroed = nullptr;               // device: hist[2][nseg][2048] u32, then SegState[nseg] -- one memset per call
    size_t zeroed_bytes = 0;
    unsigned* hist = nullptr;
    slak::SegState* state = nullptr;
    unsigned* blk_special = nullptr;      // device [nsblk]: keys == special key per 16384-element block (written when the cut bin holds it)
    uint2* cand = nullptr;                // device [total] (key, flat index); allocated by the first prune
    unsigned char* act = nullptr;         // device: one byte per four elements, bit e = (mask after the prune != 0); prune-and-grow only
    unsigned char* grown = nullptr;       // device: same layout, bit e = element is regrown
    long long* seg_nib = nullptr;         // device [nseg]: first byte of each tensor in act / grown (multiple of 4)
    size_t nib_bytes = 0;
    double* stats = nullptr;              // device [nseg][4]
    unsigned long long* checksum = nullptr;
};

namespace slak {

__device__ __forceinline__ unsigned key_w(float w) { return __float_as_uint(w) & KEY_MAX; }                 // |w|
__device__ __forceinline__ unsigned key_g(float g, float m) {
    const float gm = g * ((m == 0.0f) ? 1.0f : 0.0f);                                                       // grad * (mask == 0)
    return KEY_MAX - (__float_as_uint(gm) & KEY_MAX);                                                       // descending |.|
}
// four consecutive elements, zero-filled past the end (i % 4 == 0; a base that is not 16-byte aligned -- a view into a flat
// gradient bucket -- takes the scalar path)
__device__ __forceinline__ float4 load4(const float* __restrict__ p, long long i, long long n) {
    if (i + 3 < n && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) return *reinterpret_cast<const float4*>(p + i);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) v.x = p[i];
    if (i + 1 < n) v.y = p[i + 1];
    if (i + 2 < n) v.z = p[i + 2];
    if (i + 3 < n) v.w = p[i + 3];
    return v;
}
__device__ __forceinline__ void store4(float* __restrict__ p, long long i, long long n, float4 v) {
    if (i + 3 < n && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) { *reinterpret_cast<float4*>(p + i) = v; return; }
    if (i < n) p[i] = v.x;
    if (i + 1 < n) p[i + 1] = v.y;
    if (i + 2 < n) p[i + 2] = v.z;
    if (i + 3 < n) p[i + 3] = v.w;
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

// ---- block helpers -----------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ unsigned block_sum(unsigned v, unsigned* sh) {          // sh: NT/64 words; result valid in EVERY thread
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += sh[w];
    return t;
}

// exclusive prefix of v over the block's threads; *total = block sum (every thread)
template <int NT>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* sh, unsigned* total) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    __syncthreads();
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const unsigned x = sh[w]; if (w < wave) before += x; all += x; }
    *total = all;
    return before + incl - v;
}

// The bin of a histogram that holds the k-th smallest key (1-based k, 1 <= k <= sum of the bins).  `get(b)` reads bin b.
// Every thread owns NB/NT consecutive bins.  Results in res[0..2] (LDS): bin, rank inside the bin, count of the bin.
template <int NT, int NB, typename Get>
__device__ __forceinline__ void find_bin(Get get, unsigned long long k, unsigned* sh, unsigned* res) {
    constexpr int PER = NB / NT;
    static_assert(PER >= 1 && PER * NT == NB, "bins per thread");
    unsigned h[PER], local = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { h[j] = get(threadIdx.x * PER + j); local += h[j]; }
    unsigned total;
    const unsigned excl = block_excl_scan<NT>(local, sh, &total);
    if (threadIdx.x == 0) { res[0] = NB - 1; res[1] = 0; res[2] = 0; }               // k beyond the total: nothing below takes it
    __syncthreads();
    if (k > excl && k <= (unsigned long long)excl + local) {                           // exactly one thread
        unsigned run = excl;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (k > run && k <= (unsigned long long)run + h[j]) { res[0] = threadIdx.x * PER + j; res[1] = (unsigned)(k - run); res[2] = h[j]; }
            run += h[j];
        }
    }
    __syncthreads();
}

// Run-length aggregated LDS histogram update: neighbouring lanes that hit the same bin (ties, index-ordered candidate lists)
// issue ONE atomic.  `on` = this lane has a key.
__device__ __forceinline__ void hist_add_runs(unsigned* lh, unsigned bin, bool on) {
    const int lane = threadIdx.x & 63;
    const unsigned tag = on ? bin : 0xffffffffu;
    const unsigned prev = __shfl_up(tag, 1, 64);
    const bool head = (lane == 0) || (prev != tag);
    const unsigned long long heads = __ballot(head);
    if (head && on) {
        const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
        const int len = above ? (__ffsll((long long)above)) : (64 - lane);               // distance to the next head
        atomicAdd(&lh[bin], (unsigned)len);
    }
}

// ---- P1 / P3: histogram passes -----------------------------------------------------------------
__device__ __forceinline__ void flush_hist(const unsigned* lh, unsigned* gh, unsigned special_bin, unsigned special_count) {
#pragma unroll
    for (int j = 0; j < D1_BINS / MK_THREADS; ++j) {
        const int b = j * MK_THREADS + threadIdx.x;
        const unsigned v = lh[b];
        if (v) atomicAdd(&gh[b], v);
    }
    if (threadIdx.x == 0 && special_count) atomicAdd(&gh[special_bin], special_count);
}

// P1: |w| histogram + sum(mask)
__global__ __launch_bounds__(MK_THREADS) void mask_prune_hist_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                     const int* __restrict__ sblk_seg, const int* __restrict__ seg_sblk0,
                                                                     SegState* __restrict__ state, unsigned* __restrict__ hist) {
    __shared__ unsigned lh[D1_BINS];
    __shared__ unsigned sh[MK_THREADS / 64];
    const int s = sblk_seg[blockIdx.x];
    const slak_mask_segment_t sg = segs[s];
    const long long base = (long long)(blockIdx.x - seg_sblk0[s]) * SB_ELEMS;
#pragma unroll
    for (int j = 0; j < D1_BINS / MK_THREADS; ++j) lh[j * MK_THREADS + threadIdx.x] = 0;
    __syncthreads();
    unsigned zc = 0, on = 0;
#pragma unroll 4
    for (int it = 0; it < SB_ITERS; ++it) {
        const long long i = base + ((long long)it * MK_THREADS + threadIdx.x) * 4;
        if (i >= sg.numel) break;
        const float4 w = load4(sg.weight, i, sg.numel), m = load4(sg.mask, i, sg.numel);
        const float wv[4] = {w.x, w.y, w.z, w.w}, mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i + e < sg.numel) {
                const unsigned key = key_w(wv[e]);
                if (key == 0) ++zc; else atomicAdd(&lh[key >> D1_SHIFT], 1u);   // masked weights are exact zeros: counted in a register
                on += (mv[e] != 0.0f) ? 1u : 0u;
            }
        }
    }
    const unsigned zt = block_sum<MK_THREADS>(zc, sh);
    const unsigned ont = block_sum<MK_THREADS>(on, sh);
    unsigned* gh = hist + (size_t)s * D1_BINS;                        // hist[PH_PRUNE][s]; the regrow histograms follow at [nseg + s]
    flush_hist(lh, gh, 0u, zt);
    if (threadIdx.x == 0 && ont) atomicAdd(&state[s].cnt_mask, (unsigned long long)ont);
    SelState& sel = state[s].sel[PH_PRUNE];
    if (threadIdx.x == 0 && zt) atomicAdd(&sel.nspecial, zt);
}

// one block per tensor, after P1: funcs.py:107-109 in fp64 exactly as CPython evaluates them, then the bin of the k-th key.
// (A "last block to finish does it" epilogue inside P1 needs a device-scope fence per block: measured at ~1 us each, serialised per
// XCD -- 0.24 ms for 1.9 K blocks.  A 95-block launch costs a few us.)
__global__ __launch_bounds__(MK_THREADS) void mask_pick_prune_kernel(const slak_mask_segment_t* __restrict__ segs, SegState* __restrict__ state,
                                                                     const unsigned* __restrict__ hist, double* __restrict__ stats, double prune_rate) {
    __shared__ unsigned sh[MK_THREADS / 64];
    __shared__ unsigned res[4];
    const int s = blockIdx.x;
    SelState& sel = state[s].sel[PH_PRUNE];
    const unsigned* gh = hist + (size_t)s * D1_BINS;
    const long long numel = segs[s].numel;
    const double nonzeros = (double)state[s].cnt_mask;
    const double zeros = (double)numel - nonzeros;
    const double num_remove = ceil(prune_rate * nonzeros);          // math.ceil(masking.prune_rate*name2nonzeros)
    double kk = ceil(zeros + num_remove);                           // math.ceil(num_zeros + num_remove)
    if (kk > (double)numel) kk = (double)numel;                     // idx[:k] saturates
    const int mode = (num_remove == 0.0) ? MODE_NONZERO : (kk > 0 ? MODE_SELECT : MODE_NONE);
    if (mode == MODE_SELECT) find_bin<MK_THREADS, D1_BINS>([&](int b) { return gh[b]; }, (unsigned long long)kk, sh, res);
    if (threadIdx.x == 0) {
        stats[4 * s + 0] = nonzeros; stats[4 * s + 1] = zeros;
        sel.mode = mode;
        if (mode == MODE_SELECT) { sel.d1 = res[0]; sel.k = res[1]; }
    }
}

// P2 / P4: compact the keys of bin d1 (with their flat index)
template <int PHASE>
__global__ __launch_bounds__(MK_THREADS) void mask_compact_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                  const int* __restrict__ sblk_seg, const int* __restrict__ seg_sblk0,
                                                                  SegState* __restrict__ state, const long long* __restrict__ seg_off,
                                                                  uint2* __restrict__ cand, unsigned* __restrict__ sblk_special,
                                                                  const unsigned char* __restrict__ act, unsigned char* __restrict__ grown,
                                                                  const long long* __restrict__ seg_nib) {
    // Candidates are staged in LDS (wave-aggregated appends) and leave with ONE global atomic per block and coalesced stores; a
    // wave whose candidates no longer fit appends straight to the global list (only tie-heavy data gets there).
    __shared__ uint2 stage[CP_STAGE];
    __shared__ unsigned sh[MK_THREADS / 64];
    __shared__ unsigned lcount, lfail, gbase;
    constexpr unsigned SPECIAL = special_key(PHASE);
    const int s = sblk_seg[blockIdx.x];
    SelState& sel = state[s].sel[PHASE];
    if (sel.mode != MODE_SELECT) return;                               // block-uniform
    const slak_mask_segment_t sg = segs[s];
    const unsigned d1 = sel.d1;
    uint2* glist = cand + seg_off[s];
    const long long base = (long long)(blockIdx.x - seg_sblk0[s]) * SB_ELEMS;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (threadIdx.x == 0) { lcount = 0; lfail = 0xffffffffu; }
    __syncthreads();
    unsigned nsp = 0;
    const long long nib0 = (PHASE == PH_GROW) ? seg_nib[s] : 0;
    // (issuing the block's 16 loads per thread before the first use -- fully unrolled -- measured 3.5x SLOWER: 161 vs 46 us)
#pragma unroll 2
    for (int it = 0; it < SB_ITERS; ++it) {
        const long long i = base + ((long long)it * MK_THREADS + threadIdx.x) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned abits = 0;
        if (i < sg.numel) {
            if (PHASE == PH_PRUNE) a = load4(sg.weight, i, sg.numel);
            else { a = load4(sg.grad, i, sg.numel); abits = act[nib0 + (i >> 2)]; }
        }
        const float av[4] = {a.x, a.y, a.z, a.w};
        unsigned key[4]; bool f[4]; unsigned long long bal[4]; unsigned n = 0, sure = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            key[e] = (PHASE == PH_PRUNE) ? key_w(av[e]) : key_g(av[e], (abits >> e) & 1u ? 1.0f : 0.0f);
            const bool in = i + e < sg.numel;
            const bool in_bin = in && (key[e] >> D1_SHIFT) == d1;
            if (in && (key[e] >> D1_SHIFT) < d1) sure |= 1u << e;       // below the cut bin: taken whatever the refine finds
            f[e] = in_bin && key[e] != SPECIAL;
            nsp += (in_bin && key[e] == SPECIAL) ? 1u : 0u;
            bal[e] = __ballot(f[e]);
            n += (unsigned)__popcll(bal[e]);
        }
        if (PHASE == PH_GROW && i < sg.numel) grown[nib0 + (i >> 2)] = (unsigned char)sure;
        if (n == 0) continue;                                           // wave-uniform
        unsigned pos = 0;
        if (lane == 0) pos = atomicAdd(&lcount, n);
        pos = __shfl(pos, 0, 64);
        const bool fits = pos + n <= (unsigned)CP_STAGE;
        uint2* dst = stage;
        if (!fits) {                                                    // wave-uniform
            if (lane == 0) { atomicMin(&lfail, pos); pos = atomicAdd(&sel.cand, n); }
            pos = __shfl(pos, 0, 64);
            dst = glist;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (f[e]) dst[pos + (unsigned)__popcll(bal[e] & lt)] = make_uint2(key[e], (unsigned)(i + e));
            pos += (unsigned)__popcll(bal[e]);
        }
    }
    if (d1 == (SPECIAL >> D1_SHIFT)) {                                  // block-uniform: the cut bin holds the special key
        const unsigned t = block_sum<MK_THREADS>(nsp, sh);
        if (threadIdx.x == 0) sblk_special[blockIdx.x] = t;
    }
    __syncthreads();
    const unsigned staged = min(lcount, lfail);                         // appends are contiguous up to the first one that did not fit
    if (staged == 0) return;
    if (threadIdx.x == 0) gbase = atomicAdd(&sel.cand, staged);
    __syncthreads();
    for (unsigned j = threadIdx.x; j < staged; j += MK_THREADS) glist[gbase + j] = stage[j];
}

// R: one workgroup per tensor finishes the select on the candidate list
template <int PHASE>
__global__ __launch_bounds__(RF_THREADS) void mask_refine_kernel(const slak_mask_segment_t* __restrict__ segs, SegState* __restrict__ state,
                                                                 const long long* __restrict__ seg_off, const uint2* __restrict__ cand,
                                                                 const int* __restrict__ seg_sblk0, const unsigned* __restrict__ sblk_special,
                                                                 const unsigned char* __restrict__ act, const long long* __restrict__ seg_nib) {
    __shared__ unsigned lh[2048];
    __shared__ unsigned sh[RF_THREADS / 64];
    __shared__ unsigned res[4];
    constexpr unsigned SPECIAL = special_key(PHASE);
    const int s = blockIdx.x;
    SelState& sel = state[s].sel[PHASE];
    if (sel.mode != MODE_SELECT) return;
    const uint2* list = cand + seg_off[s];
    const unsigned c = sel.cand;
    const unsigned cpad = (c + 63u) & ~63u;                            // whole waves stay in the loops (run-length aggregation shuffles)
    unsigned long long k = sel.k;
    unsigned thr = sel.d1 << D1_SHIFT;
    const unsigned nspecial = sel.nspecial;
    // key bits 19..10, then 9..0
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
        const int shift = round == 0 ? 10 : 0;
        const unsigned dmask = 1023u;
        const unsigned himask = round == 0 ? 0xfff00000u : 0xfffffc00u;    // bit 31 marks "no entry"
        lh[threadIdx.x] = 0;
        __syncthreads();
        for (unsigned i0 = threadIdx.x; i0 < cpad; i0 += RF_BATCH * RF_THREADS) {
            unsigned key[RF_BATCH];
#pragma unroll
            for (int u = 0; u < RF_BATCH; ++u) { const unsigned i = i0 + u * RF_THREADS; key[u] = i < c ? list[i].x : 0xffffffffu; }
#pragma unroll
            for (int u = 0; u < RF_BATCH; ++u) {
                if (i0 + u * RF_THREADS < cpad) hist_add_runs(lh, (key[u] >> shift) & dmask, (key[u] & himask) == thr);   // wave-uniform guard
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && (SPECIAL & himask) == thr) lh[(SPECIAL >> shift) & dmask] += nspecial;     // the keys that are counted, not listed
        __syncthreads();
        find_bin<RF_THREADS, 1024>([&](int b) { return lh[b]; }, k, sh, res);
        thr |= res[0] << shift;
        k = res[1];
        __syncthreads();
    }
    const unsigned count_eq = res[2];
    unsigned idx_thr = 0xffffffffu;
    if (k < count_eq && thr == SPECIAL) {
        // only the k lowest flat indices among the special keys are taken.  Which 16384-element block holds the k-th one ...
        const int b0 = seg_sblk0[s], nb = seg_sblk0[s + 1] - b0;
        unsigned before = 0;                                            // special keys in the blocks before this chunk of RF_THREADS blocks
        __syncthreads();
        if (threadIdx.x == 0) res[0] = 0xffffffffu;
        __syncthreads();
#pragma unroll 1
        for (int cb = 0; cb < nb; cb += RF_THREADS) {
            const int b = cb + threadIdx.x;
            const unsigned v = b < nb ? sblk_special[b0 + b] : 0u;
            unsigned total;
            const unsigned excl = before + block_excl_scan<RF_THREADS>(v, sh, &total);
            if (k > excl && k <= (unsigned long long)excl + v) { res[0] = (unsigned)b; res[1] = (unsigned)(k - excl); }
            before += total;
            __syncthreads();
            if (res[0] != 0xffffffffu) break;                           // uniform
        }
        const unsigned bsel = res[0];
        unsigned r = res[1];                                            // 1-based rank inside the block
        __syncthreads();
        // ... and which element of that block it is (two consecutive elements per thread: thread order is index order)
        const slak_mask_segment_t sg = segs[s];
#pragma unroll 1
        for (int chunk = 0; chunk < SB_ELEMS / (2 * RF_THREADS); ++chunk) {
            const long long i0 = (long long)bsel * SB_ELEMS + (long long)chunk * 2 * RF_THREADS + 2 * threadIdx.x;
            bool f[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long long i = i0 + e;
                unsigned key = ~SPECIAL;
                if (i < sg.numel) {                                     // regrow: the pruned mask lives in the byte-per-four-elements copy (P3)
                    if (PHASE == PH_PRUNE) key = key_w(sg.weight[i]);
                    else key = key_g(sg.grad[i], (act[seg_nib[s] + (i >> 2)] >> (i & 3)) & 1u ? 1.0f : 0.0f);
                }
                f[e] = key == SPECIAL;
            }
            unsigned total;
            const unsigned excl = block_excl_scan<RF_THREADS>((f[0] ? 1u : 0u) + (f[1] ? 1u : 0u), sh, &total);
            if (r <= total) {                                           // uniform: it is in this chunk; exactly one thread holds it
                if (f[0] && excl + 1 == r) res[2] = (unsigned)i0;
                else if (f[1] && excl + (f[0] ? 1u : 0u) + 1 == r) res[2] = (unsigned)(i0 + 1);
                break;
            }
            r -= total;
        }
        __syncthreads();
        idx_thr = res[2];
    } else if (k < count_eq) {
        // only the k lowest flat indices among the listed keys == thr are taken: select the k-th smallest index, 11 bits at a time
        int bits = 1;
        while (bits < 31 && (1ll << bits) < segs[s].numel) ++bits;
        unsigned prefix = 0, done_mask = 0;
        int hi = bits;                                                  // bits [hi-1 .. lo] form this round's digit
#pragma unroll 1
        while (hi > 0) {
            const int lo = hi > 11 ? hi - 11 : 0;
            const unsigned dmask = (1u << (hi - lo)) - 1u;
            lh[threadIdx.x] = 0; lh[threadIdx.x + RF_THREADS] = 0;
            __syncthreads();
            for (unsigned i0 = threadIdx.x; i0 < cpad; i0 += RF_BATCH * RF_THREADS) {
                uint2 kv[RF_BATCH];
#pragma unroll
                for (int u = 0; u < RF_BATCH; ++u) { const unsigned i = i0 + u * RF_THREADS; kv[u] = i < c ? list[i] : make_uint2(0xffffffffu, 0u); }
#pragma unroll
                for (int u = 0; u < RF_BATCH; ++u) {
                    if (i0 + u * RF_THREADS < cpad) hist_add_runs(lh, (kv[u].y >> lo) & dmask, kv[u].x == thr && (kv[u].y & done_mask) == prefix);
                }
            }
            __syncthreads();
            find_bin<RF_THREADS, 2048>([&](int b) { return lh[b]; }, k, sh, res);
            prefix |= res[0] << lo;
            done_mask |= dmask << lo;
            k = res[1];
            __syncthreads();
            hi = lo;
        }
        idx_thr = prefix;
    }
    if (threadIdx.x == 0) { sel.thr = thr; sel.idx_thr = idx_thr; }
}

// P3: prune membership -> new mask; regrow-key histogram of the NEW mask in the same pass
template <bool GROW>
__global__ __launch_bounds__(MK_THREADS) void mask_prune_final_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                      const int* __restrict__ sblk_seg, const int* __restrict__ seg_sblk0,
                                                                      SegState* __restrict__ state, unsigned* __restrict__ hist, int nseg,
                                                                      unsigned char* __restrict__ act, const long long* __restrict__ seg_nib) {
    __shared__ unsigned lh[GROW ? D1_BINS : 1];
    __shared__ unsigned sh[MK_THREADS / 64];
    const int s = sblk_seg[blockIdx.x];
    const slak_mask_segment_t sg = segs[s];
    SelState& sp = state[s].sel[PH_PRUNE];
    const int mode = sp.mode;
    const unsigned thr = sp.thr, idx_thr = sp.idx_thr;
    const long long base = (long long)(blockIdx.x - seg_sblk0[s]) * SB_ELEMS;
    if (GROW) {
#pragma unroll
        for (int j = 0; j < D1_BINS / MK_THREADS; ++j) lh[j * MK_THREADS + threadIdx.x] = 0;
        __syncthreads();
    }
    unsigned changed = 0, top = 0;                                     // changed wraps mod 2^32 (MODE_NONZERO counts old - new, signed)
#pragma unroll 2
    for (int it = 0; it < SB_ITERS; ++it) {
        const long long i = base + ((long long)it * MK_THREADS + threadIdx.x) * 4;
        if (i >= sg.numel) break;
        const float4 m = load4(sg.mask, i, sg.numel);
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f), g = w;
        if (mode != MODE_NONE) w = load4(sg.weight, i, sg.numel);
        if (GROW) g = load4(sg.grad, i, sg.numel);
        const float wv[4] = {w.x, w.y, w.z, w.w}, mv[4] = {m.x, m.y, m.z, m.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float nm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            nm[e] = mv[e];
            if (i + e < sg.numel) {
                if (mode == MODE_SELECT) {
                    const unsigned key = key_w(wv[e]);
                    if (key < thr || (key == thr && (unsigned)(i + e) <= idx_thr)) { if (mv[e] != 0.0f) ++changed; nm[e] = 0.0f; }
                } else if (mode == MODE_NONZERO) {                       // prune returns weight.data != 0.0
                    nm[e] = (wv[e] != 0.0f) ? 1.0f : 0.0f;
                    changed += (mv[e] != 0.0f ? 1u : 0u) - (nm[e] != 0.0f ? 1u : 0u);
                }
                if (GROW) {
                    const unsigned key = key_g(gv[e], nm[e]);
                    if (key == KEY_MAX) ++top; else atomicAdd(&lh[key >> D1_SHIFT], 1u);   // active weights (|.| == 0): counted in a register
                }
            }
        }
        if (GROW) {
            // prune-and-grow: the pruned mask travels to P4 / P5 as one byte per four elements; the fp32 mask is written once, by P5
            act[seg_nib[s] + (i >> 2)] = (unsigned char)((nm[0] != 0.0f ? 1u : 0u) | (nm[1] != 0.0f ? 2u : 0u) | (nm[2] != 0.0f ? 4u : 0u) | (nm[3] != 0.0f ? 8u : 0u));
        } else if (mode != MODE_NONE) store4(sg.mask, i, sg.numel, make_float4(nm[0], nm[1], nm[2], nm[3]));
    }
    const unsigned ct = block_sum<MK_THREADS>(changed, sh);
    if (threadIdx.x == 0 && ct) atomicAdd(&sp.cnt_changed, (unsigned long long)(long long)(int)ct);
    unsigned* gh = hist + ((size_t)nseg + s) * D1_BINS;
    if (GROW) {
        const unsigned tt = block_sum<MK_THREADS>(top, sh);
        flush_hist(lh, gh, D1_BINS - 1, tt);
        if (threadIdx.x == 0 && tt) atomicAdd(&state[s].sel[PH_GROW].nspecial, tt);
    }
}

// one block per tensor, after P3: removed = name2nonzeros - new_mask.sum() (sparse_core.py:345), the regrow k and its bin
__global__ __launch_bounds__(MK_THREADS) void mask_pick_grow_kernel(SegState* __restrict__ state, const unsigned* __restrict__ hist,
                                                                    double* __restrict__ stats, int nseg, int grow) {
    __shared__ unsigned sh[MK_THREADS / 64];
    __shared__ unsigned res[4];
    const int s = blockIdx.x;
    const unsigned* gh = hist + ((size_t)nseg + s) * D1_BINS;
    const long long removed = (long long)state[s].sel[PH_PRUNE].cnt_changed;
    SelState& sel = state[s].sel[PH_GROW];
    const int gmode = (grow && removed > 0) ? MODE_SELECT : MODE_NONE;  // math.floor(removed) of an integer; nothing to regrow otherwise
    if (gmode == MODE_SELECT) find_bin<MK_THREADS, D1_BINS>([&](int b) { return gh[b]; }, (unsigned long long)removed, sh, res);
    if (threadIdx.x == 0) {
        state[s].removed = (unsigned long long)removed;
        stats[4 * s + 2] = (double)removed;
        stats[4 * s + 3] = stats[4 * s + 0] - (double)removed;          // final when nothing is regrown
        sel.mode = gmode;
        if (gmode == MODE_SELECT) { sel.d1 = res[0]; sel.k = res[1]; }
    }
}

__global__ void mask_finish_kernel(const SegState* __restrict__ state, double* __restrict__ stats, int nseg) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg || state[s].sel[PH_GROW].mode != MODE_SELECT) return;
    stats[4 * s + 3] = stats[4 * s + 0] - (double)(long long)state[s].removed + (double)state[s].sel[PH_GROW].cnt_changed;
}

// after the regrow refine: the listed candidates that made the cut
__global__ __launch_bounds__(MK_THREADS) void mask_mark_kernel(const int* __restrict__ blk_seg, const int* __restrict__ seg_blk0,
                                                               const SegState* __restrict__ state, const long long* __restrict__ seg_off,
                                                               const uint2* __restrict__ cand, unsigned* __restrict__ grown32,
                                                               const long long* __restrict__ seg_nib) {
    // 2048 list entries per block (the list is never longer than the tensor); 16 K-entry blocks measured 34 us against 17: a tensor's
    // list sits in few blocks and the kernel is latency-bound
    const int s = blk_seg[blockIdx.x];
    const SelState& sel = state[s].sel[PH_GROW];
    if (sel.mode != MODE_SELECT) return;
    const unsigned c = sel.cand, thr = sel.thr, idx_thr = sel.idx_thr;
    const unsigned j0 = (unsigned)(blockIdx.x - seg_blk0[s]) * MK_BLOCK_ELEMS;
    if (j0 >= c) return;
    const uint2* list = cand + seg_off[s];
    const long long nib0 = seg_nib[s];
#pragma unroll
    for (int q = 0; q < MK_PER_THREAD; ++q) {
        const unsigned j = j0 + q * MK_THREADS + threadIdx.x;
        if (j < c) {
            const uint2 kv = list[j];
            if (kv.x < thr || (kv.x == thr && kv.y <= idx_thr)) {
                const long long nib = nib0 + (kv.y >> 2);
                atomicOr(&grown32[nib >> 2], 1u << ((unsigned)(nib & 3) * 8u + (kv.y & 3u)));
            }
        }
    }
}

// P5: final mask = pruned mask | regrown, written as exact 0.0 / 1.0; w *= mask (+ momentum): the apply of sparse_core.py:357.
// The gradient is only read where the cut fell among the keys that are exactly 0 (counted, not listed): those are taken by index.
__global__ __launch_bounds__(MK_THREADS) void mask_grow_final_kernel(const slak_mask_segment_t* __restrict__ segs,
                                                                     const int* __restrict__ blk_seg, const int* __restrict__ seg_blk0,
                                                                     SegState* __restrict__ state, const unsigned char* __restrict__ act,
                                                                     const unsigned char* __restrict__ grown, const long long* __restrict__ seg_nib) {
    __shared__ unsigned sh[MK_THREADS / 64];
    const int s = blk_seg[blockIdx.x];
    const slak_mask_segment_t sg = segs[s];
    SelState& sel = state[s].sel[PH_GROW];
    const int mode = sel.mode;
    const unsigned idx_thr = sel.idx_thr;
    const bool ties_at_zero = mode == MODE_SELECT && sel.thr == KEY_MAX;
    const long long base = (long long)(blockIdx.x - seg_blk0[s]) * MK_BLOCK_ELEMS;
    const long long nib0 = seg_nib[s];
    unsigned changed = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long i = base + ((long long)h * MK_THREADS + threadIdx.x) * 4;
        if (i >= sg.numel) break;
        const unsigned a = act[nib0 + (i >> 2)];
        unsigned gr = (mode == MODE_SELECT) ? grown[nib0 + (i >> 2)] : 0u;
        const float4 w = load4(sg.weight, i, sg.numel);
        float4 mo = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sg.momentum) mo = load4(sg.momentum, i, sg.numel);
        if (ties_at_zero) {
            const float4 g = load4(sg.grad, i, sg.numel);
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < sg.numel && key_g(gv[e], (a >> e) & 1u ? 1.0f : 0.0f) == KEY_MAX && (unsigned)(i + e) <= idx_thr) gr |= 1u << e;
        }
        changed += (unsigned)__popc(gr & ~a & 15u);
        const unsigned on = (a | gr) & 15u;
        const float mv[4] = {on & 1u ? 1.0f : 0.0f, on & 2u ? 1.0f : 0.0f, on & 4u ? 1.0f : 0.0f, on & 8u ? 1.0f : 0.0f};
        store4(sg.mask, i, sg.numel, make_float4(mv[0], mv[1], mv[2], mv[3]));
        store4(sg.weight, i, sg.numel, make_float4(w.x * mv[0], w.y * mv[1], w.z * mv[2], w.w * mv[3]));
        if (sg.momentum) store4(sg.momentum, i, sg.numel, make_float4(mo.x * mv[0], mo.y * mv[1], mo.z * mv[2], mo.w * mv[3]));
    }
    if (mode != MODE_SELECT) return;
    const unsigned ct = block_sum<MK_THREADS>(changed, sh);
    if (threadIdx.x == 0 && ct) atomicAdd(&sel.cnt_changed, (unsigned long long)ct);
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

// memset + P1 + P2 + R + P3: the prune half (sparse_core.py:337-347) and, with GROW, the first regrow pass
template <bool GROW>
static int run_prune(slak_mask_plan* p, double prune_rate, hipStream_t st) {
    if (!p->cand || (GROW && !p->act)) return SLAK_ERR_WORKSPACE;       // (allocated by slak_mask_plan_create: nothing is hipMalloc'ed mid-training)
    HIPCHK(hipMemsetAsync(p->zeroed, 0, p->zeroed_bytes, st));
    hipLaunchKernelGGL(mask_prune_hist_kernel, dim3(p->nsblk), dim3(MK_THREADS), 0, st, p->segs, p->sblk_seg, p->seg_sblk0, p->state, p->hist);
    hipLaunchKernelGGL(mask_pick_prune_kernel, dim3(p->nseg), dim3(MK_THREADS), 0, st, p->segs, p->state, p->hist, p->stats, prune_rate);
    hipLaunchKernelGGL(mask_compact_kernel<PH_PRUNE>, dim3(p->nsblk), dim3(MK_THREADS), 0, st, p->segs, p->sblk_seg, p->seg_sblk0, p->state,
                       p->seg_off, p->cand, p->blk_special, (const unsigned char*)nullptr, (unsigned char*)nullptr, (const long long*)nullptr);
    hipLaunchKernelGGL(mask_refine_kernel<PH_PRUNE>, dim3(p->nseg), dim3(RF_THREADS), 0, st, p->segs, p->state, p->seg_off, p->cand, p->seg_sblk0,
                       p->blk_special, (const unsigned char*)nullptr, (const long long*)nullptr);
    hipLaunchKernelGGL(mask_prune_final_kernel<GROW>, dim3(p->nsblk), dim3(MK_THREADS), 0, st, p->segs, p->sblk_seg, p->seg_sblk0, p->state,
                       p->hist, p->nseg, p->act, p->seg_nib);
    hipLaunchKernelGGL(mask_pick_grow_kernel, dim3(p->nseg), dim3(MK_THREADS), 0, st, p->state, p->hist, p->stats, p->nseg, GROW ? 1 : 0);
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
    std::vector<int> blk_seg, sblk_seg, seg_sblk0(nseg + 1);
    std::vector<long long> seg_off(nseg), seg_nib(nseg);
    long long nib = 0;
    p->seg_first_blk.resize(nseg + 1);
    for (int s = 0; s < nseg; ++s) {
        if (!segs_host[s].weight || !segs_host[s].mask || segs_host[s].numel <= 0 || segs_host[s].numel >= (1ll << 31)) {
            delete p; return SLAK_ERR_INVALID_ARG;                      // flat indices are carried as 32-bit values
        }
        p->seg_first_blk[s] = (int)blk_seg.size();
        seg_sblk0[s] = (int)sblk_seg.size();
        seg_off[s] = p->total;
        seg_nib[s] = nib; nib += ((segs_host[s].numel + 3) / 4 + 3) / 4 * 4;   // a byte per four elements, tensors start at a dword
        const long long nb = (segs_host[s].numel + MK_BLOCK_ELEMS - 1) / MK_BLOCK_ELEMS;
        for (long long b = 0; b < nb; ++b) blk_seg.push_back(s);
        const long long nsb = (segs_host[s].numel + SB_ELEMS - 1) / SB_ELEMS;
        for (long long b = 0; b < nsb; ++b) sblk_seg.push_back(s);
        p->total += segs_host[s].numel;
    }
    p->seg_first_blk[nseg] = (int)blk_seg.size();
    seg_sblk0[nseg] = (int)sblk_seg.size();
    p->nblk = (int)blk_seg.size();
    p->nsblk = (int)sblk_seg.size();
    const size_t hist_bytes = sizeof(unsigned) * 2 * (size_t)nseg * D1_BINS;
    p->zeroed_bytes = hist_bytes + sizeof(SegState) * nseg;
#define ALLOC(ptr, bytes) HIPCHK(hipMalloc((void**)&(ptr), (bytes)))
    ALLOC(p->segs, sizeof(slak_mask_segment_t) * nseg);
    ALLOC(p->blk_seg, sizeof(int) * p->nblk);
    ALLOC(p->seg_blk0, sizeof(int) * (nseg + 1));
    ALLOC(p->sblk_seg, sizeof(int) * p->nsblk);
    ALLOC(p->seg_sblk0, sizeof(int) * (nseg + 1));
    ALLOC(p->seg_off, sizeof(long long) * nseg);
    ALLOC(p->seg_nib, sizeof(long long) * nseg);
    ALLOC(p->blk_special, sizeof(unsigned) * p->nsblk);
    ALLOC(p->zeroed, p->zeroed_bytes);
    ALLOC(p->stats, sizeof(double) * 4 * nseg);
    ALLOC(p->checksum, sizeof(unsigned long long));
#undef ALLOC
    p->hist = (unsigned*)p->zeroed;
    p->state = (SegState*)((char*)p->zeroed + hist_bytes);
    HIPCHK(hipHostMalloc((void**)&p->segs_pinned, sizeof(slak_mask_segment_t) * nseg));
    HIPCHK(hipEventCreateWithFlags(&p->upload_done, hipEventDisableTiming));
    HIPCHK(hipMemcpy(p->segs, p->segs_host.data(), sizeof(slak_mask_segment_t) * nseg, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->blk_seg, blk_seg.data(), sizeof(int) * p->nblk, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->seg_blk0, p->seg_first_blk.data(), sizeof(int) * (nseg + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->sblk_seg, sblk_seg.data(), sizeof(int) * p->nsblk, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->seg_sblk0, seg_sblk0.data(), sizeof(int) * (nseg + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->seg_off, seg_off.data(), sizeof(long long) * nseg, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->seg_nib, seg_nib.data(), sizeof(long long) * nseg, hipMemcpyHostToDevice));
    p->nib_bytes = (size_t)nib + 16;
    // The select's scratch is taken HERE, when the plan is made (Masking.add_module: before the first training step), not lazily at the first
    // prune-and-grow: a raw hipMalloc in the middle of training synchronises the device and can fail once the framework's caching allocator
    // has reserved the memory.  cand: worst case one (key, index) record per element (8 B/element: 245 MB for SLaK-T; the
    // lists normally hold a few per cent of a tensor); the two byte-per-four-elements arrays of the regrow half.
    HIPCHK(hipMalloc((void**)&p->cand, sizeof(uint2) * (size_t)p->total));
    HIPCHK(hipMalloc((void**)&p->act, p->nib_bytes));
    HIPCHK(hipMalloc((void**)&p->grown, p->nib_bytes));
    HIPCHK(hipMemset(p->stats, 0, sizeof(double) * 4 * nseg));
    HIPCHK(hipMemset(p->zeroed, 0, p->zeroed_bytes));
    *plan_out = p;
    return SLAK_OK;
}

// The descriptor table goes up through a pinned staging copy, asynchronously: the host only waits (on an event) when the
// PREVIOUS upload of the same plan has not been consumed yet.
static int upload_segs(slak_mask_plan* p, hipStream_t st) {
    HIPCHK(hipEventSynchronize(p->upload_done));
    std::memcpy(p->segs_pinned, p->segs_host.data(), sizeof(slak_mask_segment_t) * p->nseg);
    HIPCHK(hipMemcpyAsync(p->segs, p->segs_pinned, sizeof(slak_mask_segment_t) * p->nseg, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(p->upload_done, st));
    return SLAK_OK;
}

int slak_mask_plan_set_grads(slak_mask_plan_t* p, const void* const* grads_host, void* stream) {
    if (!p || !grads_host) return SLAK_ERR_INVALID_ARG;
    bool same = true;
    for (int s = 0; s < p->nseg; ++s) { same = same && p->segs_host[s].grad == (const float*)grads_host[s]; p->segs_host[s].grad = (const float*)grads_host[s]; }
    return same ? SLAK_OK : upload_segs(p, (hipStream_t)stream);
}

int slak_mask_plan_set_momentum(slak_mask_plan_t* p, void* const* momentum_host, void* stream) {
    if (!p || !momentum_host) return SLAK_ERR_INVALID_ARG;
    bool same = true;
    for (int s = 0; s < p->nseg; ++s) { same = same && p->segs_host[s].momentum == (float*)momentum_host[s]; p->segs_host[s].momentum = (float*)momentum_host[s]; }
    return same ? SLAK_OK : upload_segs(p, (hipStream_t)stream);
}

int slak_mask_plan_destroy(slak_mask_plan_t* p) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    void* dev[] = {p->segs, p->blk_seg, p->seg_blk0, p->sblk_seg, p->seg_sblk0, p->seg_off, p->seg_nib, p->act, p->grown, p->blk_special, p->zeroed, p->cand,
                   p->stats, p->checksum};
    for (void* d : dev) if (d) (void)hipFree(d);
    if (p->segs_pinned) (void)hipHostFree(p->segs_pinned);
    if (p->upload_done) (void)hipEventDestroy(p->upload_done);
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
    int rc = run_prune<true>(p, prune_rate, st);           // prune loop, sparse_core.py:337-347 (+ first regrow pass)
    if (rc != SLAK_OK) return rc;
    hipLaunchKernelGGL(mask_compact_kernel<PH_GROW>, dim3(p->nsblk), dim3(MK_THREADS), 0, st, p->segs, p->sblk_seg, p->seg_sblk0, p->state,
                       p->seg_off, p->cand, p->blk_special, p->act, p->grown, p->seg_nib);   // growth loop, sparse_core.py:349-355
    hipLaunchKernelGGL(mask_refine_kernel<PH_GROW>, dim3(p->nseg), dim3(RF_THREADS), 0, st, p->segs, p->state, p->seg_off, p->cand, p->seg_sblk0,
                       p->blk_special, p->act, p->seg_nib);
    hipLaunchKernelGGL(mask_mark_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, st, p->blk_seg, p->seg_blk0, p->state, p->seg_off, p->cand,
                       (unsigned*)p->grown, p->seg_nib);
    hipLaunchKernelGGL(mask_grow_final_kernel, dim3(p->nblk), dim3(MK_THREADS), 0, st, p->segs, p->blk_seg, p->seg_blk0, p->state, p->act, p->grown,
                       p->seg_nib);                         // ... and the apply of sparse_core.py:357
    hipLaunchKernelGGL(mask_finish_kernel, dim3(ceil_div(p->nseg, 64)), dim3(64), 0, st, p->state, p->stats, p->nseg);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_mask_prune(slak_mask_plan_t* p, double prune_rate, void* stream) {
    if (!p) return SLAK_ERR_INVALID_ARG;
    return run_prune<false>(p, prune_rate, (hipStream_t)stream);
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
