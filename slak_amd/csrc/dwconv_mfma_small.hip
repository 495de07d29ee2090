// slak_amd/csrc/dwconv_mfma_small.hip -- MFMA depthwise-conv forward / data-grad for SMALL planes (H, W <= 16: the
// 14x14 and 7x7 stages of SLaK), 16-bit activations.
//
// Same factorisation as dwconv_mfma_dma.hip (dense 1-D Toeplitz per short tap, five shifted B fragments into ONE
// accumulator), re-shaped for planes that are smaller than a cache line pair:
//   * a workgroup owns FOUR CONSECUTIVE CHANNELS (one per wave) and streams over the batch, so that what it reads and
//     writes per image is one contiguous 4*H*W-element block (1568 B for 14x14, 392 B for 7x7) instead of isolated 98-392 B
//     planes (the register-staged kernel of dwconv_mfma.hip measured 2x the algorithmic HBM traffic on 7x7:
//     profiles/r01_pmc_traffic.txt);
//   * v_mfma_f32_16x16x32: M = 16 output positions along the long axis, N = 16 lanes = positions along the short axis of
//     one 14-plane / two 7-planes, and the K = 32 contraction holds SEVERAL TAPS side by side (2 x 16 or 4 x 8): each
//     8-lane k-group reads the row its own tap needs (u + r - 2), so 5 taps cost 3 (2) MFMAs and one 4-register accumulator;
//   * planes are staged [short-axis position][long-axis position] with pitch 8/16 (vertical kernels are simply written
//     transposed by the staging stores: the planes are tiny), so every B fragment is one aligned ds_read_b128;
//   * zero padding by select on the fragment, never by multiplication (NaN/Inf stay inside their plane).
#include "mfma_common.h"

namespace slak {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int SM_CB = 4;                // channels per workgroup (one per wave)
constexpr int SM_NCH = 2;               // staging chunks per thread per iteration (upper bound)

struct SmallParams {
    const void* x; const float* w; void* y;
    int N, C, H, W, kh, kw, flip, KL, padL;
    int Wt, Wl, KP;        // long-axis extent, short-axis extent, padded k per tap (8 or 16)
    int NI;                // images per iteration
    int PPT;               // planes per 16-lane tile
    int in_plane;          // LDS elements per staged plane (Wl * KP)
    int images_per_wg, slices;
    int nchunks;           // staging chunks per iteration (NI * 4*H*W / V)
};

__device__ __forceinline__ f32x4_t mfma16_bf16(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma16_f16(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ f32x4_t mfma16(s16x8 a, s16x8 b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mfma16<bf16_t>(s16x8 a, s16x8 b, f32x4_t c) { return mfma16_bf16(a, b, c); }
template <> __device__ __forceinline__ f32x4_t mfma16<f16_t>(s16x8 a, s16x8 b, f32x4_t c) { return mfma16_f16(a, b, c); }

// A fragments: frag[c][m][lane][8]; lane -> output position o = lane & 15, k-group kg = lane >> 4 -> tap r = m*TP + (kg*8)/KP,
// first long-axis input position i0 = (kg*8) % KP; element e: T_r[o, i0+e] = w[r][i0 + e - o + padL]
struct SmallPackParams { const float* w; uint16_t* frags; int C, kh, kw, NM, KP, vert, flip, Wt, KL, padL, is_bf16; };

__global__ void toeplitz_pack_small_kernel(const SmallPackParams p) {
    const int total = p.C * p.NM * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, m = (idx >> 6) % p.NM, c = (idx >> 6) / p.NM;
    const int TP = 32 / p.KP, o = lane & 15, kg = lane >> 4;
    const int r = m * TP + (kg * 8) / p.KP, i0 = (kg * 8) % p.KP;
    const float* wc = p.w + (size_t)c * p.kh * p.kw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = i0 + e;
        int t = i - o + p.padL;
        const bool ok = r < MF_TAPS && o < p.Wt && i < p.Wt && t >= 0 && t < p.KL;
        int rr = r < MF_TAPS ? r : 0;
        t = ok ? t : 0;
        if (p.flip) { t = p.KL - 1 - t; rr = MF_TAPS - 1 - rr; }
        const float wv = p.vert ? wc[t * p.kw + rr] : wc[rr * p.kw + t];
        v[e] = ok ? wv : 0.f;
    }
    u32x4 out;
#pragma unroll
    for (int e = 0; e < 8; e += 2) out[e >> 1] = p.is_bf16 ? pack2<bf16_t>(v[e], v[e + 1]) : pack2<f16_t>(v[e], v[e + 1]);
    ((u32x4*)p.frags)[idx] = out;
}

// NM: MFMAs per tile (ceil(5 / taps per MFMA)); V: staging vector width (8: 16-byte chunks, 4: 8-byte chunks)
template <typename T, int NM, int V, bool VERT>
__global__ __launch_bounds__(MF_THREADS) void dwconv_mfma_small_kernel(const SmallParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int HW = p.H * p.W, blk = SM_CB * HW;               // elements of one image's 4-channel block
    uint16_t* lin = lds;                                      // [SM_CB][NI][Wl][KP]
    uint16_t* lout = lds + ((SM_CB * p.NI * p.in_plane + 7) & ~7);   // [NI][SM_CB][HW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + SM_CB - 1) / SM_CB;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c0 = cb * SM_CB, c = c0 + wave;
    const uint16_t* __restrict__ x = (const uint16_t*)p.x;
    uint16_t* __restrict__ y = (uint16_t*)p.y;
    const int n_begin = slice * p.images_per_wg;
    int n_end = n_begin + p.images_per_wg; if (n_end > p.N) n_end = p.N;
    if (n_begin >= n_end) return;
    const int iters = (n_end - n_begin + p.NI - 1) / p.NI;
    const int nch = (p.C - c0 < SM_CB) ? (p.C - c0) : SM_CB;  // channels that exist in this block
    const int valid_blk = nch * HW;                           // elements of a block that belong to this image

    // ---- staging map: chunk idx -> (image ni, element offset e0 inside the image's block) ----------------
    int ch_ni[SM_NCH], ch_e0[SM_NCH];
    const int cpi = blk / V;                                  // chunks per image block
#pragma unroll
    for (int k = 0; k < SM_NCH; ++k) {
        const int idx = tid + k * MF_THREADS;
        const bool ok = idx < p.nchunks;
        const int ni = ok ? idx / cpi : 0;
        ch_ni[k] = ok ? ni : -1;
        ch_e0[k] = ok ? (idx - ni * cpi) * V : 0;
    }
    chunk_t<V> st[SM_NCH];
    auto prefetch = [&](int it) {
        const int n0 = n_begin + it * p.NI;
#pragma unroll
        for (int k = 0; k < SM_NCH; ++k) {
            st[k] = chunk_zero<V>();
            if (ch_ni[k] >= 0 && n0 + ch_ni[k] < n_end) {
                const uint16_t* src = x + ((size_t)(n0 + ch_ni[k]) * p.C + c0) * HW + ch_e0[k];
                if (ch_e0[k] + V <= valid_blk) st[k] = chunk_load<V>(src);
                else {                                            // ragged channel block (C % 4 != 0): never read past the block
                    for (int i = 0; i < V; ++i) if (ch_e0[k] + i < valid_blk) chunk_set<V>(st[k], i, src[i]);
                }
            }
        }
    };
    // LDS element offset of every element of this thread's chunks (-1: beyond the block).  The chunk -> (channel, h, w) map does
    // not change between iterations: resolving it here keeps the per-iteration staging to V plain ds_write_b16 per chunk (the
    // div/mod + carry chain it replaces was ~12 VALU per ELEMENT, a third of the loop's instructions on 7x7 planes).
    int soff[SM_NCH][V];
#pragma unroll
    for (int k = 0; k < SM_NCH; ++k) {
        int e = ch_e0[k];
        int ch = e / HW, rem = e - ch * HW;
        int h = rem / p.W, w = rem - h * p.W;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const bool ok = ch_ni[k] >= 0 && e + i < valid_blk;
            soff[k][i] = ok ? ((ch * p.NI + ch_ni[k]) * p.Wl + (VERT ? w : h)) * p.KP + (VERT ? h : w) : -1;
            ++w;
            if (w == p.W) { w = 0; ++h; if (h == p.H) { h = 0; ++ch; } }
        }
    }
    auto stage_write = [&]() {
#pragma unroll
        for (int k = 0; k < SM_NCH; ++k) {
#pragma unroll
            for (int i = 0; i < V; ++i)
                if (soff[k][i] >= 0) lin[soff[k][i]] = chunk_get<V>(st[k], i);
        }
    };

    prefetch(0);
    for (int i = tid; i < (SM_CB * p.NI * p.in_plane + 7) / 8; i += MF_THREADS) ((u32x4*)lin)[i] = u32x4{0u, 0u, 0u, 0u};
    // A fragments of this wave's channel, built in place: lane -> output position o = lane & 15, k-group kg = lane >> 4 -> tap
    // r = m*TP + (kg*8)/KP, first long-axis input position i0 = (kg*8) % KP; element e: T_r[o, i0+e] = w[r][i0 + e - o + padL]
    s16x8 afrag[NM];
    {
        const float* wc = p.w + (size_t)(c < p.C ? c : 0) * p.kh * p.kw;
        const int o = lane & 15, kgq = lane >> 4, TPq = 32 / p.KP, i0q = (kgq * 8) % p.KP;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int r = m * TPq + (kgq * 8) / p.KP;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = i0q + e;
                int t = i - o + p.padL;
                const bool ok = r < MF_TAPS && o < p.Wt && i < p.Wt && t >= 0 && t < p.KL;
                int rr = r < MF_TAPS ? r : 0;
                t = ok ? t : 0;
                if (p.flip) { t = p.KL - 1 - t; rr = MF_TAPS - 1 - rr; }
                const float wv = VERT ? wc[t * p.kw + rr] : wc[rr * p.kw + t];
                v[e] = ok ? wv : 0.f;
            }
            u32x4 a;
#pragma unroll
            for (int e = 0; e < 8; e += 2) a[e >> 1] = pack2<T>(v[e], v[e + 1]);
            afrag[m] = __builtin_bit_cast(s16x8, a);
        }
    }
    __syncthreads();
    stage_write();
    __syncthreads();

    // ---- per-lane constants: tile lane u -> (plane in tile, position), k-group -> (tap within an MFMA, first k) ----
    const int u = lane & 15, kg = lane >> 4;
    const int pl = u / p.Wl, pos = u - pl * p.Wl;             // pl >= PPT: idle lane
    const int TP = 32 / p.KP, i0 = (kg * 8) % p.KP, rsel = (kg * 8) / p.KP;
    int boff[NM]; bool bok[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int r = m * TP + rsel, row = pos + r - 2;
        bok[m] = r < MF_TAPS && pl < p.PPT && row >= 0 && row < p.Wl;
        boff[m] = (bok[m] ? row : 0) * p.KP + i0;
    }
    const int ntiles = (p.NI + p.PPT - 1) / p.PPT;

    for (int it = 0; it < iters; ++it) {
        const int n0 = n_begin + it * p.NI;
        if (it + 1 < iters) prefetch(it + 1);
        if (c < p.C) {
            const uint16_t* cin = lin + wave * p.NI * p.in_plane;
            for (int t = 0; t < ntiles; ++t) {
                const int ni = t * p.PPT + pl;                   // image of this lane's plane
                const bool img_ok = pl < p.PPT && ni < p.NI;
                const uint16_t* pb = cin + (img_ok ? ni : 0) * p.in_plane;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    u32x4 b = *(const u32x4*)(pb + boff[m]);
                    const bool ok = bok[m] && img_ok;
                    b[0] = ok ? b[0] : 0u; b[1] = ok ? b[1] : 0u; b[2] = ok ? b[2] : 0u; b[3] = ok ? b[3] : 0u;
                    acc = mfma16<T>(afrag[m], __builtin_bit_cast(s16x8, b), acc);
                }
                // D: column = lane & 15 = tile lane u, rows o = 4*kg + e
                if (img_ok && n0 + ni < n_end) {
                    uint16_t* op = lout + (ni * SM_CB + wave) * HW;
                    const unsigned p0 = pack2<T>(acc[0], acc[1]), p1 = pack2<T>(acc[2], acc[3]);
                    const int o = 4 * kg;
                    if constexpr (VERT) {                        // long axis = H: (oh = o + e, ow = pos)
                        if (o + 0 < p.Wt) op[(o + 0) * p.W + pos] = (uint16_t)(p0 & 0xffffu);
                        if (o + 1 < p.Wt) op[(o + 1) * p.W + pos] = (uint16_t)(p0 >> 16);
                        if (o + 2 < p.Wt) op[(o + 2) * p.W + pos] = (uint16_t)(p1 & 0xffffu);
                        if (o + 3 < p.Wt) op[(o + 3) * p.W + pos] = (uint16_t)(p1 >> 16);
                    } else {                                     // long axis = W: (oh = pos, ow = o + e)
                        uint16_t* orow = op + pos * p.W + o;
                        if (o + 0 < p.Wt) orow[0] = (uint16_t)(p0 & 0xffffu);
                        if (o + 1 < p.Wt) orow[1] = (uint16_t)(p0 >> 16);
                        if (o + 2 < p.Wt) orow[2] = (uint16_t)(p1 & 0xffffu);
                        if (o + 3 < p.Wt) orow[3] = (uint16_t)(p1 >> 16);
                    }
                }
            }
        }
        __syncthreads();
        // ---- finished images -> HBM: each image's block of nch channels is contiguous --------------------
#pragma unroll
        for (int k = 0; k < SM_NCH; ++k) {
            if (ch_ni[k] >= 0 && n0 + ch_ni[k] < n_end) {
                uint16_t* dst = y + ((size_t)(n0 + ch_ni[k]) * p.C + c0) * HW + ch_e0[k];
                const uint16_t* src = lout + ch_ni[k] * blk + ch_e0[k];
                if (ch_e0[k] + V <= valid_blk) chunk_store<V>(dst, chunk_load<V>(src));
                else for (int i = 0; i < V; ++i) if (ch_e0[k] + i < valid_blk) dst[i] = src[i];
            }
        }
        if (it + 1 < iters) stage_write();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_small_params(SmallParams& p, const ConvDims& d, bool vert, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    if (p.Wt > 16 || p.Wl > 16) return false;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return false;
    p.KP = p.Wt <= 8 ? 8 : 16;
    p.PPT = 16 / p.Wl;
    p.in_plane = p.Wl * p.KP;
    const int HW = d.H * d.W, blk = SM_CB * HW;
    const int V = (blk % 8 == 0 && (HW % 2 == 0)) ? 8 : 4;
    if (blk % V) return false;
    int NI = (SM_NCH * MF_THREADS * V) / blk; if (NI < 1) return false;
    if (NI > 16) NI = 16;
    NI = NI / p.PPT * p.PPT; if (NI < p.PPT) NI = p.PPT;
    if (NI * blk / V > SM_NCH * MF_THREADS) return false;
    const int cblocks = (d.C + SM_CB - 1) / SM_CB;
    int slices = resident_wgs / cblocks; if (slices < 1) slices = 1;
    if (slices > d.N) slices = d.N;
    int per = (d.N + slices - 1) / slices;
    static const int min_iters = [] { const char* e = slak_dev_getenv("SLAK_SMALL_MIN_ITERS"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : v; }();
    if (per < min_iters * NI) per = min_iters * NI;                          // at least 3 iterations per workgroup: amortise its prologue
    if (per > d.N) per = d.N;
    if (NI > per) { NI = (per + p.PPT - 1) / p.PPT * p.PPT; }
    per = (per + NI - 1) / NI * NI;
    p.NI = NI; p.images_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.nchunks = NI * blk / V;
    return true;
}

static int small_V(const ConvDims& d) { const int HW = d.H * d.W; return ((SM_CB * HW) % 8 == 0 && HW % 2 == 0) ? 8 : 4; }
static int small_NM(const ConvDims& d, bool vert) { const int Wt = vert ? d.H : d.W; return Wt <= 8 ? 2 : 3; }

static size_t small_lds_bytes(const SmallParams& p) {
    return (size_t)((SM_CB * p.NI * p.in_plane + 7) & ~7) * 2 + (size_t)p.NI * SM_CB * p.H * p.W * 2 + 16;
}

bool dwconv_mfma_small_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt) {
    if (x_dt != y_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16) || w_dt != SLAK_F32) return false;
    SmallParams p;
    return fill_small_params(p, d, d.kh > d.kw, 2048) && small_lds_bytes(p) <= 48 * 1024;
}

size_t dwconv_mfma_small_workspace(const ConvDims& d) { (void)d; return 0; }

template <typename T, int NM, int V, bool VERT>
static int launch_small_t(SmallParams& p, const ConvDims& d, hipStream_t st) {
    auto k = dwconv_mfma_small_kernel<T, NM, V, VERT>;
    static int resident = 0;
    fill_small_params(p, d, VERT, 2048);
    const size_t lds0 = small_lds_bytes(p);
    if (resident == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, MF_THREADS, lds0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        resident = per_cu * mfma_cu_count();
    }
    fill_small_params(p, d, VERT, resident);
    const int cblocks = (d.C + SM_CB - 1) / SM_CB;
    hipLaunchKernelGGL(k, dim3((unsigned)(cblocks * p.slices)), dim3(MF_THREADS), small_lds_bytes(p), st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_small(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                             const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_small_supported(d, x_dt, w_dt, y_dt)) return SLAK_ERR_UNSUPPORTED;
    (void)ws; (void)ws_bytes;
    const bool vert = d.kh > d.kw;
    SmallParams p;
    fill_small_params(p, d, vert, 2048);
    const int NM = small_NM(d, vert), V = small_V(d);
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2; p.flip = flip_filter ? 1 : 0;
    p.x = x; p.w = (const float*)w; p.y = y;
#define SLAK_SMALL_DISPATCH(T)                                                                                                   \
    if (NM == 3 && V == 8) return vert ? launch_small_t<T, 3, 8, true>(p, d, st) : launch_small_t<T, 3, 8, false>(p, d, st);    \
    if (NM == 3 && V == 4) return vert ? launch_small_t<T, 3, 4, true>(p, d, st) : launch_small_t<T, 3, 4, false>(p, d, st);    \
    if (NM == 2 && V == 8) return vert ? launch_small_t<T, 2, 8, true>(p, d, st) : launch_small_t<T, 2, 8, false>(p, d, st);    \
    return vert ? launch_small_t<T, 2, 4, true>(p, d, st) : launch_small_t<T, 2, 4, false>(p, d, st);
    if (x_dt == SLAK_BF16) { SLAK_SMALL_DISPATCH(bf16_t) }
    SLAK_SMALL_DISPATCH(f16_t)
#undef SLAK_SMALL_DISPATCH
}

}  // namespace slak
