// slak_amd/csrc/dwconv_mfma_small_wgrad.hip -- MFMA depthwise-conv weight gradient for SMALL planes (H, W <= 16: the 14x14 and
// 7x7 stages of SLaK), 16-bit activations, fp32 result.
//
// Same arithmetic as dwconv_mfma_wgrad.hip (per-tap 1-D correlation GEMM over a stacked contraction axis + diagonal sums),
// decomposed like dwconv_mfma_small.hip: a workgroup owns FOUR CONSECUTIVE CHANNELS (one per wave) and streams over the batch, so
// that every image contributes one contiguous 4*H*W-element block of x and of dy (the per-plane kernel measured 1.3x / 2.3x the
// algorithmic HBM traffic on 14x14 / 7x7: profiles/r01_pmc_traffic.txt).
//   G_rho[o, i] = sum_{n,u} dy[o,u] * x[i, u+rho-2]      v_mfma_f32_16x16x32: M = o, N = i (<= 16 long-axis positions), K = 32 rows
// of the per-channel stack [k = 2 + n*(Wl+2) + u][long-axis position], pitch 16 (vertical kernels are written transposed by the
// staging stores).  Both operands come from ds_read_b64_tr_b16; the +-2 shift of x is a row offset.
#include "mfma_common.h"

namespace slak {

typedef __attribute__((ext_vector_type(4))) float f32x4_w;

constexpr int SW_CB = 4;                // channels per workgroup (one per wave)
constexpr int SW_NCH = 2;               // staging chunks per thread per tensor per iteration (upper bound)
constexpr int SW_P = 16;                // stack pitch (elements)

struct SmallWgradParams {
    const void* dy; const void* x; float* partial;
    float* dw; unsigned* counters;   // in-kernel slice reduction (counters == NULL: partials only, reduce kernel follows)
    int N, C, H, W, kh, kw;
    int Wt, Wl, KL, padL;
    int NI;                // images per iteration
    int NKS;               // 32-deep k-steps per iteration
    int stack_rows;        // rows of one channel's dy stack (NKS*32); the x stack has 8 more
    int images_per_wg, slices;
    int nchunks;
};

template <typename T> __device__ __forceinline__ f32x4_w mfma16w(s16x8 a, s16x8 b, f32x4_w c);
template <> __device__ __forceinline__ f32x4_w mfma16w<bf16_t>(s16x8 a, s16x8 b, f32x4_w c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_w mfma16w<f16_t>(s16x8 a, s16x8 b, f32x4_w c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <typename T, int V, bool VERT>
__global__ __launch_bounds__(MF_THREADS) void dwconv_mfma_small_wgrad_kernel(const SmallWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int HW = p.H * p.W, blk = SW_CB * HW, ntap = p.kh * p.kw;
    const int dy_elems = p.stack_rows * SW_P, x_elems = (p.stack_rows + 8) * SW_P;      // per channel
    uint16_t* dys = lds;                                      // [SW_CB][stack_rows][16]
    uint16_t* xs = lds + SW_CB * dy_elems;                    // [SW_CB][stack_rows + 8][16]
    float* scr = (float*)(xs + SW_CB * x_elems);              // per wave: [16][32] fp32 scratch for the (skewed) diagonal sums
    float* res = scr + MF_WAVES * 16 * 32;                // per wave: [ntap] results (taps no diagonal reaches stay 0)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + SW_CB - 1) / SW_CB;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c0 = cb * SW_CB, c = c0 + wave;
    const uint16_t* __restrict__ gx = (const uint16_t*)p.x;
    const uint16_t* __restrict__ gdy = (const uint16_t*)p.dy;
    const int n_begin = slice * p.images_per_wg;
    int n_end = n_begin + p.images_per_wg; if (n_end > p.N) n_end = p.N;
    const int iters = (n_end > n_begin) ? (n_end - n_begin + p.NI - 1) / p.NI : 0;
    const int nch = (p.C - c0 < SW_CB) ? (p.C - c0) : SW_CB;
    const int valid_blk = nch * HW;

    int ch_ni[SW_NCH], ch_e0[SW_NCH];
    const int cpi = blk / V;
#pragma unroll
    for (int k = 0; k < SW_NCH; ++k) {
        const int idx = tid + k * MF_THREADS;
        const bool ok = idx < p.nchunks;
        const int ni = ok ? idx / cpi : 0;
        ch_ni[k] = ok ? ni : -1;
        ch_e0[k] = ok ? (idx - ni * cpi) * V : 0;
    }
    chunk_t<V> sx[SW_NCH], sd[SW_NCH];
    auto load_chunk = [&](const uint16_t* base, int e0) -> chunk_t<V> {
        chunk_t<V> r = chunk_zero<V>();
        if (e0 + V <= valid_blk) r = chunk_load<V>(base + e0);
        else for (int i = 0; i < V; ++i) if (e0 + i < valid_blk) chunk_set<V>(r, i, base[e0 + i]);
        return r;
    };
    auto prefetch = [&](int it) {
        const int n0 = n_begin + it * p.NI;
#pragma unroll
        for (int k = 0; k < SW_NCH; ++k) {
            sx[k] = chunk_zero<V>(); sd[k] = chunk_zero<V>();
            if (ch_ni[k] >= 0 && n0 + ch_ni[k] < n_end) {
                const size_t off = ((size_t)(n0 + ch_ni[k]) * p.C + c0) * HW;
                sx[k] = load_chunk(gx + off, ch_e0[k]); sd[k] = load_chunk(gdy + off, ch_e0[k]);
            }
        }
    };
    // element (channel ch, image ni, h, w) -> stack row 2 + ni*(Wl+2) + u, column = long-axis position; zeros for dead images.
    // The LDS offsets of this thread's elements do not change between iterations: resolved once (-1: beyond the block).
    int doff[SW_NCH][V], xoff[SW_NCH][V];
#pragma unroll
    for (int k = 0; k < SW_NCH; ++k) {
        int e = ch_e0[k];
        int ch = e / HW, rem = e - ch * HW;
        int h = rem / p.W, w = rem - h * p.W;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const bool ok = ch_ni[k] >= 0 && e + i < valid_blk;
            const int row = 2 + (ok ? ch_ni[k] : 0) * (p.Wl + 2) + (VERT ? w : h), col = VERT ? h : w;
            doff[k][i] = ok ? ch * dy_elems + row * SW_P + col : -1;
            xoff[k][i] = ok ? ch * x_elems + (row + 2) * SW_P + col : -1;
            ++w;
            if (w == p.W) { w = 0; ++h; if (h == p.H) { h = 0; ++ch; } }
        }
    }
    auto stage_write = [&]() {
#pragma unroll
        for (int k = 0; k < SW_NCH; ++k) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                if (doff[k][i] >= 0) {
                    dys[doff[k][i]] = chunk_get<V>(sd[k], i);
                    xs[xoff[k][i]] = chunk_get<V>(sx[k], i);
                }
            }
        }
    };

    if (iters > 0) prefetch(0);
    for (int i = tid; i < SW_CB * (dy_elems + x_elems) / 8; i += MF_THREADS) ((u32x4*)lds)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (iters > 0) stage_write();
    __syncthreads();

    f32x4_w acc[MF_TAPS];
#pragma unroll
    for (int g = 0; g < MF_TAPS; ++g) acc[g] = f32x4_w{0.f, 0.f, 0.f, 0.f};

    // tr-read addresses: 16-lane group grp reads a 4(k) x 16 block: lane supplies (row (i16>>2), 4-column chunk i16&3) and receives
    // column i16, rows +0..3.  For the 16x16x32 fragments lane l needs column l&15, k = (l>>4)*8 + e -> two reads (4 rows each).
    const int grp = lane >> 4, i16 = lane & 15;
    const int roff = (grp * 8 + (i16 >> 2)) * SW_P + (i16 & 3) * 4;
    const uint16_t* mydy = dys + wave * dy_elems;
    const uint16_t* myx = xs + wave * x_elems;

    for (int it = 0; it < iters; ++it) {
        if (it + 1 < iters) prefetch(it + 1);
        if (c < p.C) {
            for (int ks = 0; ks < p.NKS; ++ks) {
                const uint16_t* ap = mydy + ks * 32 * SW_P + roff;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, ap));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, ap + 4 * SW_P));
                const s16x8 a = s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int g = 0; g < MF_TAPS; ++g) {
                    const uint16_t* bp = myx + (ks * 32 + g) * SW_P + roff;
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, bp));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, bp + 4 * SW_P));
                    acc[g] = mfma16w<T>(a, s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]}, acc[g]);
                }
            }
        }
        __syncthreads();
        if (it + 1 < iters) stage_write();
        __syncthreads();
    }

    // ---- diagonal sums: dw[rho][tau] = sum_o G_rho[o, o + tau - padL]; D: column i = lane & 15, rows o = 4*(lane>>4) + reg.
    //      One tap per pass through a per-wave [16][32] fp32 scratch, written SKEWED (G[o][i] -> row o, column i - o + 15: a
    //      diagonal is a column; positions beyond the plane are never written and stay zero); lane dd adds the 16 rows of column
    //      dd in order (unconditional reads at immediate offsets). ----
    if (c < p.C) {
        float* tile = scr + wave * (16 * 32);
        float* myres = res + wave * ntap;
        for (int t = lane; t < ntap; t += 64) myres[t] = 0.f;
        for (int t = lane; t < 16 * 32 / 4; t += 64) ((u32x4*)tile)[t] = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int wofs[4]; bool wok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wok[r] = i16 < p.Wt && 4 * grp + r < p.Wt;
            wofs[r] = (4 * grp + r) * 32 + (i16 - (4 * grp + r) + 15);
        }
#pragma unroll
        for (int g = 0; g < MF_TAPS; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (wok[r]) tile[wofs[r]] = acc[g][r];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 31) {
                float v[16];
#pragma unroll
                for (int o = 0; o < 16; ++o) v[o] = tile[o * 32 + lane];
                float sum = 0.f;
#pragma unroll
                for (int o = 0; o < 16; ++o) sum += v[o];
                const int tau = lane - 15 + p.padL;
                if (tau >= 0 && tau < p.KL) myres[VERT ? (tau * p.kw + g) : (g * p.kw + tau)] = sum;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        float* out = p.partial + ((size_t)slice * p.C + c) * ntap;
        for (int t = lane; t < ntap; t += 64) wgrad_store_partial(&out[t], myres[t]);
    }
    if (p.counters) wgrad_finish(p.partial, p.dw, p.counters + cb, (int*)lds, p.slices, p.C, c0, nch, ntap, tid, MF_THREADS);
}

// ------------------------------------------------------------------------------------------------------------
static int sw_V(const ConvDims& d) { const int HW = d.H * d.W; return ((SW_CB * HW) % 8 == 0 && HW % 2 == 0) ? 8 : 4; }

static bool fill_sw_params(SmallWgradParams& p, const ConvDims& d, bool vert, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    if (p.Wt > 16 || p.Wl > 16) return false;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return false;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    const int HW = d.H * d.W, blk = SW_CB * HW, V = sw_V(d);
    if (blk % V) return false;
    int NI = (SW_NCH * MF_THREADS * V) / blk; if (NI < 1) return false;
    if (NI > 16) NI = 16;
    const int cblocks = (d.C + SW_CB - 1) / SW_CB;
    int slices = resident_wgs / cblocks; if (slices < 1) slices = 1;
    if (slices > d.N) slices = d.N;
    int per = (d.N + slices - 1) / slices;
    static const int min_iters = [] { const char* e = slak_dev_getenv("SLAK_SMALL_MIN_ITERS"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : v; }();
    if (per < min_iters * NI) per = min_iters * NI;                          // amortise the prologue / epilogue of a workgroup
    if (per > d.N) per = d.N;
    if (NI > per) NI = per;
    per = (per + NI - 1) / NI * NI;
    p.NI = NI; p.images_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.nchunks = NI * blk / V;
    p.NKS = (2 + NI * (p.Wl + 2) + 31) / 32;
    p.stack_rows = p.NKS * 32;
    return true;
}

static size_t sw_lds_bytes(const SmallWgradParams& p) {
    return (size_t)SW_CB * (2 * p.stack_rows + 8) * SW_P * 2 + (size_t)MF_WAVES * 16 * 32 * 4 + (size_t)MF_WAVES * p.kh * p.kw * 4 + 32;
}

bool dwconv_mfma_small_wgrad_supported(const ConvDims& d, int dy_dt, int x_dt) {
    if (dy_dt != x_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16)) return false;
    SmallWgradParams p;
    return fill_sw_params(p, d, d.kh > d.kw, 2048) && sw_lds_bytes(p) <= 60 * 1024;
}

size_t dwconv_mfma_small_wgrad_workspace(const ConvDims& d) {
    return align_up((size_t)(d.N < 2048 ? d.N : 2048) * d.C * d.kh * d.kw * sizeof(float), 256);
}

template <typename T, int V, bool VERT>
static int launch_sw_t(SmallWgradParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_small_wgrad_kernel<T, V, VERT>;
    static int resident = 0;
    fill_sw_params(p, d, VERT, 2048);
    const size_t lds0 = sw_lds_bytes(p);
    (void)slak_set_max_lds((const void*)k, lds0);
    if (resident == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, MF_THREADS, lds0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        resident = per_cu * mfma_cu_count();
    }
    fill_sw_params(p, d, VERT, resident);
    if ((size_t)p.slices * d.C * d.kh * d.kw * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    const int cblocks = (d.C + SW_CB - 1) / SW_CB;
    hipLaunchKernelGGL(k, dim3((unsigned)(cblocks * p.slices)), dim3(MF_THREADS), sw_lds_bytes(p), st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_small_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                   const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_small_wgrad_supported(d, dy_dt, x_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    const bool vert = d.kh > d.kw;
    SmallWgradParams p;
    p.dy = dy; p.x = x; p.partial = (float*)ws;
    p.dw = dw; p.counters = wgrad_arrival_counters((d.C + SW_CB - 1) / SW_CB);
    const int V = sw_V(d);
    int rc;
    if (x_dt == SLAK_BF16) {
        if (V == 8) rc = vert ? launch_sw_t<bf16_t, 8, true>(p, d, ws_bytes, st) : launch_sw_t<bf16_t, 8, false>(p, d, ws_bytes, st);
        else rc = vert ? launch_sw_t<bf16_t, 4, true>(p, d, ws_bytes, st) : launch_sw_t<bf16_t, 4, false>(p, d, ws_bytes, st);
    } else {
        if (V == 8) rc = vert ? launch_sw_t<f16_t, 8, true>(p, d, ws_bytes, st) : launch_sw_t<f16_t, 8, false>(p, d, ws_bytes, st);
        else rc = vert ? launch_sw_t<f16_t, 4, true>(p, d, ws_bytes, st) : launch_sw_t<f16_t, 4, false>(p, d, ws_bytes, st);
    }
    if (rc != SLAK_OK || p.counters) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, d.C * d.kh * d.kw, p.slices, st);
}

}  // namespace slak
