// slak_amd/csrc/dwconv_wgrad.hip -- fp32-exact depthwise-conv weight gradient for gfx950.
//
// Replaces backward_filter_fp32/fp16 of the reference extension
// (cutlass/examples/19_large_depthwise_conv2d_torch_extension/backward_filter_fp32.cu:199-263), which
// forms a PQ x HW correlation matrix per channel on the tensor cores and atomically adds its diagonals
// into the taps (cutlass/include/cutlass/epilogue/threadblock/dwconv2d_direct_epilogue_simt.h:144-186).
// Here each tap is what it is -- a dot product over (n, h, w):
//   * a workgroup owns (channel c, batch slice); it stages G planes of x and dy in LDS (two planes
//     interleaved per float2, as in dwconv_direct.hip, so the inner op is v_pk_fma_f32);
//   * each wave owns a set of (short tap js, chunk of T long taps) and keeps those T accumulators in
//     registers across ALL planes of the slice: lanes hold partial sums over their pixels;
//   * one wavefront-wide butterfly reduction per accumulator at the very end, one plain store per tap
//     into partial[slice][c][tap]; a second tiny kernel sums the slices in a fixed order.
//   No atomics, no memset, bitwise run-to-run reproducible.  Output is always fp32
//   (backward_filter_fp16.cu:187).
#include "slak_common.h"
#include <stdlib.h>
#include <mutex>

namespace slak {

constexpr int WR = 8;                    // pixels per lane along the long axis (strip)
constexpr int WGRAD_MAX_WAVES = 16;

struct WgradParams {
    const void* dy; const void* x; float* partial;
    int N, C, H, W, kh, kw;
    int KL, KS, A, B, Ap, Bp, padL, padS;
    int Bb, nBands;           // band width along the short axis (== B unless the tiles do not fit in LDS)
    int G, npairs;
    int SAx, SBx, SAd, SBd;   // LDS strides (float2 units) of the x tile and the dy tile
    int xtile2, dtile2;
    int nslices, planes_per_slice;
    int nTapChunks, nJobs, nBatches, jobsPerBatch;   // job = (short tap js, chunk of T long taps); one per wave
    int wpad;                 // zero rows on both ends of the x tile along the long axis (= WR + T)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename Tdy, typename Tx, bool LONG_H, int T>
__global__ __launch_bounds__(WGRAD_MAX_WAVES * 64) void dwconv_wgrad_kernel(const Tdy* __restrict__ dy, const Tx* __restrict__ x,
                                                                          float* __restrict__ partial, const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                // x tile (float2 view), long axis padded by wpad each side
    float* ds = smem + 2 * p.xtile2;                 // dy tile
    const int c = blockIdx.x % p.C;
    const int slice = (blockIdx.x / p.C) % p.nslices;
    const int batch = blockIdx.x / (p.C * p.nslices);
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthreads = blockDim.x;
    const int wave = wave_id_uniform();
    const int HW = p.H * p.W;

    const int nStrips = p.Ap / WR;
    const int job = batch * p.jobsPerBatch + wave;                    // wave-uniform
    const bool has_job = wave < p.jobsPerBatch && job < p.nJobs;
    const int js = has_job ? job / p.nTapChunks : 0;
    const int tc = has_job ? job - js * p.nTapChunks : 0;
    const int t0 = tc * T;

    float2 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = float2{0.f, 0.f};

    const int n_begin = slice * p.planes_per_slice;
    int n_end = n_begin + p.planes_per_slice; if (n_end > p.N) n_end = p.N;

    for (int n0 = n_begin; n0 < n_end; n0 += p.G) {
      for (int band = 0; band < p.nBands; ++band) {
        const int b0 = band * p.Bb;
        const int bw = (p.B - b0 < p.Bb) ? (p.B - b0) : p.Bb;
        __syncthreads();                              // previous tile's readers are done
        {
            float4* z = (float4*)smem;
            const int n4 = (2 * (p.xtile2 + p.dtile2) + 3) / 4;
            for (int i = tid; i < n4; i += nthreads) z[i] = float4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        {
            // x: band + halo of padS along the short axis; dy: the band itself.  w fastest (coalesced).
            const int blo = (b0 - p.padS > 0) ? (b0 - p.padS) : 0;
            const int bhi = (b0 + bw + p.padS < p.B) ? (b0 + bw + p.padS) : p.B;
            const int h_lo = LONG_H ? 0 : blo, h_n = LONG_H ? p.H : (bhi - blo);
            const int w_lo = LONG_H ? blo : 0, w_n = LONG_H ? (bhi - blo) : p.W;
            const int per_plane = h_n * w_n;
            const int total = p.G * per_plane;
            for (int e = tid; e < total; e += nthreads) {
                const int pl = e / per_plane, rem = e - pl * per_plane;
                const int n = n0 + pl;
                if (n < n_end) {
                    const int hh = rem / w_n, h = h_lo + hh, w = w_lo + (rem - hh * w_n);
                    const int a = LONG_H ? h : w, b = LONG_H ? w : h;
                    const size_t g = ((size_t)n * p.C + c) * HW + h * p.W + w;
                    const int pp = pl >> 1, half = pl & 1;
                    xs[2 * ((a + p.wpad) * p.SAx + (pp * p.Bp + (b - b0) + p.padS) * p.SBx) + half] = to_f32(x[g]);
                    if (b >= b0 && b < b0 + bw)
                        ds[2 * (a * p.SAd + (pp * p.Bb + (b - b0)) * p.SBd) + half] = to_f32(dy[g]);
                }
            }
        }
        __syncthreads();
        if (!has_job) continue;

        const int lanesTotal = p.npairs * bw;
        const int nLaneChunks = (lanesTotal + 63) >> 6;
        const float2* __restrict__ xt = (const float2*)xs;
        const float2* __restrict__ dt = (const float2*)ds;
        for (int s = 0; s < nStrips; ++s) {
            const int a0 = s * WR;
            // x window for this strip and tap chunk: a' = a0 + t0 - padL + i, i in [0, WR+T-1)
            const int lo = a0 + t0 - p.padL;
            if (lo + WR + T - 2 < 0 || lo >= p.A) continue;          // entirely outside the image (uniform)
            for (int q = 0; q < nLaneChunks; ++q) {
                const int li = q * 64 + lane;
                const bool ok = li < lanesTotal;
                const int lic = ok ? li : 0;
                const int pp = lic / bw, bl = lic - pp * bw;
                const float2* __restrict__ xc = xt + (lo + p.wpad) * p.SAx + (pp * p.Bp + bl + js) * p.SBx;
                const float2* __restrict__ dc = dt + a0 * p.SAd + (pp * p.Bb + bl) * p.SBd;
                float2 dv[WR];
#pragma unroll
                for (int r = 0; r < WR; ++r) { float2 v = dc[r * p.SAd]; dv[r] = ok ? v : float2{0.f, 0.f}; }
                float2 xv[WR + T - 1];
#pragma unroll
                for (int i = 0; i < WR + T - 1; ++i) xv[i] = xc[i * p.SAx];
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int r = 0; r < WR; ++r) {
                        acc[t].x = __builtin_fmaf(dv[r].x, xv[r + t].x, acc[t].x);
                        acc[t].y = __builtin_fmaf(dv[r].y, xv[r + t].y, acc[t].y);
                    }
            }
        }
      }
    }

    // ---- wavefront-wide reduction, one plain store per tap ---------------------------------------
    if (has_job) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float v = wave_sum(acc[t].x + acc[t].y);
            const int tap = t0 + t;
            if (lane == 0 && tap < p.KL) {
                const int r = LONG_H ? tap : js, s_ = LONG_H ? js : tap;
                partial[((size_t)slice * p.C + c) * (p.kh * p.kw) + r * p.kw + s_] = v;
            }
        }
    }
}

__global__ void dwconv_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, int total, int nslices) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslices; ++k) s += partial[(size_t)k * total + i];   // fixed order
        dw[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
static int wgrad_slices(const ConvDims& d) {
    // enough workgroups to cover 256 CUs twice, but at least 2 planes per slice
    int want = ceil_div(640, d.C);
    int maxs = ceil_div(d.N, 2);
    int s = want < maxs ? want : maxs;
    return s < 1 ? 1 : s;
}

static int wgrad_T(const ConvDims& d) { int KL = d.kh >= d.kw ? d.kh : d.kw; return KL >= 9 ? 16 : 8; }

static bool fill_wparams(WgradParams& p, const ConvDims& d, int lds_budget) {
    const bool long_h = d.kh >= d.kw;
    const int T = wgrad_T(d);
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.KL = long_h ? d.kh : d.kw; p.KS = long_h ? d.kw : d.kh;
    p.A = long_h ? d.H : d.W; p.B = long_h ? d.W : d.H;
    p.Ap = ceil_div(p.A, WR) * WR;
    p.padL = p.KL / 2; p.padS = p.KS / 2;
    p.wpad = WR + T;
    p.nTapChunks = ceil_div(p.KL, T);
    p.nJobs = p.KS * p.nTapChunks;
    p.nBatches = ceil_div(p.nJobs, WGRAD_MAX_WAVES);
    p.jobsPerBatch = ceil_div(p.nJobs, p.nBatches);
    p.nslices = wgrad_slices(d);
    p.planes_per_slice = ceil_div(ceil_div(d.N, p.nslices), 2) * 2;
    p.nslices = ceil_div(d.N, p.planes_per_slice);
    const int ax = (p.Ap + 2 * p.wpad) | 1, ad = p.Ap | 1;
    auto sizes = [&](int G, int Bb, int& xt, int& dt) {
        int np = G / 2, Bp = Bb + 2 * p.padS;
        xt = long_h ? (p.Ap + 2 * p.wpad) * np * Bp : np * Bp * ax;
        dt = long_h ? p.Ap * np * Bb : np * Bb * ad;
    };
    int xt, dt;
    p.Bb = p.B;
    while (true) {                                   // shrink the band until one plane pair fits
        sizes(2, p.Bb, xt, dt);
        if ((xt + dt) * 8 <= 96 * 1024 || p.Bb == 1) break;
        p.Bb = (p.Bb + 1) / 2;
    }
    p.nBands = ceil_div(p.B, p.Bb);
    p.Bp = p.Bb + 2 * p.padS;
    int G = 2;
    while (true) {
        sizes(G + 2, p.Bb, xt, dt);
        if (G + 2 > p.planes_per_slice || (xt + dt) * 8 > lds_budget || (G / 2) * p.Bb >= 128) break;
        G += 2;
    }
    sizes(G, p.Bb, xt, dt);
    p.G = G; p.npairs = G / 2; p.xtile2 = xt; p.dtile2 = dt;
    if (long_h) { p.SAx = p.npairs * p.Bp; p.SBx = 1; p.SAd = p.npairs * p.Bb; p.SBd = 1; }
    else        { p.SAx = 1; p.SBx = ax; p.SAd = 1; p.SBd = ad; }
    return (size_t)(xt + dt) * 8 + 16 <= 150 * 1024;
}

size_t dwconv_wgrad_workspace(const ConvDims& d) {
    WgradParams p;
    fill_wparams(p, d, 64 * 1024);
    return align_up((size_t)p.nslices * d.C * d.kh * d.kw * sizeof(float), 256);
}

template <typename Tdy, typename Tx, bool LONG_H, int T>
static int wlaunch2(const WgradParams& p, hipStream_t st) {
    const size_t lds = (size_t)(p.xtile2 + p.dtile2) * 8 + 16;
    dim3 grid((unsigned)(p.C * p.nslices * p.nBatches));
    dim3 block((unsigned)(64 * p.jobsPerBatch));
    auto k = dwconv_wgrad_kernel<Tdy, Tx, LONG_H, T>;
    (void)slak_set_max_lds((const void*)k, lds);
    hipLaunchKernelGGL(k, grid, block, lds, st, (const Tdy*)p.dy, (const Tx*)p.x, p.partial, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename Tdy, typename Tx>
static int wlaunch(const WgradParams& p, bool long_h, int T, hipStream_t st) {
    if (long_h) return T == 16 ? wlaunch2<Tdy, Tx, true, 16>(p, st) : wlaunch2<Tdy, Tx, true, 8>(p, st);
    return T == 16 ? wlaunch2<Tdy, Tx, false, 16>(p, st) : wlaunch2<Tdy, Tx, false, 8>(p, st);
}

int launch_dwconv_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                        const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st) {
    WgradParams p;
    if (!fill_wparams(p, d, 64 * 1024)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr || ws_bytes < dwconv_wgrad_workspace(d)) return SLAK_ERR_WORKSPACE;
    if (dy_dt != x_dt) return SLAK_ERR_UNSUPPORTED;       // autograd always hands matching dtypes
    p.dy = dy; p.x = x; p.partial = (float*)ws;
    const bool long_h = d.kh >= d.kw;
    const int T = wgrad_T(d);
    int rc;
    switch (x_dt) {
        case SLAK_F32:  rc = wlaunch<float, float>(p, long_h, T, st); break;
        case SLAK_F16:  rc = wlaunch<f16_t, f16_t>(p, long_h, T, st); break;
        case SLAK_BF16: rc = wlaunch<bf16_t, bf16_t>(p, long_h, T, st); break;
        default: return SLAK_ERR_INVALID_ARG;
    }
    if (rc != SLAK_OK) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, d.C * d.kh * d.kw, p.nslices, st);
}

unsigned* wgrad_arrival_counters(int ngroups) {
    // SLOTS launches may be in flight at once before a slot's counters are handed out again (a stream serialises its own launches;
    // the slots only matter for weight-gradient launches running concurrently on different streams)
    constexpr int SLOTS = 64, MAXG = 4096, MAXDEV = 16;
    static std::mutex mu;
    static unsigned* buf[MAXDEV] = {};
    static unsigned next[MAXDEV] = {};
    if (ngroups > MAXG) return nullptr;
    static const bool force_reduce = [] { const char* e = getenv("SLAK_WGRAD_REDUCE"); return e && e[0] == '1'; }();
    if (force_reduce) return nullptr;                 // SLAK_WGRAD_REDUCE=1: partials + a separate reduce launch (A/B and debugging)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!buf[dev]) {
        unsigned* b = nullptr;
        if (hipMalloc((void**)&b, (size_t)SLOTS * MAXG * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipMemset(b, 0, (size_t)SLOTS * MAXG * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(b); return nullptr; }
        buf[dev] = b;
    }
    return buf[dev] + (size_t)(next[dev]++ % SLOTS) * MAXG;
}

int launch_wgrad_reduce(const float* partial, float* dw, int total, int nslices, hipStream_t st) {
    hipLaunchKernelGGL(dwconv_wgrad_reduce, dim3(ceil_div(total, 256)), dim3(256), 0, st, partial, dw, total, nslices);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

}  // namespace slak
