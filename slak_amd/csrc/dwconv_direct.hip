// slak_amd/csrc/dwconv_direct.hip -- fp32-exact direct depthwise conv for gfx950 (forward and data-grad).
//
// Replaces forward_fp32/fp16 + backward_data_fp32/fp16 of the reference extension
// (cutlass/examples/19_large_depthwise_conv2d_torch_extension/forward_fp32.cu:199-263,
//  backward_data_fp32.cu:199-263), which run a dense Toeplitz implicit GEMM on SIMT/Volta tensor cores.
// This is NOT that algorithm: it is a register-blocked direct convolution designed around three
// measured gfx950 facts (tools/valu_rate_probe.hip, run on MI355X):
//   * v_fmac_f32 retires 40.7 T lane-MAC/s, v_pk_fma_f32 66.9 -> every lane computes TWO planes of the
//     same channel at once (float2 accumulators, one broadcast weight), so the inner loop is v_pk_fma_f32;
//   * weights are wave-uniform (a workgroup owns one channel) -> they are fetched with scalar loads and
//     used as SGPR operands, costing no VGPRs and no LDS traffic;
//   * a whole (n,c) plane fits in LDS (<= 36 KB fp32 even at 96x96) -> planes are staged once, with the two
//     planes of a pair interleaved so that one ds_read_b64 yields one packed operand.
// Data-grad is the same kernel with the filter rotated by 180 degrees (odd kernels, "same" padding).
//
// Work decomposition.  "long axis" a = the axis of the longer filter side (H for Kx5, W for 5xK), extent A,
// KL taps; "short axis" b, extent B, KS taps.  One workgroup = channel c x G planes (G even).  A wave task =
// one strip of R consecutive positions along a, for 64 lanes spread over (plane-pair, b).  Because all lanes
// of a wave share the strip, the set of input rows that can touch it is wave-uniform: rows outside the
// image are skipped, not multiplied by zero (kernels exceed the map from stage 2 on: SURVEY.md Appendix B).
// Inner step ("chunk"): R input positions x R outputs = R*R packed FMAs against a (2R-1)-wide weight window.
#include "slak_common.h"

namespace slak {

constexpr int DR = 8;                    // outputs per lane along the long axis
constexpr int DIRECT_THREADS = 256;

static inline int klp_of(int KL) { return KL + 4 * DR; }

// ---------------------------------------------------------------------------------------------
// weight preparation: wp[c][js][q], q < KLp, wp[q] = wl[q - 2(R-1)] (zero outside [0,KL)), fp32.
// wl[t] = filter tap t along the long axis for short tap js; flipped in both axes for data-grad.
template <typename Tw>
__global__ void dwconv_prep_weights(const Tw* __restrict__ w, float* __restrict__ wp, int C, int kh, int kw,
                                    int long_is_h, int flip, int KLp) {
    const int KL = long_is_h ? kh : kw, KS = long_is_h ? kw : kh;
    const int total = C * KS * KLp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int q = i % KLp, js = (i / KLp) % KS, c = i / (KLp * KS);
        int t = q - 2 * (DR - 1);
        float v = 0.f;
        if (t >= 0 && t < KL) {
            int tt = flip ? (KL - 1 - t) : t, jj = flip ? (KS - 1 - js) : js;
            int r = long_is_h ? tt : jj, s = long_is_h ? jj : tt;
            v = to_f32(w[((size_t)c * kh + r) * kw + s]);
        }
        wp[i] = v;
    }
}

struct DirectParams {
    const void* x; const float* wp; void* y;
    int N, C, H, W;
    int KL, KS, KLp;
    int A, B, Ap, Bp, padL, padS;
    int Bb, nBands;           // band width along the short axis (== B unless a plane pair does not fit in LDS)
    int G, npairs;            // planes per workgroup (even) and pairs
    int SA, SB;               // LDS strides (in float2) along a and along (pair,b)
    int groups_per_channel;
    int tile2;                // float2 elements in the tile
};

template <typename Tin, typename Tout, bool LONG_H>
__global__ __launch_bounds__(DIRECT_THREADS) void dwconv_direct_kernel(const Tin* __restrict__ x, const float* __restrict__ wp,
                                                                     Tout* __restrict__ y, const DirectParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // float2 tile, viewed as floats
    const int c = blockIdx.x % p.C;
    const int rest = blockIdx.x / p.C;
    const int grp = rest % p.groups_per_channel;
    const int band = rest / p.groups_per_channel;
    const int n0 = grp * p.G;
    const int b0 = band * p.Bb;                                   // first short-axis coordinate of this band
    const int bw = (p.B - b0 < p.Bb) ? (p.B - b0) : p.Bb;         // its width
    const int tid = threadIdx.x;
    const int HW = p.H * p.W;

    // ---- stage: zero the tile, then scatter the G planes (coalesced along w) -------------------
    {
        float4* z = (float4*)smem;
        const int n4 = (p.tile2 * 2 + 3) / 4;
        for (int i = tid; i < n4; i += DIRECT_THREADS) z[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    {
        // rows x cols of the plane that this band needs (band + halo of padS along the short axis), w fastest
        const int blo = (b0 - p.padS > 0) ? (b0 - p.padS) : 0;
        const int bhi = (b0 + bw + p.padS < p.B) ? (b0 + bw + p.padS) : p.B;
        const int h_lo = LONG_H ? 0 : blo, h_n = LONG_H ? p.H : (bhi - blo);
        const int w_lo = LONG_H ? blo : 0, w_n = LONG_H ? (bhi - blo) : p.W;
        const int per_plane = h_n * w_n;
        const int total = p.G * per_plane;
        for (int e = tid; e < total; e += DIRECT_THREADS) {
            const int pl = e / per_plane, rem = e - pl * per_plane;
            const int n = n0 + pl;
            if (n < p.N) {
                const int hh = rem / w_n, h = h_lo + hh, w = w_lo + (rem - hh * w_n);
                const int a = LONG_H ? h : w, b = LONG_H ? w : h;
                const float v = to_f32(x[((size_t)n * p.C + c) * HW + h * p.W + w]);
                const int pp = pl >> 1, half = pl & 1;
                smem[2 * (a * p.SA + (pp * p.Bp + (b - b0) + p.padS) * p.SB) + half] = v;
            }
        }
    }
    __syncthreads();

    // ---- compute ---------------------------------------------------------------------------------
    const float2* __restrict__ tile = (const float2*)smem;
    const int lane = tid & 63;
    const int wave = wave_id_uniform();
    const int nwaves = DIRECT_THREADS / 64;
    const int nStrips = p.Ap / DR;
    const int lanesTotal = p.npairs * bw;
    const int nLaneChunks = (lanesTotal + 63) >> 6;
    const int nTasks = nStrips * nLaneChunks;
    const int backC = (p.padL + DR - 1) / DR;                 // chunks that can reach a strip from below
    const int fwdC = (p.KL - 1 - p.padL + DR - 1) / DR;        // ... and from above

    for (int task = wave; task < nTasks; task += nwaves) {
        const int s = task / nLaneChunks;                      // wave-uniform strip
        const int q = task - s * nLaneChunks;
        const int li = q * 64 + lane;
        const bool lane_ok = li < lanesTotal;
        const int lic = lane_ok ? li : 0;
        const int pp = lic / bw, bl = lic - pp * bw, b = b0 + bl;
        const int base2 = (pp * p.Bp + bl) * p.SB;
        int clo = s - backC; if (clo < 0) clo = 0;
        int chi = s + fwdC;  if (chi > nStrips - 1) chi = nStrips - 1;

        float2 acc[DR];
#pragma unroll
        for (int r = 0; r < DR; ++r) acc[r] = float2{0.f, 0.f};

        for (int js = 0; js < p.KS; ++js) {
            const float* __restrict__ wrow = wp + ((size_t)c * p.KS + js) * p.KLp + (DR - 1);
            const float2* __restrict__ tcol = tile + base2 + js * p.SB;
            for (int ci = clo; ci <= chi; ++ci) {
                const int d = (ci - s) * DR + p.padL;            // tap index when i == r
                const float* __restrict__ wq = wrow + d;         // wq[i - r + R - 1] == tap (d + i - r)
                float wv[2 * DR - 1];
#pragma unroll
                for (int m = 0; m < 2 * DR - 1; ++m) wv[m] = wq[m];   // wave-uniform -> scalar loads
                float2 xv[DR];
#pragma unroll
                for (int i = 0; i < DR; ++i) xv[i] = tcol[(ci * DR + i) * p.SA];
#pragma unroll
                for (int i = 0; i < DR; ++i) {
#pragma unroll
                    for (int r = 0; r < DR; ++r) {
                        const float wt = wv[i - r + DR - 1];
                        acc[r].x = __builtin_fmaf(wt, xv[i].x, acc[r].x);
                        acc[r].y = __builtin_fmaf(wt, xv[i].y, acc[r].y);
                    }
                }
            }
        }

        // ---- store ---------------------------------------------------------------------------------
        if (lane_ok) {
            const int na = n0 + 2 * pp, nb = na + 1;
            const int a0 = s * DR;
#pragma unroll
            for (int r = 0; r < DR; ++r) {
                const int a = a0 + r;
                if (a < p.A) {
                    const int off = LONG_H ? (a * p.W + b) : (b * p.W + a);
                    if (na < p.N) y[((size_t)na * p.C + c) * HW + off] = from_f32<Tout>(acc[r].x);
                    if (nb < p.N) y[((size_t)nb * p.C + c) * HW + off] = from_f32<Tout>(acc[r].y);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
static void fill_params(DirectParams& p, const ConvDims& d, bool long_h, int lds_budget_bytes, int hard_lds_bytes = 96 * 1024) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W;
    p.KL = long_h ? d.kh : d.kw; p.KS = long_h ? d.kw : d.kh;
    p.KLp = klp_of(p.KL);
    p.A = long_h ? d.H : d.W; p.B = long_h ? d.W : d.H;
    p.Ap = ceil_div(p.A, DR) * DR;
    p.padL = p.KL / 2; p.padS = p.KS / 2;
    const int aps = p.Ap | 1;                                    // odd stride -> conflict-free ds_read_b64 across b
    // band width: the whole short axis unless one plane pair would not fit in LDS
    p.Bb = p.B;
    auto tile2_for = [&](int G, int Bb) { int np = G / 2, Bp = Bb + 2 * p.padS; return long_h ? p.Ap * np * Bp : np * Bp * aps; };
    while (p.Bb > 1 && tile2_for(2, p.Bb) * 8 > hard_lds_bytes) p.Bb = (p.Bb + 1) / 2;
    p.nBands = ceil_div(p.B, p.Bb);
    p.Bp = p.Bb + 2 * p.padS;
    // planes per workgroup: aim at >= 2 full waves of lanes, stay inside the LDS budget, never exceed N
    int G = 2;
    const int Ncap = (d.N + 1) & ~1;
    while (G + 2 <= Ncap && tile2_for(G + 2, p.Bb) * 8 <= lds_budget_bytes && (G / 2) * p.Bb < 128) G += 2;
    p.G = G; p.npairs = G / 2;
    if (long_h) { p.SA = p.npairs * p.Bp; p.SB = 1; } else { p.SA = 1; p.SB = aps; }
    p.tile2 = tile2_for(G, p.Bb);
    p.groups_per_channel = ceil_div(d.N, G);
}

size_t dwconv_direct_workspace(const ConvDims& d) {
    const int KL = d.kh >= d.kw ? d.kh : d.kw, KS = d.kh >= d.kw ? d.kw : d.kh;
    return align_up((size_t)d.C * KS * klp_of(KL) * sizeof(float), 256);
}

template <typename Tin, typename Tout>
static int launch_typed(const DirectParams& p, bool long_h, hipStream_t st) {
    const size_t lds = (size_t)p.tile2 * 8 + 16;
    dim3 grid((unsigned)(p.C * p.groups_per_channel * p.nBands));
    if (long_h) {
        auto k = dwconv_direct_kernel<Tin, Tout, true>;
        (void)slak_set_max_lds((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(DIRECT_THREADS), lds, st, (const Tin*)p.x, p.wp, (Tout*)p.y, p);
    } else {
        auto k = dwconv_direct_kernel<Tin, Tout, false>;
        (void)slak_set_max_lds((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(DIRECT_THREADS), lds, st, (const Tin*)p.x, p.wp, (Tout*)p.y, p);
    }
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename Tin>
static int launch_in(int y_dt, const DirectParams& p, bool long_h, hipStream_t st) {
    switch (y_dt) {
        case SLAK_F32:  return launch_typed<Tin, float>(p, long_h, st);
        case SLAK_F16:  return launch_typed<Tin, f16_t>(p, long_h, st);
        case SLAK_BF16: return launch_typed<Tin, bf16_t>(p, long_h, st);
    }
    return SLAK_ERR_INVALID_ARG;
}

int launch_dwconv_direct(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                         const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws == nullptr || ws_bytes < dwconv_direct_workspace(d)) return SLAK_ERR_WORKSPACE;
    const bool long_h = d.kh >= d.kw;
    DirectParams p;
    fill_params(p, d, long_h, 40 * 1024);
    if ((size_t)p.tile2 * 8 + 16 > 150 * 1024) return SLAK_ERR_UNSUPPORTED;   // cannot happen: bands shrink until it fits
    p.x = x; p.y = y; p.wp = (const float*)ws;

    const int total = d.C * p.KS * p.KLp;
    const int pb = 256, pg = ceil_div(total, pb) < 1024 ? ceil_div(total, pb) : 1024;
    switch (w_dt) {
        case SLAK_F32:  hipLaunchKernelGGL(dwconv_prep_weights<float>,  dim3(pg), dim3(pb), 0, st, (const float*)w,  (float*)ws, d.C, d.kh, d.kw, (int)long_h, (int)flip_filter, p.KLp); break;
        case SLAK_F16:  hipLaunchKernelGGL(dwconv_prep_weights<f16_t>,  dim3(pg), dim3(pb), 0, st, (const f16_t*)w,  (float*)ws, d.C, d.kh, d.kw, (int)long_h, (int)flip_filter, p.KLp); break;
        case SLAK_BF16: hipLaunchKernelGGL(dwconv_prep_weights<bf16_t>, dim3(pg), dim3(pb), 0, st, (const bf16_t*)w, (float*)ws, d.C, d.kh, d.kw, (int)long_h, (int)flip_filter, p.KLp); break;
        default: return SLAK_ERR_INVALID_ARG;
    }
    SLAK_LAUNCH_CHECK();
    switch (x_dt) {
        case SLAK_F32:  return launch_in<float>(y_dt, p, long_h, st);
        case SLAK_F16:  return launch_in<f16_t>(y_dt, p, long_h, st);
        case SLAK_BF16: return launch_in<bf16_t>(y_dt, p, long_h, st);
    }
    return SLAK_ERR_INVALID_ARG;
}

}  // namespace slak
