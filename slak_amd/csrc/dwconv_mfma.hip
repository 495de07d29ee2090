// slak_amd/csrc/dwconv_mfma.hip -- matrix-core (MFMA) depthwise conv for gfx950: forward and data-grad of the
// SLaK branches (Kx5, 5xK, 5x5) for 16-bit activations.
//
// Replaces forward_fp16 / backward_data_fp16 of the reference extension
// (cutlass/examples/19_large_depthwise_conv2d_torch_extension/forward_fp16.cu:186-249,
//  backward_data_fp16.cu:184-246).  Those run a [batch x HW] x [HW x PQ] implicit GEMM whose filter operand
// is a 2-D Toeplitz expansion (8.6 % useful MACs for 51x5 on 56x56).  This is a different
// factorisation, chosen because the VALU cannot reach the HBM roofline at 127 flop/B (tools/valu_rate_probe:
// 67-75 T lane-MAC/s => >= 130 us for a pass whose HBM time is 24 us):
//
//   long axis t (extent Wt, KL taps, pad padL)      short axis l (extent Wl, 5 taps, pad 2)
//   Z_r[o, u] = sum_i  T_r[o, i] * X[i, u]          one SMALL DENSE 1-D Toeplitz GEMM per short tap r:
//                                                   A = T_r (Wt x Wt, T_r[o,i] = w[r][i-o+padL]), B = the plane
//                                                   itself; N runs over the short axis AND over the batch.
//   Y[o, u]   = sum_r  Z_r[o, u + r - 2]            5 shifted adds ALONG THE LANE AXIS of the accumulators
//                                                   (v_add_f32 + DPP wave_shr/wave_shl, Horner form).
// Because SLaK's long kernels are as long as the map (51 on 56, 49 on 28, 47 on 14, 13 on 7), T_r is dense:
// nothing is wasted on a band that is not there, and every B fragment read from LDS feeds 5*MT MFMAs.
// A 32-lane tile carries 2 halo lanes per side (28 useful: 56 = 2 x 28, 28 = 1 x 28 exactly); small planes
// are packed several per tile with 2 zero lanes between them (2 x 14, 3 x 7), and short Toeplitz axes pack
// several taps r into the 32 MFMA rows (2 x 16 for 14, 4 x 8 for 7).
//   * horizontal kernels (5xK): t = W (contiguous) -> B fragments are plain ds_read_b128 rows;
//   * vertical kernels (Kx5): t = H -> B fragments come from ds_read_b64_tr_b16 (LDS transpose read).
// Data-grad is the same kernel with the filter rotated by 180 degrees.  Weights are rounded to the
// activation dtype for the MFMA (what autocast does to an nn.Conv2d weight); accumulation is fp32.
#include "mfma_common.h"

namespace slak {

struct MfmaFwdParams {
    const void* x; const uint16_t* frags; void* y;
    int N, C, H, W, kh, kw, flip;
    int Wt, Wl, KL, padL;
    int G;                 // planes staged per iteration
    int ppt;               // planes per 32-lane tile (0: one plane spans ntiles tiles of 28 useful lanes)
    int TS;                // lane-axis stride between tiles
    int ntiles;            // lane tiles per iteration
    int P;                 // LDS pitch (elements) of the staged stack
    int in_elems;          // LDS elements of the stack (multiple of 8)
    int HWp;               // out-buffer plane pitch (elements, multiple of 8)
    int planes_per_wg, slices;
    int nchunks, cpp, cpr; // staging chunks per iteration / per plane / per row
};

__global__ void toeplitz_pack_kernel(const ToeplitzPackParams p) {
    const int total = p.C * p.MT * p.NG * p.KS * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63;
    int f = idx >> 6;
    const int ks = f % p.KS; f /= p.KS;
    const int g = f % p.NG; f /= p.NG;
    const int mt = f % p.MT; const int c = f / p.MT;
    const int MPAD = 32 / p.RPM, l31 = lane & 31, lhi = lane >> 5;
    const int r = g * p.RPM + l31 / MPAD, o_abs = mt * 32 + (l31 % MPAD);
    const float* wc = p.w + (size_t)c * p.kh * p.kw;
    // all 8 loads are issued unconditionally (clamped index) and masked afterwards, so they overlap
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i_abs = ks * 16 + lhi * 8 + e;
        int t = i_abs - o_abs + p.padL;
        const bool ok = r < MF_TAPS && o_abs < p.Wt && i_abs < p.Wt && t >= 0 && t < p.KL;
        int rr = r < MF_TAPS ? r : 0;
        t = ok ? t : 0;
        if (p.flip) { t = p.KL - 1 - t; rr = MF_TAPS - 1 - rr; }
        const float wv = p.vert ? wc[t * p.kw + rr] : wc[rr * p.kw + t];
        v[e] = ok ? wv : 0.f;
    }
    u32x4 out;
#pragma unroll
    for (int e = 0; e < 8; e += 2)
        out[e >> 1] = p.is_bf16 ? pack2<bf16_t>(v[e], v[e + 1]) : pack2<f16_t>(v[e], v[e + 1]);
    ((u32x4*)p.frags)[idx] = out;
}

void launch_toeplitz_pack(const ToeplitzPackParams& p, hipStream_t st) {
    const int total = p.C * p.MT * p.NG * p.KS * 64;
    hipLaunchKernelGGL(toeplitz_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, p);
}

// MT: 32-row tiles along the Toeplitz axis (wave w owns tile w % MT); KS: 16-deep k-steps (Wt <= 16*KS);
// RPM: short taps packed per MFMA (32/RPM rows each); V: staging vector width (elements); VERT: long axis = H.
template <typename T, int MT, int KS, int RPM, int V, bool VERT>
__global__ __launch_bounds__(MF_THREADS, 2) void dwconv_mfma_fwd_kernel(const MfmaFwdParams p) {
    constexpr int NG = (MF_TAPS + RPM - 1) / RPM;          // accumulators (MFMA groups) per unit
    constexpr int MPAD = 32 / RPM;                          // rows per tap inside an MFMA
    constexpr int NR = 16 / RPM;                            // output registers per lane
    constexpr int WL = MF_WAVES / MT;                       // lane-tile workers per Toeplitz tile
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* lin = lds;                                    // staged stack
    uint16_t* lout = lds + p.in_elems;                      // [G][HWp] results

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = wave % MT, wl = wave / MT;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int HW = p.H * p.W;
    const uint16_t* __restrict__ x = (const uint16_t*)p.x;
    uint16_t* __restrict__ y = (uint16_t*)p.y;

    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    if (n_begin >= n_end) return;
    const int iters = (n_end - n_begin + p.G - 1) / p.G;

    // ---- per-thread staging map (identical every iteration) -------------------------------------
    int goff[MF_NCH], loff[MF_NCH], ooff[MF_NCH], jpl[MF_NCH];
#pragma unroll
    for (int k = 0; k < MF_NCH; ++k) {
        const int idx = tid + k * MF_THREADS;
        const bool ok = idx < p.nchunks;
        const int j = ok ? idx / p.cpp : 0, rem = ok ? idx - j * p.cpp : 0;
        const int h = rem / p.cpr, w0 = (rem - h * p.cpr) * V;
        const int u0 = (p.ppt ? (j / p.ppt) * p.TS + (j % p.ppt) * (p.Wl + 2) : 0) + 2;
        jpl[k] = ok ? j : -1;
        goff[k] = j * p.C * HW + rem * V;
        ooff[k] = j * p.HWp + rem * V;
        loff[k] = VERT ? (h * p.P + u0 + w0) : ((u0 + h) * p.P + w0);
    }
    chunk_t<V> st[MF_NCH];
    auto prefetch = [&](int it) {
        const int n0 = n_begin + it * p.G;
        const uint16_t* base = x + ((size_t)n0 * p.C + c) * HW;
#pragma unroll
        for (int k = 0; k < MF_NCH; ++k) {
            if (jpl[k] >= 0 && n0 + jpl[k] < n_end) st[k] = chunk_load<V>(base + goff[k]);
            else st[k] = chunk_zero<V>();
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int k = 0; k < MF_NCH; ++k) {
            if (jpl[k] >= 0) {
                if constexpr (VERT) chunk_store_lds_a4<V>(lin + loff[k], st[k]);
                else chunk_store<V>(lin + loff[k], st[k]);
            }
        }
    };

    prefetch(0);
    // ---- zero the stack (pads stay zero for the whole kernel), fetch this channel's filter --------
    {
        u32x4* z = (u32x4*)lin;
        for (int i = tid; i < p.in_elems / 8; i += MF_THREADS) z[i] = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    stage_write();

    // ---- Toeplitz fragments: A[g][ks] holds, for MFMA row m = l31 -> (tap r = g*RPM + m/MPAD, o = m%MPAD) and
    //      k = ks*16 + lhi*8 + e -> i, the weight w[r][i - o + padL] (0 outside the filter or the plane) ----
    s16x8 afrag[NG][KS];
    bool ks_active[KS];
    load_toeplitz_frags<NG, KS>(afrag, ks_active, p.frags, c, MT, mt, lane, MPAD, p.Wt, p.KL, p.padL);
    __syncthreads();

    for (int it = 0; it < iters; ++it) {
        const int n0 = n_begin + it * p.G;
        if (it + 1 < iters) prefetch(it + 1);

        for (int tile = wl; tile < p.ntiles; tile += WL) {
            const int ustart = tile * p.TS;
            f32x16 acc[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (!ks_active[ks]) continue;
                s16x8 b;
                if constexpr (!VERT) {
                    b = *(const s16x8*)(lin + (ustart + l31) * p.P + ks * 16 + lhi * 8);
                } else {
                    // ds_read_b64_tr_b16: the 16 lanes of group g = lane>>4 read a 4(k) x 16(u) block; lane gets column lane&15
                    const int grp = lane >> 4, i16 = lane & 15;
                    const uint16_t* a0 = lin + (ks * 16 + (grp >> 1) * 8 + (i16 >> 2)) * p.P + ustart + (grp & 1) * 16 + (i16 & 3) * 4;
                    s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, a0));
                    s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, a0 + 4 * p.P));
                    b = s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = mfma32<T>(afrag[g][ks], b, acc[g]);
            }
            // ---- epilogue: Y = Z2 + shr(Z1 + shr(Z0)) + shl(Z3 + shl(Z4)) along the lane axis -----------
            float yv[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                auto Z = [&](int r) -> float { return acc[r / RPM][(r % RPM) * NR + i]; };
                const float left = Z(1) + wave_shr1(Z(0));
                const float right = Z(3) + wave_shl1(Z(4));
                yv[i] = Z(2) + wave_shr1(left) + wave_shl1(right);
            }
            // lane -> (plane j, position along the lane axis)
            bool valid = l31 >= 2 && (p.ppt != 0 || l31 < 30);   // big planes: lanes 30,31 are halo only
            int j = 0, pos = ustart + l31 - 2;
            if (p.ppt) {
                const int q = (l31 - 2) / (p.Wl + 2);
                pos = (l31 - 2) - q * (p.Wl + 2);
                j = tile * p.ppt + q;
                valid = valid && q < p.ppt && j < p.G;
            }
            valid = valid && pos < p.Wl && (n0 + j) < n_end;
            if (valid) {
                uint16_t* op = lout + j * p.HWp;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int o = (RPM == 4) ? ((i & 3) + 4 * lhi) : ((i & 3) + 8 * (i >> 2) + 4 * lhi);
                    const int o_abs = mt * 32 + o;
                    if (o_abs < p.Wt) {
                        const uint16_t bits = cvt_to_bits(yv[i], (T*)nullptr);
                        if constexpr (VERT) op[o_abs * p.W + pos] = bits;        // (oh = o, ow = lane position)
                        else op[pos * p.W + o_abs] = bits;                         // (oh = lane position, ow = o)
                    }
                }
            }
        }
        __syncthreads();
        // ---- results of this iteration -> global (coalesced), next iteration's planes -> stack --------
        {
            uint16_t* base = y + ((size_t)n0 * p.C + c) * HW;
#pragma unroll
            for (int k = 0; k < MF_NCH; ++k) {
                if (jpl[k] >= 0 && n0 + jpl[k] < n_end) chunk_store<V>(base + goff[k], chunk_load<V>(lout + ooff[k]));
            }
        }
        if (it + 1 < iters) stage_write();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
struct MfmaShape { int MT, KS, RPM, V; };

static bool mfma_fwd_shape(const ConvDims& d, bool vert, MfmaShape& s) {
    const int Wt = vert ? d.H : d.W;
    const int KS_short = vert ? d.kw : d.kh;
    if (KS_short != MF_TAPS) return false;
    if (Wt > 64) return false;
    if (Wt > 32) s = MfmaShape{2, 4, 1, 8};
    else if (Wt > 16) s = MfmaShape{1, 2, 1, 4};
    else if (Wt > 8) s = MfmaShape{1, 1, 2, 2};
    else s = MfmaShape{1, 1, 4, 1};
    if (d.W % s.V) return false;
    return true;
}

static bool fill_mfma_params(MfmaFwdParams& p, const ConvDims& d, bool vert, const MfmaShape& s, int cu_count) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    const int HW = d.H * d.W;
    p.HWp = (HW + 7) & ~7;
    const int WLW = MF_WAVES / s.MT;
    if (p.Wl + 2 * 2 <= 32 && s.MT == 1) {
        // small lane axis: ppt planes per tile, 2 zero lanes in front of each
        p.ppt = 32 / (p.Wl + 2);
        if (p.ppt * (p.Wl + 2) + 0 > 32) return false;
        p.TS = (p.ppt * (p.Wl + 2) + 3) & ~3;
        if (p.TS > 32) p.TS = 32;
        p.ntiles = WLW;                                   // one tile per lane-tile worker
        p.G = p.ntiles * p.ppt;
    } else {
        p.ppt = 0; p.TS = 28; p.G = 1;
        p.ntiles = (p.Wl + 27) / 28;
    }
    if (p.G > d.N) {                                      // tiny batches: do not stage planes that do not exist
        if (p.ppt) { p.ntiles = (d.N + p.ppt - 1) / p.ppt; p.G = p.ntiles * p.ppt; }
    }
    const int U = p.ntiles * p.TS + 32 + 4;               // lane-axis extent incl. slack read by the last tile
    if (vert) {
        p.P = (U + 3) & ~3;
        p.in_elems = s.KS * 16 * p.P;
    } else {
        p.P = s.KS * 16 + 8;
        p.in_elems = U * p.P;
    }
    p.in_elems = (p.in_elems + 7) & ~7;
    p.cpr = d.W / s.V; p.cpp = HW / s.V; p.nchunks = p.G * p.cpp;
    if (p.nchunks > MF_NCH * MF_THREADS) return false;
    // batch slices: ~2 workgroups per CU, each a multiple of G planes
    int slices = (2 * cu_count) / d.C; if (slices < 1) slices = 1;   // one resident round: never more workgroups than 2 per CU
    int per = (d.N + slices - 1) / slices; per = (per + p.G - 1) / p.G * p.G; if (per < p.G) per = p.G;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    return true;
}

static size_t mfma_fwd_lds_bytes(const MfmaFwdParams& p) {
    return (size_t)p.in_elems * 2 + (size_t)p.G * p.HWp * 2 + 16;
}

template <typename T, int MT, int KS, int RPM, int V>
static int launch_mfma_fwd_t(const MfmaFwdParams& p, bool vert, hipStream_t st) {
    const size_t lds = mfma_fwd_lds_bytes(p);
    dim3 grid((unsigned)(p.C * p.slices));
    if (vert) {
        auto k = dwconv_mfma_fwd_kernel<T, MT, KS, RPM, V, true>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, grid, dim3(MF_THREADS), lds, st, p);
    } else {
        auto k = dwconv_mfma_fwd_kernel<T, MT, KS, RPM, V, false>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, grid, dim3(MF_THREADS), lds, st, p);
    }
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T>
static int launch_mfma_fwd_shape(const MfmaFwdParams& p, const MfmaShape& s, bool vert, hipStream_t st) {
    if (s.MT == 2) return launch_mfma_fwd_t<T, 2, 4, 1, 8>(p, vert, st);
    if (s.KS == 2) return launch_mfma_fwd_t<T, 1, 2, 1, 4>(p, vert, st);
    if (s.RPM == 2) return launch_mfma_fwd_t<T, 1, 1, 2, 2>(p, vert, st);
    return launch_mfma_fwd_t<T, 1, 1, 4, 1>(p, vert, st);
}

static int g_cu_count = 0;
int mfma_cu_count() {
    if (g_cu_count == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cu_count = prop.multiProcessorCount;
        else g_cu_count = 256;
    }
    return g_cu_count;
}

// true when the MFMA path covers (dims, dtypes); fp32 filter only (the caller converts otherwise)
bool dwconv_mfma_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt) {
    if (x_dt != y_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16) || w_dt != SLAK_F32) return false;
    const bool vert = d.kh > d.kw;
    MfmaShape s; MfmaFwdParams p;
    if (!mfma_fwd_shape(d, vert, s)) return false;
    if (!fill_mfma_params(p, d, vert, s, 256)) return false;
    return mfma_fwd_lds_bytes(p) <= 64 * 1024;
}

size_t dwconv_mfma_workspace(const ConvDims& d) {
    const bool vert = d.kh > d.kw;
    MfmaShape s;
    if (!mfma_fwd_shape(d, vert, s)) return 0;
    return align_up(toeplitz_pack_bytes(d.C, s.MT, (MF_TAPS + s.RPM - 1) / s.RPM, s.KS), 256);
}

int launch_dwconv_mfma(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                       const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_supported(d, x_dt, w_dt, y_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr || ws_bytes < dwconv_mfma_workspace(d)) return SLAK_ERR_WORKSPACE;
    const bool vert = d.kh > d.kw;
    MfmaShape s; MfmaFwdParams p;
    mfma_fwd_shape(d, vert, s);
    fill_mfma_params(p, d, vert, s, mfma_cu_count());
    ToeplitzPackParams tp{(const float*)w, (uint16_t*)ws, d.C, d.kh, d.kw, s.MT, (MF_TAPS + s.RPM - 1) / s.RPM, s.KS, s.RPM,
                          vert ? 1 : 0, flip_filter ? 1 : 0, p.Wt, p.KL, p.padL, x_dt == SLAK_BF16 ? 1 : 0};
    launch_toeplitz_pack(tp, st);
    SLAK_LAUNCH_CHECK();
    p.x = x; p.frags = (const uint16_t*)ws; p.y = y; p.flip = flip_filter ? 1 : 0;
    if (x_dt == SLAK_BF16) return launch_mfma_fwd_shape<bf16_t>(p, s, vert, st);
    return launch_mfma_fwd_shape<f16_t>(p, s, vert, st);
}

}  // namespace slak
