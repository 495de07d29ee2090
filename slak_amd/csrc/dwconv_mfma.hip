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
    const uint16_t* frags_lo;   // F32 kernels: the filters' second bf16 term
    int nch, guard;             // tap chunks along the short axis (1 unless the kernel has more than five rows), zero rows in front of a plane
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
    const int total = p.C * p.MT * p.NCH * p.NG * p.KS * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63;
    int f = idx >> 6;
    const int ks = f % p.KS; f /= p.KS;
    const int g = f % p.NG; f /= p.NG;
    const int q = f % p.NCH; f /= p.NCH;
    const int mt = f % p.MT; const int c = f / p.MT;
    const int MPAD = 32 / p.RPM, l31 = lane & 31, lhi = lane >> 5;
    const int short_taps = p.vert ? p.kw : p.kh;                 // MF_TAPS unless NCH > 1
    const int rj = g * p.RPM + l31 / MPAD;                       // tap inside the chunk
    const int r = q * MF_TAPS + rj, o_abs = mt * 32 + (l31 % MPAD);
    const float* wc = p.w + (size_t)c * p.kh * p.kw;
    // all 8 loads are issued unconditionally (clamped index) and masked afterwards, so they overlap
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i_abs = ks * 16 + lhi * 8 + e;
        int t = i_abs - o_abs + p.padL;
        const bool ok = rj < MF_TAPS && r < short_taps && o_abs < p.Wt && i_abs < p.Wt && t >= 0 && t < p.KL;
        int rr = ok ? r : 0;
        t = ok ? t : 0;
        if (p.flip) { t = p.KL - 1 - t; rr = short_taps - 1 - rr; }
        const float wv = p.vert ? wc[t * p.kw + rr] : wc[rr * p.kw + t];
        v[e] = ok ? wv : 0.f;
    }
    u32x4 out;
#pragma unroll
    for (int e = 0; e < 8; e += 2)
        out[e >> 1] = p.is_bf16 ? pack2<bf16_t>(v[e], v[e + 1]) : pack2<f16_t>(v[e], v[e + 1]);
    ((u32x4*)p.frags)[idx] = out;
    if (p.frags_lo) {                                       // w = bf16(w) + bf16(w - bf16(w)) + O(2^-17 |w|)
        u32x4 lo;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            uint16_t h0, l0, h1, l1;
            split_bf16(v[e], h0, l0); split_bf16(v[e + 1], h1, l1);
            lo[e >> 1] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        ((u32x4*)p.frags_lo)[idx] = lo;
    }
}

void launch_toeplitz_pack(const ToeplitzPackParams& p, hipStream_t st) {
    const int total = p.C * p.MT * p.NCH * p.NG * p.KS * 64;
    hipLaunchKernelGGL(toeplitz_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, p);
}

// MT: 32-row tiles along the Toeplitz axis (wave w owns tile w % MT); KS: 16-deep k-steps (Wt <= 16*KS);
// RPM: short taps packed per MFMA (32/RPM rows each); V: staging vector width (elements); VERT: long axis = H.
// F32: fp32 activations and results on the bf16 matrix cores.  x = x_hi + x_lo is split ONCE per element while it is staged (two LDS stacks),
// w = w_hi + w_lo by the pack kernel; every (tap group, k-step) is three MFMAs into the same fp32 accumulator: w_hi x_hi + w_hi x_lo + w_lo x_hi
// (the dropped w_lo x_lo term and the two representation errors are each <= 2^-16 of |w||x|: ~2e-5 relative, measured in tests/test_fp32_mfma_gpu.py).
// w_hi fragments live in registers, w_lo fragments in LDS.
template <typename T, int MT, int KS, int RPM, int V, bool VERT, bool F32 = false, bool TALL = false>
__global__ __launch_bounds__(MF_THREADS, 2) void dwconv_mfma_fwd_kernel(const MfmaFwdParams p) {
    constexpr int NG = (MF_TAPS + RPM - 1) / RPM;          // accumulators (MFMA groups) per unit
    constexpr int MPAD = 32 / RPM;                          // rows per tap inside an MFMA
    constexpr int NR = 16 / RPM;                            // output registers per lane
    constexpr int WL = MF_WAVES / MT;                       // lane-tile workers per Toeplitz tile
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* lin = lds;                                    // staged stack (F32: the x_hi stack, the x_lo stack follows)
    uint16_t* lin_lo = lds + p.in_elems;
    uint16_t* lout = lds + (F32 ? 2 : 1) * p.in_elems;      // [G][HWp] results (F32: fp32)
    float* loutf = (float*)lout;
    const s16x8* alo = (const s16x8*)(loutf + (F32 ? p.G * p.HWp : 0));   // F32: [MT][NG][KS][64] w_lo fragments

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = wave % MT, wl = wave / MT;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int HW = p.H * p.W;
    const uint16_t* __restrict__ x = (const uint16_t*)p.x;
    uint16_t* __restrict__ y = (uint16_t*)p.y;
    const float* __restrict__ xf = (const float*)p.x;
    float* __restrict__ yf = (float*)p.y;

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
        const int u0 = (p.ppt ? (j / p.ppt) * p.TS + (j % p.ppt) * (p.Wl + 2) : 0) + p.guard;
        jpl[k] = ok ? j : -1;
        goff[k] = j * p.C * HW + rem * V;
        ooff[k] = j * p.HWp + rem * V;
        loff[k] = VERT ? (h * p.P + u0 + w0) : ((u0 + h) * p.P + w0);
    }
    chunk_t<V> st[F32 ? 1 : MF_NCH];
    fchunk_t<V> stf[F32 ? MF_NCH : 1];
    auto prefetch = [&](int it) {
        const int n0 = n_begin + it * p.G;
        const size_t plane0 = ((size_t)n0 * p.C + c) * HW;
#pragma unroll
        for (int k = 0; k < MF_NCH; ++k) {
            const bool on = jpl[k] >= 0 && n0 + jpl[k] < n_end;
            if constexpr (F32) stf[k] = on ? fchunk_load<V>(xf + plane0 + goff[k]) : fchunk_zero<V>();
            else st[k] = on ? chunk_load<V>(x + plane0 + goff[k]) : chunk_zero<V>();
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int k = 0; k < MF_NCH; ++k) {
            if (jpl[k] >= 0) {
                if constexpr (F32) {
                    chunk_t<V> hi, lo;
                    fchunk_split<V>(stf[k], hi, lo);
                    if constexpr (VERT) { chunk_store_lds_a4<V>(lin + loff[k], hi); chunk_store_lds_a4<V>(lin_lo + loff[k], lo); }
                    else { chunk_store<V>(lin + loff[k], hi); chunk_store<V>(lin_lo + loff[k], lo); }
                } else {
                    if constexpr (VERT) chunk_store_lds_a4<V>(lin + loff[k], st[k]);
                    else chunk_store<V>(lin + loff[k], st[k]);
                }
            }
        }
    };

    prefetch(0);
    // ---- zero the stack (pads stay zero for the whole kernel), fetch this channel's filter --------
    {
        u32x4* z = (u32x4*)lin;
        for (int i = tid; i < (F32 ? 2 : 1) * p.in_elems / 8; i += MF_THREADS) z[i] = u32x4{0u, 0u, 0u, 0u};
        if constexpr (F32) {                                // this channel's w_lo fragments: global -> LDS, register layout kept
            const s16x8* src = (const s16x8*)p.frags_lo + (size_t)c * MT * NG * KS * 64;
            s16x8* dst = const_cast<s16x8*>(alo);
            for (int i = tid; i < MT * NG * KS * 64; i += MF_THREADS) dst[i] = src[i];
        }
    }
    __syncthreads();
    stage_write();

    // ---- Toeplitz fragments: A[g][ks] holds, for MFMA row m = l31 -> (tap r = g*RPM + m/MPAD, o = m%MPAD) and
    //      k = ks*16 + lhi*8 + e -> i, the weight w[r][i - o + padL] (0 outside the filter or the plane) ----
    s16x8 afrag[NG][KS];
    bool ks_active[KS];
    load_toeplitz_frags_at<NG, KS>(afrag, ks_active, p.frags, (size_t)(c * MT + mt) * p.nch, mt, lane, MPAD, p.Wt, p.KL, p.padL);   // chunk 0
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
            // Kernels with more than five rows (square kernels, the reference's own test grid): the rows are taken five at a time.  Chunk q's
            // five-tap sum has to land 5q + 2 - kh/2 rows further along the short axis; with kh/2 zero rows in front of the plane that is the
            // stack 5q rows further down, so every chunk adds into the SAME five accumulators and the lane-shift epilogue runs once.  The
            // chunk's fragments come from the workspace (L2) each time: 5 KS loads per 5 KS MFMAs -- a path for completeness, not a hot one.
            for (int q = 0; q < (TALL ? p.nch : 1); ++q) {
            if constexpr (TALL) {
                if (q > 0) load_toeplitz_frags_at<NG, KS>(afrag, ks_active, p.frags, (size_t)(c * MT + mt) * p.nch + q, mt, lane, MPAD, p.Wt, p.KL, p.padL);
            }
            const int qrow = TALL ? q * MF_TAPS : 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (!ks_active[ks]) continue;
                auto read_b = [&](const uint16_t* stack) -> s16x8 {
                    if constexpr (!VERT) {
                        return *(const s16x8*)(stack + (ustart + l31 + qrow) * p.P + ks * 16 + lhi * 8);
                    } else {
                        // ds_read_b64_tr_b16: the 16 lanes of group g = lane>>4 read a 4(k) x 16(u) block; lane gets column lane&15
                        const int grp = lane >> 4, i16 = lane & 15;
                        const uint16_t* a0 = stack + (ks * 16 + (grp >> 1) * 8 + (i16 >> 2)) * p.P + ustart + (grp & 1) * 16 + (i16 & 3) * 4;
                        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, a0));
                        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, a0 + 4 * p.P));
                        return s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    }
                };
                const s16x8 b = read_b(lin);
                if constexpr (F32) {
                    const s16x8 bl = read_b(lin_lo);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const s16x8 al = alo[((mt * NG + g) * KS + ks) * 64 + lane];
                        acc[g] = mfma32<T>(al, b, acc[g]);               // small terms first
                        acc[g] = mfma32<T>(afrag[g][ks], bl, acc[g]);
                        acc[g] = mfma32<T>(afrag[g][ks], b, acc[g]);
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g] = mfma32<T>(afrag[g][ks], b, acc[g]);
                }
            }
            }
            if constexpr (TALL) {                                   // the next tile starts with chunk 0 again
                if (p.nch > 1) load_toeplitz_frags_at<NG, KS>(afrag, ks_active, p.frags, (size_t)(c * MT + mt) * p.nch, mt, lane, MPAD, p.Wt, p.KL, p.padL);
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
                float* opf = loutf + j * p.HWp;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int o = (RPM == 4) ? ((i & 3) + 4 * lhi) : ((i & 3) + 8 * (i >> 2) + 4 * lhi);
                    const int o_abs = mt * 32 + o;
                    if (o_abs < p.Wt) {
                        const int at = VERT ? o_abs * p.W + pos : pos * p.W + o_abs;   // (oh, ow) = (o, lane position) / (lane position, o)
                        if constexpr (F32) opf[at] = yv[i];
                        else op[at] = cvt_to_bits(yv[i], (T*)nullptr);
                    }
                }
            }
        }
        __syncthreads();
        // ---- results of this iteration -> global (coalesced), next iteration's planes -> stack --------
        {
            const size_t plane0 = ((size_t)n0 * p.C + c) * HW;
#pragma unroll
            for (int k = 0; k < MF_NCH; ++k) {
                if (jpl[k] >= 0 && n0 + jpl[k] < n_end) {
                    if constexpr (F32) fchunk_store<V>(yf + plane0 + goff[k], fchunk_load<V>(loutf + ooff[k]));
                    else chunk_store<V>(y + plane0 + goff[k], chunk_load<V>(lout + ooff[k]));
                }
            }
        }
        if (it + 1 < iters) stage_write();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
struct MfmaShape { int MT, KS, RPM, V; };

// kernels with more than five rows and at least as many columns (the square kernels of the reference's test grid, --Decom False): the
// long axis is W, the kh rows are taken in chunks of five (forward and data gradient; 16-bit tensors)
static bool mfma_tall(const ConvDims& d) { return d.kh != MF_TAPS && d.kw != MF_TAPS && d.kh <= d.kw && (d.kh & 1) && (d.kw & 1) && d.kw <= 63; }
static int mfma_nch(const ConvDims& d) { return mfma_tall(d) ? (d.kh + MF_TAPS - 1) / MF_TAPS : 1; }

static bool mfma_fwd_shape(const ConvDims& d, bool vert, MfmaShape& s) {
    const int Wt = vert ? d.H : d.W;
    const int KS_short = vert ? d.kw : d.kh;
    if (KS_short != MF_TAPS && !(mfma_tall(d) && !vert)) return false;
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
    const bool tall = mfma_tall(d) && !vert;
    p.nch = tall ? mfma_nch(d) : 1;
    p.guard = tall ? d.kh / 2 : 2;
    if (p.Wl + 2 * 2 <= 32 && s.MT == 1 && !tall) {
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
    const int U = p.ntiles * p.TS + 32 + 4 + (tall ? MF_TAPS * p.nch + p.guard : 0);   // lane-axis extent incl. slack read by the last tile (tall: + the chunks' reach)
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

static size_t mfma_fwd_lds_bytes(const MfmaFwdParams& p, const MfmaShape& s, bool f32) {
    if (!f32) return (size_t)p.in_elems * 2 + (size_t)p.G * p.HWp * 2 + 16;
    const size_t ng = (MF_TAPS + s.RPM - 1) / s.RPM;
    return (size_t)p.in_elems * 4 + (size_t)p.G * p.HWp * 4 + (size_t)s.MT * ng * s.KS * 64 * 16 + 16;     // two stacks, fp32 results, w_lo fragments
}

template <typename T, int MT, int KS, int RPM, int V, bool F32>
static int launch_mfma_fwd_t(const MfmaFwdParams& p, bool vert, hipStream_t st) {
    const size_t lds = mfma_fwd_lds_bytes(p, MfmaShape{MT, KS, RPM, V}, F32);
    dim3 grid((unsigned)(p.C * p.slices));
    if constexpr (!F32) {
        if (p.nch > 1 || p.guard != 2) {                             // more than five rows (or fewer: 3 x 3): the chunked horizontal kernel
            auto k = dwconv_mfma_fwd_kernel<T, MT, KS, RPM, V, false, false, true>;
            (void)slak_set_max_lds((const void*)k, lds);
            hipLaunchKernelGGL(k, grid, dim3(MF_THREADS), lds, st, p);
            SLAK_LAUNCH_CHECK();
            return SLAK_OK;
        }
    }
    if (vert) {
        auto k = dwconv_mfma_fwd_kernel<T, MT, KS, RPM, V, true, F32>;
        (void)slak_set_max_lds((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(MF_THREADS), lds, st, p);
    } else {
        auto k = dwconv_mfma_fwd_kernel<T, MT, KS, RPM, V, false, F32>;
        (void)slak_set_max_lds((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(MF_THREADS), lds, st, p);
    }
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, bool F32>
static int launch_mfma_fwd_shape(const MfmaFwdParams& p, const MfmaShape& s, bool vert, hipStream_t st) {
    if (s.MT == 2) return launch_mfma_fwd_t<T, 2, 4, 1, 8, F32>(p, vert, st);
    if (s.KS == 2) return launch_mfma_fwd_t<T, 1, 2, 1, 4, F32>(p, vert, st);
    if (s.RPM == 2) return launch_mfma_fwd_t<T, 1, 1, 2, 2, F32>(p, vert, st);
    return launch_mfma_fwd_t<T, 1, 1, 4, 1, F32>(p, vert, st);
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

// true when the MFMA path covers (dims, dtypes); fp32 filter only (the caller converts otherwise).  fp32 activations: the two-term
// bf16 split (three MFMAs per product term group), two workgroups per CU as long as the doubled stacks and the w_lo fragments fit 80 KB.
bool dwconv_mfma_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt) {
    if (x_dt != y_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16 && x_dt != SLAK_F32) || w_dt != SLAK_F32) return false;
    const bool vert = d.kh > d.kw;
    MfmaShape s; MfmaFwdParams p;
    if (!mfma_fwd_shape(d, vert, s)) return false;
    if (!fill_mfma_params(p, d, vert, s, 256)) return false;
    if (x_dt == SLAK_F32) return !mfma_tall(d) && mfma_fwd_lds_bytes(p, s, true) <= 80 * 1024;
    return mfma_fwd_lds_bytes(p, s, false) <= 64 * 1024;
}

size_t dwconv_mfma_workspace(const ConvDims& d) {
    const bool vert = d.kh > d.kw;
    MfmaShape s;
    if (!mfma_fwd_shape(d, vert, s)) return 0;
    return 2 * align_up((size_t)mfma_nch(d) * toeplitz_pack_bytes(d.C, s.MT, (MF_TAPS + s.RPM - 1) / s.RPM, s.KS), 256);     // w_hi (all dtypes) + w_lo (fp32 activations)
}

int launch_dwconv_mfma(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                       const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_supported(d, x_dt, w_dt, y_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr || ws_bytes < dwconv_mfma_workspace(d)) return SLAK_ERR_WORKSPACE;
    const bool vert = d.kh > d.kw;
    MfmaShape s; MfmaFwdParams p;
    mfma_fwd_shape(d, vert, s);
    fill_mfma_params(p, d, vert, s, mfma_cu_count());
    const bool f32 = x_dt == SLAK_F32;
    uint16_t* lo = f32 ? (uint16_t*)((char*)ws + dwconv_mfma_workspace(d) / 2) : nullptr;
    ToeplitzPackParams tp{(const float*)w, (uint16_t*)ws, d.C, d.kh, d.kw, s.MT, (MF_TAPS + s.RPM - 1) / s.RPM, s.KS, s.RPM,
                          vert ? 1 : 0, flip_filter ? 1 : 0, p.Wt, p.KL, p.padL, x_dt != SLAK_F16 ? 1 : 0, lo, p.nch};
    launch_toeplitz_pack(tp, st);
    SLAK_LAUNCH_CHECK();
    p.x = x; p.frags = (const uint16_t*)ws; p.frags_lo = lo; p.y = y; p.flip = flip_filter ? 1 : 0;
    if (f32) return launch_mfma_fwd_shape<bf16_t, true>(p, s, vert, st);
    if (x_dt == SLAK_BF16) return launch_mfma_fwd_shape<bf16_t, false>(p, s, vert, st);
    return launch_mfma_fwd_shape<f16_t, false>(p, s, vert, st);
}

}  // namespace slak
