// slak_amd/csrc/dwconv_mfma_wgrad.hip -- matrix-core (MFMA) depthwise-conv weight gradient for gfx950, 16-bit
// activations, fp32 result.
//
// Replaces backward_filter_fp16 of the reference extension
// (cutlass/examples/19_large_depthwise_conv2d_torch_extension/backward_filter_fp16.cu:181-243), which forms a
// PQ x HW correlation matrix per channel with K = batch and atomically adds its diagonals into the taps
// (cutlass/include/cutlass/epilogue/threadblock/dwconv2d_direct_epilogue_volta_tensor_op.h).  Here the correlation
// is 1-D along the LONG axis only, one small matrix per short tap:
//
//   long axis t (extent Wt, KL taps, pad padL)     short axis l (extent Wl, 5 taps, pad 2)
//   G_rho[o, i] = sum_{n, u} dy[o, u] * x[i, u + rho - 2]        (Wt x Wt, contraction over the short axis AND the batch)
//   dw[tau, rho] = sum_o G_rho[o, o + tau - padL]                (diagonal sums, once per workgroup)
//
// The contraction index k runs over a STACK of staged planes separated by 2 zero positions, so the +-2 shift of x is
// a plain address offset and k-steps straddle planes freely (no per-plane padding of K to 16).
//   * horizontal kernels (5xK): k = rows  -> both operands via ds_read_b64_tr_b16 (x at row offset rho);
//   * vertical kernels (Kx5):   k = cols  -> dy via aligned ds_read_b128, x via ds_read_b128 at a 2-byte-granular
//     offset rho (misaligned LDS reads are legal on gfx950: tools/mfma_probe.hip).
// Accumulators live in registers across the whole batch slice of the workgroup.  The diagonal reduction uses
// per-wave private LDS arrays and a fixed summation order: no atomics, bitwise run-to-run reproducible; a second
// tiny kernel (dwconv_wgrad_reduce) sums the batch slices in a fixed order.
#include "mfma_common.h"

namespace slak {

extern unsigned long long* g_dma_dbg;
constexpr int WG_NTR = 8;               // transpose blocks per 16-lane group (large planes: of one plane; small planes: of all staged planes)

struct MfmaWgradParams {
    const void* dy; const void* x; float* partial;
    float* dw; unsigned* counters;   // in-kernel slice reduction (counters == NULL: partials only, reduce kernel follows)
    int N, C, H, W, kh, kw;
    int Wt, Wl, KL, padL;
    int G;                 // planes staged per iteration
    int NKS;               // 16-deep k-steps per iteration
    int P;                 // LDS pitch (elements) of both stacks
    int Hi, Pi;            // vertical: rows / pitch of a staging image (H, W rounded up to 4)
    int dy_elems, x_elems; // LDS elements of the two stacks (multiples of 8)
    int planes_per_wg, slices;
    int nchunks, cpp, cpr; // staging chunks per iteration / per plane / per row
    unsigned long long* dbg;
    // filters with more than five rows (horizontal kernels only): one launch per chunk of five rows.  The kernel itself sees a 5 x kw filter
    // (kh == 5); gap = zero k-rows between stacked planes (2; kh_total / 2 here: the row shift of x is still a plain address offset), row0 =
    // rows of x the chunk's first tap lies below the top one (5 q), rows = rows of this chunk that exist (<= 5), rec = floats per channel in
    // `partial` / `dw` (kh_total * kw; both pointers arrive offset to the chunk's first row)
    int gap, row0, rows, rec;
};

// MT: 32-wide tiles along the long axis for both o and i (wave w owns (w & 1, w >> 1) when MT == 2; when MT == 1 the
// four waves split the k-steps); RPN: short taps packed per 32 MFMA columns; V: staging vector width; VERT: long axis = H.
// F32: fp32 operands on the bf16 matrix cores.  Both tensors are split x = x_hi + x_lo while they are staged (a second pair of stacks, and of
// staging images on the vertical path, `lo_off` elements behind the first); every k-step is dy_lo x_hi + dy_hi x_lo + dy_hi x_hi into the
// same fp32 accumulator (dropped: dy_lo x_lo and the representation errors, each <= 2^-16 of |dy||x|).
template <typename T, int MT, int RPN, int V, bool VERT, bool F32 = false>
__global__ __launch_bounds__(MF_THREADS, 2) void dwconv_mfma_wgrad_kernel(const MfmaWgradParams p) {
    constexpr int NG = (MF_TAPS + RPN - 1) / RPN;
    constexpr int NPAD = 32 / RPN;
    constexpr int NSET = F32 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* dys = lds;
    uint16_t* xs = lds + p.dy_elems;
    const int lo_off = p.dy_elems + p.x_elems;              // F32: [dy_hi | x_hi | dy_lo | x_lo]
    const int stack_elems = NSET * (p.dy_elems + p.x_elems) > MF_WAVES * 32 * 33 * 2 ? NSET * (p.dy_elems + p.x_elems) : MF_WAVES * 32 * 33 * 2;
    float* dwl = (float*)(lds + stack_elems);               // [MF_WAVES][kh*kw]
    uint16_t* img = (uint16_t*)(dwl + MF_WAVES * p.kh * p.kw);   // vertical only: row-major staging images [dy|x][G][Hi][Pi] (F32: a second set behind)
    const int img_lo = 2 * p.G * p.Hi * p.Pi;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = (MT == 2) ? (wave & 1) : 0, nt = (MT == 2) ? (wave >> 1) : 0;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int HW = p.H * p.W, ntap = p.kh * p.kw;
    const uint16_t* __restrict__ gx = (const uint16_t*)p.x;
    const uint16_t* __restrict__ gdy = (const uint16_t*)p.dy;
    const float* __restrict__ fx = (const float*)p.x;
    const float* __restrict__ fdy = (const float*)p.dy;

    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    const int iters = (n_end > n_begin) ? (n_end - n_begin + p.G - 1) / p.G : 0;

    // ---- per-thread staging map ---------------------------------------------------------------------
    int goff[MF_NCH], loff[MF_NCH], jpl[MF_NCH];
#pragma unroll
    for (int k = 0; k < MF_NCH; ++k) {
        const int idx = tid + k * MF_THREADS;
        const bool ok = idx < p.nchunks;
        const int j = ok ? idx / p.cpp : 0, rem = ok ? idx - j * p.cpp : 0;
        const int h = rem / p.cpr, w0 = (rem - h * p.cpr) * V;
        jpl[k] = ok ? j : -1;
        goff[k] = j * p.C * HW + rem * V;
        // horizontal: straight into the stacks, k-position of (plane j, row h) = 2 + j*(Wl+2) + h;
        // vertical: into the row-major staging image of plane j (transposed into the stacks afterwards)
        loff[k] = VERT ? ((j * p.Hi + h) * p.Pi + w0) : ((p.gap + j * (p.Wl + p.gap) + h) * p.P + w0);
    }
    // vertical: transpose map.  Block b = (tensor t, plane j, 4 image rows kb, 16 image columns cb): one ds_read_b64_tr_b16 per
    // 16-lane group (lane i16 supplies row kb*4 + i16/4, columns cb*16 + 4*(i16%4); receives column cb*16 + i16, rows kb*4..+3),
    // written as 8 bytes of stack row k = 2 + j*(Wl+2) + column, stack columns kb*4..+3.
    // The WG_NTR blocks a 16-lane group handles are fixed per thread.  Large planes (>= 8 blocks): the per-plane pattern,
    // repeated over (tensor, plane) in a loop; small planes: blocks of all 2*G planes flattened over the 16 groups.
    int tr_r[WG_NTR], tr_w[WG_NTR];
    const int tr_cbs = (p.W + 15) / 16, tr_per_plane = (p.Hi / 4) * tr_cbs;
    const bool tr_flat = tr_per_plane < 8;
    if constexpr (VERT) {
        const int g16 = lane >> 4, i16t = lane & 15;
        const int total = tr_flat ? 2 * p.G * tr_per_plane : tr_per_plane;
#pragma unroll
        for (int k = 0; k < WG_NTR; ++k) {
            const int b = (k * MF_WAVES + wave) * 4 + g16;
            const bool ok = b < total;
            const int pl = (ok && tr_flat) ? b / tr_per_plane : 0;          // flattened: (tensor t, plane j) index
            const int blk = ok ? b - pl * tr_per_plane : 0;
            const int t = pl / p.G, j = pl - t * p.G;
            const int kb = blk / tr_cbs, cb = blk - kb * tr_cbs;
            const int col = cb * 16 + i16t;
            const bool rd_ok = (cb * 16 + (i16t & 3) * 4) < p.Pi;      // source chunk inside the image pitch (else any valid address)
            const int src_plane = tr_flat ? pl * p.Hi * p.Pi : 0;
            const int dst_plane = tr_flat ? (t ? p.dy_elems + 2 * p.P : 0) + j * (p.Wl + 2) * p.P : 0;
            tr_r[k] = ok ? src_plane + (rd_ok ? (kb * 4 + (i16t >> 2)) * p.Pi + cb * 16 + (i16t & 3) * 4 : 0) : -1;
            tr_w[k] = (ok && col < p.W) ? dst_plane + (2 + col) * p.P + kb * 4 : -1;
        }
    }
    chunk_t<V> sx[F32 ? 1 : MF_NCH], sd[F32 ? 1 : MF_NCH];
    fchunk_t<V> sxf[F32 ? MF_NCH : 1], sdf[F32 ? MF_NCH : 1];
    auto prefetch = [&](int it) {
        const int n0 = n_begin + it * p.G;
        const size_t base = ((size_t)n0 * p.C + c) * HW;
#pragma unroll
        for (int k = 0; k < MF_NCH; ++k) {
            const bool on = jpl[k] >= 0 && n0 + jpl[k] < n_end;
            if constexpr (F32) {
                sxf[k] = on ? fchunk_load<V>(fx + base + goff[k]) : fchunk_zero<V>();
                sdf[k] = on ? fchunk_load<V>(fdy + base + goff[k]) : fchunk_zero<V>();
            } else {
                sx[k] = on ? chunk_load<V>(gx + base + goff[k]) : chunk_zero<V>();
                sd[k] = on ? chunk_load<V>(gdy + base + goff[k]) : chunk_zero<V>();
            }
        }
    };
    // the x stack carries 2 extra k-rows in front, so that row "k + rho" (rho = 0..4) holds x[k + rho - 2]
    auto put = [&](int k, const chunk_t<V>& d_, const chunk_t<V>& x_, int set) {
        if constexpr (VERT) { chunk_store<V>(img + set * img_lo + loff[k], d_); chunk_store<V>(img + set * img_lo + p.G * p.Hi * p.Pi + loff[k], x_); }
        else { chunk_store<V>(dys + set * lo_off + loff[k], d_); chunk_store<V>(xs + set * lo_off + loff[k] + p.gap * p.P, x_); }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int k = 0; k < MF_NCH; ++k) {
            if (jpl[k] >= 0) {
                if constexpr (F32) {
                    chunk_t<V> dh, dl, xh, xl;
                    fchunk_split<V>(sdf[k], dh, dl); fchunk_split<V>(sxf[k], xh, xl);
                    put(k, dh, xh, 0); put(k, dl, xl, 1);
                } else put(k, sd[k], sx[k], 0);
            }
        }
    };
    auto transpose_images = [&]() {                          // vertical: images -> stacks (all threads; caller syncs around it)
        if constexpr (VERT) {
#pragma unroll
            for (int set = 0; set < NSET; ++set) {
                const uint16_t* imgs = img + set * img_lo;
                uint16_t* base = lds + set * lo_off;
                if (tr_flat) {
#pragma unroll
                    for (int k = 0; k < WG_NTR; ++k) {
                        if (tr_r[k] >= 0) {
                            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, imgs + tr_r[k]));
                            if (tr_w[k] >= 0) *(s16x4*)(base + tr_w[k]) = v;
                        }
                    }
                } else {
                    for (int t = 0; t < 2; ++t) {
                        for (int j = 0; j < p.G; ++j) {
                            const uint16_t* src = imgs + (t * p.G + j) * p.Hi * p.Pi;
                            uint16_t* dst = base + (t ? p.dy_elems + 2 * p.P : 0) + j * (p.Wl + 2) * p.P;
#pragma unroll
                            for (int k = 0; k < WG_NTR; ++k) {
                                if (tr_r[k] >= 0) {
                                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, src + tr_r[k]));
                                    if (tr_w[k] >= 0) *(s16x4*)(dst + tr_w[k]) = v;
                                }
                            }
                        }
                    }
                }
            }
        }
    };

#ifdef SLAK_DMA_DEBUG                  // dev builds only: phase cycle counts of workgroup 0 into slak_debug_set_phase_buffer()'s buffer
    const bool prof = p.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
#else
    constexpr bool prof = false;
#endif
    unsigned long long t0 = prof ? __builtin_readcyclecounter() : 0ull, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (iters > 0) prefetch(0);
    {
        u32x4* z = (u32x4*)lds;
        const int n8 = NSET * (p.dy_elems + p.x_elems) / 8;
        for (int i = tid; i < n8; i += MF_THREADS) z[i] = u32x4{0u, 0u, 0u, 0u};
        for (int i = tid; i < MF_WAVES * ntap; i += MF_THREADS) dwl[i] = 0.f;
    }
    if constexpr (VERT) { for (int i = tid; i < NSET * p.G * p.Hi * p.Pi; i += MF_THREADS) ((unsigned*)img)[i] = 0u; }   // NSET x 2 images x G planes, 2 elements per dword
    __syncthreads();
    if (iters > 0) stage_write();
    __syncthreads();
    if constexpr (VERT) { transpose_images(); __syncthreads(); }

    if (prof) t1 = __builtin_readcyclecounter();
    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

    // ---- per-lane fragment addresses (element offsets at k-step 0) -----------------------------------
    const int grp = lane >> 4, i16 = lane & 15;
    int a_off, b_off[NG];
    {
        // tr-read: group grp reads a 4(k) x 16 block; lane supplies (k row = (grp>>1)*8 + (i16>>2), 4-col chunk i16&3)
        const int krow = (grp >> 1) * 8 + (i16 >> 2);
        a_off = krow * p.P + mt * 32 + (grp & 1) * 16 + (i16 & 3) * 4;
        const int ncol = (grp & 1) * 16 + (i16 & 3) * 4;          // MFMA column of this lane's chunk
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            int rho = g * RPN + ncol / NPAD; if (rho > MF_TAPS - 1) rho = MF_TAPS - 1;
            b_off[g] = (krow + rho + (VERT ? 0 : p.row0)) * p.P + nt * 32 + (ncol % NPAD);
        }
    }
    const int kstep_elems = 16 * p.P;
    const int ks_first = (MT == 2) ? 0 : wave, ks_stride = (MT == 2) ? 1 : MF_WAVES;

    for (int it = 0; it < iters; ++it) {
        if (it + 1 < iters) prefetch(it + 1);
        for (int ks = ks_first; ks < p.NKS; ks += ks_stride) {
            auto frag = [&](const uint16_t* q) -> s16x8 {
                const s16x4 f0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, q));
                const s16x4 f1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, q + 4 * p.P));
                return s16x8{f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
            };
            const uint16_t* ap = dys + a_off + ks * kstep_elems;
            const s16x8 a = frag(ap);
            if constexpr (F32) {
                const s16x8 al = frag(ap + lo_off);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const uint16_t* bp = xs + b_off[g] + ks * kstep_elems;
                    const s16x8 b = frag(bp), bl = frag(bp + lo_off);
                    acc[g] = mfma32<T>(al, b, acc[g]);                 // small terms first
                    acc[g] = mfma32<T>(a, bl, acc[g]);
                    acc[g] = mfma32<T>(a, b, acc[g]);
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = mfma32<T>(a, frag(xs + b_off[g] + ks * kstep_elems), acc[g]);
            }
        }
        __syncthreads();
        if (it + 1 < iters) {
            stage_write();
            if constexpr (VERT) { __syncthreads(); transpose_images(); }
        }
        __syncthreads();
    }

    // ---- diagonal sums.  Each wave dumps one accumulator tile at a time into a private 32x33 fp32 LDS scratch (aliased over
    //      the stacks, which are dead now) and lane (rho_sel, dd) adds up the diagonal i - o = dd - (NPAD-1) with plain
    //      loads in a fixed order: no atomics, bitwise reproducible.  Each (rho, tau) is produced by exactly one lane. ----
    if (prof) t2 = __builtin_readcyclecounter();
    __syncthreads();                                          // every wave is done reading the stacks
    float* mine = dwl + wave * ntap;
    float* tile = (float*)lds + wave * (32 * 33);
    constexpr int ND = 2 * NPAD - 1;                          // diagonals per tap sub-block
    constexpr int OROWS = (RPN == 1) ? 32 : NPAD;             // rows that can be non-zero (Wt <= NPAD when taps are packed)
    const int rsel = lane / ND, dd = lane - rsel * ND;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 33 + l31] = acc[g][r];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float ssum = 0.f;
        if (rsel < RPN) {
            // unconditional loads (clamped address, masked value) so that all OROWS loads are in flight together
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < OROWS; ++o) {
                const int il = o + dd - (NPAD - 1);
                const bool ok = il >= 0 && il < NPAD && mt * 32 + o < p.Wt && nt * 32 + il < p.Wt;
                const float v = tile[o * 33 + rsel * NPAD + (ok ? il : 0)];
                part[o & 3] += ok ? v : 0.f;
            }
            ssum = (part[0] + part[1]) + (part[2] + part[3]);
            const int rho = g * RPN + rsel;
            const int tau = dd - (NPAD - 1) + (nt - mt) * 32 + p.padL;
            if (rho < (VERT ? MF_TAPS : p.rows) && tau >= 0 && tau < p.KL) mine[VERT ? (tau * p.kw + rho) : (rho * p.kw + tau)] = ssum;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (prof) t3 = __builtin_readcyclecounter();
    __syncthreads();
    const int nvalid = VERT ? ntap : p.rows * p.kw, rec = p.rec ? p.rec : ntap;
    for (int t = tid; t < nvalid; t += MF_THREADS) {
        float s = dwl[t];
#pragma unroll
        for (int w = 1; w < MF_WAVES; ++w) s += dwl[w * ntap + t];
        wgrad_store_partial(&p.partial[((size_t)slice * p.C + c) * rec + t], s);
    }
    if (prof) t4 = __builtin_readcyclecounter();
    if (prof) { p.dbg[0] = t1 - t0; p.dbg[1] = t2 - t1; p.dbg[2] = t3 - t2; p.dbg[3] = t4 - t3; p.dbg[4] = iters; }
    if (p.counters) wgrad_finish(p.partial, p.dw, p.counters + c, (int*)lds, p.slices, p.C, c, 1, nvalid, tid, MF_THREADS, nullptr, 0, rec);
}

// ------------------------------------------------------------------------------------------------------------
struct WShape { int MT, RPN, V; };

static bool mfma_wgrad_shape(const ConvDims& d, bool vert, WShape& s);
static bool wgrad_tall(const ConvDims& d);

static bool mfma_wgrad_shape(const ConvDims& d, bool vert, WShape& s) {
    const int Wt = vert ? d.H : d.W;
    if ((vert ? d.kw : d.kh) != MF_TAPS && !(wgrad_tall(d) && !vert)) return false;
    if (Wt > 64) return false;
    if (Wt > 32) s = WShape{2, 1, 8};
    else if (Wt > 16) s = WShape{1, 1, 4};
    else if (Wt > 8) s = WShape{1, 2, 2};
    else s = WShape{1, 4, 1};
    return d.W % s.V == 0;
}

// filters with more than five rows and at least as many columns (square kernels: the reference's test grid, --Decom False), 16-bit tensors:
// one launch per chunk of five rows of the horizontal kernel (see MfmaWgradParams)
static bool wgrad_tall(const ConvDims& d) { return d.kh != MF_TAPS && d.kw != MF_TAPS && d.kh <= d.kw && (d.kh & 1) && (d.kw & 1) && d.kw <= 63; }

static bool fill_wgrad_params(MfmaWgradParams& p, const ConvDims& d_in, bool vert, const WShape& s, int cu_count) {
    const bool tall = wgrad_tall(d_in) && !vert;
    ConvDims d = d_in;
    if (tall) d.kh = MF_TAPS;                              // what one launch computes
    p.gap = tall ? d_in.kh / 2 : 2; p.row0 = 0; p.rows = tall ? (d_in.kh < MF_TAPS ? d_in.kh : MF_TAPS) : MF_TAPS; p.rec = tall ? d_in.kh * d_in.kw : 0;
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    const int HW = d.H * d.W;
    p.cpr = d.W / s.V; p.cpp = HW / s.V;
    // batch slices first: ~2 workgroups per CU
    int slices = (2 * cu_count) / d.C; if (slices < 1) slices = 1;    // one resident round (2 workgroups per CU)
    if (slices > d.N) slices = d.N;
    int per = (d.N + slices - 1) / slices;
    // planes per iteration: as many as the staging registers hold, not more than the slice
    int G = (MF_NCH * MF_THREADS) / p.cpp; if (G < 1) return false;
    if (G > per) G = per;
    if (G > 24) G = 24;
    p.G = G;
    per = (per + G - 1) / G * G;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.nchunks = G * p.cpp;
    const int K = p.gap + G * (p.Wl + p.gap);
    p.NKS = (K + 15) / 16;
    const int Kp = p.NKS * 16;
    p.P = s.MT * 32;                                       // k along rows, long-axis positions along columns
    p.dy_elems = Kp * p.P;
    p.x_elems = (Kp + (tall ? 2 * p.gap + 8 : 8)) * p.P;   // gap rows in front + the taps' reach behind
    p.Hi = (d.H + 3) & ~3; p.Pi = (d.W + 3) & ~3;
    {
        const int per_plane = (p.Hi / 4) * ((d.W + 15) / 16);
        if (vert && (per_plane < 8 ? 2 * G * per_plane : per_plane) > WG_NTR * MF_WAVES * 4) return false;
    }
    p.dy_elems = (p.dy_elems + 7) & ~7; p.x_elems = (p.x_elems + 7) & ~7;
    return true;
}

static size_t mfma_wgrad_lds_bytes(const MfmaWgradParams& p, bool vert, bool f32) {
    const size_t sets = f32 ? 2 : 1;
    size_t stacks = sets * (size_t)(p.dy_elems + p.x_elems) * 2, scratch = (size_t)MF_WAVES * 32 * 33 * 4;
    return (stacks > scratch ? stacks : scratch) + (size_t)MF_WAVES * p.kh * p.kw * 4 + (vert ? sets * (size_t)2 * p.G * p.Hi * p.Pi * 2 : 0) + 32;
}

// fp32 operands: the two-term bf16 split, doubled stacks (one workgroup per CU where they pass 80 KB)
bool dwconv_mfma_wgrad_supported(const ConvDims& d, int dy_dt, int x_dt) {
    if (dy_dt != x_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16 && x_dt != SLAK_F32)) return false;
    const bool vert = d.kh > d.kw;
    WShape s; MfmaWgradParams p;
    if (!mfma_wgrad_shape(d, vert, s)) return false;
    if (!fill_wgrad_params(p, d, vert, s, 256)) return false;
    if (x_dt == SLAK_F32) return !wgrad_tall(d) && mfma_wgrad_lds_bytes(p, vert, true) <= 160 * 1024;
    return mfma_wgrad_lds_bytes(p, vert, false) <= 72 * 1024;
}

size_t dwconv_mfma_wgrad_workspace(const ConvDims& d) {
    // upper bound over any CU count: slices <= N
    const bool vert = d.kh > d.kw;
    WShape s; MfmaWgradParams p;
    if (!mfma_wgrad_shape(d, vert, s) || !fill_wgrad_params(p, d, vert, s, mfma_cu_count())) return 0;
    return align_up((size_t)p.slices * d.C * d.kh * d.kw * sizeof(float), 256);
}

template <typename T, int MT, int RPN, int V, bool F32>
static int launch_wgrad_t(const MfmaWgradParams& p, bool vert, hipStream_t st) {
    const size_t lds = mfma_wgrad_lds_bytes(p, vert, F32);
    dim3 grid((unsigned)(p.C * p.slices));
    if (vert) {
        auto k = dwconv_mfma_wgrad_kernel<T, MT, RPN, V, true, F32>;
        (void)slak_set_max_lds((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(MF_THREADS), lds, st, p);
    } else {
        auto k = dwconv_mfma_wgrad_kernel<T, MT, RPN, V, false, F32>;
        (void)slak_set_max_lds((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(MF_THREADS), lds, st, p);
    }
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, bool F32>
static int launch_wgrad_shape(const MfmaWgradParams& p, const WShape& s, bool vert, hipStream_t st) {
    if (s.MT == 2) return launch_wgrad_t<T, 2, 1, 8, F32>(p, vert, st);
    if (s.RPN == 1) return launch_wgrad_t<T, 1, 1, 4, F32>(p, vert, st);
    if (s.RPN == 2) return launch_wgrad_t<T, 1, 2, 2, F32>(p, vert, st);
    return launch_wgrad_t<T, 1, 4, 1, F32>(p, vert, st);
}

int launch_dwconv_mfma_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                             const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_wgrad_supported(d, dy_dt, x_dt)) return SLAK_ERR_UNSUPPORTED;
    const bool vert = d.kh > d.kw;
    WShape s; MfmaWgradParams p;
    mfma_wgrad_shape(d, vert, s);
    fill_wgrad_params(p, d, vert, s, mfma_cu_count());
    if (ws == nullptr || ws_bytes < dwconv_mfma_wgrad_workspace(d)) return SLAK_ERR_WORKSPACE;
    p.dy = dy; p.x = x; p.partial = (float*)ws; p.dbg = g_dma_dbg;
    p.dw = dw; p.counters = wgrad_arrival_counters(d.C);
    int rc = SLAK_OK;
    const int nq = p.rec ? (d.kh + MF_TAPS - 1) / MF_TAPS : 1;       // chunks of five rows (one launch each; 1: the 5-tap kernels)
    for (int q = 0; q < nq && rc == SLAK_OK; ++q) {
        if (p.rec) {
            p.row0 = q * MF_TAPS;
            p.rows = d.kh - p.row0 < MF_TAPS ? d.kh - p.row0 : MF_TAPS;
            p.partial = (float*)ws + (size_t)p.row0 * d.kw;
            p.dw = dw + (size_t)p.row0 * d.kw;
        }
        rc = (x_dt == SLAK_F32) ? launch_wgrad_shape<bf16_t, true>(p, s, vert, st)
           : (x_dt == SLAK_BF16) ? launch_wgrad_shape<bf16_t, false>(p, s, vert, st) : launch_wgrad_shape<f16_t, false>(p, s, vert, st);
    }
    if (rc != SLAK_OK || p.counters) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, d.C * d.kh * d.kw, p.slices, st);
}

}  // namespace slak
