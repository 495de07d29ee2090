// slak_amd/csrc/dwconv_mfma_wgrad_vwave.hip -- MFMA weight gradients on planes of at most 32 rows (the 28 x 28 stage of SLaK), one plane per
// WAVE and step, no workgroup barrier in the loop: the VERTICAL kernels (K x 5; PAIR: the block's 5 x 5 kernel in the same launch) with
// nothing transposed, and (HORIZ) the horizontal kernels (5 x K) with transposing LDS reads.  The vertical case:
//
//   G_r[o, i] = sum_{n,u} dY[o, u] * X[i, u + r - 2]      (o, i = image rows, u = image columns)      dw[tau, r] = sum_o G_r[o, o+tau-padL]
// The arithmetic is dwconv_mfma_wgrad_vrows.hip's: the contraction index u runs along image rows, both operands are plain 16-byte LDS
// reads of eight consecutive elements of a row, and the five column-shifted operands X[i, u + s] are formed in registers from the lane's
// previous / current / next aligned chunk (dword selection for even shifts, v_alignbit_b32 for odd ones).  What differs is everything
// around it.  A plane of <= 32 rows is ONE 32 x 32 MFMA tile, so the four waves of a workgroup each take their own planes (images
// n_begin + wave, + 4, ... of the workgroup's channel) with their own LDS slot: no per-plane workgroup barrier (the row kernel of the
// 56 x 56 class spends a sixth of its time in them), all four SIMDs compute.  Rows of 2W bytes (W even, not a multiple of 8: 56 bytes
// on 28 x 28) do not split into whole 16-byte pieces, which is why this stage used to go through dwconv_mfma_wgrad_dma.hip's
// LDS -> LDS transposes (0.28 of the HBM roofline): here every lane loads the dword-aligned 16 bytes that start one of its row's
// pieces (buffer_load_dwordx4; dwords behind the tensor read 0), clears the elements that belong to the next row, and writes the
// piece into the padded LDS image (rows of CPR chunks, CPR odd, the pad chunk never written: X[i, -2..-1], X[i, W..W+1] and
// everything beyond column W read zeros).  Two planes are in flight per wave (two register sets).
// Epilogue (diagonal sums through a skewed per-wave tile, per-wave tap lists added in wave order, write-through partials,
// last-arriver reduction in slice order): dwconv_mfma_wgrad_vrows.hip's -- deterministic.
#include "mfma_common.h"
#include <stdlib.h>

namespace slak {

constexpr int VW_MAXJ = 2;              // pieces per lane and plane copy: ceil(32 rows x 4 pieces / 64)
constexpr unsigned VW_OOB = 0x80000000u;

struct WgradWaveParams {
    const void* dy; const void* x; float* partial; float* dw; unsigned* counters;
    const void* dy2; float* dw2;      // PAIR: the 5 x 5 branch of the same block (its dY, its dw): shares x and the five shifted operands
    int N, C, H, W, kh, kw, KL, padL;
    int DC;                // 16-byte pieces per image row: ceil(W / 8) (<= 4)
    int CPR;               // 16-byte chunks per LDS row (odd, >= DC + 1)
    int KS;                // 16-deep k-steps per plane: ceil(W / 16)
    int planes_per_wg, slices;
    unsigned tensor_bytes;
};

// PAIR: dw (K x 5) and dw2 (5 x 5) of one block in one launch -- G_r[o, i] of the small branch is the same correlation with its own dY,
// so x is fetched and shifted once for both (five more MFMAs per k-step, a third plane copy in the slot)
// HORIZ: the horizontal kernels (5 x K) in the same frame.  G_r[o, i] = sum_{n,y} dY[y, o] * X[y + r - 2, i] (o, i = image columns): the
// contraction runs over image ROWS, so both operands are column gathers -- two ds_read_b64_tr_b16 per fragment on an image of 64-byte
// rows (two zero rows above and below X for the tap shift, which is a row offset here) -- and dw[r, tau] = sum_o G_r[o, o + tau - padL].
template <typename T, bool PAIR, bool HORIZ>
__global__ __launch_bounds__(MF_THREADS, 2) void dwconv_mfma_wgrad_vwave_kernel(const WgradWaveParams p) {
    static_assert(!(PAIR && HORIZ), "the 5 x 5 branch rides with the vertical one");
    constexpr int NG = MF_TAPS;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const LB = (char*)lds;
    const int HW = p.H * p.W, ntap1 = p.kh * p.kw, ntap = ntap1 + (PAIR ? MF_TAPS * MF_TAPS : 0);
    const unsigned PB = HORIZ ? 64u : (unsigned)p.CPR * 16;       // LDS row pitch (bytes); HORIZ: 64 (the conflict-free pitch of the transposing reads)
    const unsigned row0_b = HORIZ ? 128u : 0u;                    // HORIZ: two zero rows in front of the image (and two behind)
    const unsigned copy_b = (HORIZ ? 36u : 32u) * PB;             // one plane copy: 32 rows (rows >= H stay zero)
    constexpr unsigned WAVE_B = 64 + 32 * 64 * 4;                 // per wave: [64 zero][slot: dY copy, X copy | epilogue tile 32 x 64 floats]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    char* const L = LB + wave * WAVE_B;
    float* dwl = (float*)(LB + MF_WAVES * WAVE_B);               // [MF_WAVES][ntap]
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    const int np = n_end > n_begin + wave ? (n_end - n_begin - wave + MF_WAVES - 1) / MF_WAVES : 0;     // planes of this wave

    for (unsigned o = tid * 16; o < MF_WAVES * WAVE_B + (unsigned)(MF_WAVES * ntap) * 4; o += MF_THREADS * 16) *(u32x4*)(LB + o) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    // ---- loads: lane -> pieces g = lane + 64 j of a plane copy: (row, piece) = (g / DC, g % DC) -----------------------------
    __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)p.tensor_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.tensor_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_d2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(PAIR ? p.dy2 : p.dy), 0, (int)p.tensor_bytes, 0x00020000);
    unsigned l_src[VW_MAXJ], l_dst[VW_MAXJ], l_m[VW_MAXJ][4];
#pragma unroll
    for (int j = 0; j < VW_MAXJ; ++j) {
        const int g = lane + 64 * j, row = g / p.DC, piece = g - row * p.DC;
        const bool ok = row < p.H;
        const int nv = p.W - piece * 8;                             // elements of the piece inside the row (the rest is the next row's)
        l_src[j] = ok ? (unsigned)(row * p.W + piece * 8) * 2 : VW_OOB;
        l_dst[j] = 64u + row0_b + (unsigned)(row < 32 ? row : 0) * PB + (unsigned)piece * 16;
#pragma unroll
        for (int d = 0; d < 4; ++d) l_m[j][d] = nv >= 2 * d + 2 ? 0xffffffffu : (nv == 2 * d + 1 ? 0xffffu : 0u);
    }
    const unsigned chan_b = (unsigned)c * (unsigned)HW * 2, gplane_b = (unsigned)(p.C * HW) * 2;
    struct Regs { u32x4 a[VW_MAXJ], x[VW_MAXJ], a2[PAIR ? VW_MAXJ : 1]; };
    auto load_plane = [&](int k, Regs& R) {                       // (a plane behind the wave's share loads nothing: one instruction count on every path)
        const unsigned gb = (unsigned)(n_begin + wave + MF_WAVES * k) * gplane_b + chan_b;
#pragma unroll
        for (int j = 0; j < VW_MAXJ; ++j) {
            const unsigned a = (k < np && l_src[j] != VW_OOB) ? gb + l_src[j] : VW_OOB;
            R.a[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, a, 0, 0);
            R.x[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, a, 0, 0);
            if constexpr (PAIR) R.a2[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_d2, a, 0, 0);
        }
    };
    auto stage = [&](const Regs& R) {
#pragma unroll
        for (int j = 0; j < VW_MAXJ; ++j) {
            if (l_src[j] != VW_OOB) {                               // (rows >= H of the image are never written: zero)
                *(u32x4*)(L + l_dst[j]) = u32x4{R.a[j][0] & l_m[j][0], R.a[j][1] & l_m[j][1], R.a[j][2] & l_m[j][2], R.a[j][3] & l_m[j][3]};
                *(u32x4*)(L + copy_b + l_dst[j]) = u32x4{R.x[j][0] & l_m[j][0], R.x[j][1] & l_m[j][1], R.x[j][2] & l_m[j][2], R.x[j][3] & l_m[j][3]};
                if constexpr (PAIR) *(u32x4*)(L + 2 * copy_b + l_dst[j]) = u32x4{R.a2[j][0] & l_m[j][0], R.a2[j][1] & l_m[j][1], R.a2[j][2] & l_m[j][2], R.a2[j][3] & l_m[j][3]};
            }
        }
    };
    Regs R0, R1;
    load_plane(0, R0);
    load_plane(1, R1);

    f32x16 acc[NG], acc2[PAIR ? NG : 1];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[g][i] = 0.f; if constexpr (PAIR) acc2[g][i] = 0.f; }

    // ---- fragment addresses: lane -> image row (o resp. i), 8 consecutive k = columns 16*ks + 8*lhi .. +7 -------------------
    const unsigned a_off = 64u + (unsigned)l31 * PB + lhi * 16;               // dY copy
    const unsigned x_off = a_off + copy_b;                                    // X copy
    auto rdq = [&](unsigned addr) -> u32x4 { return *(const u32x4*)(L + addr); };
    auto frag = [](unsigned d0, unsigned d1, unsigned d2, unsigned d3) -> s16x8 { return __builtin_bit_cast(s16x8, u32x4{d0, d1, d2, d3}); };
    auto sh = [](unsigned hi, unsigned lo) -> unsigned { return __builtin_amdgcn_alignbit(hi, lo, 16); };
    auto plane = [&](int k, Regs& R) {
        stage(R);                                                 // (the LDS queue is in order: the reads of the plane before are behind us)
        load_plane(k + 2, R);
        if (k >= np) return;
        if constexpr (HORIZ) {
            // lane -> column (o resp. i) = l31, 8 consecutive k = image rows 16*ks + 8*lhi .. +7: two transposing reads of four rows each
            // (a 16-lane group reads rows +0..3 of 16 columns; lane i16 supplies row i16 / 4, columns 4 * (i16 % 4) .. +3 and receives column i16)
            const int i16 = lane & 15, gq = lane >> 4;
            const unsigned tr = 64u + row0_b + (unsigned)(8 * lhi + (i16 >> 2)) * PB + (unsigned)(16 * (gq & 1) + 4 * (i16 & 3)) * 2;
            auto tr2 = [&](unsigned addr) -> s16x8 {
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr + 4 * PB));
                return s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            };
            const int nks = (p.H + 15) / 16;
            for (int ks = 0; ks < nks; ++ks) {
                const unsigned base = tr + (unsigned)ks * 16 * PB;
                const s16x8 a = tr2(base);
#pragma unroll
                for (int r = 0; r < NG; ++r) acc[r] = mfma32<T>(a, tr2(copy_b + base + (unsigned)r * PB - 2 * PB), acc[r]);
            }
            return;
        }
        for (int ks = 0; ks < p.KS; ++ks) {
            const s16x8 a = __builtin_bit_cast(s16x8, rdq(a_off + (unsigned)ks * 32));
            const unsigned xo = x_off + (unsigned)ks * 32;
            // (P of a row's first chunk is the pad chunk of the row above -- the 64 zero bytes for row 0 --, N of its last its own pad chunk)
            const u32x4 P = rdq(xo - 16), C = rdq(xo), N = rdq(xo + 16);
            const s16x8 b0 = frag(P[3], C[0], C[1], C[2]);                                                             // s = -2
            const s16x8 b1 = frag(sh(C[0], P[3]), sh(C[1], C[0]), sh(C[2], C[1]), sh(C[3], C[2]));                     // s = -1
            const s16x8 b2 = __builtin_bit_cast(s16x8, C);                                                             // s = 0
            const s16x8 b3 = frag(sh(C[1], C[0]), sh(C[2], C[1]), sh(C[3], C[2]), sh(N[0], C[3]));                     // s = +1
            const s16x8 b4 = frag(C[1], C[2], C[3], N[0]);                                                             // s = +2
            acc[0] = mfma32<T>(a, b0, acc[0]); acc[1] = mfma32<T>(a, b1, acc[1]); acc[2] = mfma32<T>(a, b2, acc[2]);
            acc[3] = mfma32<T>(a, b3, acc[3]); acc[4] = mfma32<T>(a, b4, acc[4]);
            if constexpr (PAIR) {
                const s16x8 a2 = __builtin_bit_cast(s16x8, rdq(a_off + 2 * copy_b + (unsigned)ks * 32));
                acc2[0] = mfma32<T>(a2, b0, acc2[0]); acc2[1] = mfma32<T>(a2, b1, acc2[1]); acc2[2] = mfma32<T>(a2, b2, acc2[2]);
                acc2[3] = mfma32<T>(a2, b3, acc2[3]); acc2[4] = mfma32<T>(a2, b4, acc2[4]);
            }
        }
    };
    for (int k = 0; k < np; k += 2) {
        plane(k, R0);
        plane(k + 1, R1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- diagonal sums through the wave's skewed tile (its slot is dead) ------------------------------------------------------
    float* mine = dwl + wave * ntap;
    float* tile = (float*)(L + 64);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 32 * 64 / 4; i += 64) ((u32x4*)tile)[i] = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool col_ok = l31 < (HORIZ ? p.W : p.H);
    const int o_max = HORIZ ? p.W : p.H;
    float* wr = tile + (4 * lhi) * 64 + (l31 - 4 * lhi + 31);
    // (written out twice instead of a lambda over the accumulator array: taking its address costs registers in the plane loop)
#define SLAK_VW_DIAG(AC, KL_, PADL_, OUT_)                                                                                             \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                                                   \
        if (col_ok) {                                                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                             \
                if ((r & 3) + 8 * (r >> 2) + 4 * lhi < o_max) wr[((r & 3) + 8 * (r >> 2)) * 63] = AC[g][r];                            \
        }                                                                                                                              \
        __builtin_amdgcn_wave_barrier();                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                             \
        if (lane < 63) {                                                                                                               \
            float part[4] = {0.f, 0.f, 0.f, 0.f};                                                                                      \
            _Pragma("unroll") for (int o = 0; o < 32; ++o) part[o & 3] += tile[o * 64 + lane];                                         \
            const int tau = lane - 31 + (PADL_);                                                                                       \
            if (tau >= 0 && tau < (KL_)) (OUT_)[HORIZ ? g * (KL_) + tau : tau * MF_TAPS + g] = (part[0] + part[1]) + (part[2] + part[3]); \
        }                                                                                                                              \
        __builtin_amdgcn_wave_barrier();                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                             \
    }
    SLAK_VW_DIAG(acc, p.KL, p.padL, mine)
    if constexpr (PAIR) { SLAK_VW_DIAG(acc2, MF_TAPS, MF_TAPS / 2, mine + ntap1) }      // (every tile entry is rewritten by each tap: no re-zeroing)
#undef SLAK_VW_DIAG
    __syncthreads();
    for (int t = tid; t < ntap; t += MF_THREADS) {
        float s = dwl[t];
#pragma unroll
        for (int w = 1; w < MF_WAVES; ++w) s += dwl[w * ntap + t];
        wgrad_store_partial(&p.partial[((size_t)slice * p.C + c) * ntap + t], s);
    }
    wgrad_finish(p.partial, p.dw, p.counters + c, (int*)lds, p.slices, p.C, c, 1, ntap, tid, MF_THREADS, PAIR ? p.dw2 : nullptr, ntap1);
}

// ------------------------------------------------------------------------------------------------------------
static bool hwave_enabled() {                  // SLAK_MFMA_HWAVE=0 keeps the stack kernel for the 5 x K weight gradient on these planes (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_MFMA_HWAVE"); return !(e && e[0] == '0'); }();
    return v;
}
static bool vwave_enabled() {                  // SLAK_MFMA_VWAVE=0 keeps the transposing kernel on these planes (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_MFMA_VWAVE"); return !(e && e[0] == '0'); }();
    return v;
}
static bool fill_vwave_params(WgradWaveParams& p, const ConvDims& d, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    const bool horiz = d.kh == MF_TAPS && d.kw > MF_TAPS;
    p.KL = horiz ? d.kw : d.kh; p.padL = p.KL / 2;
    if (horiz) { if (d.kw > 63 || !(d.kw & 1)) return false; }
    else if (d.kw != MF_TAPS || d.kh <= d.kw || d.kh > 63 || !(d.kh & 1)) return false;
    if (d.H < 15 || d.H > 32 || (d.W & 1) || d.W < 16 || d.W > 32) return false;      // (smaller planes: the plane-pair kernels)
    p.DC = (d.W + 7) / 8;
    p.CPR = p.DC + 1; if (!(p.CPR & 1)) ++p.CPR;                      // odd: conflict-free row-per-lane 16-byte reads
    p.KS = (d.W + 15) / 16;
    if (32 * p.DC > 64 * VW_MAXJ) return false;
    if (3u * 32u * (unsigned)p.CPR * 16u > 32u * 64u * 4u) return false;                // the (up to) three copies fit the wave's slot
    int slices = resident_wgs / d.C; if (slices < 1) slices = 1;
    int per = (d.N + slices - 1) / slices; per = (per + MF_WAVES - 1) / MF_WAVES * MF_WAVES;      // whole rounds of the four waves
    if (per < 2 * MF_WAVES) per = 2 * MF_WAVES;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * d.H * d.W * 2);
    return (size_t)d.N * d.C * d.H * d.W * 2 < 0x80000000ull;
}
static size_t vwave_lds_bytes(const WgradWaveParams& p) { return (size_t)MF_WAVES * (64 + 32 * 64 * 4) + (size_t)MF_WAVES * (p.kh * p.kw + MF_TAPS * MF_TAPS) * 4 + 32; }

bool dwconv_mfma_wgrad_vwave_supported(const ConvDims& d, int dy_dt, int x_dt) {
    if (!vwave_enabled() || dy_dt != x_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16)) return false;
    if (d.kh == MF_TAPS && d.kw > MF_TAPS && !hwave_enabled()) return false;
    WgradWaveParams p;
    return fill_vwave_params(p, d, 512);
}

size_t dwconv_mfma_wgrad_vwave_workspace(const ConvDims& d) {
    return align_up((size_t)((d.N + 2 * MF_WAVES - 1) / (2 * MF_WAVES) + 1) * d.C * (d.kh * d.kw + MF_TAPS * MF_TAPS) * sizeof(float), 256);     // slices <= ceil(N / 8); PAIR records
}

template <typename T, bool PAIR, bool HORIZ = false>
static int launch_vwave_t(WgradWaveParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    auto k = HORIZ ? dwconv_mfma_wgrad_vwave_kernel<T, false, true> : dwconv_mfma_wgrad_vwave_kernel<T, PAIR, false>;
    static const int wgs_per_cu = [] { const char* e = slak_dev_getenv("SLAK_VWAVE_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2; }();
    fill_vwave_params(p, d, wgs_per_cu * mfma_cu_count());
    const size_t lds = vwave_lds_bytes(p);
    (void)slak_set_max_lds((const void*)k, lds);
    if ((size_t)p.slices * d.C * (d.kh * d.kw + (PAIR ? MF_TAPS * MF_TAPS : 0)) * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

// dy2 / dw2 != nullptr: the 5 x 5 branch's weight gradient in the same launch
int launch_dwconv_mfma_wgrad_vwave(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                   const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st, const void* dy2, float* dw2) {
    if (!dwconv_mfma_wgrad_vwave_supported(d, dy_dt, x_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    WgradWaveParams p;
    p.dy = dy; p.x = x; p.partial = (float*)ws; p.dw = dw; p.dy2 = dy2; p.dw2 = dw2;
    p.counters = wgrad_arrival_counters(d.C);
    if (!p.counters) return SLAK_ERR_UNSUPPORTED;
    const bool horiz = d.kh == MF_TAPS && d.kw > MF_TAPS;
    if (horiz) {
        if (dy2 || dw2) return SLAK_ERR_UNSUPPORTED;
        return x_dt == SLAK_BF16 ? launch_vwave_t<bf16_t, false, true>(p, d, ws_bytes, st) : launch_vwave_t<f16_t, false, true>(p, d, ws_bytes, st);
    }
    if (dy2 && dw2) return x_dt == SLAK_BF16 ? launch_vwave_t<bf16_t, true>(p, d, ws_bytes, st) : launch_vwave_t<f16_t, true>(p, d, ws_bytes, st);
    return x_dt == SLAK_BF16 ? launch_vwave_t<bf16_t, false>(p, d, ws_bytes, st) : launch_vwave_t<f16_t, false>(p, d, ws_bytes, st);
}

}  // namespace slak
