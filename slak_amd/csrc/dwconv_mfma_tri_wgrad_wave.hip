// slak_amd/csrc/dwconv_mfma_tri_wgrad_wave.hip -- the THREE weight gradients of a decomposed block (K x 5, 5 x K, 5 x 5: models/SLaK.py:82-100; the
// reference runs backward_filter_fp16.cu:181-243 once per branch) in ONE launch on planes of ONE MFMA tile (15 <= H <= 32, W even, 16 <= W <= 32:
// the 28 x 28 stage, 24 x 24).  x is fetched from HBM once for the three correlations (4 plane reads per block instead of 6 / 5).
//
//   vertical   (K x 5)  G^v_r[o, i] = sum_{n,u} dYv[o, u] X[i, u + r - 2]   o, i = image ROWS,    contraction along a row     dw_v[tau, r] = sum_o G^v_r[o, o+tau-padL]
//   small      (5 x 5)  G^s_r[o, i] = sum_{n,u} dYs[o, u] X[i, u + r - 2]   (the vertical correlation with its own dY: same B operands)
//   horizontal (5 x K)  G^h_r[o, i] = sum_{n,y} dYh[y, o] X[y + r - 2, i]   o, i = image COLUMNS, contraction along a column  dw_h[r, tau] = sum_o G^h_r[o, o+tau-padL]
//
// The frame is dwconv_mfma_wgrad_vwave.hip's: a plane is one 32 x 32 tile, so every WAVE takes its own planes with its own LDS images -- no
// workgroup barrier anywhere; rows of 2 W bytes are not whole 16-byte pieces, so they travel through registers (dword-aligned
// buffer_load_dwordx4, the next row's elements cleared, ds_write_b128 into a padded image: rows of five 16-byte chunks, the last never
// written), two planes in flight per wave.  The horizontal branch reads the SAME images with transposing reads (ds_read_b64_tr_b16), its five
// row shifts are the six-dword window arithmetic of the vertical branch's column shifts applied to a twelve-row column segment.
// Fifteen accumulators (three branches x five taps) as named accumulator registers: one wave per SIMD (tri_wgrad_common.h).
//
// Work decomposition -- wave-granular and channel-aligned: the 4 x CUs waves are dealt to the channels (floor or ceil of waves / C each), a
// channel's images are split evenly among its waves; no wave ever crosses a channel boundary (nothing is summed up in mid-stream) and the
// imbalance is one plane in ~25.  A wave writes ONE partial record; the last wave of a channel to arrive adds the channel's records in wave
// order (bitwise reproducible) and scatters into the three dw tensors.
#include "tri_wgrad_common.h"
#include <stdlib.h>

namespace slak {

constexpr int TV_MAXJ = 2;              // pieces per lane and plane copy: 32 rows x 4 pieces / 64
constexpr unsigned TV_OOB = 0x80000000u;
constexpr int TV_CPR = 5;               // 16-byte chunks per LDS row (W <= 32: four data chunks + the pad chunk; odd: conflict-free row-per-lane reads)
constexpr int TV_RS = 36;               // rows per plane copy: the image's 32 + two zero rows on either side of X (the tap shift of the horizontal branch)

struct TriWaveParams {
    const void* dy[3];                  // dY of the K x 5, 5 x K, 5 x 5 branch
    const void* x; float* partial; float* dw[3]; unsigned* counters;
    int N, C, H, W, K, padL;
    int DC;                // 16-byte pieces per image row: ceil(W / 8) (<= 4)
    int waves, base, extra;   // waves in the grid; channels 0 .. extra - 1 get base + 1 of them, the others base
    unsigned tensor_bytes;
};

template <typename T>
__global__ __launch_bounds__(MF_THREADS, 1) void dwconv_mfma_tri_wgrad_wave_kernel(const TriWaveParams p) {
    constexpr unsigned PB = TV_CPR * 16, copy_b = TV_RS * PB;     // [dYv][X][dYs][dYh]: X's rows -2, -1 are the (zero) tail of dYv's copy
    constexpr unsigned WAVE_B = 64 + 4 * copy_b;                  // per wave: 64 zero bytes ("chunk -1" of the first row), the four images | later its skewed tile + tap list
    static_assert(WAVE_B >= 64 + 32 * 64 * 4 + 1024, "the epilogue tile fits the wave's slot");
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const LB = (char*)lds;
    const int HW = p.H * p.W, ntl = p.K * MF_TAPS, ntot = 2 * ntl + MF_TAPS * MF_TAPS;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    char* const L = LB + wave * WAVE_B;
    float* mine = (float*)(LB + MF_WAVES * WAVE_B) + wave * ntot;  // this wave's taps [dw_v | dw_h | dw_s]

    // ---- which channel, which of its images -------------------------------------------------------------------------------
    const int gw = blockIdx.x * MF_WAVES + wave;
    int c, part, nparts, gw0;                                     // gw0: the first wave of the channel
    {
        const int cut = p.extra * (p.base + 1);
        if (gw < cut) { c = gw / (p.base + 1); nparts = p.base + 1; gw0 = c * nparts; }
        else { const int r = gw - cut; c = p.extra + r / p.base; nparts = p.base; gw0 = cut + (c - p.extra) * nparts; }
        part = gw - gw0;
    }
    const bool idle = gw >= p.waves || c >= p.C;                  // (a grid rounded up to whole workgroups)
    const int n0 = idle ? 0 : (int)((long long)part * p.N / nparts), n1 = idle ? 0 : (int)((long long)(part + 1) * p.N / nparts);
    const int np = n1 - n0;

    for (unsigned o = lane * 16; o < WAVE_B; o += 64 * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};
    for (int t = lane; t < ntot; t += 64) mine[t] = 0.f;
    __builtin_amdgcn_wave_barrier();

    // ---- loads: lane -> pieces g = lane + 64 j of a plane copy: (row, piece) = (g / DC, g % DC) -----------------------------
    __amdgpu_buffer_rsrc_t rs[4] = {__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy[0]), 0, (int)p.tensor_bytes, 0x00020000),
                                    __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.tensor_bytes, 0x00020000),
                                    __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy[2]), 0, (int)p.tensor_bytes, 0x00020000),
                                    __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy[1]), 0, (int)p.tensor_bytes, 0x00020000)};
    unsigned l_src[TV_MAXJ], l_dst[TV_MAXJ], l_m[TV_MAXJ][4];
#pragma unroll
    for (int j = 0; j < TV_MAXJ; ++j) {
        const int g = lane + 64 * j, row = g / p.DC, piece = g - row * p.DC;
        const bool ok = row < p.H;
        const int nv = p.W - piece * 8;                             // elements of the piece inside the row (the rest is the next row's)
        l_src[j] = ok ? (unsigned)(row * p.W + piece * 8) * 2 : TV_OOB;
        l_dst[j] = 64u + (unsigned)(row < 32 ? row : 0) * PB + (unsigned)piece * 16;
#pragma unroll
        for (int d = 0; d < 4; ++d) l_m[j][d] = nv >= 2 * d + 2 ? 0xffffffffu : (nv == 2 * d + 1 ? 0xffffu : 0u);
    }
    const unsigned chan_b = (unsigned)c * (unsigned)HW * 2, gplane_b = (unsigned)(p.C * HW) * 2;
    struct Regs { u32x4 v[4][TV_MAXJ]; };
    auto load_plane = [&](int k, Regs& R) {                       // (a plane behind the wave's share loads nothing: one instruction count on every path)
        const unsigned gb = (unsigned)(n0 + k) * gplane_b + chan_b;
#pragma unroll
        for (int j = 0; j < TV_MAXJ; ++j) {
            const unsigned a = (k < np && l_src[j] != TV_OOB) ? gb + l_src[j] : TV_OOB;
#pragma unroll
            for (int t = 0; t < 4; ++t) R.v[t][j] = __builtin_amdgcn_raw_buffer_load_b128(rs[t], a, 0, 0);
        }
    };
    auto stage = [&](const Regs& R) {
#pragma unroll
        for (int j = 0; j < TV_MAXJ; ++j) {
            if (l_src[j] != TV_OOB) {                               // (rows >= H of the image are never written: zero)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    *(u32x4*)(L + (unsigned)t * copy_b + l_dst[j]) = u32x4{R.v[t][j][0] & l_m[j][0], R.v[t][j][1] & l_m[j][1], R.v[t][j][2] & l_m[j][2], R.v[t][j][3] & l_m[j][3]};
            }
        }
    };
    Regs R0, R1;
    load_plane(0, R0);
    load_plane(1, R1);

    tw_acc_claim();
    tw_acc_zero<0, 240>();

    // ---- fragment addresses ----------------------------------------------------------------------------------------------
    // vertical / small: lane -> image row l31, 8 consecutive k = columns 16 ks + 8 lhi .. +7
    const unsigned av = 64u + (unsigned)l31 * PB + lhi * 16;                                       // dYv (copy 0); dYs: + 2 copies; X: + 1 copy
    // horizontal: lane -> image column l31, 8 consecutive k = rows 16 ks + 8 lhi .. +7: a 16-lane group of a transposing read covers 4 rows x 16
    // columns (lane i16 supplies row i16 / 4, columns 4 (i16 % 4) .. +3, receives column i16)
    const int i16 = lane & 15, gq = lane >> 4;
    const unsigned trl = 64u + (unsigned)(8 * lhi + (i16 >> 2)) * PB + (unsigned)(16 * (gq & 1) + 4 * (i16 & 3)) * 2;
    const unsigned ah = 3 * copy_b + trl;                                                          // dYh (copy 3)
    const unsigned xh = copy_b + trl - 2 * PB;                                                     // X, the window starts two rows up
    auto rdq = [&](unsigned addr) -> u32x4 { return *(const u32x4*)(L + addr); };
    auto rdd = [&](unsigned addr) -> unsigned { return *(const unsigned*)(L + addr); };
    auto rdt = [&](unsigned addr) -> u32x2 { return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr))); };

    auto plane = [&](int k, Regs& R) {
        stage(R);                                                 // (the LDS queue is in order: the reads of the plane before are behind us)
        load_plane(k + 2, R);
        if (k >= np) return;
        // four k-steps: vertical + small over columns 0-15, 16-31, then horizontal over rows 0-15, 16-31; the operands of step j + 1 are read and
        // shifted while the MFMAs of step j run
        s16x8 a, a2, b[MF_TAPS], an, a2n, bn[MF_TAPS];
        {
            a = __builtin_bit_cast(s16x8, rdq(av)); a2 = __builtin_bit_cast(s16x8, rdq(av + 2 * copy_b));
            const u32x4 C = rdq(av + copy_b); const unsigned P3 = rdd(av + copy_b - 4), N0 = rdd(av + copy_b + 16);
            tw_taps(b, P3, C[0], C[1], C[2], C[3], N0);
        }
        {   // vertical step 0; operands of vertical step 1
            an = __builtin_bit_cast(s16x8, rdq(av + 32)); a2n = __builtin_bit_cast(s16x8, rdq(av + 2 * copy_b + 32));
            const u32x4 C = rdq(av + copy_b + 32); const unsigned P3 = rdd(av + copy_b + 28), N0 = rdd(av + copy_b + 48);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_V + 0>(a, b[0]); tw_mfma<T, TW_ACC_S + 0>(a2, b[0]); tw_mfma<T, TW_ACC_V + 16>(a, b[1]); tw_mfma<T, TW_ACC_S + 16>(a2, b[1]);
            tw_mfma<T, TW_ACC_V + 32>(a, b[2]); tw_mfma<T, TW_ACC_S + 32>(a2, b[2]);
            tw_taps(bn, P3, C[0], C[1], C[2], C[3], N0);
            tw_mfma<T, TW_ACC_V + 48>(a, b[3]); tw_mfma<T, TW_ACC_S + 48>(a2, b[3]); tw_mfma<T, TW_ACC_V + 64>(a, b[4]); tw_mfma<T, TW_ACC_S + 64>(a2, b[4]);
            __builtin_amdgcn_sched_barrier(0);
        }
        {   // vertical step 1; operands of horizontal step 0
            const u32x2 h0 = rdt(ah), h1 = rdt(ah + 4 * PB), w0 = rdt(xh), w1 = rdt(xh + 4 * PB), w2 = rdt(xh + 8 * PB);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_V + 0>(an, bn[0]); tw_mfma<T, TW_ACC_S + 0>(a2n, bn[0]); tw_mfma<T, TW_ACC_V + 16>(an, bn[1]); tw_mfma<T, TW_ACC_S + 16>(a2n, bn[1]);
            tw_mfma<T, TW_ACC_V + 32>(an, bn[2]); tw_mfma<T, TW_ACC_S + 32>(a2n, bn[2]);
            a = __builtin_bit_cast(s16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
            tw_taps(b, w0[0], w0[1], w1[0], w1[1], w2[0], w2[1]);
            tw_mfma<T, TW_ACC_V + 48>(an, bn[3]); tw_mfma<T, TW_ACC_S + 48>(a2n, bn[3]); tw_mfma<T, TW_ACC_V + 64>(an, bn[4]); tw_mfma<T, TW_ACC_S + 64>(a2n, bn[4]);
            __builtin_amdgcn_sched_barrier(0);
        }
        {   // horizontal step 0; operands of horizontal step 1
            constexpr unsigned ro = 16 * PB;
            const u32x2 h0 = rdt(ah + ro), h1 = rdt(ah + ro + 4 * PB), w0 = rdt(xh + ro), w1 = rdt(xh + ro + 4 * PB), w2 = rdt(xh + ro + 8 * PB);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_H + 0>(a, b[0]); tw_mfma<T, TW_ACC_H + 16>(a, b[1]); tw_mfma<T, TW_ACC_H + 32>(a, b[2]);
            an = __builtin_bit_cast(s16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
            tw_taps(bn, w0[0], w0[1], w1[0], w1[1], w2[0], w2[1]);
            tw_mfma<T, TW_ACC_H + 48>(a, b[3]); tw_mfma<T, TW_ACC_H + 64>(a, b[4]);
            __builtin_amdgcn_sched_barrier(0);
        }
        tw_mfma<T, TW_ACC_H + 0>(an, bn[0]); tw_mfma<T, TW_ACC_H + 16>(an, bn[1]); tw_mfma<T, TW_ACC_H + 32>(an, bn[2]);
        tw_mfma<T, TW_ACC_H + 48>(an, bn[3]); tw_mfma<T, TW_ACC_H + 64>(an, bn[4]);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int k = 0; k < np; k += 2) {
        plane(k, R0);
        plane(k + 1, R1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- diagonal sums through the wave's skewed tile (its slot is dead) ------------------------------------------------------
    float* tile = (float*)(L + 64);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 32 * 64 / 4; i += 64) ((u32x4*)tile)[i] = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* wr = tile + (4 * lhi) * 64 + (l31 - 4 * lhi + 31);
    {
        const int lim = l31 < p.H ? p.H - 4 * lhi : 0;
        tw_diag5<TW_ACC_V>(tile, wr, lim, lane, p.padL - 31, p.K, mine, MF_TAPS, 1);                          // dw_v[tau][r = g]
        tw_diag5<TW_ACC_S>(tile, wr, lim, lane, MF_TAPS / 2 - 31, MF_TAPS, mine + 2 * ntl, MF_TAPS, 1);       // dw_s[tau][r = g]
    }
    {
        const int lim = l31 < p.W ? p.W - 4 * lhi : 0;
        tw_diag5<TW_ACC_H>(tile, wr, lim, lane, p.padL - 31, p.K, mine + ntl, 1, p.K);                          // dw_h[r = g][tau]
    }
    // ---- this wave's partial record; the last wave of the channel to arrive adds the channel's records in wave order -----------------
    if (idle) return;
    float* out = p.partial + (size_t)gw * ntot;
    for (int t = lane; t < ntot; t += 64) wgrad_store_partial(&out[t], mine[t]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 1;
    if (nparts > 1) {
        if (lane == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.counters + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == (unsigned)(nparts - 1);
            if (last) __hip_atomic_store(p.counters + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        last = __builtin_amdgcn_readfirstlane(last);
    }
    if (!last) return;
    for (int t = lane; t < ntot; t += 64) {
        float s = 0.f;
        const float* src = p.partial + (size_t)gw0 * ntot + t;
        for (int k0 = 0; k0 < nparts; k0 += 8) {                  // 8 loads in flight, added in wave order; agent-scope loads read at the coherence point
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = k0 + j < nparts ? __hip_atomic_load(src + (size_t)(k0 + j) * ntot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        if (t < ntl) p.dw[0][(size_t)c * ntl + t] = s;
        else if (t < 2 * ntl) p.dw[1][(size_t)c * ntl + (t - ntl)] = s;
        else p.dw[2][(size_t)c * (MF_TAPS * MF_TAPS) + (t - 2 * ntl)] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------
static bool tri_wave_enabled() {               // SLAK_TRI_WAVE=0: the one-tile planes keep the two-launch weight gradient (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_TRI_WAVE"); return !(e && e[0] == '0'); }();
    return v;
}

static bool fill_tri_wave_params(TriWaveParams& p, int N, int C, int H, int W, int K, int cus) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.padL = K / 2;
    if (N <= 0 || C <= 0 || K <= MF_TAPS || !(K & 1) || K > 63) return false;
    if (H < 15 || H > 32 || (W & 1) || W < 16 || W > 32) return false;      // (smaller planes: the plane-pair kernels)
    p.DC = (W + 7) / 8;
    if (32 * p.DC > 64 * TV_MAXJ) return false;
    if (cus < 1) cus = 1;
    int waves = cus * MF_WAVES;                                       // one four-wave workgroup per CU
    if (waves > C * N) waves = C * N;                                 // never more waves than planes
    if (waves < C) return false;                                      // (more channels than waves: a wave would have to sum up in mid-stream)
    p.waves = waves; p.base = waves / C; p.extra = waves - p.base * C;
    p.tensor_bytes = (unsigned)((size_t)N * C * H * W * 2);
    return (size_t)N * C * H * W * 2 < 0x80000000ull;
}
static size_t tri_wave_lds_bytes(int K) { return (size_t)MF_WAVES * (64 + 4 * TV_RS * TV_CPR * 16) + (size_t)MF_WAVES * (2 * K * MF_TAPS + MF_TAPS * MF_TAPS) * 4 + 32; }

static int tri_wave_cus() {
    static const int cus_env = [] { const char* e = slak_dev_getenv("SLAK_TRIWAVE_CUS"); return e ? atoi(e) : 0; }();      // (dev: pretend another CU count)
    return cus_env > 0 ? cus_env : mfma_cu_count();
}

bool dwconv_mfma_tri_wgrad_wave_supported(int N, int C, int H, int W, int K, int dtype) {
    if (!tri_wave_enabled() || (dtype != SLAK_BF16 && dtype != SLAK_F16)) return false;
    TriWaveParams p;
    return fill_tri_wave_params(p, N, C, H, W, K, tri_wave_cus());     // the launch's own CU count (256 without a device): plan and launch agree
}

size_t dwconv_mfma_tri_wgrad_wave_workspace(int N, int C, int K) {  // one record per wave, sized for up to 1024 CUs
    (void)N; (void)C;
    return align_up((size_t)4096 * (2 * K * MF_TAPS + MF_TAPS * MF_TAPS) * sizeof(float), 256);
}

template <typename T>
static int launch_tri_wave_t(TriWaveParams& p, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_tri_wgrad_wave_kernel<T>;
    const size_t lds = tri_wave_lds_bytes(p.K);
    if (!slak_set_max_lds((const void*)k, lds)) return SLAK_ERR_UNSUPPORTED;      // (process-wide maximum per kernel and device: slak_common.h)
    const int grid = (p.waves + MF_WAVES - 1) / MF_WAVES;
    if ((size_t)grid * MF_WAVES * (2 * p.K * MF_TAPS + MF_TAPS * MF_TAPS) * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_tri_wgrad_wave(const void* const* dy, const void* x, float* const* dw, int dtype,
                                      int N, int C, int H, int W, int K, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_tri_wgrad_wave_supported(N, C, H, W, K, dtype)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    TriWaveParams p;
    if (!fill_tri_wave_params(p, N, C, H, W, K, tri_wave_cus())) return SLAK_ERR_UNSUPPORTED;
    for (int b = 0; b < 3; ++b) { p.dy[b] = dy[b]; p.dw[b] = dw[b]; }
    p.x = x; p.partial = (float*)ws;
    p.counters = wgrad_arrival_counters(C);
    if (!p.counters) return SLAK_ERR_UNSUPPORTED;
    return dtype == SLAK_BF16 ? launch_tri_wave_t<bf16_t>(p, ws_bytes, st) : launch_tri_wave_t<f16_t>(p, ws_bytes, st);
}

}  // namespace slak
