// slak_amd/csrc/dwconv_mfma_small_tri_wgrad.hip -- the weight gradients of the THREE branches of a decomposed large-kernel block
// (K x 5, 5 x K, 5 x 5 on the same input: models/SLaK.py:82-100) in ONE launch on the small planes (H <= 14; W even 8..14, or
// 4..7: the 14x14 and 7x7 stages of SLaK), 16-bit activations.  The wave-independent LDS-DMA streaming of
// dwconv_mfma_small_wgrad_dma.hip (same plane layout: row-major pitch-16 images in pairs, K = 32 = the 2 x 16 image rows of a
// pair, one v_mfma_f32_16x16x32 per tap and pair, both operands ds_read_b64_tr_b16) with
//   * x fetched ONCE per plane pair for the three correlations (4 DMA instructions per pair instead of 6);
//   * the 5 x K and the 5 x 5 branch share their five x fragments (the tap shift runs along the rows for both): 26 transposing reads
//     per pair instead of 36; the K x 5 branch runs on the transposed pair (dy_v^T, x^T) as in the single-branch kernel;
//   * fifteen 4-register accumulators per wave over its whole batch slice, three skewed-tile diagonal-sum epilogues, ONE partial
//     record per (slice, channel) holding the three filters back to back, last-arriver reduction in slice order into the three dw
//     tensors: bitwise reproducible;
//   * NARROW (W < 8): one 16-byte piece per row at a 2-byte aligned source; what a piece drags in beyond column W-1 only ever
//     meets correlation entries of non-existent positions, which the diagonal sums skip; the tensor's last row is fetched early
//     and shifted into place (see dwconv_mfma_small_tri.hip).
// At this size a launch is ~10 us of fixed cost: one launch for three removes two of them per block.
//
// DG (the 14 x 14 class only: W even 8..14): the block's DATA gradient dx = sum over the three branches of dy_b correlated with the 180-degree
// rotated filter (what dwconv_mfma_small_tri_kernel<T, true, false> computes: same Toeplitz fragments, same operand orders, same fp32 adds, so
// the same bits) in the SAME launch: the plane pairs of dy_v, dy_h, dy_s are in LDS for the weight gradients anyway, dy_v^T is transposed for both,
// so the whole backward of a block's depthwise convs reads 4 planes and writes 1 where the two launches read 7 and write 1 -- and a launch's
// ~10 us of fixed cost is paid once.  The filter windows of the Toeplitz fragments (set-up only) alias the transposed images / result area.
#include "mfma_common.h"
#include <type_traits>
#include <stdlib.h>

namespace slak {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int TW_NS = 3;                // ring slots (plane pairs of the four tensors) per wave
constexpr int TW_SLOT = 4096;           // bytes per slot: [dy_v pair][dy_h pair][dy_s pair][x pair], 1024 each
constexpr int TW_RING = 64;
constexpr int TW_T = TW_RING + TW_NS * TW_SLOT + 64;      // [dy_v^T pair 1024][x^T pair 1024]
constexpr int TW_RES = TW_T + 2048 + 64;                  // res: up to 10 * 63 + 25 floats
constexpr int TW_RESN = 672;
constexpr int TW_WAVE_BYTES = TW_RES + TW_RESN * 4;
// DG: filter windows as in dwconv_mfma_small_tri.hip (16 zeros, up to 63 taps, 17 zeros per short-axis tap; two copies one element apart)
constexpr int TW_WZP = 16, TW_WLEN = 96, TW_WCH = 5;
constexpr int TW_WINB = 2 * MF_TAPS * TW_WLEN * 2;       // bytes of one branch's windows
constexpr int TW_WAVE_BYTES_DG = TW_T + 3 * TW_WINB > TW_WAVE_BYTES ? TW_T + 3 * TW_WINB : TW_WAVE_BYTES;
static_assert(TW_NS * TW_SLOT >= 5 * 16 * 32 * 4, "the diagonal-sum tiles alias the ring");

struct SmallTriWgradParams {
    const void* dy[3]; const void* x; float* partial; float* dw[3]; unsigned* counters;
    int N, C, H, W, K;
    int images_per_slice, slices;
    unsigned tensor_bytes;
    int dbg;                             // dev switches of the quad kernel (SLAK_QW_DBG): 1 no compute, 2 no DMA, 4 no diagonal sums, 8 no octets
    const float* w[3]; void* dx;         // DG only: the three filters, the data gradient
};
#ifdef SLAK_QW_DEV                     // dev builds only (SLAK_BUILD_DEFS=-DSLAK_QW_DEV): the shipped kernel compiles the experiments out
#define QW_DBG(bit) (p.dbg & (bit))
#else
#define QW_DBG(bit) 0
#endif

template <typename T> __device__ __forceinline__ f32x4_t tw_mfma16(s16x8 a, s16x8 b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t tw_mfma16<bf16_t>(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t tw_mfma16<f16_t>(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// Diagonal sums of the five per-tap correlation tiles of a branch in ONE pass: tile = 5 x [16][32] floats of the wave's (dead)
// streaming area.  Every accumulator entry (o, i) of tap r that takes part lands on row o, column pos(i) - pos(o) + 15 of tile r; lane
// (r, column) then adds its column's 16 rows top to bottom (fixed order: reproducible) and hands the sum to `emit`.  One write
// phase and one read phase per branch (the per-tap version synchronised ten times per branch: its LDS latencies were 4.6 us of
// the 7 x 7 launch's 20).
constexpr int TW_DIAG_BYTES = MF_TAPS * 16 * 32 * 4;
template <typename F>
__device__ __forceinline__ void tw_diag5(float* tile, int lane, const f32x4_t (&acc)[MF_TAPS], const bool (&wok)[4], const int (&wofs)[4], F&& emit) {
#pragma unroll
    for (int k = 0; k < TW_DIAG_BYTES / 1024; ++k) *(u32x4*)((char*)tile + (k * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (wok[e]) tile[r * 512 + wofs[e]] = acc[r][e];
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = lane + 64 * k;                            // (tap, column): 5 x 31
        if (idx < MF_TAPS * 31) {
            const int r = idx / 31, col = idx - r * 31;
            float v[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) v[o] = tile[r * 512 + o * 32 + col];      // 16 independent reads, added in order below
            float sum = 0.f;
#pragma unroll
            for (int o = 0; o < 16; ++o) sum += v[o];
            emit(r, col, sum);
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <typename T, bool NARROW, bool DG>
__global__ __launch_bounds__(MF_THREADS) void dwconv_mfma_small_tri_wgrad_kernel(const SmallTriWgradParams p) {
    static_assert(!(DG && NARROW), "the data gradient rides along on the 14 x 14 class only");
    constexpr int TW_WAVE_BYTES = DG ? slak::TW_WAVE_BYTES_DG : slak::TW_WAVE_BYTES;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + 3) >> 2;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c0 = cb * 4, c = c0 + wave;
    const int nch = p.C - c0 < 4 ? p.C - c0 : 4;
    const int n_begin = slice * p.images_per_slice;
    int n_end = n_begin + p.images_per_slice; if (n_end > p.N) n_end = p.N;
    const bool live = c < p.C && n_begin < n_end;                // (every wave reaches the finish: it has workgroup barriers)
    const int npairs = live ? (n_end - n_begin + 1) >> 1 : 0;
    char* const L = (char*)lds + wave * TW_WAVE_BYTES;           // this wave's private region
    const int HW = p.H * p.W;
    const int nt_long = p.K * MF_TAPS, ntot = 2 * nt_long + 25;  // [K x 5][5 x K][5 x 5] back to back

    // DG: this channel's filters leave first (their latency runs under the zero fill)
    const int ntap = p.K * MF_TAPS;
    float fv[TW_WCH], fh[TW_WCH], fs = 0.f;
    if constexpr (DG) {
        const int cw = live ? c : 0;
#pragma unroll
        for (int k = 0; k < TW_WCH; ++k) {
            const int e = lane + 64 * k;
            fv[k] = e < ntap ? p.w[0][(size_t)cw * ntap + e] : 0.f;
            fh[k] = e < ntap ? p.w[1][(size_t)cw * ntap + e] : 0.f;
        }
        if (lane < 25) fs = p.w[2][(size_t)cw * 25 + lane];
    }
    for (int o = lane * 16; o < TW_WAVE_BYTES; o += 64 * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // zeros are in place before any DMA can land on them
    if constexpr (DG) {
        // filter windows, rotated by 180 degrees: branch b at TW_T + b * TW_WINB (the transposed images and the result area: unused until
        // the first pair has landed; cleared again below).  Written BEFORE the first DMA leaves: the filter loads are vector-memory
        // operations as well, and the wait the compiler puts in front of their first use then has nothing else to wait for.
        auto put = [&](int b, int r, int t, int KL, float v) {        // short tap r, long tap t of branch b
            r = MF_TAPS - 1 - r; t = KL - 1 - t;
            const uint16_t h = cvt_to_bits(v, (T*)nullptr);
            uint16_t* win = (uint16_t*)(L + TW_T + b * TW_WINB);
            win[r * TW_WLEN + TW_WZP + t] = h;
            win[MF_TAPS * TW_WLEN + r * TW_WLEN + TW_WZP + t - 1] = h;
        };
#pragma unroll
        for (int k = 0; k < TW_WCH; ++k) {
            const int e = lane + 64 * k;
            if (e < ntap) {
                put(0, e % MF_TAPS, e / MF_TAPS, p.K, fv[k]);        // (K,5): element [t][r]
                put(1, e / p.K, e - (e / p.K) * p.K, p.K, fh[k]);    // (5,K): element [r][t]
            }
        }
        if (lane < 25) put(2, lane / 5, lane % 5, 5, fs);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- DMA: lane -> (plane of the pair, image row, half of the row: columns 0..7 / W-8..W-1) ------------------------
    v4i_t rs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint64_t a = (uint64_t)(t < 3 ? p.dy[t] : p.x);
        rs[t][0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs[t][1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs[t][2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs[t][3] = 0x00020000;
    }
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;
    const int d_pp = lane >> 5, d_row = (lane >> 1) & 15, d_half = lane & 1;
    const unsigned d_src = (unsigned)d_pp * gplane_b + (unsigned)(d_row * p.W) * 2 + ((!NARROW && d_half) ? (unsigned)(p.W - 8) * 2 : 0u);
    const bool d_rowok = d_row < p.H && (!NARROW || d_half == 0);
    const unsigned lds_wave = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds) + wave * TW_WAVE_BYTES;
    const unsigned chan_b = (unsigned)(live ? c : 0) * (unsigned)HW * 2;
    const unsigned last_row_b = p.tensor_bytes - (unsigned)(2 * p.W);     // byte offset of the tensor's last image row
    auto issue_pair = [&](int q) {                                // four DMA instructions -> slot q % TW_NS
        const int n0 = n_begin + 2 * q;
        const unsigned gb = (unsigned)n0 * gplane_b + chan_b;
        const unsigned dst = lds_wave + TW_RING + (unsigned)(q % TW_NS) * TW_SLOT;
        if (d_rowok && n0 + d_pp < n_end) {
            unsigned so = gb + d_src;
            if (NARROW && so == last_row_b) so -= (unsigned)(16 - 2 * p.W);     // would end behind the tensor: fetched early, shifted below
#pragma unroll
            for (int t = 0; t < 4; ++t) lds_dma16(so, rs[t], __builtin_amdgcn_readfirstlane(dst + t * 1024));
        }
    };
    for (int q = 0; q < TW_NS - 1 && q < npairs; ++q) issue_pair(q);

    // ---- lane constants (see dwconv_mfma_small_wgrad_dma.hip) ----------------------------------------------------------
    const int g4 = lane >> 4, i16 = lane & 15;
    // DG: Toeplitz fragments (dwconv_mfma_small_tri.hip): lane -> (o = long-axis output position, k-group g4 -> tap-in-pair rsel, half of the
    // 16 k-slots); then the window area becomes zeros again (guard rows of the transposed images)
    const int rsel = g4 >> 1, half = g4 & 1;
    s16x8 tf[DG ? 3 : 1][3];
    if constexpr (DG) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const bool vert = b == 0;
            const int padL = (b == 2 ? 5 : p.K) / 2;
            const int i0 = half ? (vert ? 8 : p.W - 8) : 0;
            const int a = TW_WZP + i0 - i16 + padL;
            const int par = a & 1;
            const unsigned* src = (const unsigned*)(L + TW_T + b * TW_WINB + par * MF_TAPS * TW_WLEN * 2) + ((a - par) >> 1);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int r = 2 * m + rsel;
                u32x4 d;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    d[k] = r < MF_TAPS ? src[(r < MF_TAPS ? r : 0) * (TW_WLEN / 2) + k] : 0u;
                    if (!vert && half && 2 * k < 16 - p.W) d[k] = 0u;     // columns already covered by the first half
                }
                tf[b][m] = __builtin_bit_cast(s16x8, d);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int o = lane * 16; o < TW_WAVE_BYTES - TW_T; o += 64 * 16) *(u32x4*)(L + TW_T + o) = u32x4{0u, 0u, 0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // DG: fragment addresses and stores of the data gradient (output row i16, columns 4 g4 .. 4 g4 + 3)
    const unsigned xlane = (unsigned)(i16 * 32 + rsel * 32 + half * 16);
    const unsigned zlane = (unsigned)(half * 16);                    // the 64 zero bytes in front of the ring
    const unsigned ooff = (unsigned)(i16 * p.W + 4 * g4) * 2;
    const bool st0 = i16 < p.H && 4 * g4 < p.W, st1 = i16 < p.H && 4 * g4 + 2 < p.W;
    __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(DG ? p.dx : nullptr, 0, (int)p.tensor_bytes, 0x00020000);
    auto dfrag = [&](unsigned base, int m) -> s16x8 {              // MFMA m of a plane whose guarded image starts 64 bytes after `base`
        const unsigned a = (m == 2 && rsel) ? zlane : base + xlane + m * 64;
        return __builtin_bit_cast(s16x8, *(const u32x4*)(L + a));
    };
    const unsigned rd = (unsigned)((g4 >> 1) * 512 + ((g4 & 1) * 8 + (i16 >> 2)) * 32 + (i16 & 3) * 8);
    const unsigned trd = (unsigned)((4 * g4 + (i16 >> 2)) * 32 + (i16 & 3) * 8);
    const int t_row = i16 < 8 ? i16 : i16 - (16 - p.W);
    const bool twr_ok = NARROW ? i16 < p.W : (i16 < 8 || t_row >= 8);
    const unsigned twr = (unsigned)(t_row * 32 + g4 * 8);
    auto frag = [&](unsigned addr) -> s16x8 {                     // 8 k of one column: two transposing reads, 4 rows apart
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr + 128));
        return s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    };

    f32x4_t av[MF_TAPS], ah[MF_TAPS], as[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) { av[r] = f32x4_t{0.f, 0.f, 0.f, 0.f}; ah[r] = av[r]; as[r] = av[r]; }

    for (int q = 0; q < npairs; ++q) {
        {
            int dm = npairs - 1 - q; if (dm > TW_NS - 2) dm = TW_NS - 2;
            // only the DMAs of the pairs behind this one may be outstanding (DG: and the four stores of the pair in front of it, issued after its DMAs)
            wait_vmcnt_dyn(4 * dm + ((DG && q > 0) ? 4 : 0));
        }
        __builtin_amdgcn_wave_barrier();
        const int n0 = n_begin + 2 * q;
        const unsigned slot = (unsigned)TW_RING + (unsigned)(q % TW_NS) * TW_SLOT;
        if (q == npairs - 1 && ((n_end - n_begin) & 1)) {
            // odd slice: the second plane of the last pair was not fetched and its rows still hold an older plane: clear them
            // (32 lanes x 16 bytes per tensor)
#pragma unroll
            for (int t = 0; t < 2; ++t) *(u32x4*)(L + slot + 512 + ((lane >> 5) + 2 * t) * 1024 + (lane & 31) * 16) = u32x4{0u, 0u, 0u, 0u};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (NARROW && c == p.C - 1 && n0 + 1 >= p.N - 1 && n0 <= p.N - 1) {       // (wave-uniform) this pair holds the tensor's last plane
            const int ppl = p.N - 1 - n0, sh = 8 - p.W;             // its last row arrived `sh` elements late: shift it into place
            if (lane < 4) {
                char* rowp = L + slot + lane * 1024 + ppl * 512 + (p.H - 1) * 32;
                const u32x4 o = *(const u32x4*)rowp;
                const unsigned oo[6] = {o[0], o[1], o[2], o[3], 0u, 0u};
                const int wsh = (16 * sh) >> 5, bsh = (16 * sh) & 31;
                u32x4 nv;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned lo = 0u, hi = 0u;
#pragma unroll
                    for (int j = 0; j < 6; ++j) { if (j == k + wsh) lo = oo[j]; if (j == k + wsh + 1) hi = oo[j]; }
                    nv[k] = bsh ? ((lo >> bsh) | (hi << (32 - bsh))) : lo;
                }
                *(u32x4*)rowp = nv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        // vertical branch: dy_v pair and x pair transposed (tensor 0 and tensor 3 of the slot)
#pragma unroll
        for (int k = 0; k < 4; ++k) {                             // dy_v plane 0,1 -> T + 0, 512;  x plane 0,1 -> T + 1024, 1536
            const unsigned src = slot + (k < 2 ? 0u : 3072u) + (k & 1) * 512 + trd;
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + src));
            if (twr_ok) *(s16x4*)(L + TW_T + k * 512 + twr) = v;
        }
        const s16x8 a_h = frag(slot + 1024 + rd), a_s = frag(slot + 2048 + rd);
        const unsigned xb = slot + 3072 - 64 + rd;                // x pair minus two rows: the tap shift is +r rows
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            const s16x8 b = frag(xb + r * 32);
            ah[r] = tw_mfma16<T>(a_h, b, ah[r]);
            as[r] = tw_mfma16<T>(a_s, b, as[r]);
        }
        const s16x8 a_v = frag((unsigned)TW_T + rd);
        const unsigned xtb = (unsigned)TW_T + 1024 - 64 + rd;
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) av[r] = tw_mfma16<T>(a_v, frag(xtb + r * 32), av[r]);
        if constexpr (DG) {
            const unsigned bv = (unsigned)TW_T - 64;                // dy_v^T images
            const unsigned bh = slot + 1024u - 64, bs = slot + 2048u - 64;      // row-major images of dy_h, dy_s
            const unsigned gb = (unsigned)n0 * gplane_b + chan_b;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                if (pp == 1 && n0 + 1 >= n_end) break;              // (wave-uniform) odd slice: the second plane does not exist
                f32x4_t dv = {0.f, 0.f, 0.f, 0.f}, dh = dv, ds = dv;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    dv = tw_mfma16<T>(dfrag(bv + pp * 512, m), tf[0][m], dv);       // operands swapped: D^T = dy_v^T-tile x T^T
                    dh = tw_mfma16<T>(tf[1][m], dfrag(bh + pp * 512, m), dh);
                    ds = tw_mfma16<T>(tf[2][m], dfrag(bs + pp * 512, m), ds);
                }
                const f32x4_t sum = (dv + dh) + ds;                 // the three partial gradients, added in fp32
                const unsigned go = gb + pp * gplane_b;
                if (st0) __builtin_amdgcn_raw_buffer_store_b32(pack2<T>(sum[0], sum[1]), rdx, ooff, go, 0);
                if (st1) __builtin_amdgcn_raw_buffer_store_b32(pack2<T>(sum[2], sum[3]), rdx, ooff + 4, go, 0);
            }
        }
        if (q + TW_NS - 1 < npairs) issue_pair(q + TW_NS - 1);      // into the slot pair q-1 used
    }

    // ---- diagonal sums, one branch after the other, through the skewed 16 x 32 tile (the ring is dead) ------------------------
    if (live) {
        float* tile = (float*)(L + TW_RING);
        float* res = (float*)(L + TW_RES);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        auto diag = [&](const f32x4_t (&acc)[MF_TAPS], bool vert, int Wt, int KL, int kw, float* out) {
            const int padL = KL / 2;
            const int dup = vert ? 0 : 16 - Wt;
            auto pos = [&](int sl) { return sl < 8 ? sl : sl - dup; };
            auto valid = [&](int sl) { const int qq = pos(sl); return sl < 8 ? sl < Wt : (qq >= 8 && qq < Wt); };
            const int pi = pos(i16);
            const bool vi = valid(i16);
            int wofs[4]; bool wok[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int so = 4 * g4 + e;
                wok[e] = vi && valid(so);
                wofs[e] = so * 32 + (pi - pos(so) + 15);
            }
            tw_diag5(tile, lane, acc, wok, wofs, [&](int r, int col, float sum) {
                const int tau = col - 15 + padL;
                if (tau >= 0 && tau < KL) out[vert ? (tau * kw + r) : (r * kw + tau)] = sum;
            });
        };
        {   // vertical branch: its positions are the image rows, slot = position, so an entry's diagonal is its slot difference: rotating every
            // accumulator row left by its own index (DPP row_ror) lines the diagonals up in the lanes -- d >= 0 and d < 0 apart (|d| <= 13
            // wraps around 16 lanes) -- without the skewed LDS tile (which is latency bound: a third of the 4 us the three branches' sums cost)
            auto ror = [](float v, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + decltype(ctrl)::value, 0xf, 0xf, false)); };
            auto rows = [](float v) {
                int ti = __float_as_int(v);
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 12, 0x2, 0xf, false);     // 16-lane row 1: left by 4
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 8, 0x4, 0xf, false);      // row 2: left by 8
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 4, 0x8, 0xf, false);      // row 3: left by 12
                float t = __int_as_float(ti);
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                return t;
            };
            const int padL = p.K / 2;
            bool okp[4], okn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int so = 4 * g4 + e; const bool ok = i16 < p.H && so < p.H; okp[e] = ok && i16 >= so; okn[e] = ok && i16 < so; }
            const int tp = i16 + padL, tn = i16 - 16 + padL;
#pragma unroll
            for (int r = 0; r < MF_TAPS; ++r) {
                float a = okp[0] ? av[r][0] : 0.f, b = okn[0] ? av[r][0] : 0.f;
                a += ror(okp[1] ? av[r][1] : 0.f, std::integral_constant<int, 15>{}); b += ror(okn[1] ? av[r][1] : 0.f, std::integral_constant<int, 15>{});
                a += ror(okp[2] ? av[r][2] : 0.f, std::integral_constant<int, 14>{}); b += ror(okn[2] ? av[r][2] : 0.f, std::integral_constant<int, 14>{});
                a += ror(okp[3] ? av[r][3] : 0.f, std::integral_constant<int, 13>{}); b += ror(okn[3] ? av[r][3] : 0.f, std::integral_constant<int, 13>{});
                a = rows(a); b = rows(b);
                if (lane < 16 && tp < p.K) res[tp * MF_TAPS + r] = a;          // d = lane >= 0
                if (lane < 16 && tn >= 0 && tn < p.K) res[tn * MF_TAPS + r] = b;  // d = lane - 16 < 0
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        diag(ah, false, p.W, p.K, p.K, res + nt_long);
        if (!NARROW && p.W == 14) {
            // 5 x 5 branch on 14-wide planes by rotations as well: column slots 0..7 hold columns 0..7, slots 10..15 columns 8..13 (the row's
            // second 16-byte piece starts at column 6), so an entry's diagonal is its slot difference, minus 2 where the x slot is in the
            // upper half and the dY slot in the lower, plus 2 the other way round.  The halves of the dY slots are whole 16-lane rows (4 g4 + e),
            // so the correction is one more row-masked rotation of the partial sums of the lower / upper x slots; only |d| <= 2 is needed.
            auto ror = [](float v, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + decltype(ctrl)::value, 0xf, 0xf, false)); };
            auto rows = [](float v) {                              // 16-lane row g4: left by 4 g4
                int ti = __float_as_int(v);
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 12, 0x2, 0xf, false);
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 8, 0x4, 0xf, false);
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 4, 0x8, 0xf, false);
                return __int_as_float(ti);
            };
            auto vslot = [](int sl) { return sl < 8 || sl >= 10; };
            bool oklo[4], okhi[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const bool ok = vslot(i16) && vslot(4 * g4 + e); oklo[e] = ok && i16 < 8; okhi[e] = ok && i16 >= 8; }
            const int d = i16 < 8 ? i16 : i16 - 16, tau = d + 2;
            const bool emit = lane < 16 && tau >= 0 && tau < MF_TAPS;
            float* const out = res + 2 * nt_long;
#pragma unroll
            for (int r = 0; r < MF_TAPS; ++r) {
                float lo = oklo[0] ? as[r][0] : 0.f, hi = okhi[0] ? as[r][0] : 0.f;
                lo += ror(oklo[1] ? as[r][1] : 0.f, std::integral_constant<int, 15>{}); hi += ror(okhi[1] ? as[r][1] : 0.f, std::integral_constant<int, 15>{});
                lo += ror(oklo[2] ? as[r][2] : 0.f, std::integral_constant<int, 14>{}); hi += ror(okhi[2] ? as[r][2] : 0.f, std::integral_constant<int, 14>{});
                lo += ror(oklo[3] ? as[r][3] : 0.f, std::integral_constant<int, 13>{}); hi += ror(okhi[3] ? as[r][3] : 0.f, std::integral_constant<int, 13>{});
                lo = rows(lo); hi = rows(hi);
                // dY slots of rows 0, 1 are the lower half, of rows 2, 3 the upper: upper x slots against lower dY slots sit 2 lanes too high,
                // lower x slots against upper dY slots 2 lanes too low
                int hi_i = __float_as_int(hi), lo_i = __float_as_int(lo);
                hi_i = __builtin_amdgcn_update_dpp(hi_i, hi_i, 0x120 + 14, 0x3, 0xf, false);    // rows 0, 1: left by 2
                lo_i = __builtin_amdgcn_update_dpp(lo_i, lo_i, 0x120 + 2, 0xc, 0xf, false);     // rows 2, 3: right by 2
                float t = __int_as_float(hi_i) + __int_as_float(lo_i);
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (emit) out[r * MF_TAPS + tau] = t;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else
        diag(as, false, p.W, MF_TAPS, MF_TAPS, res + 2 * nt_long);
        float* out = p.partial + ((size_t)slice * p.C + c) * ntot;
        for (int t = lane; t < ntot; t += 64) wgrad_store_partial(&out[t], res[t]);       // taps no diagonal reaches stay 0
    } else if (c < p.C) {                                         // empty slice: its partial must still be zero
        float* out = p.partial + ((size_t)slice * p.C + c) * ntot;
        for (int t = lane; t < ntot; t += 64) wgrad_store_partial(&out[t], 0.f);
    }
    // ---- last arriver of the channel block adds the slices in order and scatters into the three dw tensors (see wgrad_finish) ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* lds_flag = (int*)lds;
    if (tid == 0) {
        int last = 1;
        if (p.slices > 1) {
            const unsigned old = __hip_atomic_fetch_add(p.counters + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == (unsigned)(p.slices - 1);
            if (last) __hip_atomic_store(p.counters + cb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *lds_flag = last;
    }
    __syncthreads();
    if (!*lds_flag) return;
    for (int t = tid; t < nch * ntot; t += MF_THREADS) {
        float s = 0.f;
        const float* src = p.partial + (size_t)c0 * ntot + t;
        const size_t stride = (size_t)p.C * ntot;
        for (int k0 = 0; k0 < p.slices; k0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = k0 + j < p.slices ? __hip_atomic_load(src + (size_t)(k0 + j) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        const int ch = t / ntot, e = t - ch * ntot;
        const size_t cc = (size_t)(c0 + ch);
        if (e < nt_long) p.dw[0][cc * nt_long + e] = s;
        else if (e < 2 * nt_long) p.dw[1][cc * nt_long + (e - nt_long)] = s;
        else p.dw[2][cc * 25 + (e - 2 * nt_long)] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// QUAD: planes of at most 7 x 7 -- EIGHT planes of one channel per MFMA (layout of dwconv_mfma_small_quad_kernel,
// dwconv_mfma_small_tri.hip: a 16-row x 32-byte tile holds four planes, rows 7 / 8 and two rows behind it are zero guards;
// K = 32 = the 2 x 16 rows of two such tiles).  The correlation tile C_r[dy position][x position] then holds two valid 7 x 7
// diagonal blocks (the two plane sets of a tile axis) whose diagonals are summed together; the off-diagonal blocks (products of
// different planes) are skipped by the diagonal sums.  Per eight planes: 15 MFMAs (plane-pair kernel: 60).
// Rows of 2W bytes start at 2-byte aligned addresses; the LDS-DMA takes those only through the texture unit's slow path (measured
// ~50 cycles per instruction for 28 pieces: 10.9 of the kernel's 27.6 us).  So the rows travel through registers: every lane loads
// the four dword-ALIGNED dwords that cover its row (one buffer_load_dwordx4 per tensor for the 56 rows of an octet), v_alignbit
// moves the row to bit 0, the elements behind column W-1 (the next row's) are cleared, and one ds_write_b128 puts the piece in
// place.  Two octets are in flight per wave (two register sets); a lane of a plane or row that does not exist loads through an
// out-of-range offset and writes zeros, which also keeps the guard rows zero.
constexpr int QW_TILE = 512 + 64;
constexpr int QW_TEN = 2 * QW_TILE;     // two tiles (an octet of planes) of one tensor
constexpr int QW_SLOT = 4 * QW_TEN;     // [dy_v][dy_h][dy_s][x]
constexpr int QW_T = TW_RING + QW_SLOT;                   // [dy_v^T: 2 x 512][64 zero][x^T tile 0][64 zero][x^T tile 1][64 zero]
constexpr int QW_XT = QW_T + 1024 + 64;
constexpr int QW_RES = QW_XT + 2 * QW_TILE;
constexpr int QW_WAVE_BYTES = QW_RES + TW_RESN * 4;
constexpr unsigned QW_OOB = 0x80000000u;


template <typename T>
__global__ __launch_bounds__(MF_THREADS) void dwconv_mfma_small_quad_wgrad_kernel(const SmallTriWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + 3) >> 2;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c0 = cb * 4, c = c0 + wave;
    const int nch = p.C - c0 < 4 ? p.C - c0 : 4;
    const int n_begin = slice * p.images_per_slice;               // (a multiple of 8)
    int n_end = n_begin + p.images_per_slice; if (n_end > p.N) n_end = p.N;
    const bool live = c < p.C && n_begin < n_end;                // (every wave reaches the finish: it has workgroup barriers)
    const int noct = (live && !QW_DBG(8)) ? (n_end - n_begin + 7) >> 3 : 0;
    char* const L = (char*)lds + wave * QW_WAVE_BYTES;
    const int HW = p.H * p.W;
    const int nt_long = p.K * MF_TAPS, ntot = 2 * nt_long + 25;  // [K x 5][5 x K][5 x 5] back to back

    for (int o = lane * 16; o < QW_WAVE_BYTES; o += 64 * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};

    // ---- loads: lane -> (tile of the octet, tile row, byte half) = (plane, image row) ------------------------------------
    __amdgpu_buffer_rsrc_t rs[4];
#pragma unroll
    // (range: up to the end of the dword that holds the tensor's last element -- the check is per dword, and an aligned dword that
    // holds a valid element lies inside the allocation)
    for (int t = 0; t < 4; ++t) rs[t] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(t < 3 ? p.dy[t] : p.x), 0, (int)((p.tensor_bytes + 3u) & ~3u), 0x00020000);
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;
    const int d_r = (lane >> 1) & 15, d_half = lane & 1, d_ab = d_r >= 9 ? 1 : 0, d_row = d_r - 9 * d_ab;
    const int d_pl = 4 * (lane >> 5) + d_ab + 2 * d_half;         // plane of the octet
    const bool d_ok = d_r != 7 && d_r != 8 && d_row < p.H;
    const unsigned d_row_b = (unsigned)(live ? c : 0) * (unsigned)HW * 2 + (unsigned)d_pl * gplane_b + (unsigned)(d_row * p.W) * 2;
    const unsigned d_sh = (d_row_b & 2u) * 8u;                    // the row starts in the upper half of its first dword (8 | n_begin: the same for every octet)
    const unsigned d_dst = (unsigned)TW_RING + (unsigned)((lane >> 5) * QW_TILE + (lane & 31) * 16);
    const unsigned bm2 = 5 < p.W ? 0xffffffffu : (4 < p.W ? 0xffffu : 0u), bm3 = 7 < p.W ? 0xffffffffu : (6 < p.W ? 0xffffu : 0u);
    auto load_oct = [&](int q, u32x4 (&R)[4]) {
        const int n0 = n_begin + 8 * q;
        // (an octet behind the slice loads nothing: the instruction count per step stays fixed)
        const unsigned a = (d_ok && n0 + d_pl < n_end && !QW_DBG(2)) ? (((unsigned)n0 * gplane_b + d_row_b) & ~3u) : QW_OOB;
#pragma unroll
        for (int t = 0; t < 4; ++t) R[t] = __builtin_amdgcn_raw_buffer_load_b128(rs[t], a, 0, 0);      // dwords behind the tensor: 0
    };
    auto stage = [&](const u32x4 (&R)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 v;
            v[0] = __builtin_amdgcn_alignbit(R[t][1], R[t][0], d_sh);
            v[1] = __builtin_amdgcn_alignbit(R[t][2], R[t][1], d_sh);
            v[2] = __builtin_amdgcn_alignbit(R[t][3], R[t][2], d_sh) & bm2;
            v[3] = __builtin_amdgcn_alignbit(0u, R[t][3], d_sh) & bm3;
            *(u32x4*)(L + d_dst + t * QW_TEN) = v;
        }
    };
    u32x4 R0[4], R1[4];
    load_oct(0, R0);
    load_oct(1, R1);

    // ---- lane constants ------------------------------------------------------------------------------------------------
    const int g4 = lane >> 4, i16 = lane & 15;
    const unsigned rdr = (unsigned)(((g4 & 1) * 8 + (i16 >> 2)) * 32 + (i16 & 3) * 8);
    const unsigned rdq = (unsigned)((g4 >> 1) * QW_TILE) + rdr;   // k-half -> tile of the octet (tiles QW_TILE apart)
    const unsigned rdt = (unsigned)((g4 >> 1) * 512) + rdr;       // dy_v^T tiles are 512 apart
    const unsigned trd = (unsigned)((4 * g4 + (i16 >> 2)) * 32 + (i16 & 3) * 8);
    const bool twr_ok = (i16 & 7) < p.W;
    const unsigned twr = (unsigned)(((i16 >> 3) * 9 + (i16 & 7)) * 32 + g4 * 8);   // transposed tiles carry the 9-row pitch as well
    auto frag = [&](unsigned addr) -> s16x8 {                     // 8 k of one column: two transposing reads, 4 rows apart
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr + 128));
        return s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    };

    f32x4_t av[MF_TAPS], ah[MF_TAPS], as[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) { av[r] = f32x4_t{0.f, 0.f, 0.f, 0.f}; ah[r] = av[r]; as[r] = av[r]; }

    constexpr unsigned slot = TW_RING;
    auto octet = [&](int q, u32x4 (&R)[4]) {
        stage(R);                                                 // (the LDS queue is in order: the reads of octet q-1 are behind us)
        load_oct(q + 2, R);                                       // unconditional: the compiler's vmcnt bookkeeping needs one count on every path
        if (q >= noct || QW_DBG(1)) return;
        // vertical branch: the dy_v tiles and the x tiles transposed (tensor 0 and tensor 3 of the slot)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned src = slot + (k < 2 ? 0u : 3u * QW_TEN) + (k & 1) * QW_TILE + trd;
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + src));
            if (twr_ok) *(s16x4*)(L + (k < 2 ? QW_T + (k & 1) * 512 : QW_XT + (k & 1) * QW_TILE) + twr) = v;
        }
        const s16x8 a_h = frag(slot + QW_TEN + rdq), a_s = frag(slot + 2 * QW_TEN + rdq);
        const unsigned xb = slot + 3 * QW_TEN - 64 + rdq;         // x tiles minus two rows: the tap shift is +r rows
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            const s16x8 b = frag(xb + r * 32);
            ah[r] = tw_mfma16<T>(a_h, b, ah[r]);
            as[r] = tw_mfma16<T>(a_s, b, as[r]);
        }
        const s16x8 a_v = frag((unsigned)QW_T + rdt);
        const unsigned xtb = (unsigned)QW_XT - 64 + rdq;
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) av[r] = tw_mfma16<T>(a_v, frag(xtb + r * 32), av[r]);
    };
    for (int q = 0; q < noct; q += 2) {
        octet(q, R0);
        octet(q + 1, R1);
    }

    // ---- diagonal sums of the two valid blocks, one branch after the other, through the skewed 16 x 32 tile (the slot is dead) ----
    if (live) {
        float* res = (float*)(L + QW_RES);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Diagonal sums WITHOUT the skewed LDS tile: inside a valid block the slot difference of an entry IS its diagonal (both tile axes carry
        // the same pitch), and |d| <= 6 < 8, so rotating every accumulator row left by its own row index (DPP row_ror: e by immediate, 4 * g4
        // per 16-lane row through row masks) lines diagonal d up in lane d mod 16; four adds over the registers, two over the rows.  The LDS-tile
        // version (tw_diag5) was latency bound: 4 of this launch's 20 us.
        auto ror = [](float v, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + decltype(ctrl)::value, 0xf, 0xf, false)); };
        auto diag = [&](const f32x4_t (&acc)[MF_TAPS], bool vert, int Wt, int KL, int kw, float* out) {
            const int padL = KL / 2;
            // slot -> (block, position): tile rows carry the 9 pitch (vertical branch: positions are rows), byte halves the 8 pitch
            auto blk = [&](int sl) { return vert ? (sl >= 9 ? 1 : 0) : (sl >> 3); };
            auto pos = [&](int sl) { return vert ? sl - 9 * (sl >= 9 ? 1 : 0) : (sl & 7); };
            auto valid = [&](int sl) { return (!vert || (sl != 7 && sl != 8)) && pos(sl) < Wt; };
            const bool vi = valid(i16);
            bool wok[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int so = 4 * g4 + e; wok[e] = vi && valid(so) && blk(so) == blk(i16); }   // (off-diagonal blocks: products of different planes)
            const int d = i16 < 8 ? i16 : i16 - 16;               // the diagonal that ends up in this lane
            const int tau = d + padL;
            const bool emit = lane < 16 && i16 != 8 && tau >= 0 && tau < KL;
#pragma unroll
            for (int r = 0; r < MF_TAPS; ++r) {
                // element (row so = 4 g4 + e, slot j) -> lane (j - so) mod 16: a left rotation by so = a right rotation by 16 - so
                float t = wok[0] ? acc[r][0] : 0.f;
                t += ror(wok[1] ? acc[r][1] : 0.f, std::integral_constant<int, 15>{});
                t += ror(wok[2] ? acc[r][2] : 0.f, std::integral_constant<int, 14>{});
                t += ror(wok[3] ? acc[r][3] : 0.f, std::integral_constant<int, 13>{});
                int ti = __float_as_int(t);
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 12, 0x2, 0xf, false);     // 16-lane row 1: left by 4
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 8, 0x4, 0xf, false);      // row 2: left by 8
                ti = __builtin_amdgcn_update_dpp(ti, ti, 0x120 + 4, 0x8, 0xf, false);      // row 3: left by 12
                t = __int_as_float(ti);
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (emit) out[vert ? (tau * kw + r) : (r * kw + tau)] = t;
            }
        };
        if (!QW_DBG(4)) {
        diag(av, true, p.H, p.K, MF_TAPS, res);
        diag(ah, false, p.W, p.K, p.K, res + nt_long);
        diag(as, false, p.W, MF_TAPS, MF_TAPS, res + 2 * nt_long);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float* out = p.partial + ((size_t)slice * p.C + c) * ntot;
        for (int t = lane; t < ntot; t += 64) wgrad_store_partial(&out[t], res[t]);       // taps no diagonal reaches stay 0
    } else if (c < p.C) {                                         // empty slice: its partial must still be zero
        float* out = p.partial + ((size_t)slice * p.C + c) * ntot;
        for (int t = lane; t < ntot; t += 64) wgrad_store_partial(&out[t], 0.f);
    }
    // ---- last arriver of the channel block adds the slices in order and scatters into the three dw tensors (see wgrad_finish) ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* lds_flag = (int*)lds;
    if (tid == 0) {
        int last = 1;
        if (p.slices > 1) {
            const unsigned old = __hip_atomic_fetch_add(p.counters + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == (unsigned)(p.slices - 1);
            if (last) __hip_atomic_store(p.counters + cb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *lds_flag = last;
    }
    __syncthreads();
    if (!*lds_flag) return;
    for (int t = tid; t < nch * ntot; t += MF_THREADS) {
        float s = 0.f;
        const float* src = p.partial + (size_t)c0 * ntot + t;
        const size_t stride = (size_t)p.C * ntot;
        for (int k0 = 0; k0 < p.slices; k0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = k0 + j < p.slices ? __hip_atomic_load(src + (size_t)(k0 + j) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        const int ch = t / ntot, e = t - ch * ntot;
        const size_t cc = (size_t)(c0 + ch);
        if (e < nt_long) p.dw[0][cc * nt_long + e] = s;
        else if (e < 2 * nt_long) p.dw[1][cc * nt_long + (e - nt_long)] = s;
        else p.dw[2][cc * 25 + (e - 2 * nt_long)] = s;
    }
}

static bool quad_wgrad_enabled() {             // SLAK_SMALL_QUAD=0 keeps the plane-pair kernel on 7 x 7 (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_SMALL_QUAD"); return !(e && e[0] == '0'); }();
    return v;
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_stw_params(SmallTriWgradParams& p, int N, int C, int H, int W, int K, int target_wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K;
    if (N <= 0 || C <= 0 || K <= MF_TAPS || !(K & 1) || K > 63) return false;
    if (H > 14 || H < 1) return false;
    if (W >= 8 ? (W > 14 || (W & 1)) : W < 4) return false;
    const int cblocks = (C + 3) / 4;
    int slices = target_wgs / cblocks; if (slices < 1) slices = 1;
    int per = (N + slices - 1) / slices; per = (per + 1) & ~1;      // whole pairs
    if (per < 8) per = 8;
    if (per > ((N + 1) & ~1)) per = (N + 1) & ~1;
    p.images_per_slice = per; p.slices = (N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)N * C * H * W * 2);
    return (size_t)N * C * H * W * 2 < 0xffffffffull;
}

bool dwconv_mfma_small_tri_wgrad_supported(int N, int C, int H, int W, int K, int dtype) {
    if (dtype != SLAK_BF16 && dtype != SLAK_F16) return false;
    SmallTriWgradParams p;
    return fill_stw_params(p, N, C, H, W, K, 512);
}

size_t dwconv_mfma_small_tri_wgrad_workspace(int N, int C, int K) {
    return align_up((size_t)((N + 7) / 8 + 1) * C * (2 * K * MF_TAPS + 25) * sizeof(float), 256);   // slices <= ceil(N / 8)
}

template <typename T>
static int launch_qw_t(SmallTriWgradParams& p, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_small_quad_wgrad_kernel<T>;
    static const int wgs_per_cu = [] { const char* e = slak_dev_getenv("SLAK_QW_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 3; }();
    fill_stw_params(p, p.N, p.C, p.H, p.W, p.K, wgs_per_cu * mfma_cu_count());
    int per = (p.images_per_slice + 7) & ~7;                        // whole octets
    if (per > ((p.N + 7) & ~7)) per = (p.N + 7) & ~7;
    p.images_per_slice = per; p.slices = (p.N + per - 1) / per;
    if ((size_t)(p.N + 7) * p.C * p.H * p.W * 2 >= 0xffffffffull) return SLAK_ERR_UNSUPPORTED;
    if ((size_t)p.slices * p.C * (2 * p.K * MF_TAPS + 25) * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    const size_t lds = (size_t)MF_WAVES * QW_WAVE_BYTES;
    (void)slak_set_max_lds((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(((p.C + 3) / 4) * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, bool NARROW, bool DG = false>
static int launch_stw_t(SmallTriWgradParams& p, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_small_tri_wgrad_kernel<T, NARROW, DG>;
    fill_stw_params(p, p.N, p.C, p.H, p.W, p.K, 2 * mfma_cu_count());
    if ((size_t)p.slices * p.C * (2 * p.K * MF_TAPS + 25) * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    const size_t lds = (size_t)MF_WAVES * (DG ? TW_WAVE_BYTES_DG : TW_WAVE_BYTES);
    (void)slak_set_max_lds((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(((p.C + 3) / 4) * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_small_tri_wgrad(const void* const* dy, const void* x, float* const* dw, int dtype,
                                       int N, int C, int H, int W, int K, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_small_tri_wgrad_supported(N, C, H, W, K, dtype)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    SmallTriWgradParams p;
    fill_stw_params(p, N, C, H, W, K, 512);
    for (int b = 0; b < 3; ++b) { p.dy[b] = dy[b]; p.dw[b] = dw[b]; }
    p.x = x; p.partial = (float*)ws; p.dx = nullptr; p.w[0] = p.w[1] = p.w[2] = nullptr;
#ifdef SLAK_QW_DEV
    { static const int dbg = [] { const char* e = slak_dev_getenv("SLAK_QW_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }
#else
    p.dbg = 0;
#endif
    p.counters = wgrad_arrival_counters((C + 3) / 4);
    if (!p.counters) return SLAK_ERR_UNSUPPORTED;                 // (the caller runs the three per-branch launches)
    if (H <= 7 && W <= 7 && quad_wgrad_enabled())                 // planes of at most 7 x 7: eight per MFMA
        return dtype == SLAK_BF16 ? launch_qw_t<bf16_t>(p, ws_bytes, st) : launch_qw_t<f16_t>(p, ws_bytes, st);
    if (dtype == SLAK_BF16) return W < 8 ? launch_stw_t<bf16_t, true>(p, ws_bytes, st) : launch_stw_t<bf16_t, false>(p, ws_bytes, st);
    return W < 8 ? launch_stw_t<f16_t, true>(p, ws_bytes, st) : launch_stw_t<f16_t, false>(p, ws_bytes, st);
}

// The whole backward of a block's three depthwise convs in one launch (14 x 14 class): dx and the three weight gradients.
bool dwconv_mfma_small_tri_bwd_supported(int N, int C, int H, int W, int K, int dtype) {
    static const bool on = [] { const char* e = getenv("SLAK_SMALL_TRI_BWD"); return !(e && e[0] == '0'); }();
    return on && dwconv_mfma_small_tri_wgrad_supported(N, C, H, W, K, dtype) && W >= 8 && !(H <= 7 && W <= 7);
}

int launch_dwconv_mfma_small_tri_bwd(const void* const* dy, const void* x, const float* const* w, void* dx, float* const* dw, int dtype,
                                     int N, int C, int H, int W, int K, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_small_tri_bwd_supported(N, C, H, W, K, dtype)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    SmallTriWgradParams p;
    fill_stw_params(p, N, C, H, W, K, 512);
    for (int b = 0; b < 3; ++b) { p.dy[b] = dy[b]; p.dw[b] = dw[b]; p.w[b] = w[b]; }
    p.x = x; p.dx = dx; p.partial = (float*)ws; p.dbg = 0;
    p.counters = wgrad_arrival_counters((C + 3) / 4);
    if (!p.counters) return SLAK_ERR_UNSUPPORTED;
    return dtype == SLAK_BF16 ? launch_stw_t<bf16_t, false, true>(p, ws_bytes, st) : launch_stw_t<f16_t, false, true>(p, ws_bytes, st);
}

}  // namespace slak
