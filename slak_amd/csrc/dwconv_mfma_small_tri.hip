// slak_amd/csrc/dwconv_mfma_small_tri.hip -- the THREE branches of a decomposed large-kernel block (Kx5, 5xK, 5x5:
// ReparamLargeKernelConv, models/SLaK.py:60-100) in ONE launch on the 14x14 class, with the wave-independent LDS-DMA streaming
// of dwconv_mfma_small_dma.hip (same plane layout, fragments, counted vmcnt):
//   forward        x                 -> y_v, y_h, y_s     x is fetched ONCE per plane pair instead of three times
//   data gradient  dy_v, dy_h, dy_s  -> dx                the three contributions meet in ONE accumulator (both operand orders
//                                                         leave the same lane/register map), so the two elementwise adds
//                                                         autograd would run on the three partial gradients disappear
// At this size a launch is ~10 us of fixed cost for ~8 us of HBM time, so one launch for three is also what removes most of it.
// NARROW (W < 8, odd widths included: the 7x7 planes of SLaK's last stage): a row is one 16-byte piece at a 2-byte aligned
// address (the LDS-DMA takes it: tools/dma_probe.hip), so only the "half 0" lanes fetch; the piece drags in 8 - W elements of the
// next row (or plane), which are cleared in the fragment registers (two v_and per fragment: nothing foreign, NaN or not, ever
// reaches an MFMA) and never transposed into x^T; results leave as 2-byte stores (rows of an odd width are not dword aligned).
#include "mfma_common.h"
#include <type_traits>
#include <stdlib.h>

namespace slak {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int ST_NS = 4;                // ring slots (plane pairs) per wave
constexpr int ST_WZP = 16;              // zeros in front of a filter row
constexpr int ST_WLEN = 96;             // elements per padded filter row (16 + 63 + 17)
constexpr int ST_WCH = 5;               // filter elements staged per lane (upper bound)
constexpr int ST_WINB = 2 * MF_TAPS * ST_WLEN * 2;      // bytes of one branch's filter windows (two copies)

struct SmallTriParams {
    const void* in[3];                   // forward: in[0] = x; data gradient: dy_v, dy_h, dy_s
    void* out[3];                        // forward: y_v, y_h, y_s; data gradient: out[0] = dx
    const float* w[3];                   // filters of the vertical (K,5), horizontal (5,K) and small (5,5) branch
    int N, C, H, W, K, flip;
    int images_per_slice, slices;
    unsigned tensor_bytes;
    float* stats;                        // forward only, or NULL: [slices][C][6] = per (slice, channel) sum y_b, sum y_b^2 of the ROUNDED outputs
};                                       // (the batch statistics of the three branch BatchNorms, models/SLaK.py:92-95: saves their read pass)

template <typename T> __device__ __forceinline__ f32x4_t st_mfma16(s16x8 a, s16x8 b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t st_mfma16<bf16_t>(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t st_mfma16<f16_t>(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// DGRAD: three inputs, one output, filters rotated by 180 degrees; else one input, three outputs
template <typename T, bool DGRAD, bool NARROW>
__global__ __launch_bounds__(MF_THREADS) void dwconv_mfma_small_tri_kernel(const SmallTriParams p) {
    constexpr int NT = DGRAD ? 3 : 1;                             // input tensors
    constexpr int SLOT = NT * 1024;                               // bytes per ring slot: NT x [2 planes x 16 rows x 32 B]
    // per-wave LDS region (bytes): [64 zero pad][ring][64 zero pad][x^T: 2 planes][64 zero row][3 x filter windows]
    constexpr int RING = 64, XT = RING + ST_NS * SLOT + 64, ZROW = XT + 1024, WIN = ZROW + 64, WAVE_BYTES = WIN + 3 * ST_WINB;
    constexpr int NSTORE = (DGRAD ? 4 : 12) * (NARROW ? 2 : 1);    // store instructions per complete pair
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + 3) >> 2;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c = cb * 4 + wave;
    const int n_begin = slice * p.images_per_slice;
    int n_end = n_begin + p.images_per_slice; if (n_end > p.N) n_end = p.N;
    if (c >= p.C || n_begin >= n_end) return;                     // no workgroup barrier anywhere: waves may leave
    const int npairs = (n_end - n_begin + 1) >> 1;
    char* const L = (char*)lds + wave * WAVE_BYTES;
    const int HW = p.H * p.W;

    // ---- filters (three branches), zero fill ------------------------------------------------------------------
    const int ntap = p.K * MF_TAPS;                               // vertical and horizontal branch; the small one has 25
    float wv[ST_WCH], wh[ST_WCH], wsm = 0.f;
#pragma unroll
    for (int k = 0; k < ST_WCH; ++k) {
        const int e = lane + 64 * k;
        wv[k] = e < ntap ? p.w[0][(size_t)c * ntap + e] : 0.f;
        wh[k] = e < ntap ? p.w[1][(size_t)c * ntap + e] : 0.f;
    }
    if (lane < 25) wsm = p.w[2][(size_t)c * 25 + lane];
    for (int o = lane * 16; o < WAVE_BYTES; o += 64 * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // zeros are in place before any DMA can land on them

    // ---- DMA: lane -> (plane of the pair, image row, half of the row) ------------------------------------------
    v4i_t rs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint64_t a = (uint64_t)p.in[t];
        rs[t][0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs[t][1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs[t][2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs[t][3] = 0x00020000;
    }
    __amdgpu_buffer_rsrc_t ro[3];
#pragma unroll
    for (int t = 0; t < (DGRAD ? 1 : 3); ++t) ro[t] = __builtin_amdgcn_make_buffer_rsrc(p.out[t], 0, (int)p.tensor_bytes, 0x00020000);
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;
    const int d_pp = lane >> 5, d_row = (lane >> 1) & 15, d_half = lane & 1;
    const unsigned d_src = (unsigned)d_pp * gplane_b + (unsigned)(d_row * p.W) * 2 + (d_half ? (unsigned)(p.W - 8) * 2 : 0u);
    const bool d_rowok = d_row < p.H && (!NARROW || d_half == 0);
    const unsigned lds_wave = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds) + wave * WAVE_BYTES;
    const unsigned chan_b = (unsigned)c * (unsigned)HW * 2;
    const unsigned last_row_b = p.tensor_bytes - (unsigned)(2 * p.W);     // byte offset of the tensor's last image row
    auto issue_pair = [&](int q) {
        const int n0 = n_begin + 2 * q;
        const unsigned gb = (unsigned)n0 * gplane_b + chan_b;
        const unsigned dst = lds_wave + RING + (unsigned)(q % ST_NS) * SLOT;
        if (d_rowok && n0 + d_pp < n_end) {
            unsigned so = gb + d_src;
            // NARROW: the piece of the tensor's very last row would end 16 - 2W bytes behind the tensor (the buffer range check
            // zeroes it, and nothing promises that memory exists): it is fetched that much earlier and shifted into place below
            if (NARROW && so == last_row_b) so -= (unsigned)(16 - 2 * p.W);
#pragma unroll
            for (int t = 0; t < NT; ++t) lds_dma16(so, rs[t], __builtin_amdgcn_readfirstlane(dst + t * 1024));
        }
    };
    for (int q = 0; q < ST_NS - 1 && q < npairs; ++q) issue_pair(q);

    // ---- filter windows: branch b at WIN + b*ST_WINB, two copies one element apart ---------------------------------------
    auto put = [&](int b, int r, int t, int KL, float v) {        // short tap r, long tap t of branch b
        if (p.flip) { r = MF_TAPS - 1 - r; t = KL - 1 - t; }
        const uint16_t h = cvt_to_bits(v, (T*)nullptr);
        uint16_t* win = (uint16_t*)(L + WIN + b * ST_WINB);
        win[r * ST_WLEN + ST_WZP + t] = h;
        win[MF_TAPS * ST_WLEN + r * ST_WLEN + ST_WZP + t - 1] = h;
    };
#pragma unroll
    for (int k = 0; k < ST_WCH; ++k) {
        const int e = lane + 64 * k;
        if (e < ntap) {
            put(0, e % MF_TAPS, e / MF_TAPS, p.K, wv[k]);        // (K,5): element [t][r]
            put(1, e / p.K, e - (e / p.K) * p.K, p.K, wh[k]);    // (5,K): element [r][t]
        }
    }
    if (lane < 25) put(2, lane / 5, lane % 5, 5, wsm);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // Toeplitz fragments: lane -> (o = long-axis output position, k-group kg -> tap-in-pair rsel, half of the 16 k-slots)
    const int l15 = lane & 15, kg = lane >> 4, rsel = kg >> 1, half = kg & 1;
    s16x8 tf[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const bool vert = b == 0;
        const int padL = (b == 2 ? 5 : p.K) / 2;
        const int i0 = half ? (vert ? 8 : p.W - 8) : 0;
        const int a = ST_WZP + i0 - l15 + padL;
        const int par = a & 1;
        const unsigned* src = (const unsigned*)(L + WIN + b * ST_WINB + par * MF_TAPS * ST_WLEN * 2) + ((a - par) >> 1);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int r = 2 * m + rsel;
            u32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = r < MF_TAPS ? src[(r < MF_TAPS ? r : 0) * (ST_WLEN / 2) + k] : 0u;
                if (!vert && half && 2 * k < 16 - p.W) d[k] = 0u;     // columns already covered by the first half
                if (NARROW && !vert && !half) {                       // k-slots beyond the row (W < 8): no such input
                    if (2 * k >= p.W) d[k] = 0u; else if (2 * k + 1 >= p.W) d[k] &= 0xffffu;
                }
            }
            tf[b][m] = __builtin_bit_cast(s16x8, d);
        }
    }

    // ---- lane constants of the loop (see dwconv_mfma_small_dma.hip) --------------------------------------------------
    const unsigned xlane = (unsigned)(l15 * 32 + rsel * 32 + half * 16);
    const unsigned zlane = (unsigned)ZROW + half * 16;
    const int g4 = lane >> 4;
    const unsigned trd = (unsigned)((4 * g4 + (l15 >> 2)) * 32 + (l15 & 3) * 8);
    const int xt_row = l15 < 8 ? l15 : l15 - (16 - p.W);
    const bool twr_ok = NARROW ? l15 < p.W : (l15 < 8 || xt_row >= 8);
    const unsigned twr = (unsigned)(XT + xt_row * 32 + g4 * 8);
    const unsigned ooff = (unsigned)(l15 * p.W + 4 * kg) * 2;
    const bool st0 = l15 < p.H && 4 * kg < p.W, st1 = l15 < p.H && 4 * kg + 2 < p.W;
    // NARROW: what a row's 16-byte piece holds beyond column W-1 belongs to the next row / plane: cleared in the register
    const unsigned bm2 = 5 < p.W ? 0xffffffffu : (4 < p.W ? 0xffffu : 0u), bm3 = 7 < p.W ? 0xffffffffu : (6 < p.W ? 0xffffu : 0u);
    auto frag = [&](unsigned base, int m, bool rowmajor) -> s16x8 {   // MFMA m of a plane whose guarded image starts 64 bytes after `base`
        const unsigned a = (m == 2 && rsel) ? zlane : base + xlane + m * 64;
        u32x4 v = *(const u32x4*)(L + a);
        if (NARROW && rowmajor) { v[2] &= bm2; v[3] &= bm3; }
        return __builtin_bit_cast(s16x8, v);
    };
    auto store4 = [&](const f32x4_t& v, const __amdgpu_buffer_rsrc_t& r, unsigned go) {      // 4 consecutive columns of output row l15
        if constexpr (NARROW) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (l15 < p.H && 4 * kg + j < p.W) __builtin_amdgcn_raw_buffer_store_b16((short)(pack2<T>(v[j], 0.f) & 0xffffu), r, ooff + 2 * j, go, 0);
        } else {
            if (st0) __builtin_amdgcn_raw_buffer_store_b32(pack2<T>(v[0], v[1]), r, ooff, go, 0);
            if (st1) __builtin_amdgcn_raw_buffer_store_b32(pack2<T>(v[2], v[3]), r, ooff + 4, go, 0);
        }
    };

    float bsum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto stat4 = [&](const f32x4_t& v, int b) {                   // what store4 stores, as the BatchNorm statistics see it (14 x 14 class: W even)
        if constexpr (std::is_same<T, bf16_t>::value) {
            const bf16x2_t one = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            const bf16x2_t p01 = __builtin_bit_cast(bf16x2_t, st0 ? pack2<T>(v[0], v[1]) : 0u), p23 = __builtin_bit_cast(bf16x2_t, st1 ? pack2<T>(v[2], v[3]) : 0u);
            // v_dot2c_f32_bf16 with a ZERO addend (its addend is aligned with truncation), the adds kept apart from it by the empty asm
            float t0 = __builtin_amdgcn_fdot2_f32_bf16(p01, one, 0.f, false), t1 = __builtin_amdgcn_fdot2_f32_bf16(p23, one, 0.f, false);
            float t2 = __builtin_amdgcn_fdot2_f32_bf16(p01, p01, 0.f, false), t3 = __builtin_amdgcn_fdot2_f32_bf16(p23, p23, 0.f, false);
            asm("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
            bsum[2 * b] += t0 + t1;
            bsum[2 * b + 1] += t2 + t3;
        }
    };
    for (int q = 0; q < npairs; ++q) {
        {
            const int st = (q < ST_NS - 2 ? q : ST_NS - 2) * NSTORE;
            int dm = npairs - 1 - q; if (dm > ST_NS - 2) dm = ST_NS - 2;
            wait_vmcnt_dyn(st + dm * NT);
        }
        __builtin_amdgcn_wave_barrier();
        const int n0 = n_begin + 2 * q;
        const unsigned slot = (unsigned)RING + (unsigned)(q % ST_NS) * SLOT;
        if (NARROW && c == p.C - 1 && n0 + 1 >= p.N - 1 && n0 <= p.N - 1) {       // (wave-uniform) this pair holds the tensor's last plane
            const int ppl = p.N - 1 - n0, sh = 8 - p.W;             // its last row arrived `sh` elements late: shift it into place
            if (lane < NT) {
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
        // vertical branch: its input plane pair (tensor 0) transposed into x^T
        {
            const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + slot + trd));
            const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + slot + 512 + trd));
            if (twr_ok) { *(s16x4*)(L + twr) = t0; *(s16x4*)(L + twr + 512) = t1; }
        }
        const unsigned bv = (unsigned)XT - 64;                                    // x^T images
        const unsigned bh = slot + (DGRAD ? 1024u : 0u) - 64, bs = slot + (DGRAD ? 2048u : 0u) - 64;   // row-major images
        const unsigned gb = (unsigned)n0 * gplane_b + chan_b;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            if (pp == 1 && n0 + 1 >= n_end) break;                  // (wave-uniform) odd slice: the second plane does not exist
            f32x4_t av = {0.f, 0.f, 0.f, 0.f}, ah = av, as = av;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                av = st_mfma16<T>(frag(bv + pp * 512, m, false), tf[0][m], av);       // operands swapped: D^T = X^T-tile x T^T
                ah = st_mfma16<T>(tf[1][m], frag(bh + pp * 512, m, true), ah);
                as = st_mfma16<T>(tf[2][m], frag(bs + pp * 512, m, true), as);
            }
            const unsigned go = gb + pp * gplane_b;
            if constexpr (DGRAD) {
                const f32x4_t s = (av + ah) + as;                   // the three partial gradients, added in fp32
                store4(s, ro[0], go);
            } else {
                store4(av, ro[0], go); store4(ah, ro[1], go); store4(as, ro[2], go);
                if constexpr (!NARROW && std::is_same<T, bf16_t>::value) { if (p.stats) { stat4(av, 0); stat4(ah, 1); stat4(as, 2); } }
            }
        }
        if (q + ST_NS - 1 < npairs) issue_pair(q + ST_NS - 1);
    }
    if constexpr (!DGRAD && !NARROW) {
        if (p.stats) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float v = bsum[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) p.stats[((size_t)slice * p.C + c) * 6 + k] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// QUAD: planes of at most 7 x 7 (SLaK's last stage) -- FOUR planes (images of one channel) per MFMA tile, eight per loop step.
// The kernel above spends one 16 x 16 x 32 tile on a 7 x 7 plane (8 % of it), and its memory instructions are the texture unit's
// worst case: 16-byte DMA pieces at 2-byte aligned addresses (~50 cycles per instruction) and two-byte stores (~20 cycles each,
// 24 per plane pair): 0.2 of the HBM roofline.  Here a 16-row x 32-byte LDS tile holds four planes:
//     rows 0..6  bytes 0..15 : plane 0     bytes 16..31 : plane 2          (a row = one 16-byte piece, elements >= W cleared)
//     rows 7, 8  zero guard
//     rows 9..15 bytes 0..15 : plane 1     bytes 16..31 : plane 3
// followed by two zero rows.  The Toeplitz operands are block diagonal (a plane only meets its own taps), indices on BOTH
// tile axes carry the 9-row pitch (x^T is written with it), so all three branches leave D[M = (byte half, column)][N = (row block, row)]
// in one lane / register map: nine MFMAs per four planes.
//   in:  every lane loads the four dword-ALIGNED dwords that cover one image row (one buffer_load_dwordx4 per tensor for the 56
//        rows of eight planes), v_alignbit moves the row to bit 0, the next row's elements behind column W-1 are cleared, one
//        ds_write_b128 puts the piece in place; two octets are in flight per wave (two register sets);
//   out: the results are written into LDS in the planes' own (contiguous) layout, shifted so that the dword-aligned 16-byte
//        chunks of a plane in HBM are 16-byte aligned in LDS: one buffer_store_dwordx4 (lane -> plane, chunk) and one two-byte
//        store (the element in front of / behind the chunks) per output tensor and eight planes, instead of 32 two-byte stores.
// Lanes with nothing to load or store carry an out-of-range offset (buffer range check: zeros in, dropped out): no exec masking.
// Zero operands do not stop NaN / Inf: a non-finite value in one plane reaches the other planes of its tile (0 x Inf = NaN).
// The outputs of such a step are non-finite in the reference as well (the loss sees every plane), only not in the same places.
constexpr int SQ_TILE = 512 + 64;       // bytes: 16 rows x 32 + two zero guard rows
constexpr unsigned SQ_OOB = 0x80000000u;
template <bool DGRAD> struct SqLds {
    static constexpr int NT = DGRAD ? 3 : 1, NO = DGRAD ? 1 : 3;
    // per-wave bytes: [64 zero][NT x 2 tiles, each with its guard][x^T tile][64 zero row][NO x 8 planes x 128: results, plane layout]
    static constexpr int IN = 64, XT = IN + NT * 2 * SQ_TILE, ZROW = XT + 512, OUT = ZROW + 64, END = OUT + NO * 1024;
    static constexpr int WAVE_BYTES = END > 3 * ST_WINB ? END : 3 * ST_WINB;     // the filter windows (set-up only) alias all of it
};

// DW: some plane has whole dwords left behind its 16-byte chunks (not 7 x 7: 98 = 6 x 16 + 2 bytes)
template <typename T, bool DGRAD, bool DW>
__global__ __launch_bounds__(MF_THREADS) void dwconv_mfma_small_quad_kernel(const SmallTriParams p) {
    using LY = SqLds<DGRAD>;
    constexpr int NT = LY::NT, NO = LY::NO, IN = LY::IN, XT = LY::XT, ZROW = LY::ZROW, OUT = LY::OUT, WAVE_BYTES = LY::WAVE_BYTES;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + 3) >> 2;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c = cb * 4 + wave;
    const int n_begin = slice * p.images_per_slice;               // (a multiple of 8)
    int n_end = n_begin + p.images_per_slice; if (n_end > p.N) n_end = p.N;
    if (c >= p.C || n_begin >= n_end) return;                     // no workgroup barrier anywhere: waves may leave
    const int noct = (n_end - n_begin + 7) >> 3;
    char* const L = (char*)lds + wave * WAVE_BYTES;
    const int HW = p.H * p.W;
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;
    const unsigned chan_b = (unsigned)c * (unsigned)HW * 2;

    // ---- loads: lane -> (tile of the octet, tile row, byte half) = (plane, image row); the first two octets leave right away ----
    __amdgpu_buffer_rsrc_t ri[NT], ro[NO];
    // (load range: up to the end of the dword that holds the tensor's last element -- the check is per dword, and an aligned dword
    // that holds a valid element lies inside the allocation)
#pragma unroll
    for (int t = 0; t < NT; ++t) ri[t] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in[t]), 0, (int)((p.tensor_bytes + 3u) & ~3u), 0x00020000);
#pragma unroll
    for (int t = 0; t < NO; ++t) ro[t] = __builtin_amdgcn_make_buffer_rsrc(p.out[t], 0, (int)p.tensor_bytes, 0x00020000);
    const int d_r = (lane >> 1) & 15, d_half = lane & 1, d_ab = d_r >= 9 ? 1 : 0, d_row = d_r - 9 * d_ab;
    const int d_pl = 4 * (lane >> 5) + d_ab + 2 * d_half;         // plane of the octet
    const bool d_ok = d_r != 7 && d_r != 8 && d_row < p.H;
    const unsigned d_row_b = chan_b + (unsigned)d_pl * gplane_b + (unsigned)(d_row * p.W) * 2;
    const unsigned d_sh = (d_row_b & 2u) * 8u;                    // the row starts in the upper half of its first dword (8 | n_begin: the same for every octet)
    const unsigned d_dst = (unsigned)IN + (unsigned)((lane >> 5) * SQ_TILE + (lane & 31) * 16);
    const unsigned bm2 = 5 < p.W ? 0xffffffffu : (4 < p.W ? 0xffffu : 0u), bm3 = 7 < p.W ? 0xffffffffu : (6 < p.W ? 0xffffu : 0u);
    auto load_oct = [&](int q, u32x4 (&R)[NT]) {                  // (an octet behind the slice loads nothing: one instruction count on every path)
        const int n0 = n_begin + 8 * q;
        const unsigned a = (d_ok && n0 + d_pl < n_end) ? (((unsigned)n0 * gplane_b + d_row_b) & ~3u) : SQ_OOB;
#pragma unroll
        for (int t = 0; t < NT; ++t) R[t] = __builtin_amdgcn_raw_buffer_load_b128(ri[t], a, 0, 0);
    };
    auto stage = [&](const u32x4 (&R)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            u32x4 v;
            v[0] = __builtin_amdgcn_alignbit(R[t][1], R[t][0], d_sh);
            v[1] = __builtin_amdgcn_alignbit(R[t][2], R[t][1], d_sh);
            v[2] = __builtin_amdgcn_alignbit(R[t][3], R[t][2], d_sh) & bm2;
            v[3] = __builtin_amdgcn_alignbit(0u, R[t][3], d_sh) & bm3;
            *(u32x4*)(L + d_dst + t * 2 * SQ_TILE) = v;
        }
    };
    u32x4 R0[NT], R1[NT];
    load_oct(0, R0);
    load_oct(1, R1);

    // ---- filters (three branches) -> windows: branch b at b*ST_WINB, two copies one element apart -----------------------------
    const int ntap = p.K * MF_TAPS;
    float wv[ST_WCH], wh[ST_WCH], wsm = 0.f;
#pragma unroll
    for (int k = 0; k < ST_WCH; ++k) {
        const int e = lane + 64 * k;
        wv[k] = e < ntap ? p.w[0][(size_t)c * ntap + e] : 0.f;
        wh[k] = e < ntap ? p.w[1][(size_t)c * ntap + e] : 0.f;
    }
    if (lane < 25) wsm = p.w[2][(size_t)c * 25 + lane];
    for (int o = lane * 16; o < WAVE_BYTES; o += 64 * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    auto put = [&](int b, int r, int t, int KL, float v) {        // short tap r, long tap t of branch b
        if (p.flip) { r = MF_TAPS - 1 - r; t = KL - 1 - t; }
        const uint16_t h = cvt_to_bits(v, (T*)nullptr);
        uint16_t* win = (uint16_t*)(L + b * ST_WINB);
        win[r * ST_WLEN + ST_WZP + t] = h;
        win[MF_TAPS * ST_WLEN + r * ST_WLEN + ST_WZP + t - 1] = h;
    };
#pragma unroll
    for (int k = 0; k < ST_WCH; ++k) {
        const int e = lane + 64 * k;
        if (e < ntap) {
            put(0, e % MF_TAPS, e / MF_TAPS, p.K, wv[k]);        // (K,5): element [t][r]
            put(1, e / p.K, e - (e / p.K) * p.K, p.K, wh[k]);    // (5,K): element [r][t]
        }
    }
    if (lane < 25) put(2, lane / 5, lane % 5, 5, wsm);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // block-diagonal Toeplitz fragments: lane -> (o = l15: 9-pitch index of the output position, kg -> tap-in-pair rsel, byte half = 8 k-slots)
    const int l15 = lane & 15, kg = lane >> 4, rsel = kg >> 1, half = kg & 1;
    const int oblk = l15 >= 9 ? 1 : 0;                            // block of the lane's 9-pitch index
    s16x8 tf[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const bool vert = b == 0;
        const int padL = (b == 2 ? 5 : p.K) / 2;
        // vertical: k-slot s = 8*half + e is the tile ROW (9-pitch, like o): tap s - o; others: k-slot e is column e of the plane pair `half`
        const int a = ST_WZP + (vert ? 8 * half - l15 : -(l15 - 9 * oblk)) + padL;
        const int par = a & 1;
        const unsigned* src = (const unsigned*)(L + b * ST_WINB + par * MF_TAPS * ST_WLEN * 2) + ((a - par) >> 1);
        const bool keep = half == oblk;                           // a plane meets only its own taps
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int r = 2 * m + rsel;
            u32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = (keep && r < MF_TAPS) ? src[(r < MF_TAPS ? r : 0) * (ST_WLEN / 2) + k] : 0u;
                if (!vert) {                                      // k-slots beyond the row (W < 8): no such input
                    if (2 * k >= p.W) d[k] = 0u; else if (2 * k + 1 >= p.W) d[k] &= 0xffffu;
                }
            }
            tf[b][m] = __builtin_bit_cast(s16x8, d);
        }
    }
    asm volatile("" : "+v"(tf[0][0]), "+v"(tf[0][1]), "+v"(tf[0][2]), "+v"(tf[1][0]), "+v"(tf[1][1]), "+v"(tf[1][2]), "+v"(tf[2][0]), "+v"(tf[2][1]), "+v"(tf[2][2]));
    __builtin_amdgcn_wave_barrier();                              // the fragments are in registers: the windows give way to the tiles
    for (int o = lane * 16; o < WAVE_BYTES; o += 64 * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};

    // ---- lane constants of the loop ---------------------------------------------------------------------------------
    const unsigned xlane = (unsigned)(l15 * 32 + rsel * 32 + half * 16);
    const unsigned zlane = (unsigned)ZROW + half * 16;
    const int g4 = lane >> 4;
    const unsigned trd = (unsigned)((4 * g4 + (l15 >> 2)) * 32 + (l15 & 3) * 8);
    const bool twr_ok = (l15 & 7) < p.W;
    const unsigned twr = (unsigned)(XT + ((l15 >> 3) * 9 + (l15 & 7)) * 32 + g4 * 8);
    auto frag = [&](unsigned base, int m) -> s16x8 {              // MFMA m of a tile that starts 64 bytes after `base`
        const unsigned a = (m == 2 && rsel) ? zlane : base + xlane + m * 64;
        return __builtin_bit_cast(s16x8, *(const u32x4*)(L + a));
    };
    // result element j of the lane: N = l15 = (row block, row), M = 4*kg + j = (byte half, column), both with the 9 pitch.
    // It goes to LDS offset plane*128 + 16 + 2*(row*W + col) - a, a = the plane's HBM start modulo 4 (0 or 2)
    const int o_row = l15 - 9 * oblk, o_pl = oblk + 2 * (kg >> 1);                  // (valid elements of kg 0,1 / 2,3 lie in byte half 0 / 1)
    const bool o_nok = l15 != 7 && l15 != 8 && o_row < p.H;
    const unsigned o_a = ((unsigned)o_pl * gplane_b + chan_b) & 2u;                 // (4 * gplane_b = 0 mod 8: the same for both tiles)
    unsigned wo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = 4 * kg + j, hf = m >= 9 ? 1 : 0, col = m - 9 * hf;
        const bool ok = o_nok && m != 7 && m != 8 && col < p.W;
        wo[j] = (unsigned)OUT + (unsigned)o_pl * 128u + (ok ? 16u + (unsigned)(o_row * p.W + col) * 2u - o_a : 116u + 2u * j);   // (116..: unused bytes of the plane's slot)
    }
    // stores: lane -> (plane of the octet, chunk slot)
    const int s_pl = lane >> 3, s_k = lane & 7;
    const unsigned s_g = (unsigned)s_pl * gplane_b + chan_b;                        // the plane's first byte (without the octet's base)
    const unsigned s_a = s_g & 2u, s_len = (unsigned)HW * 2u - s_a, s_nch = s_len >> 4, s_rest = s_len & 15u;
    const unsigned sc_g = (unsigned)s_k < s_nch ? s_g + s_a + 16u * s_k : SQ_OOB;
    const unsigned sc_l = (unsigned)OUT + (unsigned)s_pl * 128u + 16u + 16u * s_k;
    const unsigned sd_g = (unsigned)s_k < (s_rest >> 2) ? s_g + s_a + 16u * s_nch + 4u * s_k : SQ_OOB;
    const unsigned sd_l = (unsigned)OUT + (unsigned)s_pl * 128u + 16u + 16u * s_nch + 4u * (s_k & 3);
    const bool s_head = s_k == 0 && s_a == 2u, s_tail = s_k == 1 && (s_rest & 2u);
    const unsigned ss_g = s_head ? s_g : (s_tail ? s_g + (unsigned)HW * 2u - 2u : SQ_OOB);
    const unsigned ss_l = (unsigned)OUT + (unsigned)s_pl * 128u + (s_head ? 14u : 16u + (unsigned)HW * 2u - 2u - s_a);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    float bsum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};              // forward with p.stats: sum y_b, sum y_b^2 of what this wave stores (bf16)
    auto octet = [&](int q, u32x4 (&R)[NT]) {
        stage(R);                                                 // (the LDS queue is in order: the reads of the octet before are behind us)
        load_oct(q + 2, R);                                       // (every path issues the same memory instructions: the compiler's vmcnt
        const int n0 = n_begin + 8 * q;                           //  bookkeeping stays exact; an octet behind the slice stores nothing)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if (n0 + 4 * tt >= n_end) break;                        // (wave-uniform) the slice ends before this tile
            const unsigned tile = (unsigned)IN + tt * SQ_TILE;
            {                                                       // vertical branch: its input tile (tensor 0) transposed into x^T, rows at the 9 pitch
                const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + tile + trd));
                if (twr_ok) *(s16x4*)(L + twr) = t0;
            }
            const unsigned bv = (unsigned)XT - 64;
            const unsigned bh = tile + (DGRAD ? 2u * SQ_TILE : 0u) - 64, bs = tile + (DGRAD ? 4u * SQ_TILE : 0u) - 64;
            f32x4_t av = {0.f, 0.f, 0.f, 0.f}, ah = av, as = av;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                av = st_mfma16<T>(frag(bv, m), tf[0][m], av);                     // operands swapped: D^T = X^T-tile x T^T
                ah = st_mfma16<T>(tf[1][m], frag(bh, m), ah);
                as = st_mfma16<T>(tf[2][m], frag(bs, m), as);
            }
            auto put4 = [&](const f32x4_t& v, int t) {              // into the planes' own layout (elements that do not exist: spare bytes of the slot)
                const unsigned p01 = pack2<T>(v[0], v[1]), p23 = pack2<T>(v[2], v[3]);
                char* base = L + t * 1024 + tt * 512;
                *(uint16_t*)(base + wo[0]) = (uint16_t)(p01 & 0xffffu);
                *(uint16_t*)(base + wo[1]) = (uint16_t)(p01 >> 16);
                *(uint16_t*)(base + wo[2]) = (uint16_t)(p23 & 0xffffu);
                *(uint16_t*)(base + wo[3]) = (uint16_t)(p23 >> 16);
            };
            if constexpr (DGRAD) {
                const f32x4_t s = (av + ah) + as;                   // the three partial gradients, added in fp32
                put4(s, 0);
            } else {
                put4(av, 0); put4(ah, 1); put4(as, 2);
            }
        }
        const unsigned go = (unsigned)n0 * gplane_b;
        unsigned gc = sc_g, gd = sd_g, gs = ss_g;
        if (n0 + s_pl >= n_end) { gc = SQ_OOB; gd = SQ_OOB; gs = SQ_OOB; }                      // planes behind the slice
#pragma unroll
        for (int t = 0; t < NO; ++t) {
            const u32x4 v = *(const u32x4*)(L + t * 1024 + sc_l);
            __builtin_amdgcn_raw_buffer_store_b128(v, ro[t], gc, go, 0);
            unsigned dwv = 0u;
            if constexpr (DW) { dwv = *(const unsigned*)(L + t * 1024 + sd_l); __builtin_amdgcn_raw_buffer_store_b32(dwv, ro[t], gd, go, 0); }
            const uint16_t hs = *(const uint16_t*)(L + t * 1024 + ss_l);
            __builtin_amdgcn_raw_buffer_store_b16((short)hs, ro[t], gs, go, 0);
            if constexpr (!DGRAD && std::is_same<T, bf16_t>::value) {
                if (p.stats) {                                      // (wave-uniform) the BatchNorm statistics of exactly the stored values
                    const bf16x2_t one = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
                    const unsigned d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;        // (scalars first: see stat8 in dwconv_mfma_dma.hip)
                    const bf16x2_t x0 = __builtin_bit_cast(bf16x2_t, d0), x1 = __builtin_bit_cast(bf16x2_t, d1), x2 = __builtin_bit_cast(bf16x2_t, d2), x3 = __builtin_bit_cast(bf16x2_t, d3);
                    float a0 = __builtin_amdgcn_fdot2_f32_bf16(x0, one, 0.f, false), a1 = __builtin_amdgcn_fdot2_f32_bf16(x1, one, 0.f, false);
                    float a2 = __builtin_amdgcn_fdot2_f32_bf16(x2, one, 0.f, false), a3 = __builtin_amdgcn_fdot2_f32_bf16(x3, one, 0.f, false);
                    float q0 = __builtin_amdgcn_fdot2_f32_bf16(x0, x0, 0.f, false), q1 = __builtin_amdgcn_fdot2_f32_bf16(x1, x1, 0.f, false);
                    float q2 = __builtin_amdgcn_fdot2_f32_bf16(x2, x2, 0.f, false), q3 = __builtin_amdgcn_fdot2_f32_bf16(x3, x3, 0.f, false);
                    asm("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
                    const float e = __uint_as_float((unsigned)hs << 16);
                    const bool cok = gc != SQ_OOB, sok = gs != SQ_OOB;
                    bsum[2 * t] += (cok ? (a0 + a1) + (a2 + a3) : 0.f) + (sok ? e : 0.f);
                    bsum[2 * t + 1] += (cok ? (q0 + q1) + (q2 + q3) : 0.f) + (sok ? e * e : 0.f);
                    if constexpr (DW) {                             // the whole dwords behind the plane's 16-byte chunks
                        const float lo = __uint_as_float(dwv << 16), hi = __uint_as_float(dwv & 0xffff0000u);
                        if (gd != SQ_OOB) { bsum[2 * t] += lo + hi; bsum[2 * t + 1] += lo * lo + hi * hi; }
                    }
                }
            }
        }
    };
    for (int q = 0; q < noct; q += 2) {
        octet(q, R0);
        octet(q + 1, R1);
    }
    if constexpr (!DGRAD) {
        if (p.stats) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float v = bsum[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) p.stats[((size_t)slice * p.C + c) * 6 + k] = v;
            }
        }
    }
}

static bool quad_enabled() {                   // SLAK_SMALL_QUAD=0 keeps the one-plane-per-tile kernel on 7 x 7 (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_SMALL_QUAD"); return !(e && e[0] == '0'); }();
    return v;
}
static int quad_target_wgs() {                 // workgroups the launch aims at (dev: SLAK_SQ_WGS per CU)
    static const int wgs_per_cu = [] { const char* e = slak_dev_getenv("SLAK_SQ_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4; }();
    return wgs_per_cu * mfma_cu_count();
}
static bool fill_quad_params(SmallTriParams& p, int N, int C, int H, int W, int K, int target_wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K;
    if (N <= 0 || C <= 0 || K < 5 || !(K & 1) || K > 63 || K * MF_TAPS > ST_WCH * 64) return false;
    if (H > 7 || H < 1 || W > 7 || W < 4) return false;
    const int cblocks = (C + 3) / 4;
    int slices = target_wgs / cblocks; if (slices < 1) slices = 1;
    int per = (N + slices - 1) / slices; per = (per + 7) & ~7;      // whole octets
    if (per > ((N + 7) & ~7)) per = (N + 7) & ~7;
    p.images_per_slice = per; p.slices = (N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)N * C * H * W * 2);
    return (size_t)(N + 7) * C * H * W * 2 < 0x80000000ull;              // offsets of an octet's missing planes stay below SQ_OOB
}
template <typename T, bool DGRAD>
static int launch_quad_t(SmallTriParams& p, hipStream_t st) {
    const unsigned pb = (unsigned)(p.H * p.W) * 2u;
    const bool dw = (pb & 15u) >= 4u || ((pb - 2u) & 15u) >= 4u;
    auto k = dw ? dwconv_mfma_small_quad_kernel<T, DGRAD, true> : dwconv_mfma_small_quad_kernel<T, DGRAD, false>;
    const size_t lds = (size_t)MF_WAVES * SqLds<DGRAD>::WAVE_BYTES;
    fill_quad_params(p, p.N, p.C, p.H, p.W, p.K, quad_target_wgs());
    (void)slak_set_max_lds((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(((p.C + 3) / 4) * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_tri_params(SmallTriParams& p, int N, int C, int H, int W, int K, int target_wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K;
    if (N <= 0 || C <= 0 || K < 5 || !(K & 1) || K > 63 || K * MF_TAPS > ST_WCH * 64) return false;
    if (H > 14 || H < 1) return false;                                 // both the row-major image and its transpose need guard rows
    if (W >= 8 ? (W > 14 || (W & 1)) : W < 4) return false;            // 8..14 even: two row halves; 4..7: one piece per row (NARROW)
    const int cblocks = (C + 3) / 4;
    int slices = target_wgs / cblocks; if (slices < 1) slices = 1;
    int per = (N + slices - 1) / slices; per = (per + 1) & ~1;
    if (per < 8) per = 8;
    if (per > ((N + 1) & ~1)) per = (N + 1) & ~1;
    p.images_per_slice = per; p.slices = (N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)N * C * H * W * 2);
    return (size_t)N * C * H * W * 2 < 0xffffffffull;
}

bool dwconv_mfma_small_tri_supported(int N, int C, int H, int W, int K, int dtype) {
    if (dtype != SLAK_BF16 && dtype != SLAK_F16) return false;
    SmallTriParams p;
    return fill_tri_params(p, N, C, H, W, K, 768);
}

template <typename T, bool DGRAD, bool NARROW>
static int launch_tri_tn(SmallTriParams& p, hipStream_t st) {
    constexpr int NT = DGRAD ? 3 : 1;
    constexpr size_t WAVE_BYTES = 64 + ST_NS * NT * 1024 + 64 + 1024 + 64 + 3 * ST_WINB;
    auto k = dwconv_mfma_small_tri_kernel<T, DGRAD, NARROW>;
    fill_tri_params(p, p.N, p.C, p.H, p.W, p.K, (DGRAD ? 2 : 3) * mfma_cu_count());   // resident workgroups per CU (LDS)
    const size_t lds = (size_t)MF_WAVES * WAVE_BYTES;
    (void)slak_set_max_lds((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(((p.C + 3) / 4) * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, bool DGRAD>
static int launch_tri_t(SmallTriParams& p, hipStream_t st) {
    return p.W < 8 ? launch_tri_tn<T, DGRAD, true>(p, st) : launch_tri_tn<T, DGRAD, false>(p, st);
}

// rows of the forward kernel's statistics output ([rows][C][6]); 0 = the kernel does not take the shape
int dwconv_mfma_small_tri_stats_rows(int N, int C, int H, int W, int K, int dtype) {
    if (dtype != SLAK_BF16 || !dwconv_mfma_small_tri_supported(N, C, H, W, K, dtype)) return 0;
    SmallTriParams p;
    if (quad_enabled() && fill_quad_params(p, N, C, H, W, K, quad_target_wgs())) return p.slices;      // planes up to 7 x 7: sums in the store phase
    if (W < 8) return 0;                                               // (one plane per tile: bound by instructions per plane, the sums cost more than bn3's pass)
    fill_tri_params(p, N, C, H, W, K, 3 * mfma_cu_count());
    return p.slices;
}

int launch_dwconv_mfma_small_tri(bool dgrad, const void* const* in, void* const* out, const float* const* w, int dtype,
                                 int N, int C, int H, int W, int K, hipStream_t st, float* stats) {
    if (!dwconv_mfma_small_tri_supported(N, C, H, W, K, dtype)) return SLAK_ERR_UNSUPPORTED;
    SmallTriParams p;
    p.stats = dgrad ? nullptr : stats;
    fill_tri_params(p, N, C, H, W, K, 768);
    for (int i = 0; i < 3; ++i) { p.in[i] = in[dgrad ? i : 0]; p.out[i] = out[dgrad ? 0 : i]; p.w[i] = w[i]; }
    p.flip = dgrad ? 1 : 0;
    if (quad_enabled() && fill_quad_params(p, N, C, H, W, K, 768)) {       // planes of at most 7 x 7: four per tile
        if (dtype == SLAK_BF16) return dgrad ? launch_quad_t<bf16_t, true>(p, st) : launch_quad_t<bf16_t, false>(p, st);
        return dgrad ? launch_quad_t<f16_t, true>(p, st) : launch_quad_t<f16_t, false>(p, st);
    }
    fill_tri_params(p, N, C, H, W, K, 768);
    if (dtype == SLAK_BF16) return dgrad ? launch_tri_t<bf16_t, true>(p, st) : launch_tri_t<bf16_t, false>(p, st);
    return dgrad ? launch_tri_t<f16_t, true>(p, st) : launch_tri_t<f16_t, false>(p, st);
}

}  // namespace slak
