// slak_amd/csrc/dwconv_mfma_wide.hip -- MFMA depthwise-conv forward / data-grad for maps wider than 64 along the filter's long
// axis (96x96: SLaK at 384 px, BASELINE configs[4]; 128x128: the 512 px segmentation crops; anything with 64 < Wt <= 128,
// Wt % 16 == 0, W % 8 == 0).  The reference takes any H, W (forward_fp32.cu:199-263).
//
// Same arithmetic as dwconv_mfma_dma.hip (1-D Toeplitz GEMM per short tap, ONE accumulator, operands swapped for the vertical
// kernels) with what changes once the map is longer than the filter:
//   * the Toeplitz matrix is a BAND: of the (Wt/32) x (Wt/16) blocks only those with |i - o| <= KL/2 are multiplied (14 of 18
//     at 96, 20 of 32 at 128).  A block's fragment depends on d = ks - 2 mt only (T[o,i] = w[i - o + pad]) and KL <= 63 means
//     -2 <= d <= 3: six fragments per short tap serve every block, whatever the map size.
//   * a plane no longer fits a register-resident tile set: wave mt owns the 32 output positions [32 mt, 32 mt + 32) of the long
//     axis and walks the plane in STRIPS of 32 short-axis positions; each strip's results go through a small double-buffered LDS
//     out-buffer to HBM as 16-byte pieces.
//   * whole planes are DMA'd (`buffer_load_dwordx4 ... lds`) with a PADDED row pitch: at 192 or 256 bytes per row the
//     row-per-lane 16-byte fragment reads would be 4- / 16-way bank conflicted.  LDS-DMA writes are lane-linear, so the padding is
//     made on the source side: destination chunk q of the image takes source chunk (q / cd) * cs + q % cd, pad chunks are skipped
//     lanes (tools/dma_probe.hip: inactive lanes write nothing).
//   * vertical kernels transpose the landed plane LDS -> LDS (ds_read_b64_tr_b16 + ds_write_b64) into x^T once per plane and
//     start the next plane's DMA into the same slot right after; horizontal kernels alternate two slots.
// Zero padding: two all-zero guard rows on either side of the image along the short axis (slot rows / x^T rows, written once);
// along the long axis nothing is needed (Wt % 16 == 0: every k-step lies inside the plane, and the band fragments are zero
// outside the filter).
#include "mfma_common.h"

namespace slak {

constexpr int WD_ND = 6;                // distinct Toeplitz fragments per short tap: d = ks - 2 mt in [-2, 3]
constexpr int WD_ZP = 64;               // zeros in front of a filter row (window starts never go negative)
constexpr int WD_LEN = 192;             // elements per padded filter row
constexpr int WD_WCH = 5;               // filter elements staged per lane of the staging wave (upper bound, 64 lanes)
constexpr int WD_NCO = 2;               // 16-byte copy-out chunks per thread and strip (upper bound)

struct WideParams {
    const void* x; const float* w; void* y;
    int N, C, H, W, kh, kw, flip;
    int Wt, Wl, KL, padL;
    int MT, KS;            // 32-row output tiles / 16-deep k-steps along the long axis
    int tpp;               // strips of 32 short-axis positions per plane
    int cs, cd;            // 16-byte chunks per image row in HBM / in the LDS image (padded pitch)
    int inc_r, inc_c;      // 64 / cd, 64 % cd: how (row, chunk) of a lane's destination advances from one DMA instruction to the next
    int ninstr;            // DMA instructions per plane
    int slot_bytes, NB;    // ring slot, slots
    int PT, xt_bytes;      // vertical: pitch (elements) and size of x^T
    int opitch, out_bytes; // out-buffer row pitch / size (bytes)
    int tr_cbs, tr_q16, tr_r16;   // vertical: 16-column transpose blocks per 4-row band; 16 / tr_cbs, 16 % tr_cbs
    int planes_per_wg, slices;
    unsigned tensor_bytes;
};

// One 32x32 output tile: blocks dd = LO..HI (k-steps 2 mt - 2 + dd; `rp` already points at k-step 2 mt - 2), five short taps each,
// ONE accumulator.  Software pipeline pinned with sched_barrier: hipcc otherwise sinks every ds_read next to its MFMA (ds_read;
// s_waitcnt lgkmcnt(0); v_mfma).  The fragment of tap r for the next block is fetched right after this block's MFMA of tap r has
// issued, into the same registers.
template <typename T, bool VERT, int LO, int HI>
__device__ __forceinline__ f32x16 wide_tile_mma(const s16x8 (&afrag)[MF_TAPS][WD_ND], const char* L, unsigned rp, unsigned rpitch) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    s16x8 b[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) b[r] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + rp + (unsigned)r * rpitch + LO * 32u));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dd = LO; dd <= HI; ++dd) {
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            // vertical: operands swapped (D^T = X^T-tile x T^T) so that a lane holds 4 consecutive columns of one output row
            acc = VERT ? mfma32<T>(b[r], afrag[r][dd], acc) : mfma32<T>(afrag[r][dd], b[r], acc);
            if (dd < HI) b[r] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + rp + (unsigned)r * rpitch + (dd + 1) * 32u));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

template <typename T, bool VERT>
__global__ __launch_bounds__(MF_THREADS, 2) void dwconv_mfma_wide_kernel(const WideParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;                                      // everything below is a BYTE offset into the LDS block
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    if (n_begin >= n_end) return;
    const int nplanes = n_end - n_begin;
    const int HW = p.H * p.W;
    const unsigned pitch = (unsigned)p.cd * 16;                      // row pitch of the DMA image
    const unsigned rpitch = VERT ? (unsigned)p.PT * 2 : pitch;       // row pitch of the image the fragments are read from
    const unsigned ring_b = 0;
    const unsigned xt_b = (unsigned)(p.NB * p.slot_bytes);
    const unsigned lout_b = xt_b + (VERT ? (unsigned)p.xt_bytes : 0u);
    const unsigned win_b = lout_b;                                   // [2 copies][5 taps][WD_LEN]: prologue only, aliases the out-buffers
    constexpr unsigned win_bytes = 2 * MF_TAPS * WD_LEN * 2;

    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)p.x;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes);
        rsrc[3] = 0x00020000;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    // ---- DMA of one plane by ONE wave (wave 3 when MT == 3: it owns no output tile; else waves take turns).  The issuing wave
    // waits with vmcnt(0) right before the barrier that hands the plane over: exact, because it has issued no other DMA since.
    const int issuer_fixed = p.MT == 3 ? 3 : -1;
    const int row0 = lane / p.cd, cc0 = lane - row0 * p.cd;
    auto issue_plane = [&](int pl) {
        if (pl >= nplanes) return;
        const int issuer = issuer_fixed >= 0 ? issuer_fixed : (pl & 3);
        if (wave != issuer) return;                                   // wave-uniform
        const unsigned src0 = (unsigned)(((size_t)(n_begin + pl) * p.C + c) * HW * 2);
        unsigned dst = lds_base + ring_b + (unsigned)(pl % p.NB) * (unsigned)p.slot_bytes + (VERT ? 0u : 2u * pitch);
        int row = row0, cc = cc0;
        for (int k = 0; k < p.ninstr; ++k) {
            if (cc < p.cs && row < p.H) lds_dma16(src0 + (unsigned)(row * p.cs + cc) * 16u, rsrc, __builtin_amdgcn_readfirstlane(dst));
            dst += 1024u;
            cc += p.inc_c; row += p.inc_r;
            if (cc >= p.cd) { cc -= p.cd; ++row; }
        }
    };
    auto issuer_of = [&](int pl) { return issuer_fixed >= 0 ? issuer_fixed : (pl & 3); };

    // ---- prologue: first plane in flight, zero areas, filter windows, fragments ---------------------------------------
    issue_plane(0);
    const int stager = p.MT == 3 ? 2 : 3;                            // not the wave that has just issued plane 0
    const int ntap = p.kh * p.kw;
    float wreg[WD_WCH];
    if (wave == stager) {
#pragma unroll
        for (int k = 0; k < WD_WCH; ++k) { const int e = lane + 64 * k; wreg[k] = e < ntap ? p.w[(size_t)c * ntap + e] : 0.f; }
    }
    {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (unsigned o = tid * 16; o < win_bytes; o += MF_THREADS * 16) *(u32x4*)(L + win_b + o) = z4;
        if constexpr (VERT) {                                         // x^T: guard rows (and pad columns) stay zero
            for (unsigned o = tid * 16; o < (unsigned)p.xt_bytes; o += MF_THREADS * 16) *(u32x4*)(L + xt_b + o) = z4;
        } else {                                                      // two guard rows in front of and behind the image of every slot
            const unsigned g2 = 2u * pitch;
            for (int s = 0; s < p.NB; ++s) {
                const unsigned sb = ring_b + (unsigned)s * (unsigned)p.slot_bytes;
                for (unsigned o = tid * 16; o < g2; o += MF_THREADS * 16) {
                    *(u32x4*)(L + sb + o) = z4;
                    *(u32x4*)(L + sb + (unsigned)(p.H + 2) * pitch + o) = z4;
                }
            }
        }
    }
    wg_barrier();
    if (wave == stager) {
#pragma unroll
        for (int k = 0; k < WD_WCH; ++k) {
            const int e = lane + 64 * k;
            if (e < ntap) {
                int r = VERT ? e % MF_TAPS : e / p.kw, t = VERT ? e / MF_TAPS : e - (e / p.kw) * p.kw;      // short tap r, long tap t
                if (p.flip) { r = MF_TAPS - 1 - r; t = p.KL - 1 - t; }
                const uint16_t v = cvt_to_bits(wreg[k], (T*)nullptr);
                uint16_t* win = (uint16_t*)(L + win_b);
                win[r * WD_LEN + WD_ZP + t] = v;                                           // copy 0
                win[MF_TAPS * WD_LEN + r * WD_LEN + WD_ZP + t - 1] = v;                    // copy 1 = copy 0 shifted by one element
            }
        }
    }
    wg_barrier();
    // Toeplitz fragments: lane (l31 -> o within the tile, lhi -> k half) of block d holds the 8-element window of the padded filter
    // row that starts at 16 d + 8 lhi - l31 + padL.  Independent of mt.
    s16x8 afrag[MF_TAPS][WD_ND];
#pragma unroll
    for (int dd = 0; dd < WD_ND; ++dd) {
        const int a = WD_ZP + 16 * (dd - 2) + lhi * 8 - l31 + p.padL;   // >= 1
        const int par = a & 1;
        const unsigned* src = (const unsigned*)(L + win_b + par * MF_TAPS * WD_LEN * 2) + ((a - par) >> 1);
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            u32x4 d4;
#pragma unroll
            for (int k = 0; k < 4; ++k) d4[k] = src[r * (WD_LEN / 2) + k];
            afrag[r][dd] = __builtin_bit_cast(s16x8, d4);
        }
    }
    // this wave's tile along the long axis and its active blocks dd in [dd_lo, dd_hi] (k-step ks = 2 mt + dd - 2)
    const int mt = wave;
    const bool has_tile = wave < p.MT;
    int dd_lo = 2 - 2 * mt, dd_hi = p.KS + 1 - 2 * mt;               // 0 <= ks < KS
    {
        // block d meets the band  -padL <= i - o <= KL-1-padL  iff  16 d - 31 <= KL-1-padL  and  16 d + 15 >= -padL
        const int dmax = (p.KL - 1 - p.padL + 31) >> 4;              // floor((KL-1-padL+31)/16)
        const int dmin = -((p.padL + 15) >> 4);                      // ceil(-(padL+15)/16)
        if (dd_lo < dmin + 2) dd_lo = dmin + 2;
        if (dd_hi > dmax + 2) dd_hi = dmax + 2;
        if (dd_lo < 0) dd_lo = 0;
        if (dd_hi > WD_ND - 1) dd_hi = WD_ND - 1;
        if (!has_tile) { dd_lo = 1; dd_hi = 0; }
    }

    // ---- per-thread constants of the loops ---------------------------------------------------------------------------------
    // epilogue: lane = output row of the out-buffer, register quad q = 4 consecutive columns -> one 8-byte LDS store each
    //   horizontal: out-buffer [32 strip rows][Wt], row = short position l31, columns mt*32 + 8q + 4 lhi
    //   vertical:   out-buffer [Wt rows][32 strip columns], row = mt*32 + l31, columns 8q + 4 lhi
    const unsigned orel = VERT ? (unsigned)(mt * 32 + l31) * (unsigned)p.opitch + (unsigned)(4 * lhi) * 2
                               : (unsigned)l31 * (unsigned)p.opitch + (unsigned)(mt * 32 + 4 * lhi) * 2;
    // copy-out: 16-byte chunk idx of a strip's out-buffer -> HBM
    unsigned co_l[WD_NCO], co_g[WD_NCO]; int co_row[WD_NCO], co_col[WD_NCO];
#pragma unroll
    for (int k = 0; k < WD_NCO; ++k) {
        const int idx = tid + k * MF_THREADS;
        if constexpr (VERT) {
            const int row = idx >> 2, c4 = idx & 3;
            co_l[k] = (unsigned)row * (unsigned)p.opitch + (unsigned)c4 * 16u;
            co_g[k] = (unsigned)(row * p.W + c4 * 8) * 2u;
            co_row[k] = row < p.H ? 0 : (1 << 30);                    // valid row (constant over strips)
            co_col[k] = c4 * 8;                                       // + 32 s < W
        } else {
            const int row = idx / p.cs, cc = idx - row * p.cs;
            co_l[k] = (unsigned)row * (unsigned)p.opitch + (unsigned)cc * 16u;
            co_g[k] = (unsigned)idx * 16u;
            co_row[k] = row < 32 ? row : (1 << 30);                   // + 32 s < H
            co_col[k] = 0;
        }
    }
    // vertical: transpose of the landed plane.  Block b = (4 image rows kb, 16 image columns cb); the 16 lanes of a group read it
    // with one ds_read_b64_tr_b16 (lane i16 supplies row kb*4 + i16/4, columns cb*16 + 4*(i16%4) and receives column cb*16 + i16,
    // rows kb*4..+3) and write 8 bytes of x^T (row = image column + 2 guard rows).
    auto transpose_plane = [&]() {
        const int grp = lane >> 4, i16 = lane & 15;
        const int total = (p.H >> 2) * p.tr_cbs;
        int b = wave * 4 + grp;
        int kb = b / p.tr_cbs, cb = b - kb * p.tr_cbs;
        for (; b < total; b += 16) {                                  // uniform per 16-lane group
            const unsigned src = ring_b + (unsigned)(kb * 4 + (i16 >> 2)) * pitch + (unsigned)(cb * 32 + (i16 & 3) * 8);
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + src));
            const int col = cb * 16 + i16;
            if (col < p.W) *(s16x4*)(L + xt_b + (unsigned)((2 + col) * p.PT + kb * 4) * 2u) = v;
            cb += p.tr_r16; kb += p.tr_q16;
            if (cb >= p.tr_cbs) { cb -= p.tr_cbs; ++kb; }
        }
    };

    int cnt = 0;                                                      // strips done (out-buffer parity)
    for (int pl = 0; pl < nplanes; ++pl) {
        if (wave == issuer_of(pl)) wait_vmcnt<0>();
        wg_barrier();                                                 // plane pl has landed; everyone is done with plane pl-1
        unsigned img_b;
        if constexpr (VERT) {
            transpose_plane();
            wg_barrier();
            issue_plane(pl + 1);                                      // the slot is free again
            img_b = xt_b;
        } else {
            issue_plane(pl + 1);                                      // into the other slot
            img_b = ring_b + (unsigned)(pl % p.NB) * (unsigned)p.slot_bytes;
        }
        char* const yplane = (char*)p.y + ((size_t)(n_begin + pl) * p.C + c) * HW * 2;
        for (int s = 0; s < p.tpp; ++s, ++cnt) {
            const unsigned ob = lout_b + (unsigned)(cnt & 1) * (unsigned)p.out_bytes;
            if (has_tile) {
                // fragment of tap r, block dd: 16 bytes at row (32 s + l31 + r) of the guarded image, columns 16 ks + 8 lhi ..
                const unsigned rp = img_b + (unsigned)(s * 32 + l31) * rpitch + (unsigned)lhi * 16u + (unsigned)(2 * mt - 2) * 32u;
                f32x16 acc;
                // the block range is wave-uniform but not a compile-time constant; one straight-line instantiation per range
                // (conditions inside the pinned pipeline made hipcc shuffle the fragment registers: ~50 instructions per MFMA)
#define SLAK_WD_CASE(LO, HI) case (LO) * 8 + (HI): acc = wide_tile_mma<T, VERT, LO, HI>(afrag, L, rp, rpitch); break;
                switch (dd_lo * 8 + dd_hi) {
                    SLAK_WD_CASE(0, 2) SLAK_WD_CASE(0, 3) SLAK_WD_CASE(0, 4) SLAK_WD_CASE(0, 5)
                    SLAK_WD_CASE(1, 2) SLAK_WD_CASE(1, 3) SLAK_WD_CASE(1, 4) SLAK_WD_CASE(1, 5)
                    SLAK_WD_CASE(2, 2) SLAK_WD_CASE(2, 3) SLAK_WD_CASE(2, 4) SLAK_WD_CASE(2, 5)
                    default: acc = wide_tile_mma<T, VERT, 0, 5>(afrag, L, rp, rpitch); break;      // not reached (fill_wide_params)
                }
#undef SLAK_WD_CASE
                const bool row_ok = VERT ? (mt * 32 + l31 < p.Wt) : (s * 32 + l31 < p.Wl);
                const int col0 = VERT ? s * 32 + 4 * lhi : mt * 32 + 4 * lhi, ncol = VERT ? p.Wl : p.Wt;
                if (row_ok) {
                    char* op = L + ob + orel;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (col0 + 8 * q < ncol) {
                            u32x2 v;
                            v[0] = pack2<T>(acc[4 * q + 0], acc[4 * q + 1]);
                            v[1] = pack2<T>(acc[4 * q + 2], acc[4 * q + 3]);
                            *(u32x2*)(op + 16 * q) = v;
                        }
                    }
                }
            }
            wg_barrier();                                             // the strip's out-buffer is complete
            char* const ys = yplane + (VERT ? (size_t)s * 64 : (size_t)s * 32 * p.W * 2);
#pragma unroll
            for (int k = 0; k < WD_NCO; ++k) {
                const bool ok = VERT ? (co_row[k] == 0 && s * 32 + co_col[k] < p.W) : (s * 32 + co_row[k] < p.H);
                if (ok) *(u32x4*)(ys + co_g[k]) = *(const u32x4*)(L + ob + co_l[k]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_wide_params(WideParams& p, const ConvDims& d, bool vert, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return false;
    if (p.Wt <= 64 || p.Wt > 128 || p.Wt % 16) return false;
    if (d.W % 8 || d.H % 4 || p.Wl % 4 || p.Wl > 128) return false;
    if (p.KL > 63 || d.kh * d.kw > WD_WCH * 64) return false;
    p.MT = (p.Wt + 31) / 32; p.KS = p.Wt / 16;
    p.tpp = (p.Wl + 31) / 32;
    p.cs = d.W / 8;
    if (vert) { p.cd = p.cs; while (p.cd % 16 != 4 && p.cd % 16 != 12) ++p.cd; }      // tr-reads: pitch = +-64 bytes mod 256
    else p.cd = p.cs | 1;                                                               // row-per-lane b128 reads: odd chunk pitch
    p.inc_r = 64 / p.cd; p.inc_c = 64 % p.cd;
    p.ninstr = (d.H * p.cd + 63) / 64;
    p.slot_bytes = (vert ? d.H : d.H + 4) * p.cd * 16;
    p.NB = vert ? 1 : 2;
    p.PT = p.Wt + 8;
    p.xt_bytes = vert ? (p.Wl + 4) * p.PT * 2 : 0;
    p.opitch = vert ? 80 : (p.cs | 1) * 16;
    p.out_bytes = vert ? p.Wt * p.opitch : 32 * p.opitch;
    p.tr_cbs = (d.W + 15) / 16; p.tr_q16 = 16 / p.tr_cbs; p.tr_r16 = 16 % p.tr_cbs;
    if ((vert ? 4 * d.H : 32 * p.cs) > WD_NCO * MF_THREADS) return false;
    int slices = resident_wgs / d.C; if (slices < 1) slices = 1;      // one resident round
    if (slices > d.N) slices = d.N;
    const int per = (d.N + slices - 1) / slices;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * d.H * d.W * 2);
    return true;
}

static size_t wide_lds_bytes(const WideParams& p) {
    const size_t win = (size_t)2 * MF_TAPS * WD_LEN * 2, out2 = (size_t)2 * p.out_bytes;
    // fragment reads of lanes beyond the short-axis extent run up to 35 rows past the image: keep them inside the block
    const size_t slack = (size_t)36 * (p.xt_bytes ? p.PT * 2 : p.cd * 16);
    size_t tail = out2 > win ? out2 : win;
    if (tail < slack) tail = slack;
    return (size_t)p.NB * p.slot_bytes + p.xt_bytes + tail + 64;
}

bool dwconv_mfma_wide_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt) {
    if (x_dt != y_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16) || w_dt != SLAK_F32) return false;
    WideParams p;
    if (!fill_wide_params(p, d, d.kh > d.kw, 512)) return false;
    return wide_lds_bytes(p) <= 160 * 1024;
}

template <typename T, bool VERT>
static int launch_wide_tv(WideParams& p, const ConvDims& d, hipStream_t st) {
    auto k = dwconv_mfma_wide_kernel<T, VERT>;
    const size_t lds = wide_lds_bytes(p);
    static thread_local size_t cached_lds = 0; static thread_local int cached_per_cu = 0;     // per instantiation; queried once per (device, LDS size)
    const size_t lds_key = ((size_t)(slak_current_device() + 1) << 32) | lds;
    if (cached_lds != lds_key) {
        (void)slak_set_max_lds((const void*)k, lds);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, MF_THREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        cached_per_cu = per_cu > 8 ? 8 : per_cu; cached_lds = lds_key;
    }
    fill_wide_params(p, d, VERT, cached_per_cu * mfma_cu_count());
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_wide(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                            const ConvDims& d, bool flip_filter, hipStream_t st) {
    if (!dwconv_mfma_wide_supported(d, x_dt, w_dt, y_dt)) return SLAK_ERR_UNSUPPORTED;
    const bool vert = d.kh > d.kw;
    WideParams p;
    fill_wide_params(p, d, vert, 512);
    p.x = x; p.w = (const float*)w; p.y = y; p.flip = flip_filter ? 1 : 0;
    if (x_dt == SLAK_BF16) return vert ? launch_wide_tv<bf16_t, true>(p, d, st) : launch_wide_tv<bf16_t, false>(p, d, st);
    return vert ? launch_wide_tv<f16_t, true>(p, d, st) : launch_wide_tv<f16_t, false>(p, d, st);
}

}  // namespace slak
