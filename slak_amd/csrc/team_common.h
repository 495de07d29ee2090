// slak_amd/csrc/team_common.h -- what the four-wave-team three-branch kernels share (dwconv_mfma_team_tri.hip: one tile per plane;
// dwconv_mfma_team_half.hip: planes of 2 x 2 tiles): parameters, the cross-tile B-fragment pipeline, Toeplitz fragments from LDS
// filter windows.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "mfma_common.h"

namespace slak {

extern unsigned long long* g_dma_dbg;    // dev hook: slak_debug_set_phase_buffer()

constexpr int TT_WAVES = 4;
constexpr int TT_THREADS = TT_WAVES * 64;
constexpr int TT_ZP = 64;               // zeros in front of a filter row (window starts never go negative)
constexpr int TT_LEN = 192;             // elements per padded filter row
constexpr int TT_ZROW = 32;             // elements of the all-zero row that out-of-range k pieces point at (two 16-byte halves are read)
constexpr int TT_NTR = 4;               // transpose blocks (4 rows x 16 cols) per 16-lane group and group of planes (upper bound)
constexpr int TT_WCH = 5;               // filter elements staged per lane of a staging wave (upper bound, 64 lanes)
constexpr int TT_NCO = 2;               // 16-byte copy-out chunks per thread, tensor and group (upper bound)
constexpr unsigned TT_OOB = 0x80000000u;   // a buffer offset beyond every tensor (< 2^31 bytes): loads return zeros, stores are dropped

// Dev instrumentation (ablation bits and the s_memtime timeline of tools/time_team.py / tools/team_timeline.py) exists only in a build with
// -DSLAK_TEAM_DEV; in the shipped library the conditions are constants and the compiler drops the code.
#ifdef SLAK_TEAM_DEV
#define TT_DBG(p, bit) ((p).dbg & (bit))
#define TT_TIMELINE(p, vb) ((p).tl ? (p).tl + (size_t)(vb) * 64 : nullptr)
static inline int team_dev_flags() { static const int dbg = [] { const char* e = slak_dev_getenv("SLAK_TEAM_DBG"); return e ? atoi(e) : 0; }(); return dbg; }
#else
#define TT_DBG(p, bit) 0
#define TT_TIMELINE(p, vb) ((unsigned long long*)nullptr)
static inline int team_dev_flags() { return 0; }
#endif

constexpr int TT_SOLO_DEFAULT = 15;  // SLAK_TEAM_SOLO bit mask (dwconv_mfma_team_tri.hip): one team per workgroup measured 5-12 % faster for every (op, class)

#ifndef TT_IO_PRIO
#define TT_IO_PRIO 1
#endif

struct TeamPiece { unsigned lds_off, g_off; int info; };       // info = tensor | plane-of-group << 4 | lanes << 8 (0: no such piece)
constexpr int TT_NPW = 6;               // LDS-DMA pieces per wave and group (upper bound: dgrad)

struct TeamParams {
    const void* in[3]; void* out[3]; const float* w[3];       // branch order: vertical (K x 5), horizontal (5 x K), small (5 x 5)
    TeamPiece pieces[TT_WAVES][TT_NPW];   // piece q = (tensor t, plane j of the group, 64-chunk piece pp) belongs to wave q % 4 (host table: no SGPR arrays)
    int my_pieces[TT_WAVES];
    int N, C, H, W, K, dgrad;
    int G;                 // planes per group (iteration): 1 (two tiles per axis) or 4 (one tile per plane)
    int chunks_pp;         // 16-byte chunks per plane (HW/8)
    int ppp;               // LDS-DMA pieces (64 chunks) per plane
    int plane_lds;         // LDS elements from one plane of a ring slot to the next (HW + 2W guard rows)
    int tslot_elems;       // LDS elements of one tensor's part of a ring slot (guarded image)
    int t0_elems, t0_plane, t0_first;   // tensor 0's part, plane stride and first-plane offset: the guarded image (forward) or -- dgrad, where dy of the
                           // vertical branch is only ever transposed -- the bare planes (no guard rows: 448 bytes per slot that buy the third ring slot)
    int NT;                // tensors in a slot: 1 (forward) or 3 (dgrad)
    int NB;                // ring depth in groups
    int PT;                // pitch of the transposed image
    int xt_rows;           // rows of one transposed plane image incl. 2+2 guard rows
    int planes_per_wg, slices;   // per TEAM
    int iters_max;         // groups per team (upper bound: both teams of a workgroup run this many phase pairs)
    int team_lds;          // LDS bytes of one team
    unsigned m_cpp;        // magic multiplier: n / chunks_pp == (n * m_cpp) >> 22
    int tr_pp, tr_cbs;     // transpose blocks per plane, per 4-row band
    unsigned tensor_bytes;
    float* stats;          // forward only, or NULL: [slices * 4][C][6] partial (sum y_v, sum y_v^2, sum y_h, sum y_h^2, sum y_s, sum y_s^2)
    int dbg;               // dev (SLAK_TEAM_DBG): 1 skip the MFMA tiles, 2 skip the copy-out, 4 skip the transposes
    unsigned long long* tl; // dev: per-team timeline [team][64] (slak_debug_set_phase_buffer)
};

// The B fragment of tap r, k-step ks of a tile: 16 bytes at rp0 + r*rpitch + 32*ks.  SWAP: rows of x^T (vertical branch; pads are
// zero).  Otherwise rows of the guarded row-major image; KS = k-steps of the plane class: the last two may reach past the row end --
// pieces beyond it read the zero row instead (wlim = W - lhi*8: k-step ks of this lane lies inside the row iff ks*16 < wlim).
template <bool SWAP, bool R16, int KS>
__device__ __forceinline__ s16x8 team_load_b(const char* L, unsigned rpr, int ks, int wlim, unsigned zrow_l) {
    u32x4 b;
    if constexpr (SWAP) b = *(const u32x4*)(L + rpr + ks * 32);
    else if constexpr (R16) {
        if (ks >= KS - 2) { const unsigned q = ks * 16 < wlim ? rpr + ks * 32 : zrow_l; b = *(const u32x4*)(L + q); }
        else b = *(const u32x4*)(L + rpr + ks * 32);
    } else {                                                                          // W % 8 == 4: rows are 8-byte aligned
        if (ks >= KS - 2) {
            const unsigned q0 = ks * 16 < wlim ? rpr + ks * 32 : zrow_l, q1 = ks * 16 + 4 < wlim ? rpr + ks * 32 + 8 : zrow_l + 8;
            const u32x2 lo = *(const u32x2*)(L + q0), hi = *(const u32x2*)(L + q1);
            b = u32x4{lo[0], lo[1], hi[0], hi[1]};
        } else {
            const u32x2 lo = *(const u32x2*)(L + rpr + ks * 32), hi = *(const u32x2*)(L + rpr + ks * 32 + 8);
            b = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    }
    return __builtin_bit_cast(s16x8, b);
}
constexpr int TT_NBUF = MF_TAPS + 1;
// the first five fragments of a tile (the tile that opens a compute phase; later tiles get theirs from their predecessor)
template <bool SWAP, bool R16, int KS, int K0, int ROT>
__device__ __forceinline__ void team_tile_prefetch(s16x8 (&b)[TT_NBUF], const char* L, unsigned rp0, unsigned rpitch, int wlim, unsigned zrow_l) {
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) b[(ROT + r) % TT_NBUF] = team_load_b<SWAP, R16, KS>(L, rp0 + r * rpitch, K0, wlim, zrow_l);
}
// One 32x32 tile: NK k-steps starting at K0, five short taps each, accumulated INTO acc; SWAP: operands swapped (vertical branch:
// D^T = X^T-tile x T^T, so that a lane holds 4 consecutive columns of one output row).  A software pipeline pinned with sched_barrier:
// five fragments are in flight in SIX buffers -- the fragment fetched right after MFMA j goes into the registers MFMA j-1 read (an LDS
// load into the registers of the MFMA that has just issued waits for that MFMA to read them), fragment j of the tile lives in
// b[(ROT + j) % 6] and the first five are already there (team_tile_prefetch, or the tile before: NXT).  The pipeline runs ACROSS tiles:
// during the last k-step the first five fragments of the NEXT tile are fetched (NXT = 1: a SWAP tile, 2: a plain tile, at nrp0 /
// npitch, k-step NXT_K0), so a tile boundary costs no LDS latency (measured before: 64 cycles per MFMA with ~300 idle cycles per
// boundary, against 37 inside a tile).
struct TeamNoFill { __device__ __forceinline__ void operator()(int) const {} };
// fill(j): instructions issued in the shadow of MFMA j -- the epilogue of the tile before (pack + LDS stores of an accumulator that is complete)
template <typename T, bool SWAP, bool R16, int KS, int NK, int K0, int ROT, int NXT, int NXT_K0, typename F = TeamNoFill>
__device__ __forceinline__ void team_tile_mma(f32x16& acc, const s16x8 (&afrag)[MF_TAPS][NK], s16x8 (&b)[TT_NBUF], const char* L, unsigned rp0,
                                              unsigned rpitch, int wlim, unsigned zrow_l, unsigned nrp0, unsigned npitch, F fill = F()) {
    constexpr int NJ = MF_TAPS * NK;
    unsigned rp[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) rp[r] = rp0 + r * rpitch;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int kk = j / MF_TAPS, r = j % MF_TAPS;
        acc = SWAP ? mfma32<T>(b[(ROT + j) % TT_NBUF], afrag[r][kk], acc) : mfma32<T>(afrag[r][kk], b[(ROT + j) % TT_NBUF], acc);
        if (j + MF_TAPS < NJ) {
            b[(ROT + j + MF_TAPS) % TT_NBUF] = team_load_b<SWAP, R16, KS>(L, rp[(j + MF_TAPS) % MF_TAPS], K0 + (j + MF_TAPS) / MF_TAPS, wlim, zrow_l);
            asm volatile("" :: "v"(b[(ROT + j) % TT_NBUF]));         // the operand MFMA j is reading stays allocated across the load (no register reuse)
        } else if constexpr (NXT != 0) {
            const int i = j + MF_TAPS - NJ;                           // fragment i of the next tile
            b[(ROT + j + MF_TAPS) % TT_NBUF] = team_load_b<NXT == 1, R16, KS>(L, nrp0 + i * npitch, NXT_K0, wlim, zrow_l);
            asm volatile("" :: "v"(b[(ROT + j) % TT_NBUF]));
        }
        fill(j);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// CLS 2: the 5 x 5 tile with its Toeplitz fragments in LDS (keeping them in registers -- 60 on top of branch A's 80 -- spilled).
// A fragment depends on d = ks - 2 mt only (window start 16 d + 8 lhi - l31 + 2): twenty fragments (d = -1 .. 2 x five taps) of 1 KiB
// at sfr + ((d + 1) * 5 + r) * 1024 + lane * 16, built once per team.  Entries for inputs beyond the row need no mask: the B pieces
// there are the zero row.  Tile (MT, sub): k-steps MT .. MT + 2, i.e. d = kk - MT.  Both operands of MFMA j are fetched five MFMAs
// ahead: B into bq (chained from the tile before, as in team_tile_mma), A into sa (first five by team_small_prefetch).
constexpr int TT_SFR_BYTES = 4 * MF_TAPS * 1024;
template <int MT>
__device__ __forceinline__ void team_small_prefetch(s16x8 (&sa)[TT_NBUF], const char* L, unsigned sfr_l) {
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) sa[r] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + sfr_l + ((1 - MT) * MF_TAPS + r) * 1024));
}
template <typename T, bool R16, int KS, int MT, int ROT, int NXT, int NXT_K0, typename F = TeamNoFill>
__device__ __forceinline__ void team_small_tile_mma(f32x16& acc, s16x8 (&sa)[TT_NBUF], s16x8 (&b)[TT_NBUF], const char* L, unsigned rp0, unsigned rpitch,
                                                    int wlim, unsigned zrow_l, unsigned sfr_l, unsigned nrp0, unsigned npitch, F fill = F()) {
    constexpr int NJ = MF_TAPS * 3;
    unsigned rp[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) rp[r] = rp0 + r * rpitch;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        acc = mfma32<T>(sa[j % TT_NBUF], b[(ROT + j) % TT_NBUF], acc);
        if (j + MF_TAPS < NJ) {
            const int jn = j + MF_TAPS, kk = jn / MF_TAPS, r = jn % MF_TAPS;
            sa[jn % TT_NBUF] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + sfr_l + ((kk - MT + 1) * MF_TAPS + r) * 1024));
            b[(ROT + jn) % TT_NBUF] = team_load_b<false, R16, KS>(L, rp[r], MT + kk, wlim, zrow_l);
            asm volatile("" :: "v"(b[(ROT + j) % TT_NBUF]), "v"(sa[j % TT_NBUF]));
        } else if constexpr (NXT != 0) {
            const int i = j + MF_TAPS - NJ;                           // fragment i of the next tile
            b[(ROT + j + MF_TAPS) % TT_NBUF] = team_load_b<NXT == 1, R16, KS>(L, nrp0 + i * npitch, NXT_K0, wlim, zrow_l);
            asm volatile("" :: "v"(b[(ROT + j) % TT_NBUF]));
        }
        fill(j);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Toeplitz fragments of one branch for Toeplitz rows mt*32.., k-steps K0 .. K0+NK-1, from that branch's filter windows in LDS
template <int NK>
__device__ __forceinline__ void team_build_frags(s16x8 (&afrag)[MF_TAPS][NK], const char* win, int K0, int mt, int l31, int lhi, int Wt, int padL) {
    const int kfull = Wt >> 4;                                       // k-steps below this lie entirely inside the plane
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        const int ks = K0 + kk;
        const int a = TT_ZP + ks * 16 + lhi * 8 - (mt * 32 + l31) + padL;                 // window start (element index), >= 1
        const int par = a & 1;
        const unsigned* src = (const unsigned*)(win + par * MF_TAPS * TT_LEN * 2) + ((a - par) >> 1);
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            u32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = src[r * (TT_LEN / 2) + k];
                if (ks >= kfull && ks * 16 + lhi * 8 + 2 * k >= Wt) d[k] = 0u;          // i >= Wt: no such input
            }
            afrag[r][kk] = __builtin_bit_cast(s16x8, d);
        }
    }
}

}  // namespace slak
