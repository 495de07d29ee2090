// slak_amd/csrc/dwconv_mfma_wide_tri.hip -- the THREE branches of a decomposed block (K x 5, 5 x K, 5 x 5: models/SLaK.py:82-100) in ONE launch
// on maps with 64 < H, W <= 96 (96 x 96: SLaK at 384 px, BASELINE configs[4]; 80 x 80), forward and data gradient.  The reference runs
// forward_fp16.cu:186-249 / backward_data_fp16.cu:184-246 once per branch (x read three times, three partial gradients added by autograd).
//
// Arithmetic: dwconv_mfma_wide.hip's -- 1-D Toeplitz GEMM per short tap along the branch's long axis, the Toeplitz matrix a BAND of 16-wide
// blocks whose fragment depends on d = ks - 2 mt only (six per tap for the long branches, four for the 5 x 5 one), operands swapped for the
// vertical branch so that all three leave the SAME lane / register map of a 32 x 32 output tile (lane = row, four consecutive columns per
// register quad).  What is new:
//   * ONE staged plane serves the three branches (forward: x is read from HBM once instead of three times; data gradient: the three partial
//     gradients of a tile are added in the accumulator registers and rounded once -- autograd's two tensor adds and their traffic disappear).
//   * A wave owns output COLUMN BLOCK mt of all three branches: region (row band s, column block mt) is the horizontal tile (s-th strip of
//     tile mt), the 5 x 5 tile of the same rows and columns, and the vertical branch's tile of row band s in ITS strip mt (a vertical wave's
//     Toeplitz fragments do not depend on the row band, so it can walk the bands) -- that is what makes the three results meet in one wave.
//   * The wave keeps the fragments of all three branches (120 + 120 + 80 registers): the kernel is compiled for ONE wave per SIMD (512
//     registers).  Waves 0..2 compute; wave 3 is the plane SERVER: it issues the LDS-DMA of plane p + 1 (every input tensor), waits for it
//     and transposes the vertical branch's input into x^T while the others compute plane p -- one workgroup barrier per plane, nobody waits
//     for a transpose.
//   * Results leave through a wave-private staging tile as 16-byte stores (64 contiguous bytes per row and tile): no out-buffer barrier.
//   * Work decomposition: the C * N planes in (channel, image) order are cut into one equal RANGE per CU (whole slices per channel leave a
//     quarter of the CUs idle at C = 96); a range that crosses a channel boundary rebuilds its fragments there (three barriers, ~3 us).
#include "mfma_common.h"

namespace slak {

constexpr int WT_ND = 6;                // Toeplitz fragments per short tap of a long branch: d = ks - 2 mt in [-2, 3]
constexpr int WT_NDS = 4;               // of the 5 x 5 branch: d in [-1, 2]
constexpr int WT_ZP = 64;               // zeros in front of a filter row (window starts never go negative)
constexpr int WT_LEN = 192;             // elements per padded filter row
constexpr unsigned WT_WIN = 2u * MF_TAPS * WT_LEN * 2u;      // bytes of one branch's windows: two copies one element apart
constexpr unsigned WT_STP = 80;         // staging tile: row pitch (64 bytes of results + 16: the 8-byte epilogue writes stay conflict-free)
constexpr unsigned WT_STB = 32 * WT_STP;

struct WideTriParams {
    const void* in[3];                  // forward: in[0] = x; data gradient: dy of the K x 5, 5 x K, 5 x 5 branch
    const float* w[3];                  // (C,1,K,5), (C,1,5,K), (C,1,5,5)
    void* out[3];                       // forward: y of the three branches; data gradient: out[0] = dx
    int N, C, H, W, K, padL;
    int MTr, KSr, KSc;                  // 32-row bands; 16-deep k-steps along the rows (vertical branch) / along the columns
    int cs;                             // 16-byte chunks per image row in HBM
    int cdh, ninh, slot_h;              // row-major image (horizontal / 5 x 5 operands): chunks per LDS row (odd), DMA instructions per plane, bytes (4 guard rows)
    int cdv, ninv, slot_v;              // data gradient: the vertical branch's dy, landed compact for the transposing reads
    int PT, xt_bytes;                   // x^T: pitch (elements), bytes
    int trc;                            // 16-column transpose blocks per 4-row band
    int per, planes;                    // planes per workgroup range, C * N
    unsigned tensor_bytes;
};

// The store of a finished tile, cut into steps that ride behind the MFMAs of the NEXT tile (one wave per SIMD: nothing else hides them).
// lane = row l31 of the tile, register quad q = columns 8 q + 4 lhi .. + 3: through the wave's staging tile (a wave's LDS operations execute in
// order: the reads below see the writes above them without a wait) to two 16-byte stores per lane, 64 contiguous bytes per row.
template <typename T>
struct WtStore {
    f32x16 acc; char* stg; __amdgpu_buffer_rsrc_t rsrc; unsigned soff, go0, go1; u32x4 r0, r1;
    // rsrc: the output tensor; soff: the plane's byte offset in it (wave-uniform); go0 / go1: this lane's two 16-byte pieces in the plane (row
    // 32 s + lane / 4 resp. 16 rows further down, columns 32 mt + 8 (lane % 4)) -- or an out-of-range offset: the buffer range check drops the
    // store, so nothing in the MFMA stream is conditional (a branch inside the pinned pipeline makes hipcc spill the fragments)
    __device__ __forceinline__ void arm(const f32x16& a, char* staging, __amdgpu_buffer_rsrc_t out, unsigned plane_off, int s, int mt, int lane, int H, int W, bool live) {
        acc = a; stg = staging; rsrc = out; soff = plane_off;
        const int row0 = lane >> 2, c4 = lane & 3, gc = mt * 32 + c4 * 8;
        const unsigned g = (unsigned)((s * 32 + row0) * W + gc) * 2u;
        go0 = (live && s * 32 + row0 < H && gc < W) ? g : 0x80000000u;
        go1 = (live && s * 32 + row0 + 16 < H && gc < W) ? g + (unsigned)(16 * W) * 2u : 0x80000000u;
    }
    __device__ __forceinline__ void step(int k, int l31, int lhi, int lane) {
        if (k < 4) {
            u32x2 v;
            v[0] = pack2<T>(acc[4 * k + 0], acc[4 * k + 1]);
            v[1] = pack2<T>(acc[4 * k + 2], acc[4 * k + 3]);
            *(u32x2*)(stg + (unsigned)l31 * WT_STP + (unsigned)(8 * lhi + 16 * k)) = v;
        } else if (k == 4) {                                            // (a wave's LDS operations execute in order: no wait between the writes and these reads)
            asm volatile("" ::: "memory");
            const unsigned o = (unsigned)(lane >> 2) * WT_STP + (unsigned)(lane & 3) * 16u;
            r0 = *(const u32x4*)(stg + o); r1 = *(const u32x4*)(stg + o + 16u * WT_STP);
            asm volatile("" ::: "memory");
        } else if (k == 7) {                                            // three MFMAs later: the reads have returned
            __builtin_amdgcn_raw_buffer_store_b128(r0, rsrc, go0, soff, 0);
            __builtin_amdgcn_raw_buffer_store_b128(r1, rsrc, go1, soff, 0);
        }
    }
};
constexpr int WT_STEPS = 8;

// One 32 x 32 tile of one branch: blocks dd = LO..HI (the fragment array starts at block DOFF), five short taps each, added into `acc`.
// The pinned software pipeline of dwconv_mfma_wide.hip: the fragment of tap r for the next block is fetched right behind this block's MFMA of tap r;
// behind MFMA j also step j of the previous tile's store.
template <typename T, bool VERT, bool FILL, int ND, int DOFF, int LO, int HI>
__device__ __forceinline__ f32x16 wt_tile(const s16x8 (&af)[MF_TAPS][ND], const char* L, unsigned rp, unsigned rpitch, f32x16 acc, WtStore<T>& st, int l31, int lhi, int lane) {
    s16x8 b[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) b[r] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + rp + (unsigned)r * rpitch + LO * 32u));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dd = LO; dd <= HI; ++dd) {
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            acc = VERT ? mfma32<T>(b[r], af[r][dd - DOFF], acc) : mfma32<T>(af[r][dd - DOFF], b[r], acc);
            if (dd < HI) b[r] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + rp + (unsigned)r * rpitch + (dd + 1) * 32u));
            if constexpr (FILL) if ((dd - LO) * MF_TAPS + r < WT_STEPS) st.step((dd - LO) * MF_TAPS + r, l31, lhi, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (FILL) {
#pragma unroll
        for (int k = (HI - LO + 1) * MF_TAPS; k < WT_STEPS; ++k) st.step(k, l31, lhi, lane);  // (a tile of one block: the rest of the store)
    }
    return acc;
}

// the wave-uniform block range picks one straight-line instantiation (a run-time range inside the pinned pipeline makes hipcc shuffle the fragments)
template <typename T, bool VERT, bool FILL>
__device__ __forceinline__ f32x16 wt_long_tile(const s16x8 (&af)[MF_TAPS][WT_ND], const char* L, unsigned rp, unsigned rpitch, int lo, int hi, f32x16 acc,
                                               WtStore<T>& st, int l31, int lhi, int lane) {
#define SLAK_WT_CASE(LO, HI) case (LO) * 8 + (HI): return wt_tile<T, VERT, FILL, WT_ND, 0, LO, HI>(af, L, rp, rpitch, acc, st, l31, lhi, lane);
    switch (lo * 8 + hi) {
        SLAK_WT_CASE(0, 2) SLAK_WT_CASE(0, 3) SLAK_WT_CASE(0, 4) SLAK_WT_CASE(0, 5)
        SLAK_WT_CASE(1, 2) SLAK_WT_CASE(1, 3) SLAK_WT_CASE(1, 4) SLAK_WT_CASE(1, 5)
        SLAK_WT_CASE(2, 2) SLAK_WT_CASE(2, 3) SLAK_WT_CASE(2, 4) SLAK_WT_CASE(2, 5)
        default:                                                      // an empty range: only the pending store
            if constexpr (FILL) {
#pragma unroll
                for (int k = 0; k < WT_STEPS; ++k) st.step(k, l31, lhi, lane);
            }
            return acc;
    }
#undef SLAK_WT_CASE
}
template <typename T, bool FILL>
__device__ __forceinline__ f32x16 wt_small_tile(const s16x8 (&af)[MF_TAPS][WT_NDS], const char* L, unsigned rp, unsigned rpitch, int lo, int hi, f32x16 acc,
                                                WtStore<T>& st, int l31, int lhi, int lane) {
#define SLAK_WT_CASE(LO, HI) case (LO) * 8 + (HI): return wt_tile<T, false, FILL, WT_NDS, 1, LO, HI>(af, L, rp, rpitch, acc, st, l31, lhi, lane);
    switch (lo * 8 + hi) {
        SLAK_WT_CASE(1, 2) SLAK_WT_CASE(1, 3) SLAK_WT_CASE(1, 4) SLAK_WT_CASE(2, 2) SLAK_WT_CASE(2, 3) SLAK_WT_CASE(2, 4)
        default:
            if constexpr (FILL) {
#pragma unroll
                for (int k = 0; k < WT_STEPS; ++k) st.step(k, l31, lhi, lane);
            }
            return acc;
    }
#undef SLAK_WT_CASE
}

constexpr int WT_MAXDMA = 20;           // DMA instructions per plane copy at most (96 rows x 13 chunks / 64)
constexpr int WT_MAXTR = 36;            // transpose blocks per 16-lane group of the server wave at most (24 row bands x 6 column blocks / 4)
constexpr unsigned WT_NONE = 0xffffffffu;

template <typename T, bool DGRAD>
__global__ __launch_bounds__(MF_THREADS, 1) void dwconv_mfma_wide_tri_kernel(const WideTriParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;
    constexpr int R = DGRAD ? 2 : 3;                                  // ring slots of the row-major inputs (forward: the DMA runs two planes ahead)
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int q0 = blockIdx.x * p.per;
    int q1 = q0 + p.per; if (q1 > p.planes) q1 = p.planes;
    const int iters = q1 - q0;
    if (iters <= 0) return;
    const int HW = p.H * p.W;
    const unsigned pitch = (unsigned)p.cdh * 16u, xpitch = (unsigned)p.PT * 2u;
    // ---- LDS map (byte offsets) ---------------------------------------------------------------------------------------------
    const unsigned h_b = 0;                                           // [R] row-major slots: x (forward) / dy of the horizontal branch
    const unsigned s_b = h_b + (unsigned)R * (unsigned)p.slot_h;      // [R] data gradient: dy of the 5 x 5 branch
    const unsigned v_b = s_b + (DGRAD ? (unsigned)R * (unsigned)p.slot_h : 0u);   // data gradient: dy of the vertical branch, compact
    const unsigned xt_b = v_b + (DGRAD ? (unsigned)p.slot_v : 0u);    // [2] x^T (the vertical branch's operand)
    const unsigned st_b = xt_b + 2u * (unsigned)p.xt_bytes;           // [3] staging tiles
    const unsigned win_b = st_b + 3u * WT_STB;                        // [3 branches] filter windows
    const unsigned lds_end = win_b + 3u * WT_WIN;
    for (unsigned o = tid * 16; o < lds_end; o += MF_THREADS * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};
    wg_barrier();

    // The barriers of the two roles below pair up one to one: per plane ONE (the hand-over), and at a channel boundary of the range three more
    // (windows free -> cleared -> written), in the same order in both loops.
    if (wave == 3) {
        // =================== the plane server ===================
        v4i_t rs[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint64_t a = (uint64_t)p.in[DGRAD ? t : 0];
            rs[t][0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs[t][1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
            rs[t][2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs[t][3] = 0x00020000;
        }
        const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
        // Destination chunk g = 64 k + lane of an image with cd chunks per row takes source chunk (g / cd) * cs + g % cd; pad chunks and rows
        // beyond the image are skipped lanes (inactive lanes write nothing).  The per-lane source offsets do not depend on the plane: once.
        unsigned offh[WT_MAXDMA], offv[DGRAD ? WT_MAXDMA : 1];
#pragma unroll
        for (int k = 0; k < WT_MAXDMA; ++k) {
            const int g = k * 64 + lane;
            { const int row = g / p.cdh, cc = g - row * p.cdh; offh[k] = (k < p.ninh && cc < p.cs && row < p.H) ? (unsigned)(row * p.cs + cc) * 16u : WT_NONE; }
            if constexpr (DGRAD) { const int row = g / p.cdv, cc = g - row * p.cdv; offv[k] = (k < p.ninv && cc < p.cs && row < p.H) ? (unsigned)(row * p.cs + cc) * 16u : WT_NONE; }
        }
        // x^T: block (4 image rows kb, 16 image columns cb) by one ds_read_b64_tr_b16 of a 16-lane group (lane i16 supplies row 4 kb + i16 / 4, columns
        // 16 cb + 4 (i16 % 4) .. + 3 and receives column 16 cb + i16, rows 4 kb .. + 3) and one 8-byte write (x^T row = image column + 2 guard rows).
        // Group grp takes blocks grp, grp + 4, ..: the offsets of its blocks, once.
        const unsigned tr_sp = DGRAD ? (unsigned)p.cdv * 16u : pitch;
        unsigned tsrc[WT_MAXTR], tdst[WT_MAXTR];
        {
            const int grp = lane >> 4, i16 = lane & 15, total = (p.H >> 2) * p.trc;
#pragma unroll
            for (int j = 0; j < WT_MAXTR; ++j) {
                const int b = grp + 4 * j, kb = b / p.trc, cb = b - kb * p.trc, col = cb * 16 + i16;
                tsrc[j] = (unsigned)(kb * 4 + (i16 >> 2)) * tr_sp + (unsigned)(cb * 32 + (i16 & 3) * 8);
                tdst[j] = (b < total && col < p.W) ? (unsigned)((2 + col) * p.PT + kb * 4) * 2u : WT_NONE;
                if (b >= total) tsrc[j] = 0;
            }
        }
        auto plane_off = [&](int q) -> unsigned { const int c = q / p.N, n = q - c * p.N; return (unsigned)(((size_t)n * p.C + c) * HW * 2); };
        auto issue = [&](int it) {                                    // every input of plane q0 + it
            if (it >= iters) return;
            const unsigned src0 = plane_off(q0 + it), slot = (unsigned)(it % R) * (unsigned)p.slot_h;
#pragma unroll
            for (int k = 0; k < WT_MAXDMA; ++k) {
                if (k < p.ninh) {                                       // wave-uniform
                    const unsigned d = lds_base + h_b + slot + 2u * pitch + (unsigned)k * 1024u;
                    if (offh[k] != WT_NONE) lds_dma16(src0 + offh[k], rs[DGRAD ? 1 : 0], __builtin_amdgcn_readfirstlane(d));
                    if constexpr (DGRAD) if (offh[k] != WT_NONE) lds_dma16(src0 + offh[k], rs[2], __builtin_amdgcn_readfirstlane(d + (s_b - h_b)));
                }
                if constexpr (DGRAD) if (k < p.ninv && offv[k] != WT_NONE) lds_dma16(src0 + offv[k], rs[0], __builtin_amdgcn_readfirstlane(lds_base + v_b + (unsigned)k * 1024u));
            }
        };
        auto transpose = [&](int it) {
            if (it >= iters) return;
            const unsigned src_b = DGRAD ? v_b : h_b + (unsigned)(it % R) * (unsigned)p.slot_h + 2u * pitch;
            const unsigned dst_b = xt_b + (unsigned)(it & 1) * (unsigned)p.xt_bytes;
#pragma unroll
            for (int j0 = 0; j0 < WT_MAXTR; j0 += 6) {               // six reads in flight, then their writes
                s16x4 v[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) v[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + src_b + tsrc[j0 + j]));
#pragma unroll
                for (int j = 0; j < 6; ++j) if (tdst[j0 + j] != WT_NONE) *(s16x4*)(L + dst_b + tdst[j0 + j]) = v[j];
            }
        };
        const int dma_per_plane = DGRAD ? 2 * p.ninh + p.ninv : p.ninh;   // (every instruction has at least one active lane: it is issued)
        issue(0);
        if constexpr (!DGRAD) issue(1);
        int c_cur = -1, c = q0 / p.N, n = q0 - c * p.N;
        for (int it = 0; it < iters; ++it) {
            if (c != c_cur) {
                if (c_cur >= 0) { wg_barrier(); wg_barrier(); }
                wg_barrier();
                c_cur = c;
            }
            if (it == 0) { wait_vmcnt_dyn((!DGRAD && iters > 1) ? dma_per_plane : 0); transpose(0); }
            wg_barrier();                                             // plane `it` has landed and its x^T is complete; everyone is done with plane it - 1
            if constexpr (DGRAD) {
                issue(it + 1);                                        // into the slots plane it - 1 has left (the compact one was transposed before the barrier)
                wait_vmcnt<0>();
            } else {
                issue(it + 2);                                        // (ring of three: plane it + 1 has been on its way since the last iteration)
                wait_vmcnt_dyn(it + 2 < iters ? dma_per_plane : 0);  // loads retire in order: at most plane it + 2's are outstanding
            }
            transpose(it + 1);
            if (++n == p.N) { n = 0; ++c; }
        }
        return;
    }

    // =================== the compute waves (0..2: column block mt of all three branches) ===================
    const int mt = wave;
    // block ranges: tile t of a long axis with KS k-steps and a filter of KL taps meets blocks dd with 0 <= 2 t + dd - 2 < KS inside the band
    auto range = [](int t, int KS, int KL, int& lo, int& hi) {
        const int padL = KL / 2;
        const int dmax = (KL - 1 - padL + 31) >> 4, dmin = -((padL + 15) >> 4);
        lo = 2 - 2 * t; hi = KS + 1 - 2 * t;
        if (lo < dmin + 2) lo = dmin + 2;
        if (hi > dmax + 2) hi = dmax + 2;
        if (lo < 0) lo = 0;
        if (hi > WT_ND - 1) hi = WT_ND - 1;
    };
    int lo_h, hi_h, lo_s, hi_s;
    range(mt, p.KSc, p.K, lo_h, hi_h);
    range(mt, p.KSc, MF_TAPS, lo_s, hi_s);
    s16x8 fh[MF_TAPS][WT_ND], fv[MF_TAPS][WT_ND], fs[MF_TAPS][WT_NDS];
    char* const stg = L + st_b + (unsigned)wave * WT_STB;
    const int t192 = wave * 64 + lane;
    __amdgpu_buffer_rsrc_t ro[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) ro[t] = __builtin_amdgcn_make_buffer_rsrc(p.out[DGRAD ? 0 : t], 0, (int)p.tensor_bytes, 0x00020000);
    WtStore<T> pend;                                                  // the finished tile whose store is under way
    {
        f32x16 z0;
#pragma unroll
        for (int i = 0; i < 16; ++i) z0[i] = 0.f;
        pend.arm(z0, stg, ro[0], 0u, 0, mt, lane, p.H, p.W, false);
        pend.r0 = pend.r1 = u32x4{0u, 0u, 0u, 0u};
    }

    int c_cur = -1, c = q0 / p.N, n = q0 - c * p.N;
    for (int it = 0; it < iters; ++it) {
        if (c != c_cur) {
            // ---- a new channel: filter windows (two copies one element apart: every 8-element window is dword aligned), then this wave's fragments
            if (c_cur >= 0) {
                wg_barrier();                                         // (nobody reads the old windows any more)
                for (unsigned o = (unsigned)t192 * 16; o < 3u * WT_WIN; o += 192 * 16) *(u32x4*)(L + win_b + o) = u32x4{0u, 0u, 0u, 0u};
                wg_barrier();
            }
#pragma unroll
            for (int br = 0; br < 3; ++br) {
                const int kh = br == 0 ? p.K : MF_TAPS, kw = br == 1 ? p.K : MF_TAPS, ntap = kh * kw, KL = br == 2 ? MF_TAPS : p.K;
                uint16_t* win = (uint16_t*)(L + win_b + (unsigned)br * WT_WIN);
                for (int e = t192; e < ntap; e += 192) {
                    int r = br == 0 ? e % MF_TAPS : e / kw, t = br == 0 ? e / MF_TAPS : e - (e / kw) * kw;      // short tap r, long tap t
                    if (DGRAD) { r = MF_TAPS - 1 - r; t = KL - 1 - t; }       // the data gradient is the correlation with the filter rotated by 180 degrees
                    const uint16_t v = cvt_to_bits(p.w[br][(size_t)c * ntap + e], (T*)nullptr);
                    win[r * WT_LEN + WT_ZP + t] = v;
                    win[MF_TAPS * WT_LEN + r * WT_LEN + WT_ZP + t - 1] = v;
                }
            }
            wg_barrier();
            // lane (l31 -> o within the tile, lhi -> k half) of block d holds the 8-element window that starts at 16 d + 8 lhi - l31 + padL
            auto build = [&](auto& f, int nd, int d0, unsigned wb, int padL) {
#pragma unroll
                for (int dd = 0; dd < nd; ++dd) {
                    const int a = WT_ZP + 16 * (dd + d0 - 2) + lhi * 8 - l31 + padL;
                    const int par = a & 1;
                    const unsigned* src = (const unsigned*)(L + wb + par * MF_TAPS * WT_LEN * 2) + ((a - par) >> 1);
#pragma unroll
                    for (int r = 0; r < MF_TAPS; ++r) {
                        u32x4 d4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) d4[k] = src[r * (WT_LEN / 2) + k];
                        f[r][dd] = __builtin_bit_cast(s16x8, d4);
                    }
                }
            };
            build(fv, WT_ND, 0, win_b, p.padL);
            build(fh, WT_ND, 0, win_b + WT_WIN, p.padL);
            build(fs, WT_NDS, 1, win_b + 2u * WT_WIN, MF_TAPS / 2);
            c_cur = c;
        }
        wg_barrier();                                                 // plane `it` has landed and its x^T is complete
        {
            const unsigned slot = (unsigned)(it % R) * (unsigned)p.slot_h;
            const unsigned img_h = h_b + slot, img_s = DGRAD ? s_b + slot : img_h;
            const unsigned img_v = xt_b + (unsigned)(it & 1) * (unsigned)p.xt_bytes;
            const unsigned plane_b = (unsigned)(((size_t)n * p.C + c) * HW * 2);
            for (int s = 0; s < p.MTr; ++s) {
                // horizontal / 5 x 5: tap r, block dd = 16 bytes at row 32 s + l31 + r of the guarded image, columns 16 ks + 8 lhi ..  (ks = 2 mt - 2 + dd)
                const unsigned rp_h = (unsigned)(s * 32 + l31) * pitch + (unsigned)lhi * 16u + (unsigned)((2 * mt - 2) * 32);
                // vertical: x^T row 32 mt + l31 + r (image column + guard), image rows 16 ks + 8 lhi ..  (ks = 2 s - 2 + dd)
                const unsigned rp_v = img_v + (unsigned)(mt * 32 + l31) * xpitch + (unsigned)lhi * 16u + (unsigned)((2 * s - 2) * 32);
                int lo_v, hi_v;
                range(s, p.KSr, p.K, lo_v, hi_v);
                f32x16 z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                // every tile's store (`pend`) rides behind the MFMAs of the tile after it -- across regions and planes
                if constexpr (DGRAD) {                                // the three partial gradients of the tile in ONE accumulator, one rounding
                    f32x16 a = wt_long_tile<T, true, true>(fv, L, rp_v, xpitch, lo_v, hi_v, z, pend, l31, lhi, lane);
                    a = wt_long_tile<T, false, false>(fh, L, img_h + rp_h, pitch, lo_h, hi_h, a, pend, l31, lhi, lane);
                    a = wt_small_tile<T, false>(fs, L, img_s + rp_h, pitch, lo_s, hi_s, a, pend, l31, lhi, lane);
                    pend.arm(a, stg, ro[0], plane_b, s, mt, lane, p.H, p.W, true);
                } else {
                    const f32x16 av = wt_long_tile<T, true, true>(fv, L, rp_v, xpitch, lo_v, hi_v, z, pend, l31, lhi, lane);
                    pend.arm(av, stg, ro[0], plane_b, s, mt, lane, p.H, p.W, true);
                    const f32x16 ah = wt_long_tile<T, false, true>(fh, L, img_h + rp_h, pitch, lo_h, hi_h, z, pend, l31, lhi, lane);
                    pend.arm(ah, stg, ro[1], plane_b, s, mt, lane, p.H, p.W, true);
                    const f32x16 as = wt_small_tile<T, true>(fs, L, img_s + rp_h, pitch, lo_s, hi_s, z, pend, l31, lhi, lane);
                    pend.arm(as, stg, ro[2], plane_b, s, mt, lane, p.H, p.W, true);
                }
            }
        }
        if (++n == p.N) { n = 0; ++c; }
    }
#pragma unroll
    for (int k = 0; k < WT_STEPS; ++k) pend.step(k, l31, lhi, lane);  // the last tile's store
}

// ------------------------------------------------------------------------------------------------------------
static bool wide_tri_enabled() {               // SLAK_WIDE_TRI=0: maps beyond 64 keep the per-branch launches (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_WIDE_TRI"); return !(e && e[0] == '0'); }();
    return v;
}

static bool fill_wide_tri_params(WideTriParams& p, int N, int C, int H, int W, int K, bool dgrad, int wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.padL = K / 2;
    if (N <= 0 || C <= 0 || K <= MF_TAPS || !(K & 1) || K > 63) return false;
    if (H <= 64 || H > 96 || W <= 64 || W > 96 || (H % 16) || (W % 16)) return false;      // three column blocks = three compute waves; every k-step inside the plane
    if (K * MF_TAPS > 2 * WT_LEN) return false;
    p.MTr = (H + 31) / 32; p.KSr = H / 16; p.KSc = W / 16;
    p.cs = W / 8;
    p.cdh = p.cs | 1;                                                 // row-per-lane b128 reads: odd chunk pitch
    p.ninh = (H * p.cdh + 63) / 64;
    p.slot_h = (p.MTr * 32 + 4) * p.cdh * 16;                         // the image behind 2 guard rows; rows up to 32 MTr + 3 are read (lanes beyond H: results dropped)
    p.cdv = p.cs; while (p.cdv % 16 != 4 && p.cdv % 16 != 12) ++p.cdv;   // transposing reads: pitch = +-64 bytes mod 256
    p.ninv = (H * p.cdv + 63) / 64;
    p.slot_v = dgrad ? (int)align_up((size_t)p.ninv * 1024, 16) : 0;
    if (dgrad && p.slot_v < H * p.cdv * 16) return false;
    p.PT = p.MTr * 32 + 8;
    p.xt_bytes = (96 + 4) * p.PT * 2;                                 // x^T rows: image columns behind 2 guard rows; rows up to 32 * 3 + 3 are read
    p.trc = (W + 15) / 16;
    if (p.ninh > WT_MAXDMA || p.ninv > WT_MAXDMA || ((H >> 2) * p.trc + 3) / 4 > WT_MAXTR) return false;
    if ((H + 4) * p.cdh * 16 > p.slot_h) return false;
    const long long P = (long long)N * C;
    if (P >= 0x40000000ll) return false;
    if (wgs < 1) wgs = 1;
    p.planes = (int)P;
    p.per = (int)((P + wgs - 1) / wgs);
    p.tensor_bytes = (unsigned)((size_t)N * C * H * W * 2);
    return (size_t)N * C * H * W * 2 < 0x7fffffffull;
}

static size_t wide_tri_lds_bytes(const WideTriParams& p, bool dgrad) {
    return (size_t)(dgrad ? 4 : 3) * p.slot_h + (dgrad ? (size_t)p.slot_v : 0) + (size_t)2 * p.xt_bytes + 3 * WT_STB + 3 * WT_WIN;
}

bool dwconv_mfma_wide_tri_supported(int N, int C, int H, int W, int K, int dtype, bool dgrad) {
    if (!wide_tri_enabled() || (dtype != SLAK_BF16 && dtype != SLAK_F16)) return false;
    WideTriParams p;
    return fill_wide_tri_params(p, N, C, H, W, K, dgrad, mfma_cu_count()) && wide_tri_lds_bytes(p, dgrad) <= 160 * 1024;
}

template <typename T, bool DGRAD>
static int launch_wide_tri_t(WideTriParams& p, hipStream_t st) {
    auto k = dwconv_mfma_wide_tri_kernel<T, DGRAD>;
    const size_t lds = wide_tri_lds_bytes(p, DGRAD);
    if (!slak_set_max_lds((const void*)k, lds)) return SLAK_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)((p.planes + p.per - 1) / p.per);
    hipLaunchKernelGGL(k, dim3(grid), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_wide_tri(bool dgrad, const void* const* in, void* const* out, const float* const* w, int dtype,
                                int N, int C, int H, int W, int K, hipStream_t st) {
    if (!dwconv_mfma_wide_tri_supported(N, C, H, W, K, dtype, dgrad)) return SLAK_ERR_UNSUPPORTED;
    WideTriParams p;
    fill_wide_tri_params(p, N, C, H, W, K, dgrad, mfma_cu_count());  // one four-wave workgroup per CU (512 registers per wave)
    for (int b = 0; b < 3; ++b) { p.in[b] = in[b]; p.out[b] = out[b]; p.w[b] = w[b]; }
    const bool bf = dtype == SLAK_BF16;
    if (dgrad) return bf ? launch_wide_tri_t<bf16_t, true>(p, st) : launch_wide_tri_t<f16_t, true>(p, st);
    return bf ? launch_wide_tri_t<bf16_t, false>(p, st) : launch_wide_tri_t<f16_t, false>(p, st);
}

}  // namespace slak
