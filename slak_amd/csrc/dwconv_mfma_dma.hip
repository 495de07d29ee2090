// slak_amd/csrc/dwconv_mfma_dma.hip -- MFMA depthwise-conv forward / data-grad for the large maps (56x56, 28x28 class;
// plane bytes a multiple of 16): LDS-DMA input ring + single-accumulator Toeplitz GEMM.
//
//   long axis t (extent Wt, KL taps)    short axis l (extent Wl, 5 taps)      per plane, per channel:
//   Y[o, u] = sum_r sum_i T_r[o, i] * X[i, u + r - 2]          A = T_r (dense 1-D Toeplitz), B = rows u-2..u+2 of the plane image:
// the five short taps are five B fragments read at five row addresses and accumulated into ONE 32x32 accumulator, so
// there is no cross-lane shift at all and only 16 accumulator registers (the register-staged kernel of dwconv_mfma.hip
// keeps one accumulator per tap and combines them with DPP lane shifts, ~11 cycles each on gfx950: tools/dpp_probe.hip).
//   * horizontal kernels (5xK): the contraction runs along W, contiguous in the DMA image: B = ds_read_b128 of image rows.
//   * vertical kernels (Kx5): the contraction runs along H.  Each group is transposed once LDS->LDS (ds_read_b64_tr_b16 +
//     ds_write_b64, 13 instruction pairs per 56x56 plane) into a double-buffered x^T and runs the SAME core on x^T with the
//     MFMA operands swapped (D^T = X^T-tile x T^T), so a lane again holds 4 consecutive ow of one output row.
// Input: planes go HBM -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip), several groups deep, so every
// workgroup keeps several planes in flight.  LDS-DMA writes are lane-linear (destination = M0 base + instruction offset +
// lane*16: tools/dma_probe.hip, tools/dma_offset_probe.hip), so a plane lands as a plain row-major image of pitch W.
// Zero padding: along the short axis the ring slot (horizontal) / x^T (vertical) carries two all-zero guard rows on either
// side of every plane, written once, never touched by the DMA -- taps that fall outside the plane read zeros, no select;
// along the long axis the last two k-steps redirect pieces beyond the row end to a zero row (one select per fragment),
// which also keeps NaN/Inf of a neighbouring row from leaking in.
// Synchronisation: ONE barrier per group.  Group g is issued entirely by wave g%4 (inline asm, invisible to hipcc's waitcnt
// pass); the same wave waits for it with `s_waitcnt vmcnt(0)` just before the barrier of the iteration that consumes it --
// exact, because between issuing group g and waiting for it that wave issues no other DMA (its next one is g+4), only
// output stores that are an iteration old.  After the barrier every wave copies the previous group's results from the LDS
// out-buffer to HBM (16 bytes per lane, fully coalesced) and then computes.  Barriers are raw s_barrier + lgkmcnt(0), never
// __syncthreads() (which would drain the DMA queue).
// Toeplitz fragments: T_r[o, i] = w_r[i - o + padL] depends on i - o only, so a lane's 8 consecutive k are an 8-element
// WINDOW of the zero-padded filter row.  The filter is staged once per workgroup in LDS as bf16/fp16 in two copies (one
// shifted by an element, so every window starts dword-aligned in one of them) and each fragment is 4 ds_read_b32: no pack
// kernel, no 40 KB fragment fetch per workgroup.
// Instruction count is what bounds this kernel once the MFMA chain is software-pipelined (SQ counters: the three waves of a
// SIMD keep its issue port ~95 % busy), so everything that does not change between groups -- fragment row offsets, epilogue
// offsets and predicates, copy-out and transpose maps -- is computed once per workgroup and kept in registers.
#include "mfma_common.h"
#include <type_traits>

namespace slak {

constexpr int DMA_NB = 4;               // ring depth (groups), horizontal kernels
constexpr int DMA_NBV = 3;              // vertical kernels: 3 slots + double-buffered x^T
constexpr int WIN_ZP = 64;              // zeros in front of a filter row (window starts never go negative)
constexpr int WIN_LEN = 192;            // elements per padded filter row
constexpr int ZROW_LEN = 128;           // elements of the all-zero row that out-of-range k pieces point at

unsigned long long* g_dma_dbg = nullptr;   // dev hook: slak_debug_set_phase_buffer() (only read by -DSLAK_DMA_DEBUG builds)

struct MfmaDmaParams {
    const void* x; const float* w; void* y;
    int N, C, H, W, kh, kw, flip;
    int acc;               // y += result instead of y = result (the data gradient of the 2nd / 3rd branch of a block: autograd's add folded in)
    int Wt, Wl, KL, padL;
    int G;                 // planes per group (iteration)
    int tpp;               // 32-lane tiles per plane
    int ntiles;            // G * tpp  (<= waves along l: at most one tile per wave and group)
    int chunks_pp;         // 16-byte chunks per plane (HW/8)
    int plane_lds;         // LDS elements from one plane of a ring slot to the next (horizontal: HW + 2W guard rows; vertical: HW)
    int slot_elems;        // LDS elements per ring slot
    int PT;                // pitch of the transposed image (vertical kernels)
    int xt_rows;           // rows of one transposed plane image incl. 2+2 guard rows (vertical kernels)
    int planes_per_wg, slices;
    unsigned m_cpp;        // magic multiplier: n / chunks_pp == (n * m_cpp) >> 22
    int tr_pp, tr_cbs;     // transpose blocks per plane, per 4-row band
    unsigned tensor_bytes;
    unsigned long long* dbg;
    float* stats;          // forward only, or NULL: [slices * 4][C][2] = per (slice, wave, channel) sum y, sum y^2 of the stored outputs
};                         // (batch statistics of the branch's BatchNorm, models/SLaK.py:92-95, gathered in the copy-out: saves a read pass)

constexpr int DMA_NCO = 2;              // 16-byte copy-out chunks per thread per group (upper bound)
constexpr int DMA_NTR = 4;              // transpose blocks (4 rows x 16 cols) per lane group per group of planes (upper bound)
constexpr int DMA_WCH = 5;              // filter elements staged per lane of the staging wave (upper bound, 64 lanes)

#ifdef SLAK_DMA_DEBUG
#define DBG_STAMP(k) do { if (p.dbg && tid == 0) p.dbg[64 + blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DBG_STAMP(k) do { } while (0)
#endif

// MT: 32-row tiles along the Toeplitz axis (wave w owns tile w % MT); KS: 16-deep k-steps; VERT: long axis = H;
// BAND: the filter is much shorter than the map (5x5 branch) -> skip all-zero Toeplitz blocks (wave-uniform branches).
// R16: image rows are 16-byte aligned (W % 8 == 0); otherwise B fragments are read as two 8-byte halves.
template <typename T, int MT, int KS, bool VERT, bool BAND, bool R16>
__global__ __launch_bounds__(MF_THREADS, 3) void dwconv_mfma_dma_kernel(const MfmaDmaParams p) {
    constexpr int NG = MF_TAPS;
    constexpr int NB = VERT ? DMA_NBV : DMA_NB;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;                                      // everything below is a BYTE offset into the LDS block
    const int HW = p.H * p.W;
    const unsigned slot_b = (unsigned)p.slot_elems * 2;
    // (lanes beyond the short-axis extent read a few rows past their image; keeping the images in front means such reads stay
    // inside the block -- their results are never stored)
    const unsigned xt_buf_b = VERT ? (unsigned)(p.G * p.xt_rows * p.PT) * 2 : 0u;
    const unsigned out_buf_b = (unsigned)(p.G * HW) * 2;
    const unsigned ring_b = 0;                                       // NB slots (+ 128 bytes slack behind the last)
    const unsigned xt_b = ring_b + NB * slot_b + 128;                // vertical only: 2 x [G][xt_rows][PT]
    const unsigned lout_b = xt_b + 2 * xt_buf_b;                     // 2 x [G][HW]
    const unsigned win_b = lout_b;                                   // [2 copies][5 taps][WIN_LEN]: prologue only, aliases the out-buffers
    const unsigned win_bytes = 2 * MF_TAPS * WIN_LEN * 2;
    const unsigned zrow_b = lout_b + (2 * out_buf_b > win_bytes ? 2 * out_buf_b : win_bytes);   // ZROW_LEN zeros

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = wave % MT, wl = wave / MT;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;

    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    if (n_begin >= n_end) return;
    const int iters = (n_end - n_begin + p.G - 1) / p.G;
    DBG_STAMP(0);

    // ---- DMA: one wave issues a whole group; a plane is a contiguous run of 16-byte chunks in HBM and in its slot, and the
    // 12-bit instruction offset advances BOTH addresses, so one M0 setup serves four loads ---------------------------------
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)p.x;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes);
        rsrc[3] = 0x00020000;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    const unsigned plane_b = (unsigned)p.plane_lds * 2;              // LDS bytes from plane to plane within a slot
    const unsigned first_plane_b = VERT ? 0u : (unsigned)(2 * p.W) * 2;   // horizontal: two guard rows in front of every plane
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;              // HBM bytes from image n to image n+1 of this channel
    const int cpp_full = p.chunks_pp >> 6, cpp_rem = p.chunks_pp & 63;
    const unsigned lane16 = lane * 16;
    auto issue_group = [&](int g) {
        if (g >= iters || wave != (g & 3)) return;                   // wave-uniform
        const int n0 = n_begin + g * p.G;
        unsigned voff = (unsigned)(((size_t)n0 * p.C + c) * HW * 2) + lane16;
        unsigned m0v = lds_base + ring_b + (unsigned)(g % NB) * slot_b + first_plane_b;
        for (int j = 0; j < p.G; ++j) {
            unsigned v = voff, m = m0v;
            int f = cpp_full;
            for (; f >= 4; f -= 4) { lds_dma_run<4, 0>(v, rsrc, m); v += 4096; m += 4096; }
            if (f == 3) lds_dma_run<3, 0>(v, rsrc, m); else if (f == 2) lds_dma_run<2, 0>(v, rsrc, m); else if (f == 1) lds_dma_run<1, 0>(v, rsrc, m);
            if (lane < cpp_rem) {
                if (f == 0) lds_dma_run<1, 0>(v, rsrc, m); else if (f == 1) lds_dma_run<1, 1024>(v, rsrc, m);
                else if (f == 2) lds_dma_run<1, 2048>(v, rsrc, m); else lds_dma_run<1, 3072>(v, rsrc, m);
            }
            voff += gplane_b; m0v += plane_b;
        }
    };

    // ---- prologue: first groups in flight, zero areas, filter windows, fragments -------------------------------------
    for (int g = 0; g < 3; ++g) issue_group(g);                      // waves 0..2; wave 3 stages the filter meanwhile
    const int ntap = p.kh * p.kw;
    float wreg[DMA_WCH];
    if (wave == 3) {
#pragma unroll
        for (int k = 0; k < DMA_WCH; ++k) { const int e = lane + 64 * k; wreg[k] = e < ntap ? p.w[(size_t)c * ntap + e] : 0.f; }
    }
    {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (unsigned o = tid * 16; o < win_bytes; o += MF_THREADS * 16) *(u32x4*)(L + win_b + o) = z4;
        if (tid < ZROW_LEN * 2 / 16) *(u32x4*)(L + zrow_b + tid * 16) = z4;
        if constexpr (VERT) {                                         // x^T: guard rows and pad columns stay zero
            for (unsigned o = tid * 16; o < 2 * xt_buf_b; o += MF_THREADS * 16) *(u32x4*)(L + xt_b + o) = z4;
        } else {                                                      // ring: 2 guard rows in front of every plane + 2 behind the last
            const int ngr = NB * (p.G + 1);
            for (int q = wave; q < ngr; q += MF_WAVES) {
                const int s = q / (p.G + 1), jj = q - s * (p.G + 1);
                const unsigned gb = ring_b + s * slot_b + jj * plane_b;
                for (int o = lane; o < p.W; o += 64) *(unsigned*)(L + gb + o * 4) = 0u;      // 2W elements = W dwords
            }
        }
    }
    wg_barrier();
    if (wave == 3) {
#pragma unroll
        for (int k = 0; k < DMA_WCH; ++k) {
            const int e = lane + 64 * k;
            if (e < ntap) {
                int r = VERT ? e % MF_TAPS : e / p.kw, t = VERT ? e / MF_TAPS : e - (e / p.kw) * p.kw;      // short tap r, long tap t
                if (p.flip) { r = MF_TAPS - 1 - r; t = p.KL - 1 - t; }
                const uint16_t v = cvt_to_bits(wreg[k], (T*)nullptr);
                uint16_t* win = (uint16_t*)(L + win_b);
                win[r * WIN_LEN + WIN_ZP + t] = v;                                         // copy 0
                win[MF_TAPS * WIN_LEN + r * WIN_LEN + WIN_ZP + t - 1] = v;                 // copy 1 = copy 0 shifted by one element
            }
        }
    }
    wg_barrier();
    DBG_STAMP(5);
    s16x8 afrag[NG][KS];
    bool ks_active[KS];
    const int kfull = p.Wt >> 4;                                     // k-steps below this lie entirely inside the plane
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int i_lo = ks * 16, i_hi = ks * 16 + 15, o_lo = mt * 32, o_hi = mt * 32 + 31;
        ks_active[ks] = (i_lo < p.Wt) && (o_lo < p.Wt) && (i_lo - o_hi <= p.KL - 1 - p.padL) && (o_lo - i_hi <= p.padL);
        const int a = WIN_ZP + ks * 16 + lhi * 8 - (mt * 32 + l31) + p.padL;              // window start (element index), >= 1
        const int par = a & 1;
        const unsigned* src = (const unsigned*)(L + win_b + par * MF_TAPS * WIN_LEN * 2) + ((a - par) >> 1);
#pragma unroll
        for (int r = 0; r < NG; ++r) {
            u32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = src[r * (WIN_LEN / 2) + k];
                if (ks >= kfull && ks * 16 + lhi * 8 + 2 * k >= p.Wt) d[k] = 0u;        // i >= Wt: no such input
            }
            afrag[r][ks] = __builtin_bit_cast(s16x8, d);
        }
    }
    DBG_STAMP(6);

    // ---- per-thread constants of the loop (nothing below depends on the group) -----------------------------------------
    // this wave's tile: plane j_t of the group, 32 positions `sub_t` along the short axis
    const bool has_tile = wl < p.ntiles;
    const int j_t = has_tile ? wl / p.tpp : 0, sub_t = has_tile ? wl - j_t * p.tpp : 0;
    const int pos = sub_t * 32 + l31;                                 // lane -> position along the short (lane) axis
    // B fragment of tap r, k-step ks: 16 bytes at brel[r] + 32*ks from the slot / x^T base (row pos + r of the guarded image)
    unsigned brel[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) {
        if constexpr (VERT) brel[r] = (unsigned)((j_t * p.xt_rows + pos + r) * p.PT) * 2 + lhi * 16;
        else brel[r] = (unsigned)j_t * plane_b + (unsigned)((pos + r) * p.W) * 2 + lhi * 16;
    }
    // long-axis pieces of the last two k-steps that lie beyond the row end read the zero row instead (horizontal only)
    bool kv0[2], kv1[2];
    unsigned zadj[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ks = KS - 2 + kk;
        kv0[kk] = ks * 16 + lhi * 8 < p.Wt; kv1[kk] = ks * 16 + lhi * 8 + 4 < p.Wt;
        zadj[kk] = zrow_b + lhi * 16 - ks * 32;                       // so that the instruction offset 32*ks lands in the zero row
    }
    // epilogue: lane = output row (horizontal: short-axis position; vertical: long-axis row of this wave's 32-row tile),
    // register quad q = 4 consecutive output columns -> one 8-byte LDS store each
    unsigned orel; bool qok[4];
    {
        const int orow = VERT ? mt * 32 + l31 : pos, ocol0 = (VERT ? sub_t * 32 : mt * 32) + 4 * lhi;
        const int nrow = VERT ? p.Wt : p.Wl, ncol = VERT ? p.Wl : p.Wt;
        orel = (unsigned)(j_t * HW + orow * p.W + ocol0) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) qok[q] = has_tile && orow < nrow && ocol0 + 8 * q < ncol;
    }
    // copy-out: 16-byte chunk idx of the out-buffer -> same chunk of the group's planes in HBM
    const int TC = p.G * p.chunks_pp;
    unsigned co_g[DMA_NCO]; int co_j[DMA_NCO];
#pragma unroll
    for (int k = 0; k < DMA_NCO; ++k) {
        const unsigned idx = tid + k * MF_THREADS;
        const unsigned j = p.m_cpp ? (__umul24(idx, p.m_cpp) >> 22) : 0u, rem = idx - j * p.chunks_pp;
        co_j[k] = (int)idx < TC ? (int)j : 0x3fffffff;
        co_g[k] = j * gplane_b + rem * 16;
    }
    // vertical: transpose map.  Block b of a group = (plane j, 4 image rows kb, 16 image columns cb); the 16 lanes of a group
    // read it with one ds_read_b64_tr_b16 (lane i16 supplies row kb*4 + i16/4, columns cb*16 + 4*(i16%4) and receives column
    // cb*16 + i16, rows kb*4..+3) and write 8 bytes of x^T (row = image column + 2 guard rows).
    unsigned tr_map[DMA_NTR];                                        // (source byte offset in the slot) | (x^T byte offset << 16)
    if constexpr (VERT) {
        const int grp = lane >> 4, i16 = lane & 15;
        const int total = p.G * p.tr_pp;
#pragma unroll
        for (int k = 0; k < DMA_NTR; ++k) {
            const int b = (k * MF_WAVES + wave) * 4 + grp;
            const bool ok = b < total;                                // uniform per 16-lane group
            const int j = ok ? b / p.tr_pp : 0, rem = ok ? b - j * p.tr_pp : 0;
            const int kb = rem / p.tr_cbs, cb = rem - kb * p.tr_cbs;
            const unsigned src = (unsigned)(j * HW + (kb * 4 + (i16 >> 2)) * p.W + cb * 16 + (i16 & 3) * 4) * 2;
            const unsigned dst = (cb * 16 + i16 < p.W) ? (unsigned)((j * p.xt_rows + 2 + cb * 16 + i16) * p.PT + kb * 4) * 2 : 0xffffu;
            tr_map[k] = ok ? (src | (dst << 16)) : 0xffffffffu;
        }
    }
    auto transpose_group = [&](int g) {                              // ring slot of group g -> x^T buffer g&1
        const unsigned sb = ring_b + (unsigned)(g % NB) * slot_b, db = xt_b + (g & 1) * xt_buf_b;
#pragma unroll
        for (int k = 0; k < DMA_NTR; ++k) {
            if (tr_map[k] != 0xffffffffu) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + sb + (tr_map[k] & 0xffffu)));
                if ((tr_map[k] >> 16) != 0xffffu) *(s16x4*)(L + db + (tr_map[k] >> 16)) = v;
            }
        }
    };

#ifdef SLAK_DMA_DEBUG
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = p.dbg != nullptr && blockIdx.x == 0;
#define PH_T0() unsigned long long t__ = prof ? __builtin_readcyclecounter() : 0
#define PH_ADD(k) do { if (prof) { unsigned long long n__ = __builtin_readcyclecounter(); tph[k] += n__ - t__; t__ = n__; } } while (0)
#else
#define PH_T0() do { } while (0)
#define PH_ADD(k) do { } while (0)
#endif
    if constexpr (VERT) {                                             // group 0 has to be transposed before the loop
        if (wave == 0) wait_vmcnt<0>();
        wg_barrier();
        transpose_group(0);
    }
    DBG_STAMP(1);
#ifdef SLAK_DMA_DEBUG
    const unsigned long long cyc0 = __builtin_readcyclecounter();
#endif
    char* yg = (char*)p.y + ((size_t)n_begin * p.C + c) * HW * 2;   // HBM address of the current group's first plane
    u32x4 yold[DMA_NCO];
#pragma unroll
    for (int k = 0; k < DMA_NCO; ++k) yold[k] = u32x4{0u, 0u, 0u, 0u};
    float bs1 = 0.f, bs2 = 0.f;                                       // p.stats: sums over the chunks this thread copies out
    auto stat8 = [&](const u32x4& v) {                               // v_dot2c_f32_bf16 with a ZERO addend (its addend is aligned with truncation), adds kept apart
        if constexpr (std::is_same<T, bf16_t>::value) {
            const bf16x2_t one = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            // (elements copied to scalars first: bit-casting v[i] directly made hipcc reuse element 0's result for all four)
            const unsigned d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
            const bf16x2_t x0 = __builtin_bit_cast(bf16x2_t, d0), x1 = __builtin_bit_cast(bf16x2_t, d1), x2 = __builtin_bit_cast(bf16x2_t, d2), x3 = __builtin_bit_cast(bf16x2_t, d3);
            float a0 = __builtin_amdgcn_fdot2_f32_bf16(x0, one, 0.f, false), a1 = __builtin_amdgcn_fdot2_f32_bf16(x1, one, 0.f, false);
            float a2 = __builtin_amdgcn_fdot2_f32_bf16(x2, one, 0.f, false), a3 = __builtin_amdgcn_fdot2_f32_bf16(x3, one, 0.f, false);
            float q0 = __builtin_amdgcn_fdot2_f32_bf16(x0, x0, 0.f, false), q1 = __builtin_amdgcn_fdot2_f32_bf16(x1, x1, 0.f, false);
            float q2 = __builtin_amdgcn_fdot2_f32_bf16(x2, x2, 0.f, false), q3 = __builtin_amdgcn_fdot2_f32_bf16(x3, x3, 0.f, false);
            asm("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            bs1 += (a0 + a1) + (a2 + a3); bs2 += (q0 + q1) + (q2 + q3);
        }
    };
    int n0 = n_begin;
    for (int it = 0; it < iters; ++it) {
        PH_T0();
        // the group this iteration needs in LDS: `it` (horizontal) / `it+1` for the transpose (vertical); its issuing wave waits
        const int need = VERT ? it + 1 : it;
        if (need < iters && wave == (need & 3)) wait_vmcnt<0>();
        PH_ADD(0);
        wg_barrier();                        // group `need` landed; out-buffer it-1 and (vertical) x^T `it` complete; a ring slot is free
        PH_ADD(1);
        issue_group(VERT ? it + NB : it + NB - 1);
        const unsigned ob_prev = lout_b + ((it + 1) & 1) * out_buf_b, ob_cur = lout_b + (it & 1) * out_buf_b;
        if (it > 0) {                                                 // previous group's results: LDS out-buffer -> HBM, 16 bytes per lane
            char* yp = yg - (size_t)p.G * gplane_b;
#pragma unroll
            for (int k = 0; k < DMA_NCO; ++k)
                if (n0 - p.G + co_j[k] < n_end) {
                    u32x4 v = *(const u32x4*)(L + ob_prev + (tid + k * MF_THREADS) * 16);
                    if (p.acc) v = add_packed<T>(v, yold[k]);
                    if (p.stats) stat8(v);
                    *(u32x4*)(yp + co_g[k]) = v;
                }
        }
        if (p.acc) {                                                  // what y holds for THIS group: fetched now, added one iteration later
#pragma unroll
            for (int k = 0; k < DMA_NCO; ++k)
                if (n0 + co_j[k] < n_end) yold[k] = *(const u32x4*)(yg + co_g[k]);
        }
        PH_ADD(5);
        unsigned img_b;
        if constexpr (VERT) {
            if (it + 1 < iters) transpose_group(it + 1);
            img_b = xt_b + (it & 1) * xt_buf_b;
        } else {
            img_b = ring_b + (unsigned)(it % NB) * slot_b;
        }
        PH_ADD(4);
        if (has_tile) {
            unsigned rp[MF_TAPS];
#pragma unroll
            for (int r = 0; r < MF_TAPS; ++r) rp[r] = img_b + brel[r];
            auto load_b = [&](int r, int ks) -> s16x8 {
                u32x4 b;
                if constexpr (VERT) b = *(const u32x4*)(L + rp[r] + ks * 32);              // x^T pads are zero
                else if constexpr (R16) {
                    unsigned q = rp[r];
                    if (ks >= KS - 2) q = kv0[ks - (KS - 2)] ? q : zadj[ks - (KS - 2)];
                    b = *(const u32x4*)(L + q + ks * 32);
                } else {                                                                  // W % 8 == 4: rows are 8-byte aligned
                    unsigned q0 = rp[r], q1 = rp[r];
                    if (ks >= KS - 2) { q0 = kv0[ks - (KS - 2)] ? q0 : zadj[ks - (KS - 2)]; q1 = kv1[ks - (KS - 2)] ? q1 : zadj[ks - (KS - 2)]; }
                    const u32x2 lo = *(const u32x2*)(L + q0 + ks * 32), hi = *(const u32x2*)(L + q1 + ks * 32 + 8);
                    b = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
                return __builtin_bit_cast(s16x8, b);
            };
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            if constexpr (!BAND) {
                // software pipeline pinned with sched_barrier: hipcc otherwise sinks every ds_read next to its MFMA
                // (ds_read; s_waitcnt lgkmcnt(0); v_mfma -- the LDS latency 20 times per tile).  The fragment of tap r for the
                // next k-step is fetched right after this k-step's MFMA of tap r has issued, into the same registers.
                s16x8 b[MF_TAPS];
#pragma unroll
                for (int r = 0; r < MF_TAPS; ++r) b[r] = load_b(r, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int r = 0; r < MF_TAPS; ++r) {
                        // vertical: operands swapped (D^T = X^T-tile x T^T) so that a lane holds 4 consecutive ow of one output row
                        acc = VERT ? mfma32<T>(b[r], afrag[r][ks], acc) : mfma32<T>(afrag[r][ks], b[r], acc);
                        if (ks + 1 < KS) b[r] = load_b(r, ks + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (!ks_active[ks]) continue;
                    s16x8 b[MF_TAPS];
#pragma unroll
                    for (int r = 0; r < MF_TAPS; ++r) b[r] = load_b(r, ks);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < MF_TAPS; ++r) acc = VERT ? mfma32<T>(b[r], afrag[r][ks], acc) : mfma32<T>(afrag[r][ks], b[r], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            PH_ADD(2);
            if (n0 + j_t < n_end) {
                char* op = L + ob_cur + orel;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (qok[q]) {
                        u32x2 v;
                        v[0] = pack2<T>(acc[4 * q + 0], acc[4 * q + 1]);
                        v[1] = pack2<T>(acc[4 * q + 2], acc[4 * q + 3]);
                        *(u32x2*)(op + 16 * q) = v;
                    }
                }
            }
            PH_ADD(3);
        }
        yg += (size_t)p.G * gplane_b; n0 += p.G;
    }
    wg_barrier();
    {                                                                 // the last group's results
        char* yp = yg - (size_t)p.G * gplane_b;
        const unsigned ob_last = lout_b + ((iters - 1) & 1) * out_buf_b;
#pragma unroll
        for (int k = 0; k < DMA_NCO; ++k)
            if (n0 - p.G + co_j[k] < n_end) {
                u32x4 v = *(const u32x4*)(L + ob_last + (tid + k * MF_THREADS) * 16);
                if (p.acc) v = add_packed<T>(v, yold[k]);
                if (p.stats) stat8(v);
                *(u32x4*)(yp + co_g[k]) = v;
            }
    }
    if (p.stats) {                                                    // one partial row per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { bs1 += __shfl_xor(bs1, o, 64); bs2 += __shfl_xor(bs2, o, 64); }
        if (lane == 0) { float* r = p.stats + (((size_t)slice * MF_WAVES + wave) * p.C + c) * 2; r[0] = bs1; r[1] = bs2; }
    }
#ifdef SLAK_DMA_DEBUG
    if (p.dbg && tid == 0) { p.dbg[64 + blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[64 + blockIdx.x * 8 + 7] = __builtin_readcyclecounter() - cyc0; }
    if (prof && lane == 0) { for (int k = 0; k < 6; ++k) p.dbg[wave * 8 + k] = tph[k]; p.dbg[wave * 8 + 6] = (unsigned long long)iters; }
#endif
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_dma_params(MfmaDmaParams& p, const ConvDims& d, bool vert, int MT, int KS, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    const int HW = d.H * d.W;
    if (HW % 8 || d.W % 4 || d.H % 4) return false;
    if (p.KL > 63 || d.kh * d.kw > DMA_WCH * 64) return false;        // window layout / staging wave
    if (p.Wt <= 16 * (KS - 2)) return false;                          // only the last two k-steps may reach past the plane edge
    p.tpp = (p.Wl + 31) / 32;
    const int WLW = MF_WAVES / MT;
    if (p.tpp > WLW) return false;                                    // at most one tile per wave and group
    p.G = WLW / p.tpp;
    if (p.G > d.N) p.G = d.N;
    p.ntiles = p.G * p.tpp;
    p.chunks_pp = HW / 8;
    p.plane_lds = vert ? HW : HW + 2 * d.W;
    p.slot_elems = vert ? p.G * HW : p.G * (HW + 2 * d.W) + 2 * d.W;
    p.PT = KS * 16 + 8;
    p.xt_rows = d.W + 4;
    const int TC = p.G * p.chunks_pp;
    if (TC > DMA_NCO * MF_THREADS || TC >= 1024 || p.chunks_pp >= 1024) return false;
    p.tr_cbs = (d.W + 15) / 16; p.tr_pp = (d.H / 4) * p.tr_cbs;
    if (vert && p.G * p.tr_pp > DMA_NTR * MF_WAVES * 4) return false;
    if (vert && ((size_t)p.G * p.xt_rows * p.PT * 2 >= 65535 || (size_t)p.G * HW * 2 >= 65535)) return false;   // packed 16-bit transpose map
    int slices = resident_wgs / d.C; if (slices < 1) slices = 1;      // one resident round: never more workgroups than fit at once
    int per = (d.N + slices - 1) / slices; per = (per + p.G - 1) / p.G * p.G; if (per < p.G) per = p.G;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.m_cpp = p.G <= 1 ? 0u : (unsigned)(((1u << 22) + p.chunks_pp - 1) / p.chunks_pp);   // n / cpp == (n * m) >> 22 for n, cpp < 1024
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * HW * 2);
    return true;
}

static size_t dma_lds_bytes(const MfmaDmaParams& p, bool vert) {
    const int nb = vert ? DMA_NBV : DMA_NB;
    const size_t out2 = (size_t)2 * p.G * p.H * p.W * 2, win = (size_t)2 * MF_TAPS * WIN_LEN * 2;
    return (size_t)nb * p.slot_elems * 2 + 128 + (out2 > win ? out2 : win) + (size_t)ZROW_LEN * 2 +
           (vert ? (size_t)2 * p.G * p.xt_rows * p.PT * 2 : 0) + 16;
}

static int dma_class(const ConvDims& d, bool vert) {              // 2: MT=2/KS=4, 1: MT=1/KS=2, 0: not covered
    const int Wt = vert ? d.H : d.W;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return 0;
    if (Wt > 64 || Wt <= 16) return 0;
    return Wt > 32 ? 2 : 1;
}

bool dwconv_mfma_dma_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt) {
    if (x_dt != y_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16) || w_dt != SLAK_F32) return false;
    const bool vert = d.kh > d.kw;
    const int cls = dma_class(d, vert);
    if (!cls) return false;
    MfmaDmaParams p;
    if (!fill_dma_params(p, d, vert, cls == 2 ? 2 : 1, cls == 2 ? 4 : 2, 512)) return false;
    return dma_lds_bytes(p, vert) <= 64 * 1024;
}

template <typename K>
static int resident_workgroups(K kernel, size_t lds) {          // workgroups the chip holds at once for this kernel
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, MF_THREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    return per_cu * mfma_cu_count();
}

template <typename T, int MT, int KS, bool VERT, bool BAND, bool R16>
static int launch_dma_tv(MfmaDmaParams& p, const ConvDims& d, hipStream_t st) {
    auto k = dwconv_mfma_dma_kernel<T, MT, KS, VERT, BAND, R16>;
    fill_dma_params(p, d, VERT, MT, KS, 512);
    const size_t lds = dma_lds_bytes(p, VERT);                   // does not depend on the slice count
    static int resident = 0;                                      // per instantiation; LDS size varies little within a class
    (void)slak_set_max_lds((const void*)k, lds);
    if (resident == 0) resident = resident_workgroups(k, lds);
    fill_dma_params(p, d, VERT, MT, KS, resident);
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, int MT, int KS>
static int launch_dma_t(MfmaDmaParams& p, const ConvDims& d, bool vert, bool band, hipStream_t st) {
    if (vert) return band ? launch_dma_tv<T, MT, KS, true, true, true>(p, d, st) : launch_dma_tv<T, MT, KS, true, false, true>(p, d, st);
    if (d.W % 8 == 0) return band ? launch_dma_tv<T, MT, KS, false, true, true>(p, d, st) : launch_dma_tv<T, MT, KS, false, false, true>(p, d, st);
    return band ? launch_dma_tv<T, MT, KS, false, true, false>(p, d, st) : launch_dma_tv<T, MT, KS, false, false, false>(p, d, st);
}

int launch_dwconv_mfma_dma(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                           const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st, bool accumulate,
                           float* stats, int stats_capacity_rows, int* stats_rows) {
    (void)ws; (void)ws_bytes;                                 // no workspace: fragments are built from LDS filter windows
    if (!dwconv_mfma_dma_supported(d, x_dt, w_dt, y_dt)) return SLAK_ERR_UNSUPPORTED;
    const bool vert = d.kh > d.kw;
    const int cls = dma_class(d, vert);
    const int MT = cls == 2 ? 2 : 1, KS = cls == 2 ? 4 : 2;
    MfmaDmaParams p;
    fill_dma_params(p, d, vert, MT, KS, 512);
    p.x = x; p.w = (const float*)w; p.y = y; p.flip = flip_filter ? 1 : 0;
    p.acc = accumulate ? 1 : 0;
    p.dbg = g_dma_dbg;
    p.stats = (stats && !accumulate && !flip_filter) ? stats : nullptr;
    if (p.stats && (stats_capacity_rows < MF_WAVES * d.N || !stats_rows)) return SLAK_ERR_WORKSPACE;   // slices <= N
    // band skipping pays when some (mt, ks) Toeplitz block is empty: filter half-width + 32 < 16*(KS-1)
    const bool band = (MT == 2) && (p.padL + 31 < 16 * (KS - 1));
    int rc;
    if (x_dt == SLAK_BF16) rc = cls == 2 ? launch_dma_t<bf16_t, 2, 4>(p, d, vert, band, st) : launch_dma_t<bf16_t, 1, 2>(p, d, vert, false, st);
    else rc = cls == 2 ? launch_dma_t<f16_t, 2, 4>(p, d, vert, band, st) : launch_dma_t<f16_t, 1, 2>(p, d, vert, false, st);
    if (rc == SLAK_OK && p.stats) *stats_rows = MF_WAVES * p.slices;
    return rc;
}

}  // namespace slak
