// slak_amd/csrc/dwconv_mfma_tri.hip -- the THREE branches of a decomposed large-kernel block (K x 5, 5 x K, 5 x 5 on the same
// input: models/SLaK.py:82-100) in ONE launch for the large maps (56x56, 28x28 class), forward and data gradient.
//   forward : x is DMA'd once, three outputs are written (4 plane passes over HBM instead of 6);
//   dgrad   : the three dy planes are DMA'd, ONE dx is written: dx = sum_b corr(dy_b, rot180(w_b)) -- the two elementwise adds
//             autograd otherwise runs on the per-branch gradients (2 x (read 2 + write 1)) disappear: 4 plane passes instead of 12.
// Structure: the LDS-DMA ring kernel of dwconv_mfma_dma.hip with twelve waves instead of four: wave = (branch b, role w4); the four
// waves of a branch do exactly what the four waves of the single-branch kernel do (tile = (plane of the group, 32 short-axis
// positions), wave w4 owns Toeplitz rows mt = w4 % MT), with their branch's Toeplitz fragments in registers.  Differences:
//   * the vertical branch reads x^T; the transpose of the NEXT group (ds_read_b64_tr_b16 + ds_write_b64) is done by the waves of
//     the 5 x 5 branch, which have the fewest MFMAs (band skipping) -- the vertical waves were the slow ones of the standalone kernels;
//   * every branch writes its rounded tile to its own LDS out-buffer; the copy-out sends three planes (forward) or adds the three
//     partial planes in fp32 and rounds once more (dgrad: what autograd's two bf16 adds do, with one rounding fewer);
//   * one workgroup of twelve waves per CU (registers): the ring is deep instead -- forward 8 slots, group g issued (and later awaited
//     with vmcnt(0): it has issued no other DMA in between) by wave g % 8; dgrad 4 slots x 3 tensors, each branch's own four waves
//     issue their dy (wave w4 == g % 4).
// Zero padding, synchronisation (one barrier per group, raw s_barrier), fragment construction from LDS filter windows and the pinned
// MFMA / ds_read pipeline are those of dwconv_mfma_dma.hip.
#include <stdlib.h>

#include "mfma_common.h"

namespace slak {

constexpr int TR_NCO = 2;               // 16-byte copy-out chunks per thread per group (upper bound)
constexpr int TR_WAVES = 12;            // 3 branches x 4 roles
constexpr int TR_THREADS = TR_WAVES * 64;
constexpr int TR_ZP = 64;               // zeros in front of a filter row (window starts never go negative)
constexpr int TR_LEN = 192;             // elements per padded filter row
constexpr int TR_ZROW = 128;            // elements of the all-zero row that out-of-range k pieces point at
constexpr int TR_NTR = 4;               // transpose blocks (4 rows x 16 cols) per 16-lane group per group of planes (upper bound; 4 waves)
constexpr int TR_WCH = 5;               // filter elements staged per lane of the staging wave (upper bound, 64 lanes)

struct TriParams {
    const void* in[3]; void* out[3]; const float* w[3];       // branch order: vertical (K x 5), horizontal (5 x K), small (5 x 5)
    int N, C, H, W, K, dgrad;
    int G;                 // planes per group (iteration)
    int tpp;               // 32-lane tiles per plane
    int ntiles;            // G * tpp  (<= 4 / MT: at most one tile per wave and group)
    int chunks_pp;         // 16-byte chunks per plane (HW/8)
    int plane_lds;         // LDS elements from one plane of a ring slot to the next (HW + 2W guard rows)
    int slot_elems;        // LDS elements of one tensor's part of a ring slot
    int NT;                // tensors in a slot: 1 (forward) or 3 (dgrad)
    int NB;                // ring depth in groups == issuing waves per tensor: 8 (forward: waves 0..7 take turns) or 4 (dgrad: the branch's own four waves)
    int PT;                // pitch of the transposed image
    int xt_rows;           // rows of one transposed plane image incl. 2+2 guard rows
    int planes_per_wg, slices;
    unsigned m_cpp;        // magic multiplier: n / chunks_pp == (n * m_cpp) >> 22
    int tr_pp, tr_cbs;     // transpose blocks per plane, per 4-row band
    unsigned tensor_bytes;
    int dbg;               // dev (SLAK_TRI_DBG): 1 skip the MFMA tiles, 2 skip the copy-out, 4 skip the transposes, 8 skip DMA issue and waits, 16 every wave copies out before its tile
};

// One 32x32 tile: k-steps LO..HI, five short taps each, ONE accumulator; SWAP: operands swapped (vertical branch:
// D^T = X^T-tile x T^T, so that a lane holds 4 consecutive columns of one output row).  The fragment of tap r for the next k-step is
// fetched right after this k-step's MFMA of tap r has issued, into the same registers (pinned with sched_barrier).
// kz: the last two k-steps may reach past the row end (horizontal: pieces beyond it read the zero row instead).
template <typename T, bool SWAP, bool R16, int KS, int LO, int HI>
__device__ __forceinline__ f32x16 tri_tile_mma(const s16x8 (&afrag)[MF_TAPS][KS], const char* L, const unsigned (&rp)[MF_TAPS],
                                               const bool (&kv0)[2], const bool (&kv1)[2], const unsigned (&zadj)[2]) {
    auto load_b = [&](int r, int ks) -> s16x8 {
        u32x4 b;
        if constexpr (SWAP) b = *(const u32x4*)(L + rp[r] + ks * 32);                  // x^T pads are zero
        else if constexpr (R16) {
            unsigned q = rp[r];
            if (ks >= KS - 2) q = kv0[ks - (KS - 2)] ? q : zadj[ks - (KS - 2)];
            b = *(const u32x4*)(L + q + ks * 32);
        } else {                                                                      // W % 8 == 4: rows are 8-byte aligned
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
    s16x8 b[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) b[r] = load_b(r, LO);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = LO; ks <= HI; ++ks) {
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            acc = SWAP ? mfma32<T>(b[r], afrag[r][ks], acc) : mfma32<T>(afrag[r][ks], b[r], acc);
            if (ks < HI) b[r] = load_b(r, ks + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

// MT: 32-row tiles along the Toeplitz axis; KS: 16-deep k-steps; R16: image rows are 16-byte aligned (W % 8 == 0)
template <typename T, int MT, int KS, bool R16>
__global__ __launch_bounds__(TR_THREADS, 1) void dwconv_mfma_tri_kernel(const TriParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;                                      // everything below is a BYTE offset into the LDS block
    const int HW = p.H * p.W;
    const unsigned tslot_b = (unsigned)p.slot_elems * 2;             // one tensor's part of a slot
    const unsigned slot_b = tslot_b * (unsigned)p.NT;
    const unsigned xt_buf_b = (unsigned)(p.G * p.xt_rows * p.PT) * 2;
    const unsigned out_buf_b = (unsigned)(p.G * HW) * 2;             // one branch's out-buffer
    const unsigned ring_b = 0;                                       // TR_NB slots (+ 128 bytes slack behind the last)
    const int NB = p.NB;
    const unsigned xt_b = ring_b + (unsigned)NB * slot_b + 128;       // 2 x [G][xt_rows][PT]
    const unsigned lout_b = xt_b + 2 * xt_buf_b;                     // 2 x 3 x [G][HW]
    const unsigned win_b = lout_b;                                   // [3 branches][2 copies][5 taps][TR_LEN]: prologue only, aliases the out-buffers
    constexpr unsigned win1_bytes = 2 * MF_TAPS * TR_LEN * 2;
    const unsigned zrow_b = lout_b + (6 * out_buf_b > 3 * win1_bytes ? 6 * out_buf_b : 3 * win1_bytes);   // TR_ZROW zeros

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int br = wave >> 2, w4 = wave & 3;                         // branch (0 vertical, 1 horizontal, 2 small), role
    const bool vert = br == 0;
    const int mt = w4 % MT, wl = w4 / MT;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int Wt = vert ? p.H : p.W, Wl = vert ? p.W : p.H;          // this branch's long / short axis extents
    const int KL = br == 2 ? MF_TAPS : p.K, padL = KL / 2;
    const int kh = br == 0 ? p.K : MF_TAPS, kw = br == 1 ? p.K : MF_TAPS;

    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    if (n_begin >= n_end) return;
    const int iters = (n_end - n_begin + p.G - 1) / p.G;

    // ---- DMA: one wave issues a whole group of ONE tensor: forward waves 0..3 (the input), dgrad every branch its own dy --------
    const int my_t = p.dgrad ? br : 0;                               // tensor this wave issues
    const bool issuer = p.dgrad || wave < NB;
    const int my_id = p.dgrad ? w4 : wave;                           // group g of my tensor is mine iff g % NB == my_id
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)p.in[my_t];
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes);
        rsrc[3] = 0x00020000;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    const unsigned plane_b = (unsigned)p.plane_lds * 2;              // LDS bytes from plane to plane within a slot
    const unsigned first_plane_b = (unsigned)(2 * p.W) * 2;          // two guard rows in front of every plane
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;              // HBM bytes from image n to image n+1 of this channel
    const int cpp_full = p.chunks_pp >> 6, cpp_rem = p.chunks_pp & 63;
    const unsigned lane16 = lane * 16;
    auto issue_group = [&](int g) {
        if (!issuer || g >= iters || my_id != g % NB || (p.dbg & 8)) return;        // wave-uniform
        const int n0 = n_begin + g * p.G;
        unsigned voff = (unsigned)(((size_t)n0 * p.C + c) * HW * 2) + lane16;
        unsigned m0v = lds_base + ring_b + (unsigned)(g % NB) * slot_b + (unsigned)my_t * tslot_b + first_plane_b;
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
    for (int g = 0; g < NB - 1; ++g) issue_group(g);                 // the last issuing wave has none yet; roles w4 == 3 stage the filters
    const int ntap = kh * kw;
    float wreg[TR_WCH];
    if (w4 == 3) {
#pragma unroll
        for (int k = 0; k < TR_WCH; ++k) { const int e = lane + 64 * k; wreg[k] = e < ntap ? p.w[br][(size_t)c * ntap + e] : 0.f; }
    }
    {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (unsigned o = tid * 16; o < 3 * win1_bytes; o += TR_THREADS * 16) *(u32x4*)(L + win_b + o) = z4;
        if (tid < TR_ZROW * 2 / 16) *(u32x4*)(L + zrow_b + tid * 16) = z4;
        for (unsigned o = tid * 16; o < 2 * xt_buf_b; o += TR_THREADS * 16) *(u32x4*)(L + xt_b + o) = z4;     // x^T guard rows / pad columns
        // ring: 2 guard rows in front of every plane + 2 behind the last, of every tensor part of every slot
        const int ngr = NB * p.NT * (p.G + 1);
        for (int q = wave; q < ngr; q += TR_WAVES) {
            const int st = q / (p.G + 1), jj = q - st * (p.G + 1);                          // st = slot * NT + tensor
            const unsigned gb = ring_b + (unsigned)st * tslot_b + jj * plane_b;
            for (int o = lane; o < p.W; o += 64) *(unsigned*)(L + gb + o * 4) = 0u;       // 2W elements = W dwords
        }
    }
    wg_barrier();
    if (w4 == 3) {
#pragma unroll
        for (int k = 0; k < TR_WCH; ++k) {
            const int e = lane + 64 * k;
            if (e < ntap) {
                int r = vert ? e % MF_TAPS : e / kw, t = vert ? e / MF_TAPS : e - (e / kw) * kw;      // short tap r, long tap t
                if (p.dgrad) { r = MF_TAPS - 1 - r; t = KL - 1 - t; }                                 // filter rotated by 180 degrees
                const uint16_t v = cvt_to_bits(wreg[k], (T*)nullptr);
                uint16_t* win = (uint16_t*)(L + win_b + br * win1_bytes);
                win[r * TR_LEN + TR_ZP + t] = v;                                         // copy 0
                win[MF_TAPS * TR_LEN + r * TR_LEN + TR_ZP + t - 1] = v;                  // copy 1 = copy 0 shifted by one element
            }
        }
    }
    wg_barrier();
    s16x8 afrag[MF_TAPS][KS];
    const int kfull = Wt >> 4;                                       // k-steps below this lie entirely inside the plane
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int a = TR_ZP + ks * 16 + lhi * 8 - (mt * 32 + l31) + padL;                // window start (element index), >= 1
        const int par = a & 1;
        const unsigned* src = (const unsigned*)(L + win_b + br * win1_bytes + par * MF_TAPS * TR_LEN * 2) + ((a - par) >> 1);
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            u32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = src[r * (TR_LEN / 2) + k];
                if (ks >= kfull && ks * 16 + lhi * 8 + 2 * k >= Wt) d[k] = 0u;          // i >= Wt: no such input
            }
            afrag[r][ks] = __builtin_bit_cast(s16x8, d);
        }
    }
    // active k-steps of this wave's Toeplitz block row (wave-uniform): block (mt, ks) meets the band -padL <= i - o <= KL-1-padL
    int ks_lo = KS, ks_hi = -1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int i_lo = ks * 16, i_hi = ks * 16 + 15, o_lo = mt * 32, o_hi = mt * 32 + 31;
        const bool act = (i_lo < Wt) && (o_lo < Wt) && (i_lo - o_hi <= KL - 1 - padL) && (o_lo - i_hi <= padL);
        if (act) { if (ks < ks_lo) ks_lo = ks; ks_hi = ks; }
    }

    // ---- per-thread constants of the loop (nothing below depends on the group) -----------------------------------------
    const int tpp_b = (Wl + 31) / 32;                                // this branch's tiles per plane (== p.tpp: H and W are in one class)
    const bool has_tile = wl < p.G * tpp_b && ks_hi >= ks_lo;
    const int j_t = has_tile ? wl / tpp_b : 0, sub_t = has_tile ? wl - j_t * tpp_b : 0;
    const int pos = sub_t * 32 + l31;                                 // lane -> position along the short (lane) axis
    unsigned brel[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) {
        if (vert) brel[r] = (unsigned)((j_t * p.xt_rows + pos + r) * p.PT) * 2 + lhi * 16;
        else brel[r] = (unsigned)j_t * plane_b + (unsigned)((pos + r) * p.W) * 2 + lhi * 16;
    }
    bool kv0[2], kv1[2];
    unsigned zadj[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ks = KS - 2 + kk;
        kv0[kk] = ks * 16 + lhi * 8 < Wt; kv1[kk] = ks * 16 + lhi * 8 + 4 < Wt;
        zadj[kk] = zrow_b + lhi * 16 - ks * 32;                       // so that the instruction offset 32*ks lands in the zero row
    }
    unsigned orel; bool qok[4];
    {
        const int orow = vert ? mt * 32 + l31 : pos, ocol0 = (vert ? sub_t * 32 : mt * 32) + 4 * lhi;
        const int nrow = vert ? Wt : Wl, ncol = vert ? Wl : Wt;
        orel = (unsigned)(j_t * HW + orow * p.W + ocol0) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) qok[q] = has_tile && orow < nrow && ocol0 + 8 * q < ncol;
    }
    // copy-out: 16-byte chunk idx of the (group, branch) out-buffer -> same chunk of the group's planes in HBM.
    // forward: 3 * TC chunks (branch-major); dgrad: TC chunks, each the sum of the three branch buffers
    const int TC = p.G * p.chunks_pp;
    const int nco = p.dgrad ? TC : 3 * TC;
    unsigned co_g[TR_NCO], co_l[TR_NCO]; int co_j[TR_NCO]; char* co_y[TR_NCO];
#pragma unroll
    for (int k = 0; k < TR_NCO; ++k) {
        const unsigned idx = tid + k * TR_THREADS;
        const unsigned b3 = p.dgrad ? 0u : idx / (unsigned)TC, r3 = idx - b3 * (unsigned)TC;
        const unsigned j = p.m_cpp ? (__umul24(r3, p.m_cpp) >> 22) : 0u, rem = r3 - j * p.chunks_pp;
        co_j[k] = (int)idx < nco ? (int)j : 0x3fffffff;
        co_g[k] = j * gplane_b + rem * 16;
        co_l[k] = b3 * out_buf_b + r3 * 16;
        co_y[k] = (char*)p.out[b3 < 3 ? b3 : 0];
    }
    // transpose map of the 5x5 branch's waves (role w4, 16-lane group grp): block b of a group = (plane j, 4 image rows kb, 16 image
    // columns cb); source = the guarded image of tensor 0 (forward: x; dgrad: dy of the vertical branch)
    unsigned tr_map[TR_NTR];
    if (br == 2) {
        const int grp = lane >> 4, i16 = lane & 15;
        const int total = p.G * p.tr_pp;
#pragma unroll
        for (int k = 0; k < TR_NTR; ++k) {
            const int b = (k * 4 + w4) * 4 + grp;
            const bool ok = b < total;                                // uniform per 16-lane group
            const int j = ok ? b / p.tr_pp : 0, rem = ok ? b - j * p.tr_pp : 0;
            const int kb = rem / p.tr_cbs, cb = rem - kb * p.tr_cbs;
            const unsigned src = (unsigned)(j * p.plane_lds + 2 * p.W + (kb * 4 + (i16 >> 2)) * p.W + cb * 16 + (i16 & 3) * 4) * 2;
            const unsigned dst = (cb * 16 + i16 < p.W) ? (unsigned)((j * p.xt_rows + 2 + cb * 16 + i16) * p.PT + kb * 4) * 2 : 0xffffu;
            tr_map[k] = ok ? (src | (dst << 16)) : 0xffffffffu;
        }
    }
    auto transpose_group = [&](int g) {                              // tensor 0 of ring slot g -> x^T buffer g&1
        const unsigned sb = ring_b + (unsigned)(g % NB) * slot_b, db = xt_b + (g & 1) * xt_buf_b;
#pragma unroll
        for (int k = 0; k < TR_NTR; ++k) {
            if (tr_map[k] != 0xffffffffu) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + sb + (tr_map[k] & 0xffffu)));
                if ((tr_map[k] >> 16) != 0xffffu) *(s16x4*)(L + db + (tr_map[k] >> 16)) = v;
            }
        }
    };
    // the wave that waits for group g's tensor t: role g % 4 of the issuing branch
    auto wait_group = [&](int g) {
        if (issuer && g < iters && my_id == g % NB && !(p.dbg & 8)) wait_vmcnt<0>();
    };
    auto copy_out = [&](int it_done, size_t yoff_done, int n0_done) {   // results of group it_done: LDS -> HBM, 16 bytes per lane
        const unsigned ob = lout_b + (unsigned)(it_done & 1) * 3u * out_buf_b;
#pragma unroll
        for (int k = 0; k < TR_NCO; ++k) {
            if (n0_done + co_j[k] >= n_end) continue;
            if (!p.dgrad) {
                *(u32x4*)(co_y[k] + yoff_done + co_g[k]) = *(const u32x4*)(L + ob + co_l[k]);
            } else {
                const u32x4 a = *(const u32x4*)(L + ob + co_l[k]), b = *(const u32x4*)(L + ob + out_buf_b + co_l[k]),
                            cc = *(const u32x4*)(L + ob + 2 * out_buf_b + co_l[k]);
                u32x4 s;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float lo, hi;
                    if constexpr (dtype_of<T>::value == SLAK_BF16) {
                        lo = (__uint_as_float(a[q] << 16) + __uint_as_float(b[q] << 16)) + __uint_as_float(cc[q] << 16);
                        hi = (__uint_as_float(a[q] & 0xffff0000u) + __uint_as_float(b[q] & 0xffff0000u)) + __uint_as_float(cc[q] & 0xffff0000u);
                    } else {
                        auto h2f = [](unsigned v) { return (float)__builtin_bit_cast(_Float16, (uint16_t)v); };
                        lo = (h2f(a[q] & 0xffffu) + h2f(b[q] & 0xffffu)) + h2f(cc[q] & 0xffffu);
                        hi = (h2f(a[q] >> 16) + h2f(b[q] >> 16)) + h2f(cc[q] >> 16);
                    }
                    s[q] = pack2<T>(lo, hi);
                }
                *(u32x4*)(co_y[k] + yoff_done + co_g[k]) = s;
            }
        }
    };

    // group 0 has to be transposed before the loop
    wait_group(0);
    wg_barrier();
    if (br == 2) transpose_group(0);
    size_t yoff = ((size_t)n_begin * p.C + c) * HW * 2;              // HBM byte offset of the current group's first plane
    int n0 = n_begin;
    for (int it = 0; it < iters; ++it) {
        wait_group(it + 1);                                          // the transpose of this iteration needs group it+1
        wg_barrier();                        // group it+1 landed; out-buffers it-1 and x^T `it` complete; a ring slot is free
        issue_group(it + NB - 1);
        // phases are staggered so that they overlap inside the one resident workgroup: the 5x5 branch's waves (fewest MFMAs) copy
        // out and transpose FIRST while the other eight waves are in their MFMA chains, which copy out afterwards
        const bool copy_first = br == 2 || (p.dbg & 16);
        if (copy_first && it > 0 && !(p.dbg & 2)) copy_out(it - 1, yoff - (size_t)p.G * gplane_b, n0 - p.G);
        if (br == 2 && it + 1 < iters && !(p.dbg & 4)) transpose_group(it + 1);
        const unsigned img_b = vert ? xt_b + (it & 1) * xt_buf_b
                                    : ring_b + (unsigned)(it % NB) * slot_b + (p.dgrad ? (unsigned)br * tslot_b : 0u);
        if (has_tile && !(p.dbg & 1)) {
            unsigned rp[MF_TAPS];
#pragma unroll
            for (int r = 0; r < MF_TAPS; ++r) rp[r] = img_b + brel[r];
            f32x16 acc;
            if (vert) {
                acc = tri_tile_mma<T, true, R16, KS, 0, KS - 1>(afrag, L, rp, kv0, kv1, zadj);
            } else if (ks_lo == 0 && ks_hi == KS - 1) {
                acc = tri_tile_mma<T, false, R16, KS, 0, KS - 1>(afrag, L, rp, kv0, kv1, zadj);
            } else if constexpr (KS == 4) {
                if (ks_lo == 0 && ks_hi == 2) acc = tri_tile_mma<T, false, R16, KS, 0, 2>(afrag, L, rp, kv0, kv1, zadj);
                else if (ks_lo == 1 && ks_hi == 3) acc = tri_tile_mma<T, false, R16, KS, 1, 3>(afrag, L, rp, kv0, kv1, zadj);
                else if (ks_lo == 0 && ks_hi == 1) acc = tri_tile_mma<T, false, R16, KS, 0, 1>(afrag, L, rp, kv0, kv1, zadj);
                else if (ks_lo == 1 && ks_hi == 2) acc = tri_tile_mma<T, false, R16, KS, 1, 2>(afrag, L, rp, kv0, kv1, zadj);
                else if (ks_lo == 2 && ks_hi == 3) acc = tri_tile_mma<T, false, R16, KS, 2, 3>(afrag, L, rp, kv0, kv1, zadj);
                else acc = tri_tile_mma<T, false, R16, KS, 0, KS - 1>(afrag, L, rp, kv0, kv1, zadj);   // inactive fragments are zero
            } else {
                acc = tri_tile_mma<T, false, R16, KS, 0, KS - 1>(afrag, L, rp, kv0, kv1, zadj);
            }
            if (n0 + j_t < n_end) {
                char* op = L + lout_b + (unsigned)(it & 1) * 3u * out_buf_b + (unsigned)br * out_buf_b + orel;
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
        }
        if (!copy_first && it > 0 && !(p.dbg & 2)) copy_out(it - 1, yoff - (size_t)p.G * gplane_b, n0 - p.G);
        yoff += (size_t)p.G * gplane_b; n0 += p.G;
    }
    wg_barrier();
    copy_out(iters - 1, yoff - (size_t)p.G * gplane_b, n0 - p.G);
}

// ------------------------------------------------------------------------------------------------------------
static int tri_class(int H, int W, int K) {                       // 2: MT=2/KS=4, 1: MT=1/KS=2, 0: not covered
    auto cls = [](int Wt) { return (Wt > 64 || Wt <= 16) ? 0 : (Wt > 32 ? 2 : 1); };
    const int a = cls(H), b = cls(W);
    if (a == 0 || a != b) return 0;
    if (K <= MF_TAPS || K > 63 || (K & 1) == 0) return 0;
    return a;
}

static bool fill_tri_params(TriParams& p, int N, int C, int H, int W, int K, bool dgrad, int MT, int KS, int resident_wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.dgrad = dgrad ? 1 : 0;
    const int HW = H * W;
    if (HW % 8 || W % 4 || H % 4) return false;
    if (K * MF_TAPS > TR_WCH * 64) return false;
    const int Wmin = H < W ? H : W, Wmax = H > W ? H : W;
    if (Wmin <= 16 * (KS - 2)) return false;                          // only the last two k-steps may reach past the plane edge
    p.tpp = (Wmax + 31) / 32;
    if ((Wmin + 31) / 32 != p.tpp) return false;                      // one tile count for both orientations
    const int WLW = 4 / MT;
    if (p.tpp > WLW) return false;
    p.G = WLW / p.tpp;
    if (p.G > N) p.G = N;
    p.ntiles = p.G * p.tpp;
    p.chunks_pp = HW / 8;
    p.plane_lds = HW + 2 * W;
    p.slot_elems = p.G * (HW + 2 * W) + 2 * W;
    p.slot_elems = (p.slot_elems + 7) & ~7;                           // tensor parts stay 16-byte aligned
    p.NT = dgrad ? 3 : 1;
    p.NB = dgrad ? 4 : 8;
    p.PT = KS * 16 + 8;
    p.xt_rows = W + 4;
    const int TC = p.G * p.chunks_pp;
    if ((dgrad ? TC : 3 * TC) > TR_NCO * TR_THREADS || TC >= 1024 || p.chunks_pp >= 1024) return false;
    p.tr_cbs = (W + 15) / 16; p.tr_pp = (H / 4) * p.tr_cbs;
    if (p.G * p.tr_pp > TR_NTR * 4 * 4) return false;
    if ((size_t)p.G * p.xt_rows * p.PT * 2 >= 65535 || (size_t)p.slot_elems * 2 >= 65535) return false;   // packed 16-bit transpose map
    int slices = resident_wgs / C; if (slices < 1) slices = 1;
    int per = (N + slices - 1) / slices; per = (per + p.G - 1) / p.G * p.G; if (per < p.G) per = p.G;
    p.planes_per_wg = per; p.slices = (N + per - 1) / per;
    p.m_cpp = p.G <= 1 ? 0u : (unsigned)(((1u << 22) + p.chunks_pp - 1) / p.chunks_pp);
    p.tensor_bytes = (unsigned)((size_t)N * C * HW * 2);
    return true;
}

static size_t tri_lds_bytes(const TriParams& p) {
    const size_t out6 = (size_t)6 * p.G * p.H * p.W * 2, win = (size_t)3 * 2 * MF_TAPS * TR_LEN * 2;
    return (size_t)p.NB * p.NT * p.slot_elems * 2 + 128 + (size_t)2 * p.G * p.xt_rows * p.PT * 2 + (out6 > win ? out6 : win) + (size_t)TR_ZROW * 2 + 16;
}

bool dwconv_mfma_tri_supported(int N, int C, int H, int W, int K, int dtype, bool dgrad) {
    if (dtype != SLAK_BF16 && dtype != SLAK_F16) return false;
    if (N <= 0 || C <= 0 || (long long)N * C * H * W >= (1LL << 31)) return false;
    const int cls = tri_class(H, W, K);
    if (!cls) return false;
    TriParams p;
    if (!fill_tri_params(p, N, C, H, W, K, dgrad, cls == 2 ? 2 : 1, cls == 2 ? 4 : 2, 256)) return false;
    return tri_lds_bytes(p) <= 160 * 1024;
}

template <typename T, int MT, int KS, bool R16>
static int launch_tri_t(TriParams& p, int N, int C, int H, int W, int K, bool dgrad, hipStream_t st) {
    auto k = dwconv_mfma_tri_kernel<T, MT, KS, R16>;
    fill_tri_params(p, N, C, H, W, K, dgrad, MT, KS, 256);
    const size_t lds = tri_lds_bytes(p);
    static thread_local size_t cached_lds = 0; static thread_local int cached_per_cu = 0;
    if (cached_lds != lds) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, TR_THREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        cached_per_cu = per_cu > 8 ? 8 : per_cu; cached_lds = lds;
    }
    fill_tri_params(p, N, C, H, W, K, dgrad, MT, KS, cached_per_cu * mfma_cu_count());
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(TR_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_tri(bool dgrad, const void* const* in, void* const* out, const float* const* w, int dtype,
                           int N, int C, int H, int W, int K, hipStream_t st) {
    if (!dwconv_mfma_tri_supported(N, C, H, W, K, dtype, dgrad)) return SLAK_ERR_UNSUPPORTED;
    const int cls = tri_class(H, W, K);
    TriParams p;
    for (int b = 0; b < 3; ++b) { p.in[b] = in[b]; p.out[b] = out[b]; p.w[b] = w[b]; }
    { const char* e = getenv("SLAK_TRI_DBG"); p.dbg = e ? atoi(e) : 0; }
    const bool r16 = W % 8 == 0;
    if (dtype == SLAK_BF16) {
        if (cls == 2) return r16 ? launch_tri_t<bf16_t, 2, 4, true>(p, N, C, H, W, K, dgrad, st) : launch_tri_t<bf16_t, 2, 4, false>(p, N, C, H, W, K, dgrad, st);
        return r16 ? launch_tri_t<bf16_t, 1, 2, true>(p, N, C, H, W, K, dgrad, st) : launch_tri_t<bf16_t, 1, 2, false>(p, N, C, H, W, K, dgrad, st);
    }
    if (cls == 2) return r16 ? launch_tri_t<f16_t, 2, 4, true>(p, N, C, H, W, K, dgrad, st) : launch_tri_t<f16_t, 2, 4, false>(p, N, C, H, W, K, dgrad, st);
    return r16 ? launch_tri_t<f16_t, 1, 2, true>(p, N, C, H, W, K, dgrad, st) : launch_tri_t<f16_t, 1, 2, false>(p, N, C, H, W, K, dgrad, st);
}

}  // namespace slak
