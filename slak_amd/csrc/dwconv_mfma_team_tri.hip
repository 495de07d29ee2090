// slak_amd/csrc/dwconv_mfma_team_tri.hip -- the THREE branches of a decomposed large-kernel block (K x 5, 5 x K, 5 x 5 on the same
// input: models/SLaK.py:82-100) in ONE launch on the large maps (56x56 / 28x28 class), forward and data gradient.  Round 3; replaces
// the twelve-wave kernel of round 2 (dwconv_mfma_tri.hip), which lost to three launches (134 vs 110 us) because its twelve waves ran
// every phase in lockstep behind one issuing wave.
//   forward : x travels HBM -> LDS once, three outputs are written (4 plane passes over HBM instead of 6 per block);
//   dgrad   : the three dy planes are fetched, ONE dx is written: dx = sum_b corr(dy_b, rot180(w_b)) -- autograd's two elementwise
//             adds on the per-branch gradients disappear (4 plane passes instead of 3 launches x (read dy [+ read dx] + write dx)).
// A TEAM of four waves owns one channel and a slice of the batch; a workgroup is TWO teams (<= 256 registers, <= 80 KB LDS each: one
// workgroup per CU) that share nothing but the workgroup barrier -- and through it run in ANTI-PHASE: between two barriers one team
// computes (matrix pipe) while the other moves data (HBM, LDS), then they swap.  (Two independent four-wave workgroups per CU fell
// into lockstep -- both computing, then both moving data: 115 us = 50 MFMA + 65 IO with nothing overlapped; a start-up stagger was
// absorbed by the first wait for data.)  Every wave carries the same number of MFMAs per group of planes:
//   * 56x56 class (planes of 2 x 2 tiles, one plane per group): wave (g, mt) computes the two tiles (mt, sub 0 / 1) of the vertical
//     (g = 0) or the horizontal (g = 1) branch -- 40 MFMAs -- and tile (mt, sub = g) of the 5 x 5 branch -- 15 (band): 55 each;
//   * 28x28 class (one tile per plane, four planes per group): wave w computes the three branches of plane w: 30 MFMAs each; in the
//     data gradient the three contributions go into ONE accumulator (both operand orders leave the same lane/register map).
// Per group two workgroup barriers separate a COMPUTE phase (tiles -> rounded results in LDS out-buffers) from an IO phase (out-buffers
// -> HBM in 16-byte pieces; the next group's vertical operand transposed LDS -> LDS with ds_read_b64_tr_b16; one more group requested
// from HBM): every LDS buffer is single -- what is saved buys a deep input ring.  The LDS-DMA of a group is issued by ALL four waves
// (piece q by wave q % 4; one wave issuing seven 1-KiB pieces costs ~700 cycles of its iteration, at ~100 cycles per piece), each wave
// waits for its own pieces with a COUNTED s_waitcnt: every wave issues the same vector-memory instructions every iteration
// (out-of-range planes carry out-of-range buffer offsets: loads bring zeros, stores are dropped), so the number of younger
// operations is a per-wave constant.
// Toeplitz fragments from LDS filter windows, zero guard rows / zero row, the pinned single-accumulator MFMA / ds_read pipeline and
// the transposed-image formulation of the vertical branch (D^T = X^T-tile x T^T) are those of dwconv_mfma_dma.hip.
#include "team_common.h"

namespace slak {

// CLS 2: planes of 2 x 2 tiles (32 < H, W <= 64), KS = 4 k-steps, one plane per group.  CLS 1: planes of one tile (16 < H, W <= 32),
// KS = 2, four planes per group.  R16: image rows are 16-byte aligned (W % 8 == 0).
// SOLO: the workgroup IS one team (four waves, two workgroups per CU): it runs its own compute and I/O phases one after the other, and the
// two workgroups of a CU drift against each other freely instead of alternating at workgroup barriers.
template <typename T, int CLS, bool DGRAD, bool R16, bool SOLO>
__global__ __launch_bounds__(SOLO ? TT_THREADS : 2 * TT_THREADS, SOLO ? 2 : 1) void dwconv_mfma_team_tri_kernel(const TeamParams p) {
    constexpr int KS = CLS == 2 ? 4 : 2;
    constexpr int NKS = CLS == 2 ? 3 : 2;                            // k-steps of a 5 x 5 tile (band)
    constexpr int NT = DGRAD ? 3 : 1;                                // input tensors
    constexpr int NO = DGRAD ? 1 : 3;                                // output tensors
    constexpr int NOB = DGRAD ? (CLS == 1 ? 1 : 2) : 3;              // LDS out-buffers (dgrad: the 5 x 5 tile shares an accumulator with a K-tap tile of the same position;
                                                                     // CLS 1: all three branches do)
    constexpr int NPW = DGRAD ? TT_NPW : 2;                          // LDS-DMA pieces per wave and group (upper bound)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int team = SOLO ? 0 : (wave_id_uniform() >> 2);            // 0 / 1: the two teams of the workgroup
    char* const L = (char*)lds + (size_t)team * p.team_lds;          // everything below is a BYTE offset into the TEAM's LDS block
    const int HW = p.H * p.W;
    const unsigned tslot_b = (unsigned)p.tslot_elems * 2;            // one tensor's part of a slot
    const unsigned t0_b = (unsigned)p.t0_elems * 2;                  // tensor 0's part; tensors 1, 2 follow at t0_b, t0_b + tslot_b
    const unsigned slot_b = t0_b + tslot_b * (NT - 1);
    const unsigned xt_buf_b = (unsigned)(p.G * p.xt_rows * p.PT) * 2;
    const unsigned out_buf_b = (unsigned)(p.G * HW) * 2;             // one out-buffer
    const int NB = p.NB;
    const unsigned ring_b = 0;                                       // NB slots (+ 128 bytes slack behind the last)
    const unsigned xt_b = ring_b + (unsigned)NB * slot_b + 128;       // [G][xt_rows][PT]
    const unsigned lout_b = xt_b + xt_buf_b;                         // NOB x [G][HW]
    const unsigned win_b = lout_b;                                   // [3 branches][2 copies][5 taps][TT_LEN]: prologue only, aliases the out-buffers
    constexpr unsigned win1_bytes = 2 * MF_TAPS * TT_LEN * 2;
    const unsigned zrow_b = lout_b + (NOB * out_buf_b > 3 * win1_bytes ? NOB * out_buf_b : 3 * win1_bytes);   // TT_ZROW zeros
    const unsigned sfr_b = zrow_b + TT_ZROW * 2;                     // CLS 2: the 5 x 5 branch's twenty Toeplitz fragments

    const int tid = threadIdx.x & (TT_THREADS - 1), lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;   // thread / wave index WITHIN the team
    const int wave = wave_id_uniform() & 3;
    const int vb = SOLO ? (int)blockIdx.x : (int)blockIdx.x * 2 + team;   // the team's (channel, slice)
    const bool has_work = vb < p.C * p.slices;
    const int c = has_work ? vb % p.C : 0, slice = has_work ? vb / p.C : 0;
    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    if (!has_work) n_end = n_begin;
    const int iters = (n_end - n_begin + p.G - 1) / p.G;             // (a team without planes keeps the workgroup's barriers company)

    // ---- LDS-DMA: piece q = (tensor t, plane j of the group, 64-chunk piece pp) is issued by wave q % 4 ----------------------------
    v4i_t rsrc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint64_t a = (uint64_t)p.in[t];
        rsrc[t][0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        rsrc[t][1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[t][2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes);
        rsrc[t][3] = 0x00020000;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds) + (unsigned)team * (unsigned)p.team_lds;
    const unsigned plane_b = (unsigned)p.plane_lds * 2;              // LDS bytes from plane to plane within a slot
    const unsigned first_plane_b = (unsigned)(2 * p.W) * 2;          // two guard rows in front of every plane
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;              // HBM bytes from image n to image n+1 of this channel
    const int my_pieces = p.my_pieces[wave];
    auto issue_group = [&](int g) {                                  // ALWAYS issues my_pieces instructions (planes beyond the slice: zeros)
        const int n0 = n_begin + g * p.G;
        const unsigned gb = (unsigned)(((size_t)n0 * p.C + c) * HW * 2);
        const unsigned sb = lds_base + ring_b + (unsigned)(g % NB) * slot_b;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const TeamPiece pc = p.pieces[wave][k];
            const int nl = pc.info >> 8, t = pc.info & 15, j = (pc.info >> 4) & 15;
            if (nl == 0) continue;                                   // wave-uniform: no such piece
            const unsigned voff = (n0 + j < n_end) ? gb + pc.g_off + (unsigned)lane * 16u : TT_OOB;
            const unsigned m0 = __builtin_amdgcn_readfirstlane(sb + pc.lds_off);
            if (lane < nl) {
                if (NT == 1 || t == 0) lds_dma16(voff, rsrc[0], m0);
                else if (t == 1) lds_dma16(voff, rsrc[NT > 1 ? 1 : 0], m0);
                else lds_dma16(voff, rsrc[NT > 2 ? 2 : 0], m0);
            }
        }
    };

    // ---- prologue: the ring filled, zero areas, filter windows, fragments ---------------------------------------------------------
    for (int g = 0; g < NB; ++g) issue_group(g);
    const int KL = p.K;
    float wreg[TT_WCH];
    const int st_ntap = wave < 2 ? KL * MF_TAPS : MF_TAPS * MF_TAPS;  // wave b (< 3) stages branch b's filter
    if (wave < 3) {
#pragma unroll
        for (int k = 0; k < TT_WCH; ++k) { const int e = lane + 64 * k; wreg[k] = e < st_ntap ? p.w[wave][(size_t)c * st_ntap + e] : 0.f; }
    }
    {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (unsigned o = tid * 16; o < 3 * win1_bytes; o += TT_THREADS * 16) *(u32x4*)(L + win_b + o) = z4;
        if (tid < TT_ZROW * 2 / 16) *(u32x4*)(L + zrow_b + tid * 16) = z4;
        for (unsigned o = tid * 16; o < xt_buf_b; o += TT_THREADS * 16) *(u32x4*)(L + xt_b + o) = z4;         // x^T guard rows / pad columns
        // ring: 2 guard rows in front of every plane + 2 behind the last, of every tensor part of every slot
        constexpr int NGT = DGRAD ? 2 : 1;                            // guarded tensors per slot (dgrad: tensors 1, 2)
        const int ngr = NB * NGT * (p.G + 1);
        for (int q = wave; q < ngr; q += TT_WAVES) {
            const int st = q / (p.G + 1), jj = q - st * (p.G + 1);                          // st = slot * NGT + guarded tensor
            const int sl = st / NGT, gt = st - sl * NGT;
            const unsigned gb = ring_b + (unsigned)sl * slot_b + (DGRAD ? t0_b + (unsigned)gt * tslot_b : 0u) + jj * plane_b;
            for (int o = lane; o < p.W; o += 64) *(unsigned*)(L + gb + o * 4) = 0u;       // 2W elements = W dwords
        }
    }
    wg_barrier();
    if (wave < 3) {
        const bool vert = wave == 0;
        const int kw = wave == 1 ? KL : MF_TAPS, kl = wave == 2 ? MF_TAPS : KL;
#pragma unroll
        for (int k = 0; k < TT_WCH; ++k) {
            const int e = lane + 64 * k;
            if (e < st_ntap) {
                int r = vert ? e % MF_TAPS : e / kw, t = vert ? e / MF_TAPS : e - (e / kw) * kw;      // short tap r, long tap t
                if (DGRAD) { r = MF_TAPS - 1 - r; t = kl - 1 - t; }                                   // filter rotated by 180 degrees
                const uint16_t v = cvt_to_bits(wreg[k], (T*)nullptr);
                uint16_t* win = (uint16_t*)(L + win_b + wave * win1_bytes);
                win[r * TT_LEN + TT_ZP + t] = v;                                         // copy 0
                win[MF_TAPS * TT_LEN + r * TT_LEN + TT_ZP + t - 1] = v;                  // copy 1 = copy 0 shifted by one element
            }
        }
    }
    wg_barrier();

    // ---- roles and fragments ---------------------------------------------------------------------------------------------------
    // CLS 2: mt = wave & 1, g2 = wave >> 1: branch A = vertical (g2 == 0) / horizontal (g2 == 1), tiles (mt, sub 0), (mt, sub 1);
    //        5 x 5 tile (mt, sub = s_sub), k-steps mt .. mt + 2 (the band of |i - o| <= 2 around the 32 rows of mt).  Forward: s_sub = g2.
    //        Data gradient: the 5 x 5 tile is the one that covers the same pixels as one of the wave's A tiles -- vertical tile (mt, sub)
    //        covers rows 32 mt.., columns 32 sub..; horizontal / small tile (mt, sub) covers columns 32 mt.., rows 32 sub.. -- i.e.
    //        s_sub = mt for the vertical waves, 1 - mt for the horizontal ones, and its MFMAs continue that A tile's accumulator
    //        (both operand orders leave the same lane / register map): two partial planes instead of three.
    // CLS 1: plane j = wave, the three branches' single tile.
    const int mt = CLS == 2 ? (wave & 1) : 0, g2 = CLS == 2 ? (wave >> 1) : 0;
    const bool a_vert = CLS == 2 ? (g2 == 0) : true;
    const int s_sub = CLS == 2 ? (DGRAD ? (g2 == 0 ? mt : 1 - mt) : g2) : 0;
    s16x8 fragA[MF_TAPS][KS];                                        // CLS 2: branch A;  CLS 1: vertical
    s16x8 fragS[MF_TAPS][CLS == 1 ? NKS : 1];                        // CLS 1: 5 x 5 (CLS 2 keeps them in LDS: team_small_tile_mma)
    s16x8 fragH[MF_TAPS][CLS == 1 ? KS : 1];                         // CLS 1: horizontal
    if constexpr (CLS == 2) {
        team_build_frags<KS>(fragA, L + win_b + (a_vert ? 0u : win1_bytes), 0, mt, l31, lhi, a_vert ? p.H : p.W, KL / 2);
        {                                                             // wave w builds the five fragments of d = w - 1
            const int a = TT_ZP + 16 * (wave - 1) + lhi * 8 - l31 + MF_TAPS / 2;          // window start (element index), >= 1
            const int par = a & 1;
            const unsigned* src = (const unsigned*)(L + win_b + 2 * win1_bytes + par * MF_TAPS * TT_LEN * 2) + ((a - par) >> 1);
#pragma unroll
            for (int r = 0; r < MF_TAPS; ++r) {
                u32x4 d;
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = src[r * (TT_LEN / 2) + k];
                *(u32x4*)(L + sfr_b + (wave * MF_TAPS + r) * 1024 + lane * 16) = d;
            }
        }
    } else {
        team_build_frags<KS>(fragA, L + win_b, 0, 0, l31, lhi, p.H, KL / 2);
        team_build_frags<KS>(fragH, L + win_b + win1_bytes, 0, 0, l31, lhi, p.W, KL / 2);
        team_build_frags<NKS>(fragS, L + win_b + 2 * win1_bytes, 0, 0, l31, lhi, p.W, MF_TAPS / 2);
    }

    // ---- per-thread constants of the loop (nothing below depends on the group) -----------------------------------------------------
    // tile (plane j, 32 short-axis positions sub): B fragment of tap r, k-step ks = 16 bytes at rel0 + r*pitch + 32*ks from the image base
    const unsigned pitch_v = (unsigned)p.PT * 2, pitch_h = (unsigned)p.W * 2;                 // vertical: rows of x^T; horizontal / small: image rows
    auto rel_v = [&](int j, int sub) { return (unsigned)((j * p.xt_rows + sub * 32 + l31) * p.PT) * 2 + lhi * 16; };
    auto rel_h = [&](int j, int sub) { return (unsigned)j * plane_b + (unsigned)((sub * 32 + l31) * p.W) * 2 + lhi * 16; };
    const int wlim = p.W - lhi * 8;                                   // horizontal / small: k-step ks lies inside the row iff ks*16 < wlim
    const unsigned zrow_l = zrow_b + lhi * 16;
    // epilogue of a tile whose lanes are output rows `orow0 + l31` and whose register quads are columns `ocol0 + 4*lhi + 8q`:
    // LDS offset of the lane's first quad, and ocmax = columns left in the row from there (0: the lane's row is outside the plane)
    auto out_rel = [&](int j, int orow0, int ocol0, unsigned& orel, int& ocmax) {
        const int orow = orow0 + l31, oc = ocol0 + 4 * lhi;
        orel = (unsigned)(j * HW + orow * p.W + oc) * 2;
        ocmax = orow < p.H ? p.W - oc : 0;
    };
    auto store_quad = [&](const f32x16& acc, unsigned ob, unsigned orel, int ocmax, int q) {
        if (8 * q < ocmax) {
            u32x2 v;
            v[0] = pack2<T>(acc[4 * q + 0], acc[4 * q + 1]);
            v[1] = pack2<T>(acc[4 * q + 2], acc[4 * q + 3]);
            *(u32x2*)(L + ob + orel + 16 * q) = v;
        }
    };
    auto store_tile = [&](const f32x16& acc, unsigned ob, unsigned orel, int ocmax) {
#pragma unroll
        for (int q = 0; q < 4; ++q) store_quad(acc, ob, orel, ocmax, q);
    };
    // the same epilogue as fillers of the next tile's MFMAs 2 .. 5 (its accumulator is a different one and has been complete for a while)
    auto deferred = [&](const f32x16& acc, unsigned ob, unsigned orel, int ocmax) {
        return [&acc, ob, orel, ocmax, &store_quad](int j) { if (j >= 2 && j < 6) store_quad(acc, ob, orel, ocmax, j - 2); };
    };
    unsigned relA, relS, orelA0, orelA1, orelS; int ocA0, ocA1, ocS;
    if constexpr (CLS == 2) {
        if (a_vert) { relA = rel_v(0, 0); out_rel(0, mt * 32, 0, orelA0, ocA0); out_rel(0, mt * 32, 32, orelA1, ocA1); }
        else { relA = rel_h(0, 0); out_rel(0, 0, mt * 32, orelA0, ocA0); out_rel(0, 32, mt * 32, orelA1, ocA1); }
        relS = rel_h(0, s_sub); out_rel(0, s_sub * 32, mt * 32, orelS, ocS);
    } else {
        relA = rel_v(wave, 0); relS = rel_h(wave, 0);                 // relS: the horizontal operand (5 x K and 5 x 5 read the same rows)
        out_rel(wave, 0, 0, orelA0, ocA0);
        orelA1 = orelA0; orelS = orelA0; ocA1 = ocA0; ocS = ocA0;
    }
    // copy-out: chunk idx (< TC = G * chunks_pp) of an out-buffer -> the same chunk of the group's planes in HBM
    const int TC = p.G * p.chunks_pp;
    unsigned co_g[TT_NCO], co_l[TT_NCO]; int co_j[TT_NCO];
#pragma unroll
    for (int k = 0; k < TT_NCO; ++k) {
        const unsigned idx = tid + k * TT_THREADS;
        const unsigned j = p.m_cpp ? (__umul24(idx, p.m_cpp) >> 22) : 0u, rem = idx - j * p.chunks_pp;
        co_j[k] = (int)idx < TC ? (int)j : 0x3fffffff;
        co_g[k] = j * gplane_b + rem * 16;
        co_l[k] = idx * 16;
    }
    __amdgpu_buffer_rsrc_t ro[NO];
#pragma unroll
    for (int t = 0; t < NO; ++t) ro[t] = __builtin_amdgcn_make_buffer_rsrc(p.out[t], 0, (int)p.tensor_bytes, 0x00020000);
    // transpose map: block b of a group = (plane j, 4 image rows kb, 16 image columns cb); source = the guarded image of tensor 0
    // (forward: x; dgrad: dy of the vertical branch); 16 lane groups of 16 lanes
    unsigned tr_map[TT_NTR];
    {
        const int grp = lane >> 4, i16 = lane & 15;
        const int total = p.G * p.tr_pp;
#pragma unroll
        for (int k = 0; k < TT_NTR; ++k) {
            const int b = (k * TT_WAVES + wave) * 4 + grp;
            const bool ok = b < total;                                // uniform per 16-lane group
            const int j = ok ? b / p.tr_pp : 0, rem = ok ? b - j * p.tr_pp : 0;
            const int kb = rem / p.tr_cbs, cb = rem - kb * p.tr_cbs;
            const unsigned src = (unsigned)(j * p.t0_plane + p.t0_first + (kb * 4 + (i16 >> 2)) * p.W + cb * 16 + (i16 & 3) * 4) * 2;
            const unsigned dst = (cb * 16 + i16 < p.W) ? (unsigned)((j * p.xt_rows + 2 + cb * 16 + i16) * p.PT + kb * 4) * 2 : 0xffffu;
            tr_map[k] = ok ? (src | (dst << 16)) : 0xffffffffu;
        }
    }
    auto transpose_group = [&](int g) {                              // tensor 0 of ring slot g -> x^T
        const unsigned sb = ring_b + (unsigned)(g % NB) * slot_b;
#pragma unroll
        for (int k = 0; k < TT_NTR; ++k) {
            if (tr_map[k] != 0xffffffffu) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + sb + (tr_map[k] & 0xffffu)));
                if ((tr_map[k] >> 16) != 0xffffu) *(s16x4*)(L + xt_b + (tr_map[k] >> 16)) = v;
            }
        }
    };
    // counted wait: the vector-memory operations this wave issues per group are my_pieces LDS-DMA pieces + NST stores, always
    constexpr int NST = NO * TT_NCO;
    float bs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                    // p.stats: sums over the chunks this thread copies out
    auto stat8 = [&](const u32x4& v, float& s1, float& s2) {         // v_dot2c_f32_bf16 with a ZERO addend (its addend is aligned with truncation)
        if constexpr (std::is_same<T, bf16_t>::value) {
            const bf16x2_t one = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            const unsigned d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;   // (elements copied to scalars first: see dwconv_mfma_dma.hip)
            const bf16x2_t x0 = __builtin_bit_cast(bf16x2_t, d0), x1 = __builtin_bit_cast(bf16x2_t, d1), x2 = __builtin_bit_cast(bf16x2_t, d2), x3 = __builtin_bit_cast(bf16x2_t, d3);
            float a0 = __builtin_amdgcn_fdot2_f32_bf16(x0, one, 0.f, false), a1 = __builtin_amdgcn_fdot2_f32_bf16(x1, one, 0.f, false);
            float a2 = __builtin_amdgcn_fdot2_f32_bf16(x2, one, 0.f, false), a3 = __builtin_amdgcn_fdot2_f32_bf16(x3, one, 0.f, false);
            float q0 = __builtin_amdgcn_fdot2_f32_bf16(x0, x0, 0.f, false), q1 = __builtin_amdgcn_fdot2_f32_bf16(x1, x1, 0.f, false);
            float q2 = __builtin_amdgcn_fdot2_f32_bf16(x2, x2, 0.f, false), q3 = __builtin_amdgcn_fdot2_f32_bf16(x3, x3, 0.f, false);
            asm("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            s1 += (a0 + a1) + (a2 + a3); s2 += (q0 + q1) + (q2 + q3);
        }
    };
    // results of group it_done: LDS -> HBM, 16 bytes per lane; NST stores, always.  Two halves, so that the I/O phase can put every LDS read it
    // needs (these and the transposing reads of the next group) in flight before it consumes the first: the phase is a chain of LDS
    // round trips under the other workgroup's fragment traffic, not work.
    u32x4 cbuf[TT_NCO][NOB];
    auto copy_out_read = [&]() {
#pragma unroll
        for (int k = 0; k < TT_NCO; ++k) {
            const unsigned lo = co_j[k] == 0x3fffffff ? 0u : co_l[k];                       // lanes beyond the buffer read its first chunk (dropped)
#pragma unroll
            for (int t = 0; t < NOB; ++t) cbuf[k][t] = *(const u32x4*)(L + lout_b + t * out_buf_b + lo);
        }
    };
    auto copy_out_store = [&](int it_done) {
        const int n0 = n_begin + it_done * p.G;
        const unsigned gb = (unsigned)(((size_t)n0 * p.C + c) * HW * 2);
#pragma unroll
        for (int k = 0; k < TT_NCO; ++k) {
            const bool live = n0 + co_j[k] < n_end;
            const unsigned go = live ? gb + co_g[k] : TT_OOB;
            if constexpr (!DGRAD) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (p.stats && live) stat8(cbuf[k][t], bs[2 * t], bs[2 * t + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(cbuf[k][t], ro[t], go, 0, 0);
                }
            } else if constexpr (NOB == 1) {
                __builtin_amdgcn_raw_buffer_store_b128(cbuf[k][0], ro[0], go, 0, 0);
            } else {
                // dx = (vertical partial) + (horizontal + small partial): a tensor add of two rounded planes (what autograd's adds do)
                __builtin_amdgcn_raw_buffer_store_b128(add_packed<T>(cbuf[k][0], cbuf[k][1]), ro[0], go, 0, 0);
            }
        }
    };
    s16x4 tbuf[TT_NTR];
    auto transpose_read = [&](int g) {                               // tensor 0 of ring slot g
        const unsigned sb = ring_b + (unsigned)(g % NB) * slot_b;
#pragma unroll
        for (int k = 0; k < TT_NTR; ++k)
            if (tr_map[k] != 0xffffffffu) tbuf[k] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + sb + (tr_map[k] & 0xffffu)));
    };
    auto transpose_write = [&]() {                                   // -> x^T
#pragma unroll
        for (int k = 0; k < TT_NTR; ++k)
            if (tr_map[k] != 0xffffffffu && (tr_map[k] >> 16) != 0xffffu) *(s16x4*)(L + xt_b + (tr_map[k] >> 16)) = tbuf[k];
    };

    unsigned long long* tl = TT_TIMELINE(p, vb);
    if (tl && tid == 0) {
        tl[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);        // HW_ID
        tl[1] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);        // XCC_ID
        tl[2] = __builtin_amdgcn_s_memrealtime();
        tl[60] = __builtin_readcyclecounter();
    }
    // group 0 has to be transposed before the loop: every wave waits for ITS pieces of group 0 (NB - 1 groups are younger)
    wait_vmcnt_dyn((NB - 1) * my_pieces);
    wg_barrier();
    if (!TT_DBG(p, 4)) transpose_group(0);
    auto compute_phase = [&](int it) {
        if (tl && tid == 0 && it < 27) tl[3 + 2 * it] = __builtin_amdgcn_s_memrealtime();
        if (tl && tid == 0 && it == 10) tl[57] = __builtin_readcyclecounter();
        // ---------------- compute phase: this wave's tiles of group `it` -> out-buffers.  One MFMA stream over the wave's tiles: the
        // B-fragment pipeline runs across tile boundaries, consecutive tiles alternate between two accumulators and a tile's epilogue
        // (pack + LDS stores) is issued behind the NEXT tile's MFMAs, when its accumulator has long been complete.
        const unsigned img_b = ring_b + (unsigned)(it % NB) * slot_b;
        if (!TT_DBG(p, 1)) {
            s16x8 bq[TT_NBUF];
            f32x16 acc0;
            auto zero = [](f32x16& a) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = 0.f;
            };
            if constexpr (CLS == 2) {
                const unsigned a_ob = lout_b + (a_vert ? 0u : out_buf_b);
                const unsigned s_rp = img_b + (DGRAD ? t0_b + tslot_b : 0u) + relS;
                // (AV, MT) are wave-uniform run-time values: four straight-line instantiations
                auto run = [&](auto av_c, auto mt_c) {
                    constexpr bool AV = decltype(av_c)::value;
                    constexpr int MT = decltype(mt_c)::value;
                    constexpr int NA = 1 + !AV;                       // NXT code of an A tile: 1 = SWAP (vertical), 2 = plain
                    const unsigned a_rp = (AV ? xt_b : img_b + (DGRAD ? t0_b : 0u)) + relA;
                    const unsigned a_p = AV ? pitch_v : pitch_h, a_rp1 = a_rp + 32u * a_p;
                    team_tile_prefetch<AV, R16, KS, 0, 0>(bq, L, a_rp, a_p, wlim, zrow_l);
                    const unsigned sfr_l = sfr_b + lane * 16;
                    s16x8 sa[TT_NBUF];
                    if constexpr (!DGRAD) {                           // A(sub 0) -> acc0, A(sub 1) -> acc1, S -> acc0: an epilogue is issued behind the next tile's MFMAs
                        f32x16 acc1;
                        zero(acc0);
                        if (tl && tid == 0 && it == 10) tl[55] = __builtin_readcyclecounter();
                        team_tile_mma<T, AV, R16, KS, KS, 0, 0, NA, 0>(acc0, fragA, bq, L, a_rp, a_p, wlim, zrow_l, a_rp1, a_p);
                        if (tl && tid == 0 && it == 10) tl[56] = __builtin_readcyclecounter();
                        team_small_prefetch<MT>(sa, L, sfr_l);
                        zero(acc1);
                        team_tile_mma<T, AV, R16, KS, KS, 0, (MF_TAPS * KS) % TT_NBUF, 2, MT>(acc1, fragA, bq, L, a_rp1, a_p, wlim, zrow_l, s_rp, pitch_h,
                                                                                                 deferred(acc0, a_ob, orelA0, ocA0));
                        f32x16 acc2;                                  // (a third name: the first tile's stores are fillers of the second tile)
                        zero(acc2);
                        team_small_tile_mma<T, R16, KS, MT, (2 * MF_TAPS * KS) % TT_NBUF, 0, 0>(acc2, sa, bq, L, s_rp, pitch_h, wlim, zrow_l, sfr_l, 0u, 0u,
                                                                                              deferred(acc1, a_ob, orelA1, ocA1));
                        store_tile(acc2, lout_b + 2 * out_buf_b, orelS, ocS);
                    } else {
                        // the 5 x 5 tile continues the accumulator of the A tile that covers the same pixels: sub = MT (vertical waves), 1 - MT (horizontal)
                        constexpr int SS = AV ? MT : 1 - MT;
                        f32x16 acc1;
                        zero(acc0);
                        if constexpr (SS == 0) {                      // A(sub 0) + S -> acc0, A(sub 1) -> acc1
                            team_small_prefetch<MT>(sa, L, sfr_l);
                            team_tile_mma<T, AV, R16, KS, KS, 0, 0, 2, MT>(acc0, fragA, bq, L, a_rp, a_p, wlim, zrow_l, s_rp, pitch_h);
                            team_small_tile_mma<T, R16, KS, MT, (MF_TAPS * KS) % TT_NBUF, NA, 0>(acc0, sa, bq, L, s_rp, pitch_h, wlim, zrow_l, sfr_l, a_rp1, a_p);
                            zero(acc1);
                            team_tile_mma<T, AV, R16, KS, KS, 0, (MF_TAPS * (KS + 3)) % TT_NBUF, 0, 0>(acc1, fragA, bq, L, a_rp1, a_p, wlim, zrow_l, 0u, 0u,
                                                                                                      deferred(acc0, a_ob, orelA0, ocA0));
                            store_tile(acc1, a_ob, orelA1, ocA1);
                        } else {                                      // A(sub 0) -> acc0, A(sub 1) + S -> acc1
                            team_tile_mma<T, AV, R16, KS, KS, 0, 0, NA, 0>(acc0, fragA, bq, L, a_rp, a_p, wlim, zrow_l, a_rp1, a_p);
                            team_small_prefetch<MT>(sa, L, sfr_l);
                            zero(acc1);
                            team_tile_mma<T, AV, R16, KS, KS, 0, (MF_TAPS * KS) % TT_NBUF, 2, MT>(acc1, fragA, bq, L, a_rp1, a_p, wlim, zrow_l, s_rp, pitch_h,
                                                                                                     deferred(acc0, a_ob, orelA0, ocA0));
                            team_small_tile_mma<T, R16, KS, MT, (2 * MF_TAPS * KS) % TT_NBUF, 0, 0>(acc1, sa, bq, L, s_rp, pitch_h, wlim, zrow_l, sfr_l, 0u, 0u);
                            store_tile(acc1, a_ob, orelA1, ocA1);
                        }
                    }
                };
                using std::integral_constant;
                if (a_vert) { if (mt == 0) run(integral_constant<bool, true>{}, integral_constant<int, 0>{}); else run(integral_constant<bool, true>{}, integral_constant<int, 1>{}); }
                else { if (mt == 0) run(integral_constant<bool, false>{}, integral_constant<int, 0>{}); else run(integral_constant<bool, false>{}, integral_constant<int, 1>{}); }
            } else {
                const unsigned v_rp = xt_b + relA, h_rp = img_b + (DGRAD ? t0_b : 0u) + relS, s_rp = img_b + (DGRAD ? t0_b + tslot_b : 0u) + relS;
                team_tile_prefetch<true, R16, KS, 0, 0>(bq, L, v_rp, pitch_v, wlim, zrow_l);
                zero(acc0);
                team_tile_mma<T, true, R16, KS, KS, 0, 0, 2, 0>(acc0, fragA, bq, L, v_rp, pitch_v, wlim, zrow_l, h_rp, pitch_h);
                if constexpr (!DGRAD) {                               // V -> acc0, H -> acc1, S -> acc0: an epilogue is issued behind the next tile's MFMAs
                    f32x16 acc1;
                    zero(acc1);
                    team_tile_mma<T, false, R16, KS, KS, 0, (MF_TAPS * KS) % TT_NBUF, 2, 0>(acc1, fragH, bq, L, h_rp, pitch_h, wlim, zrow_l, s_rp, pitch_h,
                                                                                               deferred(acc0, lout_b, orelA0, ocA0));
                    f32x16 acc2;
                    zero(acc2);
                    team_tile_mma<T, false, R16, KS, NKS, 0, (2 * MF_TAPS * KS) % TT_NBUF, 0, 0>(acc2, fragS, bq, L, s_rp, pitch_h, wlim, zrow_l, 0u, 0u,
                                                                                                 deferred(acc1, lout_b + out_buf_b, orelA0, ocA0));
                    store_tile(acc2, lout_b + 2 * out_buf_b, orelA0, ocA0);
                } else {                                              // the three branches in ONE accumulator
                    team_tile_mma<T, false, R16, KS, KS, 0, (MF_TAPS * KS) % TT_NBUF, 2, 0>(acc0, fragH, bq, L, h_rp, pitch_h, wlim, zrow_l, s_rp, pitch_h);
                    team_tile_mma<T, false, R16, KS, NKS, 0, (2 * MF_TAPS * KS) % TT_NBUF, 0, 0>(acc0, fragS, bq, L, s_rp, pitch_h, wlim, zrow_l, 0u, 0u);
                    store_tile(acc0, lout_b, orelA0, ocA0);
                }
            }
        }
        // the transposes of the IO phase read group it+1: this wave's pieces of it.  Younger operations of this wave:
        //   it + 1 <  NB (issued in the prologue): the prologue's later groups + every IO phase so far
        //   it + 1 >= NB (issued in IO phase it + 1 - NB): that phase's stores + NB - 2 whole phases
        {
            if (tl && tid == 0 && it == 10) tl[58] = __builtin_readcyclecounter();
            if (it + 1 < NB) wait_vmcnt_dyn((NB - 2 - it) * my_pieces + it * (my_pieces + NST));
            else if (NB >= 3) wait_vmcnt<2 * NST + 1>();             // <= NST + (NB - 2) * (my_pieces + NST): a compile-time bound (no jump table in the loop)
            else wait_vmcnt<NST>();
            if (tl && tid == 0 && it == 10) tl[59] = __builtin_readcyclecounter();
        }
    };
    auto io_phase = [&](int it) {
        if (tl && tid == 0 && it < 27) tl[4 + 2 * it] = __builtin_amdgcn_s_memrealtime();
        // ---------------- IO phase
        if (tl && tid == 0 && it == 10) tl[52] = __builtin_readcyclecounter();
        // The I/O phase is ~500 cycles of instructions whose LDS round trips chain; a SIMD issues strictly oldest-wave-first
        // (tools/mfma_share_probe.hip: of two MFMA-streaming waves the younger one does not move until the older one stalls), so beside
        // a computing wave of the CU's other workgroup this wave would only get the slots that one leaves: raise its priority for the phase.
        if (TT_IO_PRIO) __builtin_amdgcn_s_setprio(3);
        const bool tr = it + 1 < iters && !TT_DBG(p, 4);
        if (!TT_DBG(p, 2)) copy_out_read();
        if (tr) transpose_read(it + 1);
        issue_group(it + NB);                                         // into the slot group `it` just left
        if (tl && tid == 0 && it == 10) tl[53] = __builtin_readcyclecounter();
        if (!TT_DBG(p, 2)) copy_out_store(it);
        else {
#pragma unroll
            for (int k = 0; k < NST; ++k) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, ro[0], TT_OOB, 0, 0);   // keeps the count
        }
        if (tl && tid == 0 && it == 10) tl[54] = __builtin_readcyclecounter();
        if (tr) transpose_write();
        if (TT_IO_PRIO) __builtin_amdgcn_s_setprio(0);
        if (tl && tid == 0 && it == 10) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tl[51] = __builtin_readcyclecounter(); }
    };
    // Anti-phase schedule: between two workgroup barriers team 0 computes group i while team 1 moves the data of its group i-1,
    // then team 0 moves the data of group i while team 1 computes its group i.  A phase the team has no group for is skipped; the
    // barriers are executed by everybody (p.iters_max + 1 pairs).
    if constexpr (SOLO) {
        for (int i = 0; i < iters; ++i) {
            wg_barrier();
            compute_phase(i);
            wg_barrier();
            io_phase(i);
        }
    } else {
        for (int i = 0; i <= p.iters_max; ++i) {
            wg_barrier();
            if (team == 0) { if (i < iters) compute_phase(i); }
            else if (i >= 1 && i - 1 < iters) io_phase(i - 1);
            wg_barrier();
            if (team == 0) { if (i < iters) io_phase(i); }
            else if (i < iters) compute_phase(i);
        }
    }
    if (tl && tid == 0) { tl[63] = __builtin_amdgcn_s_memrealtime(); tl[61] = __builtin_readcyclecounter(); }
    wait_vmcnt<0>();                                                 // nothing of this wave (an LDS-DMA into a slot nobody reads any more) may outlive it
    if (p.stats && has_work) {                                        // one partial row per wave
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) bs[k] += __shfl_xor(bs[k], o, 64);
        }
        if (lane == 0) {
            float* r = p.stats + (((size_t)slice * TT_WAVES + wave) * p.C + c) * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) r[k] = bs[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
static int team_class(int H, int W, int K) {                      // 2: planes of 2 x 2 tiles, 1: one tile, 0: not covered
    auto cls = [](int Wt) { return (Wt > 64 || Wt <= 16) ? 0 : (Wt > 32 ? 2 : 1); };
    const int a = cls(H), b = cls(W);
    if (a == 0 || a != b) return 0;
    if (K <= MF_TAPS || K > 63 || (K & 1) == 0) return 0;
    return a;
}

static bool fill_team_params(TeamParams& p, int N, int C, int H, int W, int K, bool dgrad, int cls, int resident_wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.dgrad = dgrad ? 1 : 0;
    const int HW = H * W, KS = cls == 2 ? 4 : 2;
    if (HW % 8 || W % 4 || H % 4) return false;
    if (K * MF_TAPS > TT_WCH * 64) return false;
    const int Wmin = H < W ? H : W;
    if (Wmin <= 16 * (KS - 2)) return false;                          // only the last two k-steps may reach past the plane edge
    p.G = cls == 2 ? 1 : 4;
    p.chunks_pp = HW / 8;
    p.ppp = (p.chunks_pp + 63) / 64;
    p.NT = dgrad ? 3 : 1;
    p.plane_lds = HW + 2 * W;
    p.tslot_elems = (p.G * (HW + 2 * W) + 2 * W + 7) & ~7;            // tensor parts stay 16-byte aligned
    p.t0_elems = dgrad ? ((p.G * HW + 7) & ~7) : p.tslot_elems;
    p.t0_plane = dgrad ? HW : p.plane_lds;
    p.t0_first = dgrad ? 0 : 2 * W;
    p.PT = KS * 16 + 8;
    p.xt_rows = W + 4;
    const int TC = p.G * p.chunks_pp;
    if (TC > TT_NCO * TT_THREADS || TC >= 1024 || p.chunks_pp >= 1024) return false;
    p.tr_cbs = (W + 15) / 16; p.tr_pp = (H / 4) * p.tr_cbs;
    if (p.G * p.tr_pp > TT_NTR * TT_WAVES * 4) return false;
    if ((size_t)p.G * p.xt_rows * p.PT * 2 >= 65535 || (size_t)p.tslot_elems * 2 >= 65535) return false;   // packed 16-bit transpose map
    int slices = resident_wgs / C; if (slices < 1) slices = 1;
    int per = (N + slices - 1) / slices; per = (per + p.G - 1) / p.G * p.G; if (per < p.G) per = p.G;
    p.planes_per_wg = per; p.slices = (N + per - 1) / per;
    p.m_cpp = p.G <= 1 ? 0u : (unsigned)(((1u << 22) + p.chunks_pp - 1) / p.chunks_pp);
    p.tensor_bytes = (unsigned)((size_t)N * C * HW * 2);
    // LDS-DMA pieces: q = (t * G + j) * ppp + pp -> wave q % 4, its q / 4-th piece
    const int npieces = p.NT * p.G * p.ppp;
    if ((npieces + TT_WAVES - 1) / TT_WAVES > (dgrad ? TT_NPW : 2)) return false;
    for (int w = 0; w < TT_WAVES; ++w) {
        p.my_pieces[w] = 0;
        for (int k = 0; k < TT_NPW; ++k) {
            TeamPiece& pc = p.pieces[w][k];
            const int q = w + TT_WAVES * k;
            if (q >= npieces) { pc.lds_off = 0; pc.g_off = 0; pc.info = 0; continue; }
            const int t = q / (p.G * p.ppp), rem = q - t * (p.G * p.ppp), j = rem / p.ppp, pp = rem - j * p.ppp;
            const int lanes = pp == p.ppp - 1 ? p.chunks_pp - 64 * pp : 64;
            pc.lds_off = t == 0 ? (unsigned)p.t0_first * 2u + (unsigned)j * (unsigned)p.t0_plane * 2u + (unsigned)pp * 1024u
                                : (unsigned)p.t0_elems * 2u + (unsigned)(t - 1) * (unsigned)p.tslot_elems * 2u + (unsigned)(2 * W) * 2u + (unsigned)j * (unsigned)p.plane_lds * 2u + (unsigned)pp * 1024u;
            pc.g_off = (unsigned)j * (unsigned)(C * HW) * 2u + (unsigned)pp * 1024u;
            pc.info = t | (j << 4) | (lanes << 8);
            ++p.my_pieces[w];
        }
    }
    return true;
}

static size_t team_lds_bytes(const TeamParams& p, int cls) {
    const int nob = p.dgrad ? (cls == 1 ? 1 : 2) : 3;
    const size_t outb = (size_t)nob * p.G * p.H * p.W * 2, win = (size_t)3 * 2 * MF_TAPS * TT_LEN * 2;
    return (size_t)p.NB * ((size_t)p.t0_elems + (size_t)(p.NT - 1) * p.tslot_elems) * 2 + 128 + (size_t)p.G * p.xt_rows * p.PT * 2 + (outb > win ? outb : win) + (size_t)TT_ZROW * 2 +
           (cls == 2 ? (size_t)TT_SFR_BYTES : 0) + 16;
}

// ring depth: as deep as two workgroups per CU allow (80 KB each), at most 6 (forward) / 3 (dgrad: three tensors per slot)
static bool team_pick_ring(TeamParams& p, int cls) {
    static const int forced = [] { const char* e = slak_dev_getenv("SLAK_TEAM_NB"); return e ? atoi(e) : 0; }();
    const int hi = p.dgrad ? 3 : 6;
    for (int nb = (forced >= 2 && forced <= 8) ? forced : hi; nb >= 2; --nb) {
        p.NB = nb;
        if (team_lds_bytes(p, cls) <= 80 * 1024) return true;
        if (forced >= 2) break;
    }
    return false;
}

bool dwconv_mfma_team_tri_supported(int N, int C, int H, int W, int K, int dtype, bool dgrad) {
    static const bool off = [] { const char* e = getenv("SLAK_TEAM_TRI"); return e && e[0] == '0'; }();
    if (off) return false;
    if (dtype != SLAK_BF16 && dtype != SLAK_F16) return false;
    if (N <= 0 || C <= 0 || (long long)N * C * H * W >= (1LL << 30)) return false;       // byte offsets stay below TT_OOB
    const int cls = team_class(H, W, K);
    if (!cls) return false;
    if (cls == 2 && W % 8 != 0) return false;                         // (two-half fragment reads do not fit this class's register budget)
    TeamParams p;
    if (!fill_team_params(p, N, C, H, W, K, dgrad, cls, 512)) return false;
    return team_pick_ring(p, cls);
}

int dwconv_mfma_team_tri_stats_rows(int N, int C, int H, int W, int K, int dtype) {
    if (dtype != SLAK_BF16 || !dwconv_mfma_team_tri_supported(N, C, H, W, K, dtype, false)) return 0;
    TeamParams p;
    fill_team_params(p, N, C, H, W, K, false, team_class(H, W, K), 2 * mfma_cu_count());
    return p.slices * TT_WAVES;
}

// which (op, class) runs one team per workgroup: bit 0 forward one-tile planes, 1 forward 2 x 2 tiles, 2 dgrad one-tile, 3 dgrad 2 x 2
static int team_solo_mask() {
    static const int m = [] { const char* e = slak_dev_getenv("SLAK_TEAM_SOLO"); return e ? atoi(e) : TT_SOLO_DEFAULT; }();
    return m;
}

template <typename T, int CLS, bool DGRAD, bool R16, bool SOLO>
static int launch_team_s(TeamParams& p, int N, int C, int H, int W, int K, hipStream_t st) {
    auto k = dwconv_mfma_team_tri_kernel<T, CLS, DGRAD, R16, SOLO>;
    fill_team_params(p, N, C, H, W, K, DGRAD, CLS, 2 * mfma_cu_count());
    if (!team_pick_ring(p, CLS)) return SLAK_ERR_UNSUPPORTED;
    p.team_lds = (int)((team_lds_bytes(p, CLS) + 15) & ~(size_t)15);
    p.iters_max = (p.planes_per_wg + p.G - 1) / p.G;
    const size_t lds = (size_t)(SOLO ? 1 : 2) * p.team_lds;
    static thread_local size_t cached_key = 0;                    // (device + 1, LDS size): the attribute is per device
    const size_t key = ((size_t)(slak_current_device() + 1) << 32) | lds;
    if (cached_key != key) {
        (void)slak_set_max_lds((const void*)k, lds);
        cached_key = key;
    }
    if (SOLO) hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(TT_THREADS), lds, st, p);
    else hipLaunchKernelGGL(k, dim3((unsigned)((p.C * p.slices + 1) / 2)), dim3(2 * TT_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, int CLS, bool DGRAD, bool R16>
static int launch_team_t(TeamParams& p, int N, int C, int H, int W, int K, hipStream_t st) {
    const int bit = (DGRAD ? 2 : 0) + (CLS == 2 ? 1 : 0);
    if ((team_solo_mask() >> bit) & 1) return launch_team_s<T, CLS, DGRAD, R16, true>(p, N, C, H, W, K, st);
    return launch_team_s<T, CLS, DGRAD, R16, false>(p, N, C, H, W, K, st);
}

template <typename T, int CLS>
static int launch_team_c(TeamParams& p, bool dgrad, bool r16, int N, int C, int H, int W, int K, hipStream_t st) {
    if (dgrad) return r16 ? launch_team_t<T, CLS, true, true>(p, N, C, H, W, K, st) : launch_team_t<T, CLS, true, false>(p, N, C, H, W, K, st);
    return r16 ? launch_team_t<T, CLS, false, true>(p, N, C, H, W, K, st) : launch_team_t<T, CLS, false, false>(p, N, C, H, W, K, st);
}

int launch_dwconv_mfma_team_tri(bool dgrad, const void* const* in, void* const* out, const float* const* w, int dtype,
                                int N, int C, int H, int W, int K, hipStream_t st, float* stats) {
    if (!dwconv_mfma_team_tri_supported(N, C, H, W, K, dtype, dgrad)) return SLAK_ERR_UNSUPPORTED;
    const int cls = team_class(H, W, K);
    TeamParams p;
    for (int b = 0; b < 3; ++b) { p.in[b] = in[b]; p.out[b] = out[b]; p.w[b] = w[b]; }
    p.stats = (stats && !dgrad && dtype == SLAK_BF16) ? stats : nullptr;
    p.dbg = team_dev_flags();
    p.tl = TT_DBG(p, 16) ? g_dma_dbg : nullptr;
    const bool r16 = W % 8 == 0;
    if (dtype == SLAK_BF16) return cls == 2 ? launch_team_c<bf16_t, 2>(p, dgrad, r16, N, C, H, W, K, st) : launch_team_c<bf16_t, 1>(p, dgrad, r16, N, C, H, W, K, st);
    return cls == 2 ? launch_team_c<f16_t, 2>(p, dgrad, r16, N, C, H, W, K, st) : launch_team_c<f16_t, 1>(p, dgrad, r16, N, C, H, W, K, st);
}

}  // namespace slak
