// slak_amd/csrc/dwconv_mfma_stream_tri.hip -- FORWARD of the three branches of a decomposed large-kernel block (K x 5, 5 x K, 5 x 5 on
// the same input: models/SLaK.py:82-100) in ONE launch on planes of 2 x 2 MFMA tiles (32 < H, W <= 64, W % 8 == 0: the 56 x 56 stage,
// 48 x 48 at 384 px): x travels HBM -> LDS once, three outputs are written.  Round 3.
//
// What the earlier one-launch kernels taught (tools/team_timeline.py, tools/mfma_chain_probe.hip):
//   * a wave's MFMA stream runs at 37-39 cycles per 32-cycle MFMA, so the matrix pipe needs TWO computing waves per SIMD;
//   * a separate data-movement phase (results LDS -> HBM, next plane's transpose, LDS-DMA requests) between workgroup barriers is never
//     hidden: two workgroups per CU fall into lockstep (115 us = 50 MFMA + 65 IO), two anti-phased teams hide it but leave one
//     computing wave per SIMD (126 us);
//   * the instruction ISSUE of that data movement is cheap (a few hundred cycles per plane) once it is spread out.
// Here there is no data-movement phase.  A workgroup is four waves, two workgroups per CU (two computing waves per SIMD), ONE workgroup
// barrier per plane.  Each wave computes three tiles per plane -- (mt, sub 0), (mt, sub 1) of the vertical (waves 0, 1) or horizontal
// (waves 2, 3) branch and the 5 x 5 tile (mt, sub = wave / 2) -- as one MFMA stream (B fragments prefetched across tile boundaries,
// three accumulators in turn), and everything else rides in the shadow of those MFMAs as fillers:
//   * a tile's epilogue (pack, ds_write into the wave's OWN 32 x 32 staging tile) and its HBM stores (ds_read 16-byte pieces of the
//     staging tile back, buffer_store) are fillers of the NEXT tile -- no out-buffer is shared between waves, so results need no barrier
//     (a tile row is 64 / 48 bytes of an image row: the two halves of a 128-byte line come from two waves a few hundred cycles apart and
//     meet in L2);
//   * this wave's share of the next plane's transpose (ds_read_b64_tr_b16 + ds_write_b64 into the other x^T buffer) and its LDS-DMA
//     pieces of plane i + 2 are fillers too.
// Shared between the waves are only the landed input plane and x^T: the barrier at the end of plane i publishes plane i + 2's LDS-DMA
// (each wave waits for its own pieces with a counted vmcnt) and plane i + 1's transpose.  The 5 x 5 branch's Toeplitz fragments live in
// LDS (team_small_tile_mma): 80 fragment registers per wave instead of 140.
// The data gradient (three inputs, ONE output that sums tiles of different waves) keeps the phase structure of
// dwconv_mfma_team_tri.hip.
#include "team_common.h"

namespace slak {

constexpr int ST_PITCH = 80;            // bytes per row of a 32 x 32 staging tile (64 + 16: conflict-free 8-byte epilogue writes, 16-byte aligned rows)
constexpr int ST_TILE = 32 * ST_PITCH;

template <typename T>
__global__ __launch_bounds__(TT_THREADS, 2) void dwconv_mfma_stream_tri_kernel(const TeamParams p) {
    constexpr int KS = 4;
    constexpr bool R16 = true;
    constexpr int NB = 3;                                            // ring slots: plane i (read), i + 1 (being transposed), i + 2 (in flight)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;
    const int HW = p.H * p.W;
    const unsigned slot_b = (unsigned)p.tslot_elems * 2;
    const unsigned xt_buf_b = (unsigned)(p.xt_rows * p.PT) * 2;
    const unsigned ring_b = 0;                                       // NB slots (+ 128 bytes slack)
    const unsigned xt_b = ring_b + NB * slot_b + 128;                 // 2 x [xt_rows][PT]
    const unsigned stg_b = xt_b + 2 * xt_buf_b;                      // [4 waves][2][32][ST_PITCH]
    const unsigned zrow_b = stg_b + TT_WAVES * 2 * ST_TILE;          // TT_ZROW zeros
    const unsigned sfr_b = zrow_b + TT_ZROW * 2;                     // the 5 x 5 branch's twenty Toeplitz fragments
    const unsigned win_b = stg_b;                                    // [3 branches][2 copies][5 taps][TT_LEN]: prologue only, aliases the staging tiles
    constexpr unsigned win1_bytes = 2 * MF_TAPS * TT_LEN * 2;
    static_assert(3 * win1_bytes <= TT_WAVES * 2 * ST_TILE, "filter windows alias the staging tiles");

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    if (n_begin >= n_end) return;
    const int iters = n_end - n_begin;

    // ---- LDS-DMA: piece q (64 chunks of 16 bytes) of a plane is issued by wave q % 4 -------------------------------------------------
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)p.in[0];
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes);
        rsrc[3] = 0x00020000;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;              // HBM bytes from image n to image n+1 of this channel
    const TeamPiece pc0 = p.pieces[wave][0], pc1 = p.pieces[wave][1];
    const int my_pieces = p.my_pieces[wave];
    auto issue_piece = [&](const TeamPiece& pc, int g) {             // always issued (a plane beyond the slice: zeros)
        const int nl = pc.info >> 8;
        if (nl == 0) return;                                         // wave-uniform
        const int n0 = n_begin + g;
        const unsigned voff = (n0 < n_end) ? (unsigned)(((size_t)n0 * p.C + c) * HW * 2) + pc.g_off + (unsigned)lane * 16u : TT_OOB;
        const unsigned m0 = __builtin_amdgcn_readfirstlane(lds_base + ring_b + (unsigned)(g % NB) * slot_b + pc.lds_off);
        if (lane < nl) lds_dma16(voff, rsrc, m0);
    };

    // ---- prologue: planes 0, 1 requested, zero areas, filter windows, fragments ---------------------------------------------------
    issue_piece(pc0, 0); issue_piece(pc1, 0); issue_piece(pc0, 1); issue_piece(pc1, 1);
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
        for (unsigned o = tid * 16; o < 2 * xt_buf_b; o += TT_THREADS * 16) *(u32x4*)(L + xt_b + o) = z4;     // x^T guard rows / pad columns
        for (int q = wave; q < NB * 2; q += TT_WAVES) {              // ring: 2 guard rows in front of the plane, 2 behind
            const unsigned gb = ring_b + (unsigned)(q >> 1) * slot_b + (q & 1) * (unsigned)p.plane_lds * 2;
            for (int o = lane; o < p.W; o += 64) *(unsigned*)(L + gb + o * 4) = 0u;       // 2W elements = W dwords
        }
    }
    wg_barrier();
    if (wave < 3) {
        const bool vert = wave == 0;
        const int kw = wave == 1 ? KL : MF_TAPS;
#pragma unroll
        for (int k = 0; k < TT_WCH; ++k) {
            const int e = lane + 64 * k;
            if (e < st_ntap) {
                const int r = vert ? e % MF_TAPS : e / kw, t = vert ? e / MF_TAPS : e - (e / kw) * kw;      // short tap r, long tap t
                const uint16_t v = cvt_to_bits(wreg[k], (T*)nullptr);
                uint16_t* win = (uint16_t*)(L + win_b + wave * win1_bytes);
                win[r * TT_LEN + TT_ZP + t] = v;                                         // copy 0
                win[MF_TAPS * TT_LEN + r * TT_LEN + TT_ZP + t - 1] = v;                  // copy 1 = copy 0 shifted by one element
            }
        }
    }
    wg_barrier();
    const int mt = wave & 1, g2 = wave >> 1;                         // A = vertical (g2 == 0) / horizontal; 5 x 5 tile (mt, sub = g2)
    const bool a_vert = g2 == 0;
    s16x8 fragA[MF_TAPS][KS];
    team_build_frags<KS>(fragA, L + win_b + (a_vert ? 0u : win1_bytes), 0, mt, l31, lhi, a_vert ? p.H : p.W, KL / 2);
    {                                                                 // wave w builds the five LDS fragments of d = w - 1 of the 5 x 5 branch
        const int a = TT_ZP + 16 * (wave - 1) + lhi * 8 - l31 + MF_TAPS / 2;
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

    // ---- per-thread constants -------------------------------------------------------------------------------------------------------
    const unsigned pitch_v = (unsigned)p.PT * 2, pitch_h = (unsigned)p.W * 2;
    const unsigned a_pitch = a_vert ? pitch_v : pitch_h;
    const int wlim = p.W - lhi * 8;
    const unsigned zrow_l = zrow_b + lhi * 16;
    const unsigned relA = (unsigned)(l31 * (a_vert ? p.PT : p.W)) * 2 + lhi * 16;          // operand rows of A tile sub 0; sub 1 is 32 rows further
    const unsigned relS = (unsigned)((g2 * 32 + l31) * p.W) * 2 + lhi * 16;
    const unsigned sfr_l = sfr_b + lane * 16;
    // epilogue into this wave's staging tiles: lane = tile row l31, register quad q = tile columns 4 lhi + 8 q .. + 3
    const unsigned stg_w = stg_b + (unsigned)wave * 2 * ST_TILE, stg_wr = (unsigned)l31 * ST_PITCH + lhi * 8;
    // HBM side of a tile: thread -> (tile row lane / 4 + 16 k, 16-byte piece lane % 4).  Tile (row0, col0): vertical (mt, sub): (32 mt, 32 sub);
    // horizontal / small (mt, sub): (32 sub, 32 mt).  goff[t][k]: byte offset in the plane, or TT_OOB outside the plane
    const unsigned stg_rd = (unsigned)(lane >> 2) * ST_PITCH + (lane & 3) * 16;
    unsigned goff[3][2];
    {
        const int row0[3] = {a_vert ? mt * 32 : 0, a_vert ? mt * 32 : 32, g2 * 32};
        const int col0[3] = {a_vert ? 0 : mt * 32, a_vert ? 32 : mt * 32, mt * 32};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int row = row0[t] + (lane >> 2) + 16 * k, col = col0[t] + (lane & 3) * 8;
                goff[t][k] = (row < p.H && col < p.W) ? (unsigned)(row * p.W + col) * 2 : TT_OOB;
            }
    }
    __amdgpu_buffer_rsrc_t ro[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) ro[t] = __builtin_amdgcn_make_buffer_rsrc(p.out[t], 0, (int)p.tensor_bytes, 0x00020000);
    // transposes: 16-lane group (wave, lane / 16) takes blocks b = 16 k + 4 wave + lane / 16 (4 image rows kb, 16 image columns cb)
    unsigned tr_map[TT_NTR];
    {
        const int grp = lane >> 4, i16 = lane & 15;
#pragma unroll
        for (int k = 0; k < TT_NTR; ++k) {
            const int b = (k * TT_WAVES + wave) * 4 + grp;
            const bool ok = b < p.tr_pp;
            const int kb = ok ? b / p.tr_cbs : 0, cb = ok ? b - kb * p.tr_cbs : 0;
            const unsigned src = (unsigned)(2 * p.W + (kb * 4 + (i16 >> 2)) * p.W + cb * 16 + (i16 & 3) * 4) * 2;
            const unsigned dst = (cb * 16 + i16 < p.W) ? (unsigned)((2 + cb * 16 + i16) * p.PT + kb * 4) * 2 : 0xffffu;
            tr_map[k] = ok ? (src | (dst << 16)) : 0xffffffffu;
        }
    }
    float bs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                    // p.stats: sums over the pieces this thread stores (branch of tile t: A, A, small)
    auto stat8 = [&](const u32x4& v, float& s1, float& s2) {
        if constexpr (std::is_same<T, bf16_t>::value) {
            const bf16x2_t one = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            const unsigned d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
            const bf16x2_t x0 = __builtin_bit_cast(bf16x2_t, d0), x1 = __builtin_bit_cast(bf16x2_t, d1), x2 = __builtin_bit_cast(bf16x2_t, d2), x3 = __builtin_bit_cast(bf16x2_t, d3);
            float a0 = __builtin_amdgcn_fdot2_f32_bf16(x0, one, 0.f, false), a1 = __builtin_amdgcn_fdot2_f32_bf16(x1, one, 0.f, false);
            float a2 = __builtin_amdgcn_fdot2_f32_bf16(x2, one, 0.f, false), a3 = __builtin_amdgcn_fdot2_f32_bf16(x3, one, 0.f, false);
            float q0 = __builtin_amdgcn_fdot2_f32_bf16(x0, x0, 0.f, false), q1 = __builtin_amdgcn_fdot2_f32_bf16(x1, x1, 0.f, false);
            float q2 = __builtin_amdgcn_fdot2_f32_bf16(x2, x2, 0.f, false), q3 = __builtin_amdgcn_fdot2_f32_bf16(x3, x3, 0.f, false);
            asm("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            s1 += (a0 + a1) + (a2 + a3); s2 += (q0 + q1) + (q2 + q3);
        }
    };

    // planes 0 and 1 landed (every wave: its own pieces), plane 0 transposed
    wait_vmcnt<0>();
    wg_barrier();
#pragma unroll
    for (int k = 0; k < TT_NTR; ++k) {
        if (tr_map[k] != 0xffffffffu) {
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + ring_b + (tr_map[k] & 0xffffu)));
            if ((tr_map[k] >> 16) != 0xffffu) *(s16x4*)(L + xt_b + (tr_map[k] >> 16)) = v;
        }
    }
    wg_barrier();

    // ---- the stream -------------------------------------------------------------------------------------------------------------------
    // Tile sequence number q = 3 i + t (t = 0, 1: A tiles; 2: 5 x 5); tile q's results go through staging tile q & 1 while tile q + 1 runs.
    s16x8 bq[TT_NBUF], sa[TT_NBUF];
    f32x16 accA, accB, accS;                                         // tile t = 0 / 1 / 2 of a plane
    auto zero = [](f32x16& a) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 0.f;
    };
    // fillers of the tile that FOLLOWS tile (plane pi, t): j = 2..5 pack + ds_write of the four register quads; j = 7, 8 read two 16-byte pieces
    // back; j = 11, 12 store them.  pend = false: nothing pending (first tile of the launch)
    u32x4 r0, r1;
    auto finish = [&](const f32x16& acc, int t, int par, bool pend, unsigned gb, int j) __attribute__((always_inline)) {
        // (pend == false -- the launch's first tile has no predecessor -- still issues its two stores, out of range: the counted wait
        // at the end of a plane relies on 6 stores per plane)
        const unsigned sb = stg_w + (unsigned)par * ST_TILE;
        if (j >= 2 && j < 6) {
            const int q = j - 2;
            u32x2 v;
            v[0] = pack2<T>(acc[4 * q + 0], acc[4 * q + 1]);
            v[1] = pack2<T>(acc[4 * q + 2], acc[4 * q + 3]);
            *(u32x2*)(L + sb + stg_wr + 16 * q) = v;
        } else if (j == 7) r0 = *(const u32x4*)(L + sb + stg_rd);
        else if (j == 8) r1 = *(const u32x4*)(L + sb + stg_rd + 16 * ST_PITCH);
        else if (j == 11 || j == 12) {
            const int k = j - 11;
            const u32x4 v = k ? r1 : r0;
            const unsigned go = (!pend || goff[t][k] == TT_OOB || TT_DBG(p, 2)) ? TT_OOB : gb + goff[t][k];
            const int br = t == 2 ? 2 : (a_vert ? 0 : 1);            // output tensor of the tile
            if (p.stats && go != TT_OOB) { if (br == 0) stat8(v, bs[0], bs[1]); else if (br == 1) stat8(v, bs[2], bs[3]); else stat8(v, bs[4], bs[5]); }
            if (br == 0) __builtin_amdgcn_raw_buffer_store_b128(v, ro[0], go, 0, 0);
            else if (br == 1) __builtin_amdgcn_raw_buffer_store_b128(v, ro[1], go, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, ro[2], go, 0, 0);
        }
    };
    s16x4 tv0, tv1;
    auto transposes = [&](int k0, int pl, int j) __attribute__((always_inline)) {      // blocks k0, k0 + 1 of plane pl: reads at j = 14, 15, writes at j = 17, 18
        if (pl >= iters || TT_DBG(p, 4)) return;
        const unsigned sb = ring_b + (unsigned)(pl % NB) * slot_b, db = xt_b + (unsigned)(pl & 1) * xt_buf_b;
        if (j == 14 && tr_map[k0] != 0xffffffffu) tv0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + sb + (tr_map[k0] & 0xffffu)));
        if (j == 15 && tr_map[k0 + 1] != 0xffffffffu) tv1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + sb + (tr_map[k0 + 1] & 0xffffu)));
        if (j == 17 && tr_map[k0] != 0xffffffffu && (tr_map[k0] >> 16) != 0xffffu) *(s16x4*)(L + db + (tr_map[k0] >> 16)) = tv0;
        if (j == 18 && tr_map[k0 + 1] != 0xffffffffu && (tr_map[k0 + 1] >> 16) != 0xffffu) *(s16x4*)(L + db + (tr_map[k0 + 1] >> 16)) = tv1;
    };
    auto run = [&](auto av_c, auto mt_c) __attribute__((always_inline)) {
        constexpr bool AV = decltype(av_c)::value;
        constexpr int MT = decltype(mt_c)::value;
        constexpr int NA = 1 + !AV;                                   // NXT code of an A tile: 1 = SWAP (vertical), 2 = plain
        constexpr int R1 = (MF_TAPS * KS) % TT_NBUF, R2 = (2 * MF_TAPS * KS) % TT_NBUF;      // B-buffer rotation of tiles 1, 2 (tile 0 starts at 0 every plane)
        for (int i = 0; i < iters; ++i) {
            const unsigned img_b = ring_b + (unsigned)(i % NB) * slot_b;
            const unsigned a_rp = (AV ? xt_b + (unsigned)(i & 1) * xt_buf_b : img_b) + relA, a_rp1 = a_rp + 32u * a_pitch;
            const unsigned s_rp = img_b + relS;
            const unsigned gb = (unsigned)(((size_t)(n_begin + i) * p.C + c) * HW * 2), gbp = gb - gplane_b;
            const bool first = i == 0;
            // the plane's first five B fragments (x^T / the landed image were published by the barrier that ended the previous plane)
            team_tile_prefetch<AV, R16, KS, 0, 0>(bq, L, a_rp, a_pitch, wlim, zrow_l);
            // ---- tile 0: A(MT, sub 0).  Fillers: the LDS-DMA pieces of plane i + 2, the previous plane's 5 x 5 tile, two transpose blocks
            zero(accA);
            auto f0 = [&](int j) __attribute__((always_inline)) {
                if (j == 0 && !TT_DBG(p, 64)) issue_piece(pc0, i + 2);
                if (j == 1 && !TT_DBG(p, 64)) issue_piece(pc1, i + 2);
                finish(accS, 2, 0, !first, gbp, j);
                transposes(0, i + 1, j);
            };
            team_tile_mma<T, AV, R16, KS, KS, 0, 0, NA, 0>(accA, fragA, bq, L, a_rp, a_pitch, wlim, zrow_l, a_rp1, a_pitch, f0);
            // ---- tile 1: A(MT, sub 1).  Fillers: tile 0's results, the other two transpose blocks
            team_small_prefetch<MT>(sa, L, sfr_l);
            zero(accB);
            auto f1 = [&](int j) __attribute__((always_inline)) { finish(accA, 0, 0, true, gb, j); transposes(2, i + 1, j); };
            team_tile_mma<T, AV, R16, KS, KS, 0, R1, 2, MT>(accB, fragA, bq, L, a_rp1, a_pitch, wlim, zrow_l, s_rp, pitch_h, f1);
            // ---- tile 2: S(MT, sub = g2).  Fillers: tile 1's results
            zero(accS);
            auto f2 = [&](int j) __attribute__((always_inline)) { finish(accB, 1, 1, true, gb, j); };
            team_small_tile_mma<T, R16, KS, MT, R2, 0, 0>(accS, sa, bq, L, s_rp, pitch_h, wlim, zrow_l, sfr_l, 0u, 0u, f2);
            // ---- end of plane i: this wave's pieces of plane i + 2 have landed (the 6 stores are younger); publish them and plane i + 1's transpose
            if (!TT_DBG(p, 8)) wait_vmcnt<6>();
            if (!TT_DBG(p, 32)) wg_barrier();
        }
        {                                                             // the last plane's 5 x 5 tile
            const unsigned gb = (unsigned)(((size_t)(n_begin + iters - 1) * p.C + c) * HW * 2);
#pragma unroll
            for (int j = 2; j <= 12; ++j) finish(accS, 2, 0, true, gb, j);
        }
    };
    using std::integral_constant;
    if (a_vert) { if (mt == 0) run(integral_constant<bool, true>{}, integral_constant<int, 0>{}); else run(integral_constant<bool, true>{}, integral_constant<int, 1>{}); }
    else { if (mt == 0) run(integral_constant<bool, false>{}, integral_constant<int, 0>{}); else run(integral_constant<bool, false>{}, integral_constant<int, 1>{}); }
    wait_vmcnt<0>();                                                 // no LDS-DMA of this wave may outlive it
    if (p.stats) {                                                    // one partial row per wave
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
static bool fill_stream_params(TeamParams& p, int N, int C, int H, int W, int K, int resident_wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.dgrad = 0;
    const int HW = H * W;
    if (H <= 32 || H > 64 || W <= 32 || W > 64 || W % 8 || H % 4) return false;
    if (K <= MF_TAPS || K > 63 || (K & 1) == 0 || K * MF_TAPS > TT_WCH * 64) return false;
    p.G = 1; p.NT = 1; p.NB = 3;
    p.chunks_pp = HW / 8;
    p.ppp = (p.chunks_pp + 63) / 64;
    if (p.ppp > 2 * TT_WAVES) return false;                           // two pieces per wave
    p.plane_lds = HW + 2 * W;
    p.tslot_elems = (HW + 4 * W + 7) & ~7;
    p.t0_elems = p.tslot_elems; p.t0_plane = p.plane_lds; p.t0_first = 2 * W;
    p.PT = 4 * 16 + 8;
    p.xt_rows = W + 4;
    p.tr_cbs = (W + 15) / 16; p.tr_pp = (H / 4) * p.tr_cbs;
    if (p.tr_pp > TT_NTR * TT_WAVES * 4) return false;
    if ((size_t)p.xt_rows * p.PT * 2 >= 65535 || (size_t)p.tslot_elems * 2 >= 65535) return false;
    int slices = resident_wgs / C; if (slices < 1) slices = 1;
    int per = (N + slices - 1) / slices; if (per < 1) per = 1;
    p.planes_per_wg = per; p.slices = (N + per - 1) / per;
    p.iters_max = per; p.m_cpp = 0u;
    p.tensor_bytes = (unsigned)((size_t)N * C * HW * 2);
    for (int w = 0; w < TT_WAVES; ++w) {
        p.my_pieces[w] = 0;
        for (int k = 0; k < TT_NPW; ++k) {
            TeamPiece& pc = p.pieces[w][k];
            const int q = w + TT_WAVES * k;
            if (k >= 2 || q >= p.ppp) { pc.lds_off = 0; pc.g_off = 0; pc.info = 0; continue; }
            const int lanes = q == p.ppp - 1 ? p.chunks_pp - 64 * q : 64;
            pc.lds_off = (unsigned)(2 * W) * 2u + (unsigned)q * 1024u;
            pc.g_off = (unsigned)q * 1024u;
            pc.info = lanes << 8;
            ++p.my_pieces[w];
        }
    }
    return true;
}

static size_t stream_lds_bytes(const TeamParams& p) {
    return (size_t)3 * p.tslot_elems * 2 + 128 + (size_t)2 * p.xt_rows * p.PT * 2 + (size_t)TT_WAVES * 2 * ST_TILE + (size_t)TT_ZROW * 2 + TT_SFR_BYTES + 16;
}

bool dwconv_mfma_stream_tri_supported(int N, int C, int H, int W, int K, int dtype) {
    static const bool off = [] { const char* e = getenv("SLAK_STREAM_TRI"); return e && e[0] == '0'; }();
    if (off) return false;
    if (dtype != SLAK_BF16 && dtype != SLAK_F16) return false;
    if (N <= 0 || C <= 0 || (long long)N * C * H * W >= (1LL << 30)) return false;
    TeamParams p;
    if (!fill_stream_params(p, N, C, H, W, K, 512)) return false;
    return stream_lds_bytes(p) <= 80 * 1024;
}

int dwconv_mfma_stream_tri_stats_rows(int N, int C, int H, int W, int K, int dtype) {
    if (dtype != SLAK_BF16 || !dwconv_mfma_stream_tri_supported(N, C, H, W, K, dtype)) return 0;
    TeamParams p;
    fill_stream_params(p, N, C, H, W, K, 2 * mfma_cu_count());
    return p.slices * TT_WAVES;
}

template <typename T>
static int launch_stream_t(TeamParams& p, int N, int C, int H, int W, int K, hipStream_t st) {
    auto k = dwconv_mfma_stream_tri_kernel<T>;
    fill_stream_params(p, N, C, H, W, K, 2 * mfma_cu_count());
    const size_t lds = stream_lds_bytes(p);
    static thread_local size_t cached_key = 0;                    // (device + 1, LDS size): the attribute is per device
    const size_t key = ((size_t)(slak_current_device() + 1) << 32) | lds;
    if (cached_key != key) {
        (void)slak_set_max_lds((const void*)k, lds);
        cached_key = key;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(TT_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_stream_tri(const void* x, void* const* out, const float* const* w, int dtype,
                                  int N, int C, int H, int W, int K, hipStream_t st, float* stats) {
    if (!dwconv_mfma_stream_tri_supported(N, C, H, W, K, dtype)) return SLAK_ERR_UNSUPPORTED;
    TeamParams p;
    for (int b = 0; b < 3; ++b) { p.in[b] = x; p.out[b] = out[b]; p.w[b] = w[b]; }
    p.stats = (stats && dtype == SLAK_BF16) ? stats : nullptr;
    p.dbg = team_dev_flags();
    p.tl = nullptr;
    return dtype == SLAK_BF16 ? launch_stream_t<bf16_t>(p, N, C, H, W, K, st) : launch_stream_t<f16_t>(p, N, C, H, W, K, st);
}

}  // namespace slak
