// slak_amd/csrc/linear_gemm.hip -- the block's pointwise Linear layers on stages 2-4 (reference: models/SLaK.py:156-165: pwconv1 -> GELU -> pwconv2 on
// the NHWC activation, and their data gradients), as ONE launch per GEMM with the elementwise step in the epilogue:
//     EPI_BIAS   out[M][N]  = bf16(A[M][K] . B[N][K]^T + bias)                                   (pwconv2, dy1 . W1)
//     EPI_GELU   out        = bf16(A . B^T + bias),  out2 = GELU(out)                              (pwconv1 + nn.GELU: both are kept for the backward)
//     EPI_DGELU  out        = bf16(bf16(A . B^T) * gelu'(y1)),  dbias[N] = column sums of out      (dz . W2, GELU', pwconv1's bias gradient)
// A, B, out, out2, y1 bf16 row-major, fp32 accumulation; K in {192, 256, 384} with N a multiple of 256 (the B fragments of a wave's 32 columns stay in its
// registers: K / 4 of them), K in {512, 768} with N a multiple of 128 (two teams of four waves share K), any M >= 1.  (EPI_BIAS -- the plain GEMMs with K = 4C -- stays with the library: 0.85 PFLOP/s there, and B does not fit registers.)
//
// Why an own GEMM: at K = 192 ... 768 the library's kernels spend a tile's time in its prologue and epilogue (3 ... 12 k-iterations per 256 x 256 tile: 0.66
// PFLOP/s measured), and the GELU / GELU' passes that follow move the [M][4C] intermediate through HBM twice more.  These GEMMs are WRITE-bound: 29.6 GFLOP
// against 77 ... 308 MB of output, so the kernel is built around the store stream, not around the matrix pipe:
//   * the operand traffic decides: with both operands streamed through LDS per 128 x 128 tile (the first versions) the LDS-DMA stream alone took 35 us
//     (462 MB of L2 -> LDS traffic per call at 13 TB/s).  So B NEVER MOVES: a workgroup is eight waves, wave w owns 32 columns of a 256-column panel for ALL k and
//     keeps their K / 16 B fragments in K / 4 registers for the whole launch; only A streams (once per panel: 115 MB instead of 462);
//   * A has K contiguous: 64-k chunks of the tile's 128 rows arrive by LDS-DMA into a 3-stage ring whose 16-byte slots are XOR-swizzled on the source side
//     (no padding), two chunks in flight beside the one being multiplied -- also ACROSS tiles: the next tile's first chunks go out before this tile's
//     stores, and every wait is a counted vmcnt (vector-memory operations retire in issue order), so no tile waits for another tile's stores; fragments are
//     plain ds_read_b128, conflict-free under the swizzle, fetched one k-step ahead (pinned with sched_barrier);
//   * operands are swapped (D^T = B . A^T) so that a lane holds FOUR consecutive output columns of one row: bias, rounding, GELU by table and packing happen
//     on register pairs, the tile goes through a per-wave LDS staging tile and leaves as 16-byte stores of full 128-byte lines;
//   * a workgroup walks the row tiles of a SLAB for one 128-column panel (persistent): bias and the per-lane column partials of EPI_DGELU stay in registers
//     over the slab, one fixed-order cross-lane reduction and one partial row per workgroup at the end (deterministic: slabs are added in order).
#include <math.h>
#include <string.h>

#include <mutex>
#include <type_traits>
#include <vector>

#include "slak_common.h"
#include "mfma_common.h"
#include "gelu_grad.h"

namespace slak {

constexpr int LG2_SP = 80;                 // staging pitch: 32 columns x 2 B + 16
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_DGELU = 2 };

// forward GELU table (bf16 -> bf16, linear_skinny.hip documents it; the same entries)
constexpr unsigned G2_LO = 109u << 7, G2_N = 22u << 7;
constexpr int G2_BYTES = 2 * (int)G2_N * 2;
__device__ __forceinline__ unsigned g2_lut(const uint16_t* __restrict__ T, unsigned b) {
    const unsigned mag = b & 0x7fffu, neg = b >> 15;
    const unsigned idx = mag - G2_LO;
    const bool in = idx < G2_N;
    const unsigned t = T[(in ? idx : 0u) + neg * G2_N];
    const unsigned small = mag >= 0x100u ? b - 0x80u : (b & 0x8000u);
    const unsigned big = neg ? (mag > 0x7f7fu ? (b | 0x40u) : 0x8000u) : b;
    return in ? t : (mag < G2_LO ? small : big);
}
__device__ __forceinline__ void g2_lut2x8(const uint16_t* __restrict__ T, const unsigned (&y)[8], unsigned (&g)[8]) {
    unsigned il[8], ih[8];
    bool out = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        il[k] = (y[k] & 0x7fffu) - G2_LO; ih[k] = ((y[k] >> 16) & 0x7fffu) - G2_LO;
        out = out || il[k] >= G2_N || ih[k] >= G2_N;
    }
    if (__builtin_amdgcn_ballot_w64(out) == 0) {
        unsigned lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { lo[k] = T[il[k] + ((y[k] >> 15) & 1u) * G2_N]; hi[k] = T[ih[k] + (y[k] >> 31) * G2_N]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = lo[k] | (hi[k] << 16);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = g2_lut(T, y[k] & 0xffffu) | (g2_lut(T, y[k] >> 16) << 16);
    }
}

struct Lg2Params {
    const uint16_t* a; const uint16_t* b; const uint16_t* bias; uint16_t* out; uint16_t* out2; const uint16_t* y1; float* part; const void* table;
    int M, N, K, tiles_m, panels, slabs, tps;          // tps = row tiles per slab
    int xcd_map;                                       // 1: the panels of a slab share an XCD (slabs in whole groups of eight); 0: plain (slab, panel) order
#ifdef SLAK_LG2_DEV
    int dbg;                                           // dev builds only (SLAK_BUILD_DEFS=-DSLAK_LG2_DEV): 1 no DMA, 2 no fragment reads / MFMAs, 4 no epilogue
#endif
};
#ifdef SLAK_LG2_DEV
#define LG2_DBG(bit) (p.dbg & (bit))
#else
#define LG2_DBG(bit) 0
#endif

constexpr int LG2_TM = 128;                            // tile rows
constexpr int LG2_WAVES = 8;                           // a wave owns 32 output columns for ALL its k: its B fragments never leave its registers
constexpr int LG2_NS = 3;                              // ring stages of the A stream (two chunks in flight beside the one being multiplied)
constexpr int LG2_STAGE = LG2_TM * 128;                // 16 KB per stage: 128 rows x 64 k (SPLIT 1) or 2 k-halves x 128 rows x 32 k (SPLIT 2), XOR-swizzled slots
constexpr int LG2_NPW = LG2_STAGE / 1024 / LG2_WAVES;  // 1-KiB DMA pieces per wave and chunk
static_assert(LG2_STAGE % (1024 * LG2_WAVES) == 0, "whole pieces, the same number per wave");

// SPLIT 1 (K <= 384): eight waves x 32 columns = a 256-column panel, every wave multiplies the whole K.
// SPLIT 2 (K = 512, 768: the B fragments of a whole K do not fit a wave's registers): two teams of four waves take the two HALVES of K for the same
//   128-column panel (wave w: columns 32 (w % 4), k-half w / 4); a ring stage holds both halves' chunk; at the end of a tile the teams swap two of their
//   four row tiles' partial sums through LDS, so each wave finishes (adds, epilogue) two row tiles of its 32 columns.
template <int EPI, int KS, int SPLIT>                  // KS = K / 16 MFMA k-steps
__global__ __launch_bounds__(512, 1) void linear_gemm_kernel(const Lg2Params p) {
    constexpr int TM = LG2_TM, RM = TM / 32;           // every wave multiplies all RM row tiles with its own 32 columns
    constexpr int KST = KS / SPLIT;                    // k-steps a wave multiplies (B fragments held: 4 KST registers)
    constexpr int KC = SPLIT == 2 ? 32 : 64;           // k per chunk and k-half
    constexpr int SLOTS = KC / 8;                      // 16-byte slots per row of a chunk
    constexpr int TN = 256 / SPLIT;                    // panel columns
    constexpr int RE = RM / SPLIT;                     // row tiles a wave finishes
    constexpr int NPW = LG2_NPW;
    constexpr int NK = KST * 16 / KC;                  // chunks per tile
    constexpr int TBL = EPI == EPI_GELU ? G2_BYTES : GD_BYTES;
    constexpr int NE = EPI == EPI_GELU ? 2 * RE * 2 : RE * 2;      // epilogue stores per wave and tile
    constexpr int NY = EPI == EPI_DGELU ? RE * 2 : 0;              // y1 loads per wave and tile
    static_assert(NK >= 3 && KS % SPLIT == 0 && KST * 16 % KC == 0, "three ring stages, whole chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem + TBL;
    unsigned char* const stg = ring + LG2_NS * LG2_STAGE + wave_id_uniform() * (32 * LG2_SP);     // per-wave staging tile: 32 rows x (32 columns + pad)
    float* const exch = (float*)(ring + LG2_NS * LG2_STAGE + LG2_WAVES * 32 * LG2_SP);            // SPLIT 2: [8 waves][2 row tiles][16][64] partial sums
    const unsigned lds0 = (unsigned)(uintptr_t)SLAK_LDS(unsigned char, ring);
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id_uniform();
    const int l31 = lane & 31, lhi = lane >> 5;
    const int team = SPLIT == 2 ? wave >> 2 : 0, cgrp = SPLIT == 2 ? (wave & 3) : wave;

    // workgroup -> (slab, panel): the panels of a slab run on one XCD (they stream the same rows of A through its L2); the grid holds
    // ceil8(slabs) x panels workgroups, the ones behind the last slab have nothing to do
    const int xcd = blockIdx.x & 7, rr0 = blockIdx.x >> 3;
    const int slab = p.xcd_map ? xcd + 8 * (rr0 / p.panels) : (int)blockIdx.x / p.panels;
    const int panel = p.xcd_map ? rr0 % p.panels : (int)blockIdx.x % p.panels;
    if (slab >= p.slabs) return;
    for (int i = tid; i < TBL / 16; i += 512) ((u32x4*)smem)[i] = ((const u32x4*)p.table)[i];
    const int t_begin = slab * p.tps, t_end = min(t_begin + p.tps, p.tiles_m);
    const int col0 = panel * TN + cgrp * 32;           // this wave's first output column
    const int khalf = team * (KST * 16);               // first k of this wave's share

    // this wave's B fragments: row n = col0 + l31 of b, k = khalf + 16 ks + 8 lhi .. + 8
    s16x8 bfrag[KST];
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) bfrag[ks] = *(const s16x8*)(p.b + (size_t)(col0 + l31) * p.K + khalf + ks * 16 + lhi * 8);

    // swizzle: a row's 16-byte chunk cc lives in slot cc ^ swz(row): with swz = (row >> 1) & 7 (8 slots, 128-byte rows) resp. (row >> 2) & 3 (4 slots,
    // 64-byte rows) the fragment reads (one row per lane) are conflict-free in every 16-lane group of ds_read_b128
    auto swz = [](int r) { return SLOTS == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    // DMA plan (loop invariant): piece pi = wave + 8 k covers destination slots q = 64 pi + lane -> (k-half, tile row, slot)
    unsigned psrc[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int q = 64 * (wave + LG2_WAVES * k) + lane;
        const int half = q / (TM * SLOTS), qq = q - half * (TM * SLOTS), r = qq / SLOTS, cc = (qq % SLOTS) ^ swz(r);
        psrc[k] = (unsigned)r * (unsigned)p.K * 2u + (unsigned)(half * KST * 16 * 2) + (unsigned)cc * 16u;
    }
    auto sgpr4 = [](v4i_t v) { return v4i_t{__builtin_amdgcn_readfirstlane(v[0]), __builtin_amdgcn_readfirstlane(v[1]), __builtin_amdgcn_readfirstlane(v[2]), __builtin_amdgcn_readfirstlane(v[3])}; };
    // chunk c of the tile whose first row is row0 -> stage s (rows behind M: the descriptor's byte count ends at row M, zeros land)
    auto issue = [&](int row0, int c, int s) {
        if (LG2_DBG(1)) return;
        const unsigned long long abase = (unsigned long long)(p.a + (size_t)row0 * p.K) + (unsigned)(c * KC * 2);
        const long long arows = (long long)p.M - row0;
        const unsigned abytes = arows > 0 ? (unsigned)min((long long)TM, arows) * (unsigned)p.K * 2u - (unsigned)(c * KC * 2) : 0u;
        const v4i_t ra = sgpr4(v4i_t{(int)(unsigned)abase, (int)((unsigned)(abase >> 32) & 0xffffu), (int)abytes, 0x00020000});
#pragma unroll
        for (int k = 0; k < NPW; ++k)
            lds_dma16(psrc[k], ra, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)s * LG2_STAGE + (unsigned)(wave + LG2_WAVES * k) * 1024u));
    };

    // per-lane epilogue constants: this lane's 16 columns are col0 + 8 q + 4 lhi + e  (q < 4, e < 4)
    float bias_f[16];
    if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint2 bv = p.bias ? *(const uint2*)(p.bias + col0 + 8 * q + 4 * lhi) : uint2{0u, 0u};
            bias_f[4 * q + 0] = __uint_as_float(bv.x << 16); bias_f[4 * q + 1] = __uint_as_float(bv.x & 0xffff0000u);
            bias_f[4 * q + 2] = __uint_as_float(bv.y << 16); bias_f[4 * q + 3] = __uint_as_float(bv.y & 0xffff0000u);
        }
    }
    float colsum[16];
    if constexpr (EPI == EPI_DGELU) {
#pragma unroll
        for (int e = 0; e < 16; ++e) colsum[e] = 0.f;
    }
    // y1 rows of the row tiles this wave finishes (tile-local row tiles team * RE + j), in the flush layout (16 bytes per lane: row it*16 + lane/4, chunk lane%4)
    u32x4 yreg[RE][2];
    auto load_y1 = [&](int row0) {
        if constexpr (EPI == EPI_DGELU) {
#pragma unroll
            for (int j = 0; j < RE; ++j)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int r = row0 + (team * RE + j) * 32 + it * 16 + (lane >> 2);
                    const int rc = r < p.M ? r : p.M - 1;                                 // (rows behind M: any valid row -- they only meet zeros; keeps the load count fixed)
                    yreg[j][it] = *(const u32x4*)(p.y1 + (size_t)rc * p.N + col0 + (lane & 3) * 8);
                }
        }
    };
    const unsigned obytes = (unsigned)((size_t)p.M * p.N * 2);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == EPI_GELU ? p.out2 : p.out), 0, (int)obytes, 0x00020000);
    wait_vmcnt<0>();
    __syncthreads();                                                                      // the table is in place

    // fragment address of (row tile i, k-step ks of a chunk): k-half `team`, row = 32 i + l31, chunk cc = 2 ks + lhi at slot cc ^ swz(row)   (swz(32 i + r) == swz(r))
    const unsigned frow = (unsigned)(team * TM * KC * 2) + (unsigned)l31 * (unsigned)(KC * 2), fswz = (unsigned)swz(l31);

    // the chunks of the slab's tiles form ONE stream through the ring: chunk g lives in stage g % 3 and is issued two iterations ahead -- also across
    // tile boundaries (the next tile's first two chunks go out at the start of this tile's epilogue)
    int g = 0;                                                                            // chunks consumed so far
    issue(t_begin * TM, 0, 0);
    issue(t_begin * TM, 1, 1);
    load_y1(t_begin * TM);
    for (int t = t_begin; t < t_end; ++t) {
        const int row0 = t * TM;
        f32x16 acc[RM];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int c = 0; c < NK; ++c, ++g) {
            // my pieces of chunk c have landed.  Vector-memory operations retire in issue order, so "at most n younger ones outstanding" is exact:
            // younger than chunk c are chunk c+1's pieces and, on the first two chunks of a tile, the previous epilogue's stores and this tile's y1 loads
            if (c < 2) { if (t == t_begin) wait_vmcnt<NPW + NY>(); else wait_vmcnt<NPW + NY + NE>(); }
            else if (c + 1 < NK) wait_vmcnt<NPW>();
            else wait_vmcnt<0>();
            wg_barrier();                                                                 // everyone's have; everyone is done with chunk g-1's stage
            if (c + 2 < NK) issue(row0, c + 2, (g + 2) % LG2_NS);
            const unsigned char* const L = ring + (g % LG2_NS) * LG2_STAGE;
            if (!LG2_DBG(2)) {
                // software pipeline, pinned (hipcc otherwise sinks every ds_read to right in front of its MFMA: read, wait, multiply): the fragment of row tile i
                // for k-step ks + 1 is fetched right behind the MFMA that has just consumed row tile i of k-step ks, into the same registers
                s16x8 fa[RM];
#pragma unroll
                for (int i = 0; i < RM; ++i) fa[i] = *SLAK_LDS(const s16x8, L + i * 32 * (KC * 2) + frow + (((unsigned)lhi ^ fswz) << 4));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KC / 16; ++ks) {
#pragma unroll
                    for (int i = 0; i < RM; ++i) {
                        acc[i] = mfma32<bf16_t>(bfrag[c * (KC / 16) + ks], fa[i], acc[i]);   // D^T: acc[4q+e] = D[row l31][col 8q + 4 lhi + e]
                        if (ks + 1 < KC / 16) fa[i] = *SLAK_LDS(const s16x8, L + i * 32 * (KC * 2) + frow + (((unsigned)(2 * (ks + 1) + lhi) ^ fswz) << 4));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        // the next tile's first two chunks go out BEFORE this tile's stores (the ring is free but for the last chunk's stage, which other
        // waves may still be reading: the stages written now are the two others)
        if (t + 1 < t_end) {
            issue(row0 + TM, 0, g % LG2_NS);
            issue(row0 + TM, 1, (g + 1) % LG2_NS);
        }
        // the tail of a tile for a wave of team TEAM (a compile-time constant inside: a run-time index into acc[] would move the accumulators to scratch --
        // the first version of SPLIT 2 did, 343 us instead of 70)
        auto tail = [&](auto team_c) {
            constexpr int TEAM = decltype(team_c)::value;
            if constexpr (SPLIT == 2) {
                // the two teams hold partial sums over their k-halves: each gives the other two of its four row tiles (team 0 gives 2, 3; team 1 gives 0, 1)
                // through LDS and adds what it receives -- (k-half 0) + (k-half 1) on both sides
#pragma unroll
                for (int j = 0; j < RE; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) exch[((wave * RE + j) * 16 + e) * 64 + lane] = acc[(1 - TEAM) * RE + j][e];
                wg_barrier();
                const int partner = (1 - TEAM) * 4 + cgrp;
#pragma unroll
                for (int j = 0; j < RE; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float mine = acc[TEAM * RE + j][e], theirs = exch[((partner * RE + j) * 16 + e) * 64 + lane];
                        acc[TEAM * RE + j][e] = TEAM == 0 ? mine + theirs : theirs + mine;
                    }
            }
            // ---- epilogue: per 32-row tile, wave-private (no workgroup barrier: the other waves are already multiplying the next tile)
            if (!LG2_DBG(4))
#pragma unroll
            for (int j = 0; j < RE; ++j) {
                const int i = TEAM * RE + j;                                              // tile-local row tile this wave finishes (compile-time after unrolling)
                const int r0 = row0 + i * 32;                                                 // its first row
                unsigned py[8];
                if constexpr (EPI == EPI_DGELU) {
                    // y1 tile -> staging in the flush layout, then each lane reads its own (row l31, 4-column groups)
    #pragma unroll
                    for (int it = 0; it < 2; ++it) *SLAK_LDS(u32x4, stg + (it * 16 + (lane >> 2)) * LG2_SP + (lane & 3) * 16) = yreg[j][it];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
                    const float* const T = (const float*)smem;
                    const bool rok = r0 + l31 < p.M;                                          // rows behind M: zeros came in, but keep them out of the sums anyway
                    // eight elements at a time (column groups q, q + 1): dact is what the stand-alone GEMM would have STORED -- rounded to bf16 first
                    // (slak_linear_nt_gelu_bwd's rule) --, gelu' by table with ONE wave-uniform range test per eight gathers (gelu_grad.h), the column sums
                    // add the ROUNDED products
    #pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        const u32x2 ya = *SLAK_LDS(const u32x2, stg + l31 * LG2_SP + (8 * q + 4 * lhi) * 2);
                        const u32x2 yb = *SLAK_LDS(const u32x2, stg + l31 * LG2_SP + (8 * (q + 1) + 4 * lhi) * 2);
                        // rows behind M: their dact is 0 (A's rows behind M land as zeros), but the y1 row they were handed is another row's and may hold anything --
                        // with y = 0 the product is 0 * 0.5 = 0 exactly, so the column sums need no per-element select and accumulate in place
                        const uint4 yv = rok ? uint4{ya[0], ya[1], yb[0], yb[1]} : uint4{0u, 0u, 0u, 0u};
                        const uint4 gv = uint4{pack2<bf16_t>(acc[i][4 * q], acc[i][4 * q + 1]), pack2<bf16_t>(acc[i][4 * q + 2], acc[i][4 * q + 3]),
                                               pack2<bf16_t>(acc[i][4 * q + 4], acc[i][4 * q + 5]), pack2<bf16_t>(acc[i][4 * q + 6], acc[i][4 * q + 7])};
                        uint4 ov;
                        float cs[8];
    #pragma unroll
                        for (int e = 0; e < 8; ++e) cs[e] = colsum[4 * q + e];
                        float tv[8];
                        if (__builtin_amdgcn_ballot_w64(!gelu_grad_gather8(T, yv, tv)) == 0) gelu_bwd8_apply(gv, tv, ov, cs);
                        else gelu_bwd8(T, gv, yv, ov, cs);
                        py[2 * q] = ov.x; py[2 * q + 1] = ov.y; py[2 * q + 2] = ov.z; py[2 * q + 3] = ov.w;
    #pragma unroll
                        for (int e = 0; e < 8; ++e) colsum[4 * q + e] = cs[e];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
                } else {
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        py[2 * q] = pack2<bf16_t>(acc[i][4 * q] + bias_f[4 * q], acc[i][4 * q + 1] + bias_f[4 * q + 1]);
                        py[2 * q + 1] = pack2<bf16_t>(acc[i][4 * q + 2] + bias_f[4 * q + 2], acc[i][4 * q + 3] + bias_f[4 * q + 3]);
                    }
                }
                auto flush = [&](const unsigned (&v)[8], const __amdgpu_buffer_rsrc_t dst) {
    #pragma unroll
                    for (int q = 0; q < 4; ++q) *SLAK_LDS(u32x2, stg + l31 * LG2_SP + (8 * q + 4 * lhi) * 2) = u32x2{v[2 * q], v[2 * q + 1]};
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
    #pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int rr = it * 16 + (lane >> 2), ch = lane & 3;
                        const u32x4 v4 = *SLAK_LDS(const u32x4, stg + rr * LG2_SP + ch * 16);
                        // a raw buffer store over [out, out + M*N): rows behind M are dropped by the range check, the instruction is ALWAYS issued (the
                        // counted waits above rely on a fixed number of stores per tile)
                        __builtin_amdgcn_raw_buffer_store_b128(v4, dst, (unsigned)(r0 + rr) * (unsigned)p.N * 2u + (unsigned)(col0 + ch * 8) * 2u, 0, 0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
                };
                flush(py, rs_out);
                if constexpr (EPI == EPI_GELU) {
                    unsigned pg[8];
                    g2_lut2x8((const uint16_t*)smem, py, pg);
                    flush(pg, rs_out2);
                }
            }
        };
        if constexpr (SPLIT == 2) { if (team == 0) tail(std::integral_constant<int, 0>{}); else tail(std::integral_constant<int, 1>{}); }
        else tail(std::integral_constant<int, 0>{});
        if (t + 1 < t_end) load_y1(row0 + TM);                                            // (behind the stores: the registers are free only now)
    }
    wait_vmcnt<0>();

    if constexpr (EPI == EPI_DGELU) {
        // column sums of the slab: lanes l31 = 0..31 hold the same columns (rows differ) -> add over the 32 lanes (and the two teams) in a fixed order
        // through LDS; one partial row [TN] per workgroup: part[slab][N]
        __syncthreads();
        float* const red = (float*)ring;                                                  // [8 waves][32 rows][32 cols + 1]
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(wave * 32 + l31) * 33 + 8 * q + 4 * lhi + e] = colsum[4 * q + e];
        __syncthreads();
        if (tid < TN) {                                                                   // thread -> (column group tid / 32, column tid % 32)
            const int gq = tid >> 5, cc = tid & 31;
            float s2 = 0.f;
#pragma unroll
            for (int tm = 0; tm < SPLIT; ++tm)
                for (int r = 0; r < 32; ++r) s2 += red[((tm * (LG2_WAVES / SPLIT) + gq) * 32 + r) * 33 + cc];
            p.part[(size_t)slab * p.N + panel * TN + gq * 32 + cc] = s2;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct Lg2Plan { int tiles_m, panels, slabs, tps, split, xcd_map, grid; size_t lds; };

static bool lg2_plan(int M, int N, int K, int epi, Lg2Plan& pl) {
    if (M < 1 || (epi != EPI_GELU && epi != EPI_DGELU)) return false;
    if (K == 192 || K == 256 || K == 384) pl.split = 1;
    else if (K == 512 || K == 768) pl.split = 2;               // two teams of four waves share K (the B fragments of a whole K = 768 would need 192 registers)
    else return false;
    const int tn = 256 / pl.split;
    if (N < tn || N % tn) return false;
    // 32-bit byte offsets: the epilogue issues its stores for the tail rows of the LAST row tile too (rows up to tiles_m * 128, dropped by the buffer
    // range check) -- bound the rounded-up extent, or an offset within 127 rows of 4 GiB would wrap back into `out`
    const long long m_up = ((long long)M + LG2_TM - 1) / LG2_TM * LG2_TM;
    if (m_up * N * 2 >= (1LL << 32) || m_up * K * 2 >= (1LL << 32)) return false;
    const size_t tbl = epi == EPI_GELU ? (size_t)G2_BYTES : (size_t)GD_BYTES;
    pl.lds = tbl + (size_t)LG2_NS * LG2_STAGE + (size_t)LG2_WAVES * 32 * LG2_SP + (pl.split == 2 ? (size_t)LG2_WAVES * 2 * 16 * 64 * 4 : 0);
    pl.tiles_m = (M + LG2_TM - 1) / LG2_TM;
    pl.panels = N / tn;
    const int slots = mfma_cu_count();                          // one workgroup (eight waves) per CU
    int S = slots / pl.panels; if (S < 1) S = 1;
    if (S > pl.tiles_m) S = pl.tiles_m;
    pl.tps = (pl.tiles_m + S - 1) / S;
    // whole XCD groups of slabs (their panels then stream the same rows of A through one L2) -- unless rounding the slab count down to a multiple of eight
    // would lengthen every workgroup's walk (stage 4 of SLaK-T: 49 row tiles for 24 panels = 10 slabs of 5 tiles; 8 slabs would mean 7 tiles each on 2/3 of
    // the CUs: 78 -> 56 us measured)
    const int S8 = S - S % 8;
    if (S8 >= 8 && (pl.tiles_m + S8 - 1) / S8 == pl.tps) {
        pl.slabs = (pl.tiles_m + pl.tps - 1) / pl.tps;          // no empty slabs
        pl.xcd_map = 1; pl.grid = (pl.slabs + 7) / 8 * 8 * pl.panels;
    } else {
        pl.slabs = (pl.tiles_m + pl.tps - 1) / pl.tps;
        pl.xcd_map = 0; pl.grid = pl.slabs * pl.panels;
    }
    return true;
}

static const uint16_t* g2_table_device() {
    static std::mutex mu;
    static const uint16_t* tab[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (tab[dev]) return tab[dev];
    std::vector<uint16_t> h(2 * G2_N);
    for (unsigned sgn = 0; sgn < 2; ++sgn)
        for (unsigned i = 0; i < G2_N; ++i) {
            const uint32_t bits = ((sgn << 15) | (G2_LO + i)) << 16;
            float xf; memcpy(&xf, &bits, 4);
            const double x = xf, g = 0.5 * x * erfc(-x * 0.70710678118654752440);
            const float gf = (float)g;
            uint32_t u; memcpy(&u, &gf, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            h[sgn * G2_N + i] = (uint16_t)(u >> 16);
        }
    void* d = nullptr;
    if (hipMalloc(&d, h.size() * 2) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    tab[dev] = (const uint16_t*)d;
    return tab[dev];
}

}  // namespace slak

using namespace slak;

extern "C" {

int slak_linear_gemm_supported(int M, int N, int K, int epilogue) {
    static const bool on = [] { const char* e = getenv("SLAK_LINEAR_GEMM"); return !(e && e[0] == '0'); }();
    Lg2Plan pl;
    return (on && lg2_plan(M, N, K, epilogue, pl)) ? 1 : 0;
}

size_t slak_linear_gemm_workspace_bytes(int M, int N, int K, int epilogue) {
    Lg2Plan pl;
    if (!lg2_plan(M, N, K, epilogue, pl) || epilogue != EPI_DGELU) return 0;
    return align_up((size_t)pl.slabs * N * sizeof(float), 256);
}

int slak_linear_gemm(const void* a, const void* b, const void* bias, void* out, void* out2, const void* y1, float* dbias, int M, int N, int K, int epilogue,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!a || !b || !out) return SLAK_ERR_INVALID_ARG;
    Lg2Plan pl;
    if (!slak_linear_gemm_supported(M, N, K, epilogue) || !lg2_plan(M, N, K, epilogue, pl)) return SLAK_ERR_UNSUPPORTED;
    if (epilogue == EPI_GELU && !out2) return SLAK_ERR_INVALID_ARG;
    if (epilogue == EPI_DGELU && (!y1 || !dbias)) return SLAK_ERR_INVALID_ARG;
    if (epilogue == EPI_DGELU && (!workspace || workspace_bytes < slak_linear_gemm_workspace_bytes(M, N, K, epilogue))) return SLAK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Lg2Params p;
    p.a = (const uint16_t*)a; p.b = (const uint16_t*)b; p.bias = (const uint16_t*)bias; p.out = (uint16_t*)out; p.out2 = (uint16_t*)out2;
    p.y1 = (const uint16_t*)y1; p.part = (float*)workspace; p.table = nullptr;
#ifdef SLAK_LG2_DEV
    { const char* e = slak_dev_getenv("SLAK_LG2_DBG"); p.dbg = e ? atoi(e) : 0; }
#endif
    p.M = M; p.N = N; p.K = K; p.tiles_m = pl.tiles_m; p.panels = pl.panels; p.slabs = pl.slabs; p.tps = pl.tps; p.xcd_map = pl.xcd_map;
    if (epilogue == EPI_GELU) { p.table = g2_table_device(); if (!p.table) return SLAK_ERR_LAUNCH; }
    if (epilogue == EPI_DGELU) { p.table = gelu_grad_table_device(); if (!p.table) return SLAK_ERR_LAUNCH; }
    const dim3 grid((unsigned)pl.grid);
#define SLAK_LG2_LAUNCH(E, KS, SP)                                                      \
    do {                                                                                \
        auto k = linear_gemm_kernel<E, KS, SP>;                                         \
        if (!slak_set_max_lds((const void*)k, pl.lds)) return SLAK_ERR_LAUNCH;          \
        hipLaunchKernelGGL(k, grid, dim3(512), pl.lds, st, p);                          \
    } while (0)
#define SLAK_LG2_BY_K(E)                                                                \
    do {                                                                                \
        if (K == 192) SLAK_LG2_LAUNCH(E, 12, 1); else if (K == 256) SLAK_LG2_LAUNCH(E, 16, 1); else if (K == 384) SLAK_LG2_LAUNCH(E, 24, 1); \
        else if (K == 512) SLAK_LG2_LAUNCH(E, 32, 2); else SLAK_LG2_LAUNCH(E, 48, 2);   \
    } while (0)
    if (epilogue == EPI_GELU) SLAK_LG2_BY_K(EPI_GELU); else SLAK_LG2_BY_K(EPI_DGELU);
#undef SLAK_LG2_BY_K
#undef SLAK_LG2_LAUNCH
    SLAK_LAUNCH_CHECK();
    if (epilogue == EPI_DGELU) return tail_reduce_columns((const float*)workspace, dbias, pl.slabs, N, st);
    return SLAK_OK;
}

}  // extern "C"
