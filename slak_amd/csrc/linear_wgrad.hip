// slak_amd/csrc/linear_wgrad.hip -- weight gradient of the block's pointwise Linear layers (reference: models/SLaK.py:117-118 pwconv1 /
// pwconv2 are nn.Linear on the NHWC activation; their weight gradients are dW = dY^T X with the reduction over all N*H*W pixel rows):
//     D[N1][N2] (fp32) = X1^T X2,   X1 [M][N1], X2 [M][N2] bf16 row-major, M = 6 k ... 400 k rows, N1, N2 in {96 ... 3072}.
// Both operands have the REDUCTION index as their slow dimension, so neither is in MFMA fragment order; the library's TN kernels run
// this at ~0.38 PFLOP/s (128x128 macro tiles re-read the operands through L2 12x / 3x).  Here:
//   * a workgroup (4 waves, one per SIMD) owns a 192x192 (or 384x96 / 96x384; round 6: 256x128 / 128x256 for SLaK-B's widths) output tile = four
//     96x96 (128x64) wave tiles (nine / eight 32x32x16 MFMA accumulators each) and one SLAB of the rows; the row chunks (32 rows x both operand tiles) stream HBM/L2 -> LDS by LDS-DMA in a
//     4-stage ring (3 chunks in flight), rows padded to a pitch = +-64 B mod 256 by the per-lane source permutation of the DMA, so that
//   * both MFMA operands are formed by ds_read_b64_tr_b16 (the transposing LDS read) straight from the row-major image, conflict-free;
//   * the slabs' partial tiles (fp32) go to a workspace and a second kernel adds them in slab order: deterministic, fp32 throughout
//     (the torch path this replaces rounded each split's partial sum to bf16).
// Traffic per call: the operands once from HBM (tiles of one slab are co-scheduled on one XCD so that their re-reads hit its L2) plus
// 2 x (#workgroups x tile) of fp32 partials; the 192x192 tile per CU is the balance between the two.
#include "slak_common.h"
#include "mfma_common.h"
#include <stdlib.h>

namespace slak {

constexpr int LW_KC = 32;                  // rows of the reduction per chunk (two k = 16 MFMA steps)
constexpr int LW_NS = 4;                   // LDS stages

struct LwParams {
    const uint16_t* x1; const uint16_t* x2; float* part;      // part: [S][N1][N2] (or the result itself when S == 1)
    int M, N1, N2, tiles1, tiles2, S, cps;                     // cps = chunks per slab
    int xcd_map;                                               // 1: workgroup id -> (slab, tile) keeps a slab's tiles on one XCD
};

__host__ __device__ constexpr int lw_pitch(int T) {            // bytes per LDS row of a T-column bf16 tile: >= 2T, = +-64 mod 256, multiple of 16
    int p = 2 * T;
    while (p % 256 != 64 && p % 256 != 192) p += 16;
    return p;
}

__device__ __forceinline__ v4i_t lw_desc(const void* ptr, long long bytes) {            // raw buffer over [ptr, ptr + bytes), bytes < 4 GiB
    const unsigned long long a = (unsigned long long)ptr;
    return v4i_t{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)(unsigned)(bytes > 0 ? bytes : 0), 0x00020000};
}

// NB1 x NB2: the wave tile in 32 x 32 MFMA blocks -- 3 x 3 (96 x 96: widths in multiples of 96, SLaK-T / -S / -L); round 6: 4 x 2 / 2 x 4 (128 x 64: SLaK-B's
// widths 128 * 2^k; 4 x 4 = sixteen accumulators fill the accumulator file exactly and hipcc then shuffles them through VGPRs and scratch)
template <int W1, int W2, int NB1 = 3, int NB2 = 3>
__global__ __launch_bounds__(256, 1) void linear_wgrad_kernel(const LwParams p) {
    static_assert(W1 * W2 == 4, "four waves");
    constexpr int NM = NB1 * NB2;                             // MFMAs per k step
    constexpr int WT1 = 32 * NB1, WT2 = 32 * NB2;
    constexpr int T1 = WT1 * W1, T2 = WT2 * W2;
    constexpr int PA = lw_pitch(T1), PB = lw_pitch(T2);
    constexpr int CDA = PA / 16, CDB = PB / 16;                // destination chunks per row
    constexpr int NA = LW_KC * CDA / 64, NB = LW_KC * CDB / 64;   // DMA pieces (1 KiB) per chunk and operand
    static_assert(LW_KC * CDA % 64 == 0 && LW_KC * CDB % 64 == 0, "whole pieces");
    constexpr int NP = NA + NB, NPW = (NP + 3) / 4;            // pieces per wave and chunk (round-robin over the four waves)
    constexpr int BOFF = LW_KC * PA, STAGE = LW_KC * (PA + PB);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)SLAK_LDS(unsigned char, smem);
    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int w1 = wave / W2, w2 = wave - w1 * W2;

    // workgroup -> (slab, tile)
    const int ntiles = p.tiles1 * p.tiles2;
    int slab, tile;
    if (p.xcd_map) { const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3; slab = xcd + 8 * (r / ntiles); tile = r % ntiles; }
    else { slab = blockIdx.x / ntiles; tile = blockIdx.x - slab * ntiles; }
    const int t1 = tile / p.tiles2, t2 = tile - t1 * p.tiles2;
    const int nchunks_total = (p.M + LW_KC - 1) / LW_KC;
    const int c_begin = slab * p.cps, c_end = min(c_begin + p.cps, nchunks_total), nc = max(c_end - c_begin, 0);

    // DMA plan: piece pi = wave + 4 k; destination chunk q = 64 (pi or pi - NA) + lane -> (row q / CD, chunk q % CD) of the operand image
    unsigned psrc[NPW]; unsigned pdst[NPW]; bool pisA[NPW]; bool pact[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int pi = wave + 4 * k;
        pact[k] = pi < NP; pisA[k] = pi < NA;
        const int pj = pisA[k] ? pi : pi - NA;
        const int q = 64 * pj + lane;
        const int CD = pisA[k] ? CDA : CDB, ld = pisA[k] ? p.N1 : p.N2, T = pisA[k] ? T1 : T2, tc = pisA[k] ? t1 : t2;
        const int r = q / CD, cc = q - r * CD;
        psrc[k] = cc < T / 8 ? (unsigned)r * (unsigned)ld * 2u + (unsigned)(tc * T + cc * 8) * 2u : 0x80000000u;   // padding chunks: out of range -> zeros
        pdst[k] = (pisA[k] ? 0u : (unsigned)BOFF) + (unsigned)pj * 1024u;
    }
    // running descriptors of the next chunk to fetch: bases advance by one chunk per issue; the byte counts cover the rows that remain in
    // the SLAB (0 past its end: such a chunk fetches nothing, every piece is written as zeros -- the piece counts stay fixed)
    unsigned long long baseA = (unsigned long long)(p.x1 + (size_t)c_begin * LW_KC * p.N1), baseB = (unsigned long long)(p.x2 + (size_t)c_begin * LW_KC * p.N2);
    int rows_left = min(c_end * LW_KC, p.M) - c_begin * LW_KC;
    const unsigned stepA = (unsigned)(LW_KC * p.N1 * 2), stepB = (unsigned)(LW_KC * p.N2 * 2), rowA = (unsigned)p.N1 * 2u, rowB = (unsigned)p.N2 * 2u;
    int n_issued = 0;
    v4i_t rA, rB; unsigned sb;
    auto next_desc = [&] {
        const unsigned rl = (unsigned)max(rows_left, 0);
        rA = v4i_t{(int)(unsigned)baseA, (int)((unsigned)(baseA >> 32) & 0xffffu), (int)(rl * rowA), 0x00020000};
        rB = v4i_t{(int)(unsigned)baseB, (int)((unsigned)(baseB >> 32) & 0xffffu), (int)(rl * rowB), 0x00020000};
        sb = lds0 + (unsigned)(n_issued & (LW_NS - 1)) * STAGE;
        baseA += stepA; baseB += stepB; rows_left -= LW_KC; ++n_issued;
    };
    auto sgpr4 = [](v4i_t v) { return v4i_t{__builtin_amdgcn_readfirstlane(v[0]), __builtin_amdgcn_readfirstlane(v[1]), __builtin_amdgcn_readfirstlane(v[2]), __builtin_amdgcn_readfirstlane(v[3])}; };
    auto piece = [&](int k) { if (NP % 4 == 0 || pact[k]) lds_dma16(psrc[k], sgpr4(pisA[k] ? rA : rB), __builtin_amdgcn_readfirstlane(sb + pdst[k])); };

    f32x16 acc[NB1][NB2];
#pragma unroll
    for (int i = 0; i < NB1; ++i)
#pragma unroll
        for (int j = 0; j < NB2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing (ds_read_b64_tr_b16): 16-lane group g4 reads a [4 rows][16 cols] block, lane (i16 >> 2) = row, (i16 & 3) = 4-column chunk
    const int g4 = lane >> 4, i16 = lane & 15;
    const unsigned fa = (unsigned)((8 * (g4 >> 1) + (i16 >> 2)) * PA + (w1 * WT1 + (g4 & 1) * 16 + (i16 & 3) * 4) * 2);
    const unsigned fb = (unsigned)(BOFF + (8 * (g4 >> 1) + (i16 >> 2)) * PB + (w2 * WT2 + (g4 & 1) * 16 + (i16 & 3) * 4) * 2);
    auto frag = [&](const unsigned char* L, unsigned addr, int pitch) -> s16x8 {     // 8 k of one column: two transposing reads, 4 rows apart
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + addr + 4 * pitch));
        return s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    };
    // fragment f = 0 .. 2 (NB1 + NB2) - 1 of a chunk: k step f / (NB1 + NB2), then the NB1 blocks of the first operand, then the NB2 of the second
    struct Frags { s16x8 a[2][NB1], b[2][NB2]; };
    auto load_frag = [&](Frags& F, const unsigned char* L, int f) {
        const int ks = f / (NB1 + NB2), g = f % (NB1 + NB2);
        if (g < NB1) F.a[ks][g] = frag(L, fa + (unsigned)(ks * 16 * PA + g * 64), PA);
        else F.b[ks][g - NB1] = frag(L, fb + (unsigned)(ks * 16 * PB + (g - NB1) * 64), PB);
    };
    auto wait_chunk = [&](int ahead) {                          // my pieces of the oldest outstanding chunk have landed; `ahead` younger chunks may be in flight
        if constexpr (NP % 4 == 0) { if (ahead == 3) wait_vmcnt<3 * NPW>(); else wait_vmcnt<2 * NPW>(); }
        else wait_vmcnt_dyn(ahead * (wave < NP % 4 ? NPW : NPW - 1));
    };
#define LW_MMA(F, ks, m) acc[(m) / NB2][(m) % NB2] = mfma32<bf16_t>(F.a[ks][(m) / NB2], F.b[ks][(m) % NB2], acc[(m) / NB2][(m) % NB2])
#define LW_SB() __builtin_amdgcn_sched_barrier(0)
    // One chunk (one wave per SIMD: nothing else hides latencies, so the stream is laid out by hand, <= 5 fillers per 32-cycle MFMA):
    //   first k step : 2 MFMAs | chunk c+1 confirmed landed (counted vmcnt) + barrier (every wave now holds ALL of chunk c in registers, so
    //                  its stage is free) | 7 MFMAs, each followed by one DMA piece of chunk c+4 into that stage
    //   second k step: 9 MFMAs, each followed by fragment reads of chunk c+1 (into the other register set)
    auto chunk = [&](int c, Frags& cur, Frags& nxt) {
        LW_MMA(cur, 0, 0); LW_MMA(cur, 0, 1); LW_SB();
        wait_chunk(2);
        wg_barrier(); LW_SB();
        next_desc();
#pragma unroll
        for (int m = 2; m < NM; ++m) {                          // NM - 2 slots for the NPW pieces of chunk c + 4 (the last slot takes what is left)
            LW_MMA(cur, 0, m);
            if (m - 2 < NPW) piece(m - 2);
            if (m == NM - 1) {
#pragma unroll
                for (int k = NM - 2; k < NPW; ++k) piece(k);
            }
            LW_SB();
        }
        const unsigned char* const L = smem + ((c + 1) & (LW_NS - 1)) * STAGE;
        constexpr int NF = 2 * (NB1 + NB2), DBL = NF > NM ? NF - NM : 0;   // fragment reads of chunk c + 1: NF over NM MFMAs (the first DBL MFMAs take two)
        static_assert(NF <= 2 * NM, "fragment reads fit behind the MFMAs");
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            LW_MMA(cur, 1, m);
            if (m < DBL) { load_frag(nxt, L, 2 * m); load_frag(nxt, L, 2 * m + 1); } else if (m + DBL < NF) load_frag(nxt, L, m + DBL);
            LW_SB();
        }
    };
    static_assert(NPW <= NM - 1, "piece / MFMA interleave");
    if (nc > 0) {
        Frags F0, F1;
        for (int c = 0; c < LW_NS; ++c) {
            next_desc();
#pragma unroll
            for (int k = 0; k < NPW; ++k) piece(k);
        }
        wait_chunk(3);
        wg_barrier();
#pragma unroll
        for (int f = 0; f < 2 * (NB1 + NB2); ++f) load_frag(F0, smem, f);
        for (int c = 0; c < nc; c += 2) {                      // (the last chunk's prefetch reads a stage that was fetched empty: unused)
            chunk(c, F0, F1);
            if (c + 1 < nc) chunk(c + 1, F1, F0);
        }
    }
#undef LW_MMA
#undef LW_SB
    wait_vmcnt<0>();                                            // the trailing (empty) pieces
    wg_barrier();                                               // every wave is done with the ring: it becomes the epilogue's staging space

    // partial tile -> part[slab] through LDS, so that the stores are full lines (16 bytes per lane; row-per-lane dword stores are
    // store-issue bound): acc[i][j][r] = D[32 i + 8 (r / 4) + 4 (lane / 32) + r % 4][32 j + lane % 32]
    float* const out = p.part + (size_t)slab * p.N1 * p.N2 + (size_t)(t1 * T1 + w1 * WT1) * p.N2 + t2 * T2 + w2 * WT2;
    constexpr int EP = WT2 * 4 + 16;                            // staging pitch (bytes): 32 rows x WT2 floats per pass
    unsigned char* const E = smem + wave * (32 * EP);
    static_assert(4 * 32 * EP <= LW_NS * STAGE, "epilogue staging fits the ring");
    const int er = 4 * (lane >> 5), ec = lane & 31;
#pragma unroll
    for (int i = 0; i < NB1; ++i) {
#pragma unroll
        for (int j = 0; j < NB2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) *(float*)(E + (er + 8 * (r >> 2) + (r & 3)) * EP + (32 * j + ec) * 4) = acc[i][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < WT2 / 8; ++k) {                     // 32 rows x WT2 / 4 float4
            const int idx = k * 64 + lane, row = idx / (WT2 / 4), c4 = idx - row * (WT2 / 4);
            *(float4*)(out + (size_t)(32 * i + row) * p.N2 + c4 * 4) = *(const float4*)(E + row * EP + c4 * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
    }
}

// D[e] = sum_s part[s][e]: 16 float4 elements x 16 slab groups per workgroup (group g adds slabs g, g + 16, ... in order; the groups are
// added in order through LDS): a fixed summation order, and enough parallelism when S is in the hundreds and the result is small
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int n4, int S, size_t stride4) {
    __shared__ float4 red[16][16];
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    float4 a = float4{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        const float4* p = (const float4*)part + i;
        for (int s = g; s < S; s += 16) { const float4 v = p[(size_t)s * stride4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    }
    red[g][e] = a;
    __syncthreads();
    if (g == 0 && i < n4) {
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][e]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        ((float4*)out)[i] = a;
    }
}

// few slabs, large result: one thread per float4, the slabs in order
__global__ __launch_bounds__(256) void linear_wgrad_reduce_few_kernel(const float* __restrict__ part, float* __restrict__ out, int n4, int S, size_t stride4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4* p = (const float4*)part + i;
    float4 a = p[0];
    for (int s = 1; s < S; ++s) { const float4 v = p[(size_t)s * stride4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    ((float4*)out)[i] = a;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct LwPlan { int w1, w2, nb1, nb2, tiles1, tiles2, S, cps, xcd_map; size_t lds; };

static bool lw_wide_tiles() {                 // SLAK_LINEAR_WGRAD_128=0: the 128 x 64 wave tiles off (A/B: SLaK-B's weight gradients back on the library's split-K GEMM)
    static const bool v = [] { const char* e = getenv("SLAK_LINEAR_WGRAD_128"); return !(e && e[0] == '0'); }();
    return v;
}

static bool lw_plan(int M, int N1, int N2, LwPlan& pl) {
    if (M < LW_KC || N1 <= 0 || N2 <= 0) return false;
    pl.nb1 = pl.nb2 = 3;
    if (N1 % 192 == 0 && N2 % 192 == 0) { pl.w1 = 2; pl.w2 = 2; }
    else if (N2 == 96 && N1 % 384 == 0) { pl.w1 = 4; pl.w2 = 1; }
    else if (N1 == 96 && N2 % 384 == 0) { pl.w1 = 1; pl.w2 = 4; }
    // round 6: SLaK-B's widths (128, 256, 512, 1024 and four times that): 128 x 64 wave tiles (eight accumulators), the long side on the wider operand
    else if (lw_wide_tiles() && N1 >= N2 && N1 % 256 == 0 && N2 % 128 == 0) { pl.w1 = 2; pl.w2 = 2; pl.nb1 = 4; pl.nb2 = 2; }
    else if (lw_wide_tiles() && N1 < N2 && N1 % 128 == 0 && N2 % 256 == 0) { pl.w1 = 2; pl.w2 = 2; pl.nb1 = 2; pl.nb2 = 4; }
    else return false;
    pl.tiles1 = N1 / (32 * pl.nb1 * pl.w1); pl.tiles2 = N2 / (32 * pl.nb2 * pl.w2);
    const int ntiles = pl.tiles1 * pl.tiles2, nchunks = (M + LW_KC - 1) / LW_KC;
    const int cus = mfma_cu_count();
    int S = cus / ntiles; if (S < 1) S = 1;
    if (S >= 8) S -= S % 8;                                   // whole XCD groups
    if (S > nchunks / 8) S = nchunks / 8 > 0 ? nchunks / 8 : 1;   // at least 8 chunks per slab
    pl.cps = (nchunks + S - 1) / S;
    pl.S = (nchunks + pl.cps - 1) / pl.cps;                    // no empty slabs
    pl.xcd_map = (pl.S % 8 == 0) ? 1 : 0;
    const int T1 = 32 * pl.nb1 * pl.w1, T2 = 32 * pl.nb2 * pl.w2;
    pl.lds = (size_t)LW_NS * LW_KC * (lw_pitch(T1) + lw_pitch(T2));
    if (pl.lds > 160 * 1024) return false;
    return (long long)M * (N1 > N2 ? N1 : N2) * 2 < (1LL << 32);
}

bool linear_wgrad_supported(int M, int N1, int N2) { LwPlan pl; return lw_plan(M, N1, N2, pl); }
size_t linear_wgrad_workspace(int M, int N1, int N2) {
    LwPlan pl; if (!lw_plan(M, N1, N2, pl)) return 0;
    return pl.S > 1 ? align_up((size_t)pl.S * N1 * N2 * sizeof(float), 256) : 0;
}

int launch_linear_wgrad(const void* x1, const void* x2, float* d, int M, int N1, int N2, void* ws, size_t ws_bytes, hipStream_t st) {
    LwPlan pl;
    if (!lw_plan(M, N1, N2, pl)) return SLAK_ERR_UNSUPPORTED;
    if (pl.S > 1 && (!ws || ws_bytes < linear_wgrad_workspace(M, N1, N2))) return SLAK_ERR_WORKSPACE;
    LwParams p;
    p.x1 = (const uint16_t*)x1; p.x2 = (const uint16_t*)x2; p.part = pl.S > 1 ? (float*)ws : d;
    p.M = M; p.N1 = N1; p.N2 = N2; p.tiles1 = pl.tiles1; p.tiles2 = pl.tiles2; p.S = pl.S; p.cps = pl.cps; p.xcd_map = pl.xcd_map;
    const dim3 grid((unsigned)(pl.tiles1 * pl.tiles2 * pl.S));
#define SLAK_LW_LAUNCH(A, B, NBA, NBB)                                                                                                 \
    do {                                                                                                                               \
        auto k = linear_wgrad_kernel<A, B, NBA, NBB>;                                                                                  \
        if (!slak_set_max_lds((const void*)k, pl.lds)) return SLAK_ERR_LAUNCH; \
        hipLaunchKernelGGL(k, grid, dim3(256), pl.lds, st, p);                                                                         \
    } while (0)
    if (pl.nb1 == 4) SLAK_LW_LAUNCH(2, 2, 4, 2);
    else if (pl.nb1 == 2) SLAK_LW_LAUNCH(2, 2, 2, 4);
    else if (pl.w1 == 2) SLAK_LW_LAUNCH(2, 2, 3, 3);
    else if (pl.w1 == 4) SLAK_LW_LAUNCH(4, 1, 3, 3);
    else SLAK_LW_LAUNCH(1, 4, 3, 3);
#undef SLAK_LW_LAUNCH
    SLAK_LAUNCH_CHECK();
    if (pl.S > 1) {
        const int n4 = N1 * N2 / 4;
        if (reduce_defer_push(pl.S <= 8 ? 2 : 1, (const float*)ws, d, d, 0, pl.S, n4, st)) return SLAK_OK;      // (slak_defer_reductions_begin)
        if (pl.S <= 8) hipLaunchKernelGGL(linear_wgrad_reduce_few_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, d, n4, pl.S, (size_t)n4);
        else hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, (const float*)ws, d, n4, pl.S, (size_t)n4);
        SLAK_LAUNCH_CHECK();
    }
    return SLAK_OK;
}

}  // namespace slak

using namespace slak;

extern "C" {

int slak_linear_wgrad_supported(int M, int N1, int N2) { return linear_wgrad_supported(M, N1, N2) ? 1 : 0; }
size_t slak_linear_wgrad_workspace_bytes(int M, int N1, int N2) { return linear_wgrad_workspace(M, N1, N2); }
int slak_linear_wgrad(const void* x1_bf16, const void* x2_bf16, float* d, int M, int N1, int N2, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x1_bf16 || !x2_bf16 || !d) return SLAK_ERR_INVALID_ARG;
    return launch_linear_wgrad(x1_bf16, x2_bf16, d, M, N1, N2, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
