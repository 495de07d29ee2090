// slak_amd/csrc/linear_gemm.hip -- the expanding pointwise GEMM of a block's MLP on stages 2-4 with the GELU work in its epilogue
// (reference: models/SLaK.py:158-160  x = pwconv1(x); x = act(x) and, backward, autograd's dY1 = (dZ W2) * gelu'(Y1)):
//     Y[M][N] = X[M][K] . Wt[N][K]^T       X, Wt bf16 row-major (K contiguous), fp32 accumulate, N = 4C >= 768, K = C in {192, 384, 768}
//   EPI 1 (forward) : y1 = bf16(Y + bias), a = GELU(y1)           -- two bf16 outputs, no separate GELU pass over the 4C-wide tensor
//   EPI 2 (backward): dy1 = bf16(Y * gelu'(y1)), colsum += dy1    -- the GELU-backward pass (read dact, read y1, write dy1) and the round trip
//                     of the 4C-wide `dact` through HBM disappear; the column sums are pwconv1's bias gradient
// (EPI 0: y = bf16(Y + bias).)  The library GEMM + elementwise kernels this replaces spend 74 us (forward) / 99 us (backward) per stage-3
// call, of which 29 / 54 us are the elementwise passes.
// Structure = linear_wgrad.hip's: one workgroup per CU, four waves in a 2 x 2 arrangement of 96 x 96 wave tiles (nine 32x32x16 MFMA
// accumulators each) = a 192 (n) x 192 (m) output tile; 32-deep K chunks of both operand tiles stream HBM/L2 -> LDS by LDS-DMA into a
// three-stage ring (rows of 64 B padded to an 80-byte pitch on the source side: row-per-lane ds_read_b128 fragments, conflict-free), the
// instruction stream of a chunk is laid out by hand.  Differences: both operands are K-contiguous (plain 16-byte fragment reads), there is
// no split over K; a workgroup keeps its n-tile (weight rows, bias, column sums) and walks m-tiles, and the chunk stream runs ACROSS tiles:
// while a tile's epilogue runs, the first chunks of the next tile are already in flight.  MFMA operands: A = weight rows (n), B = x rows
// (m), so a lane owns one output row m and four consecutive n per accumulator quad -- the epilogue is per lane, values leave through a
// per-wave LDS staging tile as 16-byte full-line stores (and y1 enters through it for EPI 2).
#include "slak_common.h"
#include "mfma_common.h"
#include <cmath>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

namespace slak {

constexpr int LG_KC = 32;                  // K elements per chunk
constexpr int LG_NS = 3;                   // ring stages
constexpr int LG_T = 192;                  // tile edge (both m and n)
constexpr int LG_RP = 80;                  // LDS row pitch of a chunk row (64 B data + 16 B pad: 5 chunks, odd)
constexpr int LG_STAGE = 2 * LG_T * LG_RP; // W rows then X rows: 30,720 B
constexpr int LG_NPW = 8;                  // DMA pieces per wave and chunk (30 pieces; the two spare slots re-fetch pieces 28, 29)
constexpr int LG_OP = 208;                 // staging pitch: 96 bf16 + pad (13 chunks, odd)
constexpr int LG_OBUF = 32 * LG_OP;        // per-wave staging tile: 32 rows x 96 columns
constexpr unsigned LG_GLO = 109u << 7, LG_GN = 22u << 7;     // value tables: bf16 magnitudes 2^-18 .. 16 (see linear_skinny.hip / block_tail.hip)

struct LgParams {
    const uint16_t* x; const uint16_t* wt; const uint16_t* bias;      // bias (bf16, [N]) or NULL
    uint16_t* y; uint16_t* g;                                         // EPI 0/1: y (and gelu(y)); EPI 2: y = dy1, g unused
    const uint16_t* y1;                                               // EPI 2: the saved pre-activation
    float* colsum;                                                    // EPI 2: partial column sums [2 * SM][N]
    const void* table;                                                // EPI 1: bf16 GELU table; EPI 2: fp32 gelu' table
    int M, N, K, SM;                                                  // SM = workgroups per n-tile (m-tiles are dealt round-robin)
    int dbg;                                                          // SLAK_LG_DBG (timing experiments): 1 = no epilogue
};

// value tables (same layout as gelu_lut / gelu_grad_lut: [sign][magnitude - LG_GLO])
__device__ __forceinline__ unsigned lg_gelu(const uint16_t* __restrict__ T, unsigned b) {
    const unsigned mag = b & 0x7fffu, neg = b >> 15;
    const unsigned idx = mag - LG_GLO;
    const bool in = idx < LG_GN;
    const unsigned t = T[(in ? idx : 0u) + neg * LG_GN];
    const unsigned small = mag >= 0x100u ? b - 0x80u : (b & 0x8000u);
    const unsigned big = neg ? (mag > 0x7f7fu ? (b | 0x40u) : 0x8000u) : b;
    return in ? t : (mag < LG_GLO ? small : big);
}
__device__ __forceinline__ float lg_gelu_grad(const float* __restrict__ T, unsigned b) {
    const unsigned mag = b & 0x7fffu, neg = b >> 15;
    const unsigned idx = mag - LG_GLO;
    const bool in = idx < LG_GN;
    const float t = T[(in ? idx : 0u) + neg * LG_GN];
    const float x = __uint_as_float(b << 16);
    const float lo = 0.5f + 0.79788456080286536f * x;
    const float hi = mag > 0x7f80u ? x : (neg ? 0.0f : 1.0f);
    return in ? t : (mag < LG_GLO ? lo : hi);
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void linear_gemm_kernel(const LgParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)SLAK_LDS(unsigned char, smem);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5, wave = wave_id_uniform();
    const int wn = wave >> 1, wm = wave & 1;                              // wave tile: n block wn, m block wm of the 192 x 192 tile
    unsigned char* const ST = smem + LG_NS * LG_STAGE + wave * LG_OBUF;   // this wave's staging tile
    uint16_t* const Lbias = (uint16_t*)(smem + LG_NS * LG_STAGE + 4 * LG_OBUF);       // [192] bf16
    unsigned char* const Ltab = smem + LG_NS * LG_STAGE + 4 * LG_OBUF + 512;          // value table (EPI 1: 11 KB, EPI 2: 22 KB)

    const int ntn = p.N / LG_T, nt = blockIdx.x % ntn, sm = blockIdx.x / ntn;
    const int n0 = nt * LG_T;
    const int mtiles = (p.M + LG_T - 1) / LG_T;
    const int my_tiles = sm < mtiles ? (mtiles - sm + p.SM - 1) / p.SM : 0;           // m-tiles sm, sm + SM, ...
    const int KCH = p.K / LG_KC;
    const int total = my_tiles * KCH;                                                 // chunks of this workgroup's stream

    // ---- one-time staging: bias slice, value table --------------------------------------------------------------------------------
    for (int i = tid; i < LG_T; i += 256) Lbias[i] = (EPI != 2 && p.bias) ? p.bias[n0 + i] : (uint16_t)0;
    if constexpr (EPI == 1) for (int i = tid; i < (int)(2 * LG_GN * 2) / 16; i += 256) ((u32x4*)Ltab)[i] = ((const u32x4*)p.table)[i];
    if constexpr (EPI == 2) for (int i = tid; i < (int)(2 * LG_GN * 4) / 16; i += 256) ((u32x4*)Ltab)[i] = ((const u32x4*)p.table)[i];

    // ---- DMA plan: 30 pieces per chunk (W rows: 15, X rows: 15), piece pi = wave + 4 k; destination chunk q -> (row q / 5, chunk q % 5) ------
    unsigned psrc[LG_NPW], pdst[LG_NPW]; bool pisW[LG_NPW];
#pragma unroll
    for (int k = 0; k < LG_NPW; ++k) {
        int pi = wave + 4 * k; if (pi >= 30) pi -= 2;                     // slots 30, 31 repeat pieces 28, 29 (same data to the same place)
        pisW[k] = pi < 15;
        const int pj = pisW[k] ? pi : pi - 15;
        const int q = 64 * pj + lane, r = q / 5, cc = q - r * 5;
        psrc[k] = cc < 4 ? (unsigned)r * (unsigned)p.K * 2u + (unsigned)cc * 16u : 0x80000000u;
        pdst[k] = (pisW[k] ? 0u : (unsigned)(LG_T * LG_RP)) + (unsigned)pj * 1024u;
    }
    // running state of the chunk to fetch next: (tile index ti, chunk kc); past the last tile the byte counts are 0 (zeros, fixed piece counts)
    int iss_t = 0, iss_kc = 0, n_issued = 0;
    v4i_t rW, rX; unsigned sb;
    auto next_desc = [&] {
        const int mt = sm + iss_t * p.SM;
        const bool real = iss_t < my_tiles;
        const long long m0 = (long long)mt * LG_T;
        const int rows = real ? (int)min((long long)LG_T, (long long)p.M - m0) : 0;
        const unsigned long long bw = (unsigned long long)(p.wt + (size_t)n0 * p.K + iss_kc * LG_KC);
        const unsigned long long bx = (unsigned long long)(p.x + (real ? (size_t)m0 * p.K : 0) + iss_kc * LG_KC);
        const int remw = real ? LG_T * p.K * 2 - iss_kc * (LG_KC * 2) : 0, remx = rows > 0 ? rows * p.K * 2 - iss_kc * (LG_KC * 2) : 0;
        rW = v4i_t{(int)(unsigned)bw, (int)((unsigned)(bw >> 32) & 0xffffu), remw, 0x00020000};
        rX = v4i_t{(int)(unsigned)bx, (int)((unsigned)(bx >> 32) & 0xffffu), remx, 0x00020000};
        sb = lds0 + (unsigned)(n_issued % LG_NS) * LG_STAGE;
        ++n_issued;
        if (++iss_kc == KCH) { iss_kc = 0; ++iss_t; }
    };
    auto sgpr4 = [](v4i_t v) { return v4i_t{__builtin_amdgcn_readfirstlane(v[0]), __builtin_amdgcn_readfirstlane(v[1]), __builtin_amdgcn_readfirstlane(v[2]), __builtin_amdgcn_readfirstlane(v[3])}; };
    auto piece = [&](int k) { lds_dma16(psrc[k], sgpr4(pisW[k] ? rW : rX), __builtin_amdgcn_readfirstlane(sb + pdst[k])); };

    f32x16 acc[3][3];
    auto zero_acc = [&] {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();
    float cs[3][16];                                                       // EPI 2: column sums of this lane's (i, quad, element) columns
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) cs[i][e] = 0.f;

    // fragments: lane -> row (l31) of a 32-row block, 8 consecutive k (lhi half of a 16-deep k step): one 16-byte read
    const unsigned fw = (unsigned)((wn * 96 + l31) * LG_RP + lhi * 16);
    const unsigned fx = (unsigned)(LG_T * LG_RP + (wm * 96 + l31) * LG_RP + lhi * 16);
    struct Frags { s16x8 a[2][3], b[2][3]; };
    auto load_frag = [&](Frags& F, const unsigned char* L, int f) {        // f = 0..11: (k step f / 6, operand (f % 6) / 3, 32-row block f % 3)
        const int ks = f / 6, op = (f % 6) / 3, blk = f % 3;
        if (op == 0) F.a[ks][blk] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + fw + blk * 32 * LG_RP + ks * 32));
        else F.b[ks][blk] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + fx + blk * 32 * LG_RP + ks * 32));
    };
#define LG_MMA(F, ks, m) acc[(m) / 3][(m) % 3] = mfma32<bf16_t>(F.a[ks][(m) / 3], F.b[ks][(m) % 3], acc[(m) / 3][(m) % 3])
#define LG_SB() __builtin_amdgcn_sched_barrier(0)
    // One chunk (see linear_wgrad.hip): 2 MFMAs | chunk g+1 confirmed landed + barrier (chunk g is in everyone's registers: its stage is free)
    // | 7 MFMAs each followed by a DMA piece of chunk g+3 (the eighth piece after the ninth MFMA) | 9 MFMAs each followed by fragment reads of g+1
    // EXTRA: the first two chunks after an epilogue also leave that epilogue's store instructions outstanding (they are YOUNGER than the pieces
    // waited for; a plain vmcnt(8) would wait for the stores to drain: 1-2 us per tile)
    constexpr int EPI_STORES = EPI == 1 ? 36 : 18;
    auto chunk = [&](int g, Frags& cur, Frags& nxt, auto extra_tag) {
        constexpr int EXTRA = decltype(extra_tag)::value ? EPI_STORES : 0;
        LG_MMA(cur, 0, 0); LG_MMA(cur, 0, 1); LG_SB();
        wait_vmcnt<1 * LG_NPW + EXTRA>();                                 // my pieces of chunk g+1 (only g+2's, and the stores, may be outstanding)
        wg_barrier(); LG_SB();
        next_desc();
#pragma unroll
        for (int m = 2; m < 9; ++m) {
            LG_MMA(cur, 0, m); piece(m - 2);
            if (m == 8) piece(7);
            LG_SB();
        }
        const unsigned char* const L = smem + ((g + 1) % LG_NS) * LG_STAGE;
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            LG_MMA(cur, 1, m);
            if (m < 3) { load_frag(nxt, L, 2 * m); load_frag(nxt, L, 2 * m + 1); } else load_frag(nxt, L, m + 3);
            LG_SB();
        }
    };

    // ---- epilogue of one tile (accumulators -> global), per wave; acc[i][j][4q + e] = Y[m = 32 j + l31][n = 32 i + 8 q + 4 lhi + e] ---------
    auto wsync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    auto flush = [&](uint16_t* __restrict__ dst, long long mrow0, int rows_ok) {       // staging tile -> 32 rows x 96 columns of dst (16 B per lane)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int idx = k * 64 + lane, row = idx / 12, c = idx - row * 12;
            if (row < rows_ok) *(u32x4*)(dst + (size_t)(mrow0 + row) * p.N + n0 + wn * 96 + c * 8) = *(const u32x4*)(ST + row * LG_OP + c * 16);
        }
    };
    auto epilogue = [&](int ti) {
        const long long m0 = (long long)(sm + ti * p.SM) * LG_T + wm * 96;
        u32x4 yreg[EPI == 2 ? 3 : 1][6];
        if constexpr (EPI == 2) {                                          // every y1 load of the tile before its first store (a load queued behind
#pragma unroll                                                             // a store would wait for that store to complete)
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int idx = k * 64 + lane, row = idx / 12, c = idx - row * 12;
                    yreg[j][k] = u32x4{0u, 0u, 0u, 0u};
                    if (m0 + 32 * j + row < p.M) yreg[j][k] = *(const u32x4*)(p.y1 + (size_t)(m0 + 32 * j + row) * p.N + n0 + wn * 96 + c * 8);
                }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const long long mr0 = m0 + 32 * j;
            const int rows_ok = (int)max(0LL, min(32LL, (long long)p.M - mr0));
            if constexpr (EPI == 2) {
                // the y1 block enters through the staging tile
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int idx = k * 64 + lane, row = idx / 12, c = idx - row * 12;
                    *(u32x4*)(ST + row * LG_OP + c * 16) = yreg[j][k];
                }
                wsync();
            }
            unsigned pg[3][8];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned char* const sp = ST + l31 * LG_OP + (32 * i + 8 * q + 4 * lhi) * 2;
                    float v[4];
                    if constexpr (EPI == 2) {
                        const u32x2 yy = *(const u32x2*)sp;
                        const float* const T = (const float*)Ltab;
                        v[0] = acc[i][j][4 * q + 0] * lg_gelu_grad(T, yy[0] & 0xffffu); v[1] = acc[i][j][4 * q + 1] * lg_gelu_grad(T, yy[0] >> 16);
                        v[2] = acc[i][j][4 * q + 2] * lg_gelu_grad(T, yy[1] & 0xffffu); v[3] = acc[i][j][4 * q + 3] * lg_gelu_grad(T, yy[1] >> 16);
                    } else {
                        const u32x2 bb = *(const u32x2*)(Lbias + wn * 96 + 32 * i + 8 * q + 4 * lhi);
                        v[0] = acc[i][j][4 * q + 0] + __uint_as_float(bb[0] << 16); v[1] = acc[i][j][4 * q + 1] + __uint_as_float(bb[0] & 0xffff0000u);
                        v[2] = acc[i][j][4 * q + 2] + __uint_as_float(bb[1] << 16); v[3] = acc[i][j][4 * q + 3] + __uint_as_float(bb[1] & 0xffff0000u);
                    }
                    const unsigned y01 = pack2<bf16_t>(v[0], v[1]), y23 = pack2<bf16_t>(v[2], v[3]);
                    if constexpr (EPI == 2) {                              // the bias gradient sums the ROUNDED values (rows past M hold zeros: y1 = 0 there, acc = 0)
                        cs[i][4 * q + 0] += __uint_as_float(y01 << 16); cs[i][4 * q + 1] += __uint_as_float(y01 & 0xffff0000u);
                        cs[i][4 * q + 2] += __uint_as_float(y23 << 16); cs[i][4 * q + 3] += __uint_as_float(y23 & 0xffff0000u);
                    }
                    if constexpr (EPI == 1) {
                        const uint16_t* const T = (const uint16_t*)Ltab;
                        pg[i][2 * q] = lg_gelu(T, y01 & 0xffffu) | (lg_gelu(T, y01 >> 16) << 16);
                        pg[i][2 * q + 1] = lg_gelu(T, y23 & 0xffffu) | (lg_gelu(T, y23 >> 16) << 16);
                    }
                    *(u32x2*)sp = u32x2{y01, y23};
                }
            wsync();
            flush(p.y, mr0, rows_ok);
            if constexpr (EPI == 1) {
                wsync();
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(u32x2*)(ST + l31 * LG_OP + (32 * i + 8 * q + 4 * lhi) * 2) = u32x2{pg[i][2 * q], pg[i][2 * q + 1]};
                wsync();
                flush(p.g, mr0, rows_ok);
            }
            wsync();
        }
        zero_acc();
    };

    __syncthreads();                                                      // bias / table staged
    if (total > 0) {
        Frags F0, F1;
        for (int c = 0; c < LG_NS; ++c) {
            next_desc();
#pragma unroll
            for (int k = 0; k < LG_NPW; ++k) piece(k);
        }
        wait_vmcnt<2 * LG_NPW>();
        wg_barrier();
#pragma unroll
        for (int f = 0; f < 12; ++f) load_frag(F0, smem, f);
        int kc = 0, ti = 0;
        for (int g = 0; g < total; g += 2) {                             // KCH is even: a tile ends after a pair
            if (kc == 0 && ti > 0) { chunk(g, F0, F1, std::true_type{}); chunk(g + 1, F1, F0, std::true_type{}); }
            else { chunk(g, F0, F1, std::false_type{}); chunk(g + 1, F1, F0, std::false_type{}); }
            kc += 2;
            if (kc == KCH) { if (p.dbg & 1) zero_acc(); else epilogue(ti); kc = 0; ++ti; }
        }
    }
#undef LG_MMA
#undef LG_SB
    wait_vmcnt<0>();
    if constexpr (EPI == 2) {
        // column sums: over the 32 lanes (rows) of each half wave, then one partial row per (workgroup, wm): colsum[(sm * 2 + wm)][n]
        float* const out = p.colsum + ((size_t)(sm * 2 + wm)) * p.N + n0 + wn * 96;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = cs[i][e];
#pragma unroll
                for (int k = 1; k < 32; k <<= 1) v += __shfl_xor(v, k, 64);
                if (l31 == 0) out[32 * i + 8 * (e >> 2) + 4 * lhi + (e & 3)] = v;
            }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static bool lg_shape_ok(int M, int N, int K) {
    return M >= 1 && N >= LG_T && N % LG_T == 0 && K >= 2 * LG_KC && K % (2 * LG_KC) == 0 && K <= 4096 &&
           (long long)M * N * 2 < (1LL << 32) && (long long)M * K * 2 < (1LL << 31) && (long long)LG_T * K * 2 < (1LL << 31);
}
static int lg_sm(int M, int N) {
    const int ntn = N / LG_T, mtiles = (M + LG_T - 1) / LG_T;
    int sm = mfma_cu_count() / ntn; if (sm < 1) sm = 1; if (sm > mtiles) sm = mtiles;
    return sm;
}
static size_t lg_lds(int epi) { return (size_t)LG_NS * LG_STAGE + 4 * LG_OBUF + 512 + (epi == 1 ? 2 * LG_GN * 2 : epi == 2 ? 2 * LG_GN * 4 : 0); }

// value tables in device memory, one copy per device (kind 0: bf16 GELU, 1: fp32 gelu')
static const void* lg_table_device(int kind) {
    static std::mutex mu;
    static const void* tab[2][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (tab[kind][dev]) return tab[kind][dev];
    std::vector<uint16_t> h16(2 * LG_GN); std::vector<float> h32(2 * LG_GN);
    for (unsigned sgn = 0; sgn < 2; ++sgn)
        for (unsigned i = 0; i < LG_GN; ++i) {
            const uint32_t bits = ((sgn << 15) | (LG_GLO + i)) << 16;
            float xf; memcpy(&xf, &bits, 4);
            const double x = xf, cdf = 0.5 * erfc(-x * 0.70710678118654752440);
            const float gf = (float)(x * cdf);
            uint32_t u; memcpy(&u, &gf, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            h16[sgn * LG_GN + i] = (uint16_t)(u >> 16);
            h32[sgn * LG_GN + i] = (float)(cdf + x * 0.39894228040143267794 * exp(-0.5 * x * x));
        }
    void* d = nullptr;
    const size_t bytes = kind == 0 ? h16.size() * 2 : h32.size() * 4;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    if (hipMemcpy(d, kind == 0 ? (const void*)h16.data() : (const void*)h32.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    tab[kind][dev] = d;
    return d;
}

template <int EPI> static int lg_launch(LgParams& p, hipStream_t st) {
    const size_t lds = lg_lds(EPI);
    auto k = linear_gemm_kernel<EPI>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SLAK_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)((p.N / LG_T) * p.SM)), dim3(256), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

}  // namespace slak

using namespace slak;

extern "C" {

int slak_linear_gemm_supported(int M, int N, int K) { return lg_shape_ok(M, N, K) ? 1 : 0; }
/* rows of the partial column-sum buffer of slak_linear_gemm_dgelu: colsum[rows][N], summed by the caller */
int slak_linear_gemm_colsum_rows(int M, int N) { return 2 * lg_sm(M, N); }

int slak_linear_gemm_gelu(const void* x, const void* wt, const void* bias, void* y, void* gelu_out, int M, int N, int K, void* stream) {
    if (!x || !wt || !y) return SLAK_ERR_INVALID_ARG;
    if (!lg_shape_ok(M, N, K)) return SLAK_ERR_UNSUPPORTED;
    LgParams p{};
    p.x = (const uint16_t*)x; p.wt = (const uint16_t*)wt; p.bias = (const uint16_t*)bias; p.y = (uint16_t*)y; p.g = (uint16_t*)gelu_out;
    p.M = M; p.N = N; p.K = K; p.SM = lg_sm(M, N);
    { static const int dbg = [] { const char* e = getenv("SLAK_LG_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }
    if (gelu_out) { p.table = lg_table_device(0); if (!p.table) return SLAK_ERR_LAUNCH; return lg_launch<1>(p, (hipStream_t)stream); }
    return lg_launch<0>(p, (hipStream_t)stream);
}

int slak_linear_gemm_dgelu(const void* dz, const void* wt, const void* y1, void* dy1, float* colsum, int M, int N, int K, void* stream) {
    if (!dz || !wt || !y1 || !dy1 || !colsum) return SLAK_ERR_INVALID_ARG;
    if (!lg_shape_ok(M, N, K)) return SLAK_ERR_UNSUPPORTED;
    LgParams p{};
    p.x = (const uint16_t*)dz; p.wt = (const uint16_t*)wt; p.y = (uint16_t*)dy1; p.y1 = (const uint16_t*)y1; p.colsum = colsum;
    p.M = M; p.N = N; p.K = K; p.SM = lg_sm(M, N);
    p.table = lg_table_device(1); if (!p.table) return SLAK_ERR_LAUNCH;
    return lg_launch<2>(p, (hipStream_t)stream);
}

}  // extern "C"
