// slak_amd/csrc/linear_skinny.hip -- the pointwise (1x1) convolutions of a SLaK block on the LARGEST map, where they are not GEMMs in
// any compute sense: pwconv1 / pwconv2 (models/SLaK.py:158-160) and their data gradients at stage 1 of SLaK-T multiply a
// 401,408 x 96 (x 384) activation matrix by a 74 KB weight: 2-12 flop per byte moved, pure HBM streaming.  hipBLASLt's tiles run them
// at 1.8-2.8 TB/s (tools/time_gemm.py).
//
//   Y[M x N] = X[M x K] . Wt[N x K]^T (+ bias[N]),  optionally  G = gelu(Y)  written alongside (nn.GELU(), exact erf form, evaluated
//   on the ROUNDED y as autocast does: F.gelu of a bf16 tensor)
//
// X, Wt, bias, Y, G bf16, fp32 accumulate, "NT" (both operands K-contiguous).  One persistent 8-wave workgroup per CU:
//   * the whole weight sits in LDS (74 KB, rows padded to an odd number of 16-byte chunks: row-per-lane ds_read_b128 fragments are
//     then bank-conflict free), staged once;
//   * a wave owns 32 rows of X at a time.  Their 96-column block is one LDS-DMA burst (`buffer_load_dwordx4 ... lds`; the padded pitch
//     is made on the source side: destination chunk q takes source chunk (q / 13) * ld + q % 13, pad chunks are skipped lanes), the
//     six operand fragments are read into registers, and the NEXT block's DMA is issued right away into the same buffer -- the
//     registers are the second buffer.  No workgroup barrier after the prologue.
//   * operands swapped (D^T = Wt-tile x X-tile^T) so that a lane holds 4 consecutive output columns of ONE row; two
//     v_permlane32_swap per register pair turn that into 8 consecutive columns = one 16-byte store (row-per-lane 8-byte stores are
//     issue-bound at ~7 B/clk/CU: MI355X_MICROARCH.md, store tail).
//   (A first version read the fragments straight from global memory, 16 bytes per lane at a 192-byte stride: address-coalescer bound,
//   no faster than the library.)
//   linear_nt_k96:  K = 96,  N = 32 NT <= 384  (pwconv1 forward [+ GELU], dz . W2):   walks the N/32 column tiles per row block
//   linear_nt_n96:  N = 96,  K = 96 NKC <= 384 (pwconv2 forward, dy1 . W1):           three accumulators, walks K in 96-column blocks
#include "mfma_common.h"
#include "gelu_grad.h"
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace slak {

constexpr int LS_WAVES = 8, LS_THREADS = LS_WAVES * 64;
constexpr int LS_XP = 208;                       // pitch (bytes) of a 96-column block in LDS: 13 chunks
constexpr int LS_XBUF = 32 * LS_XP;              // one wave's X block: 6,656 B

// nn.GELU() (exact erf form) with erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7 absolute, far below the bf16 rounding of the result;
// the same evaluation as the backward kernel of block_tail.hip): one exp, one rcp, and the lower tail without cancellation
// (Phi(x) = poly e / 2 for x < 0).  ocml's erff costs 2-3x as many VALU instructions: the fused pwconv1 kernel was VALU-bound on it.
__device__ __forceinline__ float gelu_erf(float x) {
    const float e = __expf(-0.5f * x * x);
    const float t = __frcp_rn(1.0f + 0.3275911f * (fabsf(x) * 0.70710678118654752f));
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float half = 0.5f * poly * e;
    return x * (x >= 0.f ? 1.0f - half : half);
}
// GELU of a bf16 VALUE by table: the fused pwconv1 kernel applies nn.GELU() to the ROUNDED pre-activation, a bf16 number, and returns a
// bf16 number -- a function on 65,536 points.  Below |x| = 2^-9 GELU(x) rounds to x/2, above 16 to x (x > 0) or -0; the table covers
// 2^-18 <= |x| < 16 (22 exponents x 128 mantissas x 2 signs = 5,632 entries, 11 KB of LDS: what is left beside the weight) so that a
// whole wave is almost never outside it and takes the short path: four integer VALU instructions + one 2-byte LDS read per element
// against ~30 fp32 instructions with exp and rcp for gelu_erf.  The table holds the correctly rounded value (host, double precision).
// The kernel was VALU-bound on the evaluation: 275 us with gelu_erf, 162 us with none, per 154 M elements.
constexpr unsigned GL_LO = 109u << 7, GL_N = 22u << 7;             // first table magnitude (2^-18), entries per sign
constexpr int GL_BYTES = 2 * (int)GL_N * 2;
__device__ __forceinline__ unsigned gelu_lut(const uint16_t* __restrict__ T, unsigned b) {      // b: bf16 bits (upper 16 bits of the register zero)
    const unsigned mag = b & 0x7fffu, neg = b >> 15;
    const unsigned idx = mag - GL_LO;                             // wraps for |x| < 2^-9
    const bool in = idx < GL_N;
    const unsigned t = T[(in ? idx : 0u) + neg * GL_N];
    const unsigned small = mag >= 0x100u ? b - 0x80u : (b & 0x8000u);                           // x / 2 (exact; subnormal inputs: signed zero)
    const unsigned big = neg ? (mag > 0x7f7fu ? (b | 0x40u) : 0x8000u) : b;                     // x, -0, NaN for -inf / NaN
    return in ? t : (mag < GL_LO ? small : big);
}
__device__ __forceinline__ unsigned gelu_lut2(const uint16_t* __restrict__ T, unsigned pair) {
    const unsigned lo = pair & 0xffffu, hi = pair >> 16;
    const unsigned il = (lo & 0x7fffu) - GL_LO, ih = (hi & 0x7fffu) - GL_LO;
    if (__builtin_amdgcn_ballot_w64(il >= GL_N || ih >= GL_N) == 0)                               // (wave-uniform) everything inside the table
        return (unsigned)T[il + (lo >> 15) * GL_N] | ((unsigned)T[ih + (hi >> 15) * GL_N] << 16);
    return gelu_lut(T, lo) | (gelu_lut(T, hi) << 16);
}
// eight pairs at once: ONE wave-uniform range test and sixteen gathers in flight together (pair by pair, every pair's two gathers sit behind their
// own branch and expose a full LDS latency: 96 of them per 32 x 384 block were a third of the fused kernels' time)
__device__ __forceinline__ void gelu_lut2x8(const uint16_t* __restrict__ T, const unsigned (&y)[8], unsigned (&g)[8]) {
    unsigned il[8], ih[8];
    bool out = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        il[k] = (y[k] & 0x7fffu) - GL_LO; ih[k] = ((y[k] >> 16) & 0x7fffu) - GL_LO;
        out = out || il[k] >= GL_N || ih[k] >= GL_N;
    }
    if (__builtin_amdgcn_ballot_w64(out) == 0) {                     // (wave-uniform) everything inside the table
        unsigned lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { lo[k] = T[il[k] + ((y[k] >> 15) & 1u) * GL_N]; hi[k] = T[ih[k] + (y[k] >> 31) * GL_N]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = lo[k] | (hi[k] << 16);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = gelu_lut(T, y[k] & 0xffffu) | (gelu_lut(T, y[k] >> 16) << 16);
    }
}
__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// One 32 x 32 output tile in registers (lane = row l31, acc[4q + j] = column col0 + 8q + 4 lhi + j) -> bias, rounding, optional GELU,
// 16-byte stores.
template <bool GELU>
__device__ __forceinline__ void store_tile(const f32x16& acc, const uint16_t* __restrict__ bias, uint16_t* __restrict__ Y,
                                           uint16_t* __restrict__ G, size_t row_off, int col0, int lhi, bool row_ok) {
    unsigned py[8], pg[8];                                     // packed pairs: quad q -> py[2q], py[2q+1]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = col0 + 8 * q + 4 * lhi;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (bias) {
            const u32x2 bb = *(const u32x2*)(bias + c);
            b0 = bf16_lo(bb[0]); b1 = bf16_hi(bb[0]); b2 = bf16_lo(bb[1]); b3 = bf16_hi(bb[1]);
        }
        const unsigned y01 = pack2<bf16_t>(acc[4 * q + 0] + b0, acc[4 * q + 1] + b1);
        const unsigned y23 = pack2<bf16_t>(acc[4 * q + 2] + b2, acc[4 * q + 3] + b3);
        py[2 * q] = y01; py[2 * q + 1] = y23;
        if constexpr (GELU) {
            pg[2 * q] = pack2<bf16_t>(gelu_erf(bf16_lo(y01)), gelu_erf(bf16_hi(y01)));
            pg[2 * q + 1] = pack2<bf16_t>(gelu_erf(bf16_lo(y23)), gelu_erf(bf16_hi(y23)));
        }
    }
    // lanes l31 (lhi 0) and l31 + 32 (lhi 1) hold the same row: quads (0,1) -> lhi 0 keeps columns 0..7, lhi 1 gets 8..15; quads (2,3)
    // likewise 16..23 / 24..31.  v_permlane32_swap(a, b): a of lanes 32..63 <-> b of lanes 0..31.
    // (the builtin, not inline asm: the compiler then inserts the wait states the swap needs after a VALU write of its operands)
    auto swap = [](unsigned& a, unsigned& b) { const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        swap(py[4 * h + 0], py[4 * h + 2]); swap(py[4 * h + 1], py[4 * h + 3]);
        if constexpr (GELU) { swap(pg[4 * h + 0], pg[4 * h + 2]); swap(pg[4 * h + 1], pg[4 * h + 3]); }
    }
    if (row_ok) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t o = row_off + (size_t)(col0 + 16 * h + 8 * lhi);
            *(u32x4*)(Y + o) = u32x4{py[4 * h + 0], py[4 * h + 1], py[4 * h + 2], py[4 * h + 3]};
            if constexpr (GELU) *(u32x4*)(G + o) = u32x4{pg[4 * h + 0], pg[4 * h + 1], pg[4 * h + 2], pg[4 * h + 3]};
        }
    }
}

// Stage Wt [N][K] (row-major, K*2 bytes per row) into LDS rows of pitch WP bytes.
__device__ __forceinline__ void stage_weight(char* Lw, const uint16_t* __restrict__ Wt, int N, int K, int WP, int tid, int nthreads) {
    const int cpr = K >> 3, total = N * cpr;                          // 16-byte chunks per row
    for (int q = tid; q < total; q += nthreads) {
        const int r = q / cpr, cc = q - r * cpr;
        *(u32x4*)(Lw + (size_t)r * WP + cc * 16) = *(const u32x4*)(Wt + (size_t)r * K + cc * 8);
    }
}

// DMA of one 32-row x 96-column block of X (row stride ldx elements, starting at element column kc0) into a wave's LDS buffer:
// destination chunk q = 64 k + lane -> (row q / 13, chunk q % 13); 7 instructions cover the 416 chunks.  Rows >= M are skipped.
struct XDma {
    unsigned src[7];                 // source byte offset relative to the block's first element, 0xffffffff = skip
    int row[7];
};
__device__ __forceinline__ void xdma_plan(XDma& d, int lane, unsigned ldx_bytes) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int q = 64 * k + lane, r = q / 13, cc = q - r * 13;
        const bool ok = r < 32 && cc < 12;
        d.src[k] = ok ? (unsigned)r * ldx_bytes + (unsigned)cc * 16u : 0xffffffffu;
        d.row[k] = r;
    }
}
__device__ __forceinline__ void xdma_issue(const XDma& d, unsigned base_off, int rows_valid, v4i_t rsrc, unsigned lds_dst) {
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (d.src[k] != 0xffffffffu && d.row[k] < rows_valid) lds_dma16(base_off + d.src[k], rsrc, __builtin_amdgcn_readfirstlane(lds_dst + k * 1024));
}

constexpr int LK_WAVES = 6, LK_THREADS = LK_WAVES * 64;          // k96: the weight (80 KB) + per-wave X block and out tile leave room for six waves
constexpr int LK_OP = 144;                                         // pitch (bytes) of a wave's 32 x 64 out tile: 128 + 16
constexpr int LK_OBUF = 32 * LK_OP;

// packed results of one 32 x 32 tile -> the wave's out tile (columns half*32 ..), 8 bytes per quad
__device__ __forceinline__ void put_tile(char* ot, const unsigned (&p)[8], int l31, int lhi, int half) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *(u32x2*)(ot + l31 * LK_OP + half * 64 + (8 * q + 4 * lhi) * 2) = u32x2{p[2 * q], p[2 * q + 1]};
}
// the out tile (32 rows x 128 bytes) -> HBM: lane = (row lane/8, 16-byte chunk lane%8): eight FULL 128-byte lines per store instruction
// (row-per-lane 16-byte stores wrote 32 bytes per row and instruction: 2.6 TB/s, what the library's epilogue reaches too)
__device__ __forceinline__ void flush_tile(const char* ot, uint16_t* __restrict__ dst, int tm, int M, int N, int col0, int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4 v = *(const u32x4*)(ot + r * LK_OP + c * 16);
        const int row = tm * 32 + r;
        if (row < M) *(u32x4*)(dst + (size_t)row * N + col0 + c * 8) = v;
    }
}

template <bool GELU>
__global__ __launch_bounds__(LK_THREADS, 1) void linear_nt_k96_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                                    const uint16_t* __restrict__ bias, uint16_t* __restrict__ Y,
                                                                    uint16_t* __restrict__ G, int M, int N, unsigned x_bytes,
                                                                    const uint16_t* __restrict__ gelu_table) {
    constexpr int K = 96, KS = 6;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    char* const Lw = L;                                               // [N][LS_XP]
    const unsigned bias_b = (unsigned)N * LS_XP;                      // [N] bf16 (1 KB): bias reads must not touch vmcnt
    const unsigned xbuf = bias_b + 1024u + (unsigned)wave * LS_XBUF;
    char* const ot = L + bias_b + 1024u + LK_WAVES * LS_XBUF + wave * LK_OBUF;
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)X;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)x_bytes); rsrc[3] = 0x00020000;
    }
    XDma plan; xdma_plan(plan, lane, K * 2);
    const int ntiles_m = (M + 31) >> 5, stride = gridDim.x * LK_WAVES;
    int tm = blockIdx.x * LK_WAVES + wave;
    if (tm < ntiles_m) xdma_issue(plan, (unsigned)tm * 32u * K * 2u, M - tm * 32, rsrc, lds_base + xbuf);
    stage_weight(Lw, Wt, N, K, LS_XP, tid, LK_THREADS);
    for (int i = tid; i < N; i += LK_THREADS) ((uint16_t*)(L + bias_b))[i] = bias ? bias[i] : (uint16_t)0;
    const uint16_t* const glut = (const uint16_t*)(L + bias_b + 1024u + LK_WAVES * (LS_XBUF + LK_OBUF));
    if constexpr (GELU) for (int i = tid; i < GL_BYTES / 16; i += LK_THREADS) ((u32x4*)glut)[i] = ((const u32x4*)gelu_table)[i];
    __syncthreads();                                                  // the only workgroup barrier
    const uint16_t* const lbias = (const uint16_t*)(L + bias_b);
    const int npairs = N >> 6;                                        // column tiles are flushed in pairs (64 columns = one 128-byte line per row)
    const int nst = npairs * 4 * (GELU ? 2 : 1);                      // store instructions of one row block (all issued: every block has a valid row)
    int pending = 0;
    const unsigned wlane = (unsigned)l31 * LS_XP + (unsigned)lhi * 16u;
    auto lsync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    for (; tm < ntiles_m; tm += stride) {
        wait_vmcnt_dyn(pending);                                      // my block has landed: only the stores issued after its DMA may be outstanding
        __builtin_amdgcn_wave_barrier();
        s16x8 xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xbuf + wlane + ks * 32));
        lsync();                                                      // every lane has its fragments: the buffer is free
        pending = nst;
        const int tn = tm + stride;
        if (tn < ntiles_m) xdma_issue(plan, (unsigned)tn * 32u * K * 2u, M - tn * 32, rsrc, lds_base + xbuf);
        for (int pr = 0; pr < npairs; ++pr) {
            unsigned py[2][8], pg[2][8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int nt = 2 * pr + half;
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                const char* wt = Lw + (size_t)nt * 32 * LS_XP + wlane;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = mfma32<bf16_t>(__builtin_bit_cast(s16x8, *(const u32x4*)(wt + ks * 32)), xf[ks], acc);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x2 bb = *(const u32x2*)(lbias + nt * 32 + 8 * q + 4 * lhi);
                    const unsigned y01 = pack2<bf16_t>(acc[4 * q + 0] + bf16_lo(bb[0]), acc[4 * q + 1] + bf16_hi(bb[0]));
                    const unsigned y23 = pack2<bf16_t>(acc[4 * q + 2] + bf16_lo(bb[1]), acc[4 * q + 3] + bf16_hi(bb[1]));
                    py[half][2 * q] = y01; py[half][2 * q + 1] = y23;
                }
                if constexpr (GELU) gelu_lut2x8(glut, py[half], pg[half]);
            }
            put_tile(ot, py[0], l31, lhi, 0); put_tile(ot, py[1], l31, lhi, 1);
            lsync();
            flush_tile(ot, Y, tm, M, N, pr * 64, lane);
            if constexpr (GELU) {
                lsync();                                              // the tile has been read
                put_tile(ot, pg[0], l31, lhi, 0); put_tile(ot, pg[1], l31, lhi, 1);
                lsync();
                flush_tile(ot, G, tm, M, N, pr * 64, lane);
            }
            lsync();
        }
    }
}

// pwconv1 -> GELU -> pwconv2 of stage 1 in ONE pass (models/SLaK.py:158-160: x = pwconv2(act(pwconv1(x))), C = 96): the k96 kernel above with the
// second product in its epilogue.  A 32 x 32 tile of a = gelu(y1) leaves the first product as packed bf16 pairs with lane = row, eight
// k-values per lane and half-wave -- exactly an MFMA B operand of the second product z^T = W2 . a^T if the matching W2 fragment takes its k in
// the same (permuted) order: lane (row n2, half h) of k-step u of column tile nt holds W2[n2][32 nt + 16 u + 4 h + {0..3, 8..11}].  Those 72
// fragments are loop invariants: each wave keeps ALL of W2 (96 x 384) in 288 registers -- four waves per workgroup, one per SIMD, 512
// registers each; LDS holds W1, the biases, the GELU table and the waves' X blocks / out tiles as before.  y1 and a are still written (the
// backward reads them), but a is not read again and pwconv2 is not a launch: 385 MB and ~110 us less per block.  z differs from the
// two-launch result by the order in which an MFMA adds its 16 products (the k-permutation): fp32 rounding noise, far below z's bf16 rounding.
constexpr int LM_WAVES = 4, LM_THREADS = LM_WAVES * 64, LM_NP = 6;   // N1 = 64 LM_NP = 384, N2 = K = 96

__global__ __launch_bounds__(LM_THREADS, 1) void linear_mlp_fwd_k96_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W1,
                                                                         const uint16_t* __restrict__ B1, const uint16_t* __restrict__ W2,
                                                                         const uint16_t* __restrict__ B2, uint16_t* __restrict__ Y1,
                                                                         uint16_t* __restrict__ A, uint16_t* __restrict__ Z, int M, unsigned x_bytes,
                                                                         const uint16_t* __restrict__ gelu_table) {
    constexpr int K = 96, KS = 6, N = 64 * LM_NP, N2 = 96;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    char* const Lw = L;                                               // [N][LS_XP]
    const unsigned bias_b = (unsigned)N * LS_XP;                      // [N] bf16 bias of pwconv1, [96] of pwconv2 behind it (1 KB together)
    const unsigned xbuf = bias_b + 1024u + (unsigned)wave * LS_XBUF;
    char* const ot = L + bias_b + 1024u + LM_WAVES * LS_XBUF + wave * 2 * LK_OBUF;      // two out tiles: y1's and a's 32 x 64 block
    char* const og = ot + LK_OBUF;
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    // W2 -> registers (loop invariant), k in the order the a tiles arrive in
    s16x8 w2f[3][2 * LM_NP][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int nt = 0; nt < 2 * LM_NP; ++nt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint16_t* src = W2 + (size_t)(j * 32 + l31) * N + nt * 32 + 16 * u + 4 * lhi;
                const u32x2 lo = *(const u32x2*)src, hi = *(const u32x2*)(src + 8);
                w2f[j][nt][u] = __builtin_bit_cast(s16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)X;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)x_bytes); rsrc[3] = 0x00020000;
    }
    XDma plan; xdma_plan(plan, lane, K * 2);
    const int ntiles_m = (M + 31) >> 5, stride = gridDim.x * LM_WAVES;
    int tm = blockIdx.x * LM_WAVES + wave;
    if (tm < ntiles_m) xdma_issue(plan, (unsigned)tm * 32u * K * 2u, M - tm * 32, rsrc, lds_base + xbuf);
    stage_weight(Lw, W1, N, K, LS_XP, tid, LM_THREADS);
    for (int i = tid; i < N + N2; i += LM_THREADS) ((uint16_t*)(L + bias_b))[i] = i < N ? (B1 ? B1[i] : (uint16_t)0) : (B2 ? B2[i - N] : (uint16_t)0);
    const uint16_t* const glut = (const uint16_t*)(L + bias_b + 1024u + LM_WAVES * (LS_XBUF + 2 * LK_OBUF));
    for (int i = tid; i < GL_BYTES / 16; i += LM_THREADS) ((u32x4*)glut)[i] = ((const u32x4*)gelu_table)[i];
    __syncthreads();                                                  // the only workgroup barrier
    const uint16_t* const lbias = (const uint16_t*)(L + bias_b);
    constexpr int nst = LM_NP * 8 + 6;                                // store instructions of one row block: y1 and a (4 per pair each), z (2 per 32 columns)
    int pending = 0;
    const unsigned wlane = (unsigned)l31 * LS_XP + (unsigned)lhi * 16u;
    auto lsync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    for (; tm < ntiles_m; tm += stride) {
        wait_vmcnt_dyn(pending);                                      // my block has landed: only the stores issued after its DMA may be outstanding
        __builtin_amdgcn_wave_barrier();
        s16x8 xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xbuf + wlane + ks * 32));
        lsync();                                                      // every lane has its fragments: the buffer is free
        pending = nst;
        const int tn = tm + stride;
        if (tn < ntiles_m) xdma_issue(plan, (unsigned)tn * 32u * K * 2u, M - tn * 32, rsrc, lds_base + xbuf);
        f32x16 zacc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) zacc[j][i] = 0.f;
#pragma unroll
        for (int pr = 0; pr < LM_NP; ++pr) {
            unsigned py[2][8], pg[2][8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int nt = 2 * pr + half;
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                const char* wt = Lw + (size_t)nt * 32 * LS_XP + wlane;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = mfma32<bf16_t>(__builtin_bit_cast(s16x8, *(const u32x4*)(wt + ks * 32)), xf[ks], acc);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x2 bb = *(const u32x2*)(lbias + nt * 32 + 8 * q + 4 * lhi);
                    const unsigned y01 = pack2<bf16_t>(acc[4 * q + 0] + bf16_lo(bb[0]), acc[4 * q + 1] + bf16_hi(bb[0]));
                    const unsigned y23 = pack2<bf16_t>(acc[4 * q + 2] + bf16_lo(bb[1]), acc[4 * q + 3] + bf16_hi(bb[1]));
                    py[half][2 * q] = y01; py[half][2 * q + 1] = y23;
                }
                gelu_lut2x8(glut, py[half], pg[half]);
                // second product: this tile's 32 columns of a are 32 of its k
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const s16x8 bf = __builtin_bit_cast(s16x8, u32x4{pg[half][4 * u], pg[half][4 * u + 1], pg[half][4 * u + 2], pg[half][4 * u + 3]});
#pragma unroll
                    for (int j = 0; j < 3; ++j) zacc[j] = mfma32<bf16_t>(w2f[j][nt][u], bf, zacc[j]);
                }
            }
            // (no waits around the tiles: a wave's LDS operations execute in program order, so the flush reads below see these writes and the
            // next pair's writes come after them; the compiler keeps that order -- same base pointer -- and is otherwise free to run the next
            // pair's MFMAs under this pair's epilogue.  With a wait on either side the wave, alone on its SIMD, idled through every LDS round trip.)
            put_tile(ot, py[0], l31, lhi, 0); put_tile(ot, py[1], l31, lhi, 1);
            put_tile(og, pg[0], l31, lhi, 0); put_tile(og, pg[1], l31, lhi, 1);
            flush_tile(ot, Y1, tm, M, N, pr * 64, lane);
            flush_tile(og, A, tm, M, N, pr * 64, lane);
        }
        const int row = tm * 32 + l31;
#pragma unroll
        for (int j = 0; j < 3; ++j) store_tile<false>(zacc[j], lbias + N, Z, nullptr, (size_t)row * N2, j * 32, lhi, row < M);
    }
}

// dz . W2 WITH nn.GELU()'s backward in the epilogue (models/SLaK.py:159-160 backwards): dy1 = round(dz . W2) * gelu'(y1), rounded, and the column
// sums of dy1 (pwconv1's bias gradient) -- the k96 kernel above, whose output tile (dact, rounded to bf16 as the stand-alone GEMM stores it)
// meets the stored pre-activation y1 in the flush layout (lane = row lane/8, 16-byte chunk lane%8: full 128-byte lines in both directions)
// instead of making a round trip through HBM: dact is never written, the stand-alone gelu_bwd_bias pass (3 x M x N x 2 bytes) disappears.
// Same evaluation as gelu_bwd_bias_kernel (gelu_grad.h): the same dy1 bits.  FOUR waves, one per SIMD with the whole register file: the 24
// y1 loads (96 registers) of the NEXT row block leave together with its X DMA, a block ahead of their use, into a second register set; the wait at the top of a block -- only the previous block's 24 stores may be outstanding --
// covers both.  Column sums: eight per lane and column pair, reduced over the eight row lanes at the end, one partial row per wave;
// tail_reduce_columns adds the rows in a fixed order.
constexpr int LG_WAVES = 4, LG_THREADS = LG_WAVES * 64, LG_NP = 6;   // N = 64 LG_NP = 384

__global__ __launch_bounds__(LG_THREADS, 1) void linear_nt_k96_gbwd_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                                         const uint16_t* __restrict__ Y1, uint16_t* __restrict__ DY,
                                                                         float* __restrict__ part, int M, unsigned x_bytes, unsigned y_bytes,
                                                                         const float* __restrict__ table) {
    constexpr int K = 96, KS = 6, N = 64 * LG_NP;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    char* const Lw = L;                                               // [N][LS_XP]
    const unsigned xbuf = (unsigned)N * LS_XP + (unsigned)wave * LS_XBUF;
    char* const ot = L + (unsigned)N * LS_XP + LG_WAVES * LS_XBUF + wave * LK_OBUF;
    const float* const T = (const float*)(L + (unsigned)N * LS_XP + LG_WAVES * (LS_XBUF + LK_OBUF));
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)X;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)x_bytes); rsrc[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Y1), 0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(DY, 0, (int)y_bytes, 0x00020000);
    XDma plan; xdma_plan(plan, lane, K * 2);
    const int ntiles_m = M >> 5, stride = gridDim.x * LG_WAVES;       // (M % 32 == 0: every row of every block exists)
    const int fr = lane >> 3, fc = lane & 7;                          // flush layout: row fr + 8 it, 16-byte chunk fc of the 64-column pair
    const unsigned flane = (unsigned)fr * (unsigned)(N * 2) + (unsigned)fc * 16u;
    // the 24 y1 loads of row block t, always issued (a block behind the matrix: out of range, zeros).  (A macro, not a lambda: register arrays
    // taken by reference end up in scratch memory.)
#define SLAK_LG_LOAD_Y(t_, yv_) { \
        const unsigned g0_ = (t_) < ntiles_m ? (unsigned)(t_) * 32u * (unsigned)(N * 2) + flane : 0x80000000u; \
        _Pragma("unroll") for (int pr = 0; pr < LG_NP; ++pr) \
            _Pragma("unroll") for (int it = 0; it < 4; ++it) yv_[pr][it] = __builtin_amdgcn_raw_buffer_load_b128(ry, g0_ + (unsigned)(it * 8 * N * 2), pr * 128, 0); }
    int tm = blockIdx.x * LG_WAVES + wave;
    u32x4 cur[LG_NP][4], nxt[LG_NP][4];
    if (tm < ntiles_m) xdma_issue(plan, (unsigned)tm * 32u * K * 2u, 32, rsrc, lds_base + xbuf);
    SLAK_LG_LOAD_Y(tm, cur)
    stage_weight(Lw, Wt, N, K, LS_XP, tid, LG_THREADS);
    for (int i = tid; i < GD_BYTES / 16; i += LG_THREADS) ((u32x4*)T)[i] = ((const u32x4*)table)[i];
    __syncthreads();                                                  // the only workgroup barrier
    constexpr int nst = LG_NP * 4;                                    // store instructions of one row block
    int pending = 0;
    const unsigned wlane = (unsigned)l31 * LS_XP + (unsigned)lhi * 16u;
    float acc[LG_NP][8];
#pragma unroll
    for (int pr = 0; pr < LG_NP; ++pr)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[pr][k] = 0.f;
    auto lsync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    for (; tm < ntiles_m; tm += stride) {
        // my X block and my y1 registers have landed: their DMA / loads left a block ago, only the previous block's stores came after them
        wait_vmcnt_dyn(pending);
        __builtin_amdgcn_wave_barrier();
        s16x8 xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xbuf + wlane + ks * 32));
        lsync();                                                      // every lane has its fragments: the buffer is free
        pending = nst;
        const int tn = tm + stride;
        if (tn < ntiles_m) xdma_issue(plan, (unsigned)tn * 32u * K * 2u, 32, rsrc, lds_base + xbuf);
        SLAK_LG_LOAD_Y(tn, nxt)
        const unsigned g0 = (unsigned)tm * 32u * (unsigned)(N * 2) + flane;
#pragma unroll
        for (int pr = 0; pr < LG_NP; ++pr) {
            unsigned py[2][8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int nt = 2 * pr + half;
                f32x16 a;
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = 0.f;
                const char* wt = Lw + (size_t)nt * 32 * LS_XP + wlane;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a = mfma32<bf16_t>(__builtin_bit_cast(s16x8, *(const u32x4*)(wt + ks * 32)), xf[ks], a);
#pragma unroll
                for (int q = 0; q < 4; ++q) { py[half][2 * q] = pack2<bf16_t>(a[4 * q + 0], a[4 * q + 1]); py[half][2 * q + 1] = pack2<bf16_t>(a[4 * q + 2], a[4 * q + 3]); }
            }
            put_tile(ot, py[0], l31, lhi, 0); put_tile(ot, py[1], l31, lhi, 1);
            lsync();
            // the pair's four 8-element chunks with ONE wave-uniform range test: 32 table gathers in flight together
            uint4 gq[4], vq[4];
            bool ok = true;
            float a2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a2[k] = acc[pr][k];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const u32x4 g4 = *(const u32x4*)(ot + (it * 8 + fr) * LK_OP + fc * 16);
                gq[it] = uint4{g4[0], g4[1], g4[2], g4[3]};
            }
            float tq[4][8];
#pragma unroll
            for (int it = 0; it < 4; ++it) ok = gelu_grad_gather8(T, uint4{cur[pr][it][0], cur[pr][it][1], cur[pr][it][2], cur[pr][it][3]}, tq[it]) && ok;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
                for (int it = 0; it < 4; ++it) gelu_bwd8_apply(gq[it], tq[it], vq[it], a2);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[pr][k] = a2[k];
            } else {                                                  // (wave-uniform, rare) an element outside the table: the general evaluation
#pragma unroll
                for (int it = 0; it < 4; ++it) gelu_bwd8(T, gq[it], uint4{cur[pr][it][0], cur[pr][it][1], cur[pr][it][2], cur[pr][it][3]}, vq[it], acc[pr]);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{vq[it].x, vq[it].y, vq[it].z, vq[it].w}, rd, g0 + (unsigned)(it * 8 * N * 2), pr * 128, 0);
            lsync();
        }
        // the next block's y1 becomes the current one (96 register moves: ~4 % of a block; the compiler's wait in front of them counts the
        // 24 stores above, and those loads left before this block's first MFMA)
#pragma unroll
        for (int pr = 0; pr < LG_NP; ++pr)
#pragma unroll
            for (int it = 0; it < 4; ++it) cur[pr][it] = nxt[pr][it];
    }
#undef SLAK_LG_LOAD_Y
    // column sums: over the eight row lanes, then one partial row per wave
#pragma unroll
    for (int pr = 0; pr < LG_NP; ++pr)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = acc[pr][k];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 8) part[((size_t)blockIdx.x * LG_WAVES + wave) * N + pr * 64 + lane * 8 + k] = v;
        }
}

// The kernel above with the NEXT product of the block's backward in it (round 5): dt = dy1 . W1 (pwconv1's data gradient, K = 384 -> 96 columns).
// A pair's dy1 chunks go back into the wave's out tile, where a row's 64 columns lie k-contiguous -- the MFMA B operand of dt^T = W1^T . dy1^T;
// W1^T's twelve fragments of the pair (3 row tiles x 4 k-steps, fragment-major in global memory: 1 KB per load, L2 resident) are fetched at the
// top of the pair, a GEMM and a table pass ahead of their use.  The stand-alone slak_linear_nt launch with its 308 MB re-read of dy1 is gone.
// What had to give is registers -- the VALU side of this kernel (y1, table values, column sums) lives in the 256 architectural VGPRs, and a first
// version that only added the fragments and the accumulators spilled 30-100 of them to scratch (every scratch access drains vmcnt: 338 us
// against 290 for the two launches).  Here: y1 is prefetched THREE pairs ahead (a rolling window of three register sets, 48 registers, instead of
// a whole row block in two sets, 192), and the GELU' pass runs in two halves of two chunks (16 table values in flight instead of 32).
__global__ __launch_bounds__(LG_THREADS, 1) void linear_nt_k96_gbwd_dt_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                                            const uint16_t* __restrict__ Y1, const uint16_t* __restrict__ W1p,
                                                                            uint16_t* __restrict__ DY, uint16_t* __restrict__ DTO,
                                                                            float* __restrict__ part, int M, unsigned x_bytes, unsigned y_bytes,
                                                                            const float* __restrict__ table) {
    constexpr int K = 96, KS = 6, N = 64 * LG_NP;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    char* const Lw = L;                                               // [N][LS_XP]
    const unsigned xbuf = (unsigned)N * LS_XP + (unsigned)wave * LS_XBUF;
    char* const ot = L + (unsigned)N * LS_XP + LG_WAVES * LS_XBUF + wave * LK_OBUF;
    const float* const T = (const float*)(L + (unsigned)N * LS_XP + LG_WAVES * (LS_XBUF + LK_OBUF));
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)X;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)x_bytes); rsrc[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Y1), 0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(DY, 0, (int)y_bytes, 0x00020000);
    XDma plan; xdma_plan(plan, lane, K * 2);
    const int ntiles_m = M >> 5, stride = gridDim.x * LG_WAVES;       // (M % 32 == 0)
    const int fr = lane >> 3, fc = lane & 7;                          // flush layout: row fr + 8 it, 16-byte chunk fc of the 64-column pair
    const unsigned flane = (unsigned)fr * (unsigned)(N * 2) + (unsigned)fc * 16u;
    int tm = blockIdx.x * LG_WAVES + wave;
    u32x4 yw[3][4];                                                   // y1 of the pairs pr, pr + 1, pr + 2 (slot pr % 3)
    if (tm < ntiles_m) xdma_issue(plan, (unsigned)tm * 32u * K * 2u, 32, rsrc, lds_base + xbuf);
    {
        const unsigned g00 = tm < ntiles_m ? (unsigned)tm * 32u * (unsigned)(N * 2) + flane : 0x80000000u;
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int it = 0; it < 4; ++it) yw[sl][it] = __builtin_amdgcn_raw_buffer_load_b128(ry, g00 + (unsigned)(it * 8 * N * 2), sl * 128, 0);
    }
    stage_weight(Lw, Wt, N, K, LS_XP, tid, LG_THREADS);
    for (int i = tid; i < GD_BYTES / 16; i += LG_THREADS) ((u32x4*)T)[i] = ((const u32x4*)table)[i];
    __syncthreads();                                                  // the only workgroup barrier
    // at the top of a row block its X DMA must have landed: it was issued a block ago, in front of > 100 loads and stores of which the last
    // pair's fragment loads have been waited for -- any bound below 63 is met by then
    int pending = 0;
    const unsigned wlane = (unsigned)l31 * LS_XP + (unsigned)lhi * 16u;
    float acc[LG_NP][8];
#pragma unroll
    for (int pr = 0; pr < LG_NP; ++pr)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[pr][k] = 0.f;
    auto lsync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    for (; tm < ntiles_m; tm += stride) {
        wait_vmcnt_dyn(pending);
        __builtin_amdgcn_wave_barrier();
        s16x8 xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xbuf + wlane + ks * 32));
        lsync();                                                      // every lane has its fragments: the buffer is free
        pending = 60;
        const int tn = tm + stride;
        if (tn < ntiles_m) xdma_issue(plan, (unsigned)tn * 32u * K * 2u, 32, rsrc, lds_base + xbuf);
        const unsigned g0 = (unsigned)tm * 32u * (unsigned)(N * 2) + flane;
        const unsigned g0n = tn < ntiles_m ? (unsigned)tn * 32u * (unsigned)(N * 2) + flane : 0x80000000u;
        f32x16 zacc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) zacc[j][i] = 0.f;
#pragma unroll
        for (int pr = 0; pr < LG_NP; ++pr) {
            const int sl = pr % 3;
            u32x4 w1f[3][4];                                          // W1^T[32 j + l31][64 pr + 16 u + 8 lhi .. +8], packed [pr][u][j][lane] by the caller
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) w1f[j][u] = ((const u32x4*)W1p)[(((pr * 4 + u) * 3 + j) << 6) + lane];
            unsigned py[2][8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int nt = 2 * pr + half;
                f32x16 a;
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = 0.f;
                const char* wt = Lw + (size_t)nt * 32 * LS_XP + wlane;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a = mfma32<bf16_t>(__builtin_bit_cast(s16x8, *(const u32x4*)(wt + ks * 32)), xf[ks], a);
#pragma unroll
                for (int q = 0; q < 4; ++q) { py[half][2 * q] = pack2<bf16_t>(a[4 * q + 0], a[4 * q + 1]); py[half][2 * q + 1] = pack2<bf16_t>(a[4 * q + 2], a[4 * q + 3]); }
            }
            put_tile(ot, py[0], l31, lhi, 0); put_tile(ot, py[1], l31, lhi, 1);
            lsync();
            // GELU' in two halves of two 8-element chunks; dy1 goes to HBM and back into the tile (each lane the chunks it read)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                uint4 gq[2], vq[2];
                float tq[2][8];
                bool ok = true;
                float a2[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) a2[k] = acc[pr][k];
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    const int it = 2 * h2 + i2;
                    const u32x4 g4 = *(const u32x4*)(ot + (it * 8 + fr) * LK_OP + fc * 16);
                    gq[i2] = uint4{g4[0], g4[1], g4[2], g4[3]};
                }
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    const int it = 2 * h2 + i2;
                    ok = gelu_grad_gather8(T, uint4{yw[sl][it][0], yw[sl][it][1], yw[sl][it][2], yw[sl][it][3]}, tq[i2]) && ok;
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
                    for (int i2 = 0; i2 < 2; ++i2) gelu_bwd8_apply(gq[i2], tq[i2], vq[i2], a2);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[pr][k] = a2[k];
                } else {                                              // (wave-uniform, rare) an element outside the table: the general evaluation
#pragma unroll
                    for (int i2 = 0; i2 < 2; ++i2) {
                        const int it = 2 * h2 + i2;
                        gelu_bwd8(T, gq[i2], uint4{yw[sl][it][0], yw[sl][it][1], yw[sl][it][2], yw[sl][it][3]}, vq[i2], acc[pr]);
                    }
                }
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    const int it = 2 * h2 + i2;
                    const u32x4 v = u32x4{vq[i2].x, vq[i2].y, vq[i2].z, vq[i2].w};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rd, g0 + (unsigned)(it * 8 * N * 2), pr * 128, 0);
                    *(u32x4*)(ot + (it * 8 + fr) * LK_OP + fc * 16) = v;
                    // the slot's next tenant: pair pr + 3 of this row block, or pair pr - 3 of the next one
                    yw[sl][it] = __builtin_amdgcn_raw_buffer_load_b128(ry, (pr < 3 ? g0 : g0n) + (unsigned)(it * 8 * N * 2), (pr < 3 ? pr + 3 : pr - 3) * 128, 0);
                }
            }
            lsync();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const s16x8 bf = __builtin_bit_cast(s16x8, *(const u32x4*)(ot + l31 * LK_OP + 32 * u + 16 * lhi));
#pragma unroll
                for (int j = 0; j < 3; ++j) zacc[j] = mfma32<bf16_t>(__builtin_bit_cast(s16x8, w1f[j][u]), bf, zacc[j]);
            }
            lsync();
        }
        const int row = tm * 32 + l31;
#pragma unroll
        for (int j = 0; j < 3; ++j) store_tile<false>(zacc[j], nullptr, DTO, nullptr, (size_t)row * K, j * 32, lhi, true);
    }
    // column sums: over the eight row lanes, then one partial row per wave
#pragma unroll
    for (int pr = 0; pr < LG_NP; ++pr)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = acc[pr][k];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 8) part[((size_t)blockIdx.x * LG_WAVES + wave) * N + pr * 64 + lane * 8 + k] = v;
        }
}

template <int NKC>                 // K = 96 NKC
__global__ __launch_bounds__(LS_THREADS, 1) void linear_nt_n96_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                                    const uint16_t* __restrict__ bias, uint16_t* __restrict__ Y,
                                                                    int M, unsigned x_bytes) {
    constexpr int N = 96, NT = 3, K = 96 * NKC, KS = 6, WP = K * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const L = (char*)lds;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    char* const Lw = L;                                               // [96][WP]
    const unsigned bias_b = (unsigned)N * WP;
    const unsigned xbuf = bias_b + 1024u + (unsigned)wave * LS_XBUF;
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)X;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)x_bytes); rsrc[3] = 0x00020000;
    }
    XDma plan; xdma_plan(plan, lane, K * 2);
    const int ntiles_m = (M + 31) >> 5, stride = gridDim.x * LS_WAVES;
    int tm = blockIdx.x * LS_WAVES + wave;
    if (tm < ntiles_m) xdma_issue(plan, (unsigned)tm * 32u * K * 2u, M - tm * 32, rsrc, lds_base + xbuf);
    stage_weight(Lw, Wt, N, K, WP, tid, LS_THREADS);
    for (int i = tid; i < N; i += LS_THREADS) ((uint16_t*)(L + bias_b))[i] = bias ? bias[i] : (uint16_t)0;
    __syncthreads();
    const uint16_t* const lbias = (const uint16_t*)(L + bias_b);
    int pending = 0;
    const unsigned xlane = (unsigned)l31 * LS_XP + (unsigned)lhi * 16u;
    const unsigned wlane = (unsigned)l31 * WP + (unsigned)lhi * 16u;
    for (; tm < ntiles_m; tm += stride) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            if (kc == 0) wait_vmcnt_dyn(pending); else wait_vmcnt<0>();   // block kc has landed (kc = 0: the previous rows' 2 NT stores came after its DMA)
            __builtin_amdgcn_wave_barrier();
            s16x8 xf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[ks] = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xbuf + xlane + ks * 32));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // the next 96-column block of these rows, or the first block of my next rows
            if (kc + 1 < NKC) xdma_issue(plan, (unsigned)tm * 32u * K * 2u + (unsigned)(kc + 1) * 192u, M - tm * 32, rsrc, lds_base + xbuf);
            else if (tm + stride < ntiles_m) xdma_issue(plan, (unsigned)(tm + stride) * 32u * K * 2u, M - (tm + stride) * 32, rsrc, lds_base + xbuf);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const s16x8 wf = __builtin_bit_cast(s16x8, *(const u32x4*)(Lw + (size_t)t * 32 * WP + wlane + kc * 192 + ks * 32));
                    acc[t] = mfma32<bf16_t>(wf, xf[ks], acc[t]);
                }
            }
        }
        const int row = tm * 32 + l31;
        const bool row_ok = row < M;
        const size_t row_off = (size_t)row * N;
#pragma unroll
        for (int t = 0; t < NT; ++t) store_tile<false>(acc[t], lbias, Y, nullptr, row_off, t * 32, lhi, row_ok);
        pending = 2 * NT;
    }
}

// slak_pack_w1t_fragments: see the entry point
__global__ __launch_bounds__(256) void pack_w1t_fragments_kernel(const u32x4* __restrict__ w1t, u32x4* __restrict__ w1p) {
    const int t = blockIdx.x * 256 + threadIdx.x;               // source chunk: row k (96), chunk c (48) of the row
    if (t >= 96 * 48) return;
    const int k = t / 48, c = t - k * 48;
    const int j = k >> 5, l31 = k & 31, pr = c >> 3, u = (c >> 1) & 3, lhi = c & 1;
    w1p[((((pr * 4 + u) * 3 + j) * 2 + lhi) * 32) + l31] = w1t[t];
}

}  // namespace slak

using namespace slak;

// the GELU table of gelu_lut in device memory (one copy per device, built on first use): entry [sign * GL_N + (mag - GL_LO)] = bf16(GELU(x))
static const uint16_t* gelu_table_device() {
    static std::mutex mu;
    static const uint16_t* tab[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (tab[dev]) return tab[dev];
    std::vector<uint16_t> h(2 * GL_N);
    for (unsigned sgn = 0; sgn < 2; ++sgn)
        for (unsigned i = 0; i < GL_N; ++i) {
            const uint32_t bits = ((sgn << 15) | (GL_LO + i)) << 16;
            float xf; memcpy(&xf, &bits, 4);
            const double x = xf, g = 0.5 * x * erfc(-x * 0.70710678118654752440);
            const float gf = (float)g;
            uint32_t u; memcpy(&u, &gf, 4);
            u += 0x7fffu + ((u >> 16) & 1u);                      // round to nearest even
            h[sgn * GL_N + i] = (uint16_t)(u >> 16);
        }
    void* d = nullptr;
    if (hipMalloc(&d, h.size() * 2) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    tab[dev] = (const uint16_t*)d;
    return tab[dev];
}

extern "C" {

int slak_linear_nt_supported(int M, int N, int K, int gelu) {
    if (M <= 0 || N <= 0 || K <= 0 || (long long)M * (N > K ? N : K) * 2 >= (1LL << 32)) return 0;
    if (K == 96 && N % 64 == 0 && N <= 384) return 1;                                     // whole weight + six waves' row blocks and out tiles in LDS
    if (!gelu && N == 96 && (K == 96 || K == 192 || K == 288 || K == 384)) return 1;
    return 0;
}

int slak_linear_nt(const void* x, const void* wt, const void* bias, void* y, void* gelu_out, int M, int N, int K, void* stream) {
    if (!x || !wt || !y) return SLAK_ERR_INVALID_ARG;
    if (!slak_linear_nt_supported(M, N, K, gelu_out != nullptr)) return SLAK_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const uint16_t *X = (const uint16_t*)x, *W = (const uint16_t*)wt, *B = (const uint16_t*)bias;
    uint16_t *Y = (uint16_t*)y, *G = (uint16_t*)gelu_out;
    const int tiles = (M + 31) / 32;
    int wgs = mfma_cu_count(); if (wgs * LS_WAVES > tiles) wgs = (tiles + LS_WAVES - 1) / LS_WAVES;
    const unsigned xb = (unsigned)((size_t)M * K * 2);
    auto set_lds = [](const void* k, size_t lds) { return slak_set_max_lds(k, lds); };
    if (K == 96 && N % 64 == 0) {
        const size_t lds = (size_t)N * LS_XP + 1024 + (size_t)LK_WAVES * (LS_XBUF + LK_OBUF) + (G ? GL_BYTES : 0);
        const uint16_t* lut = G ? gelu_table_device() : nullptr;
        if (G && !lut) return SLAK_ERR_LAUNCH;
        int wk = mfma_cu_count(); if (wk * LK_WAVES > tiles) wk = (tiles + LK_WAVES - 1) / LK_WAVES;
        if (G) {
            if (!set_lds((const void*)linear_nt_k96_kernel<true>, lds)) return SLAK_ERR_LAUNCH;
            hipLaunchKernelGGL((linear_nt_k96_kernel<true>), dim3(wk), dim3(LK_THREADS), lds, st, X, W, B, Y, G, M, N, xb, lut);
        } else {
            if (!set_lds((const void*)linear_nt_k96_kernel<false>, lds)) return SLAK_ERR_LAUNCH;
            hipLaunchKernelGGL((linear_nt_k96_kernel<false>), dim3(wk), dim3(LK_THREADS), lds, st, X, W, B, Y, G, M, N, xb, lut);
        }
    } else {
        const size_t lds = (size_t)96 * (K * 2 + 16) + 1024 + (size_t)LS_WAVES * LS_XBUF;
#define SLAK_LS_N96(NKC) { if (!set_lds((const void*)linear_nt_n96_kernel<NKC>, lds)) return SLAK_ERR_LAUNCH; \
                           hipLaunchKernelGGL((linear_nt_n96_kernel<NKC>), dim3(wgs), dim3(LS_THREADS), lds, st, X, W, B, Y, M, xb); }
        if (K == 96) SLAK_LS_N96(1) else if (K == 192) SLAK_LS_N96(2) else if (K == 288) SLAK_LS_N96(3) else SLAK_LS_N96(4)
#undef SLAK_LS_N96
    }
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

/* pwconv1 -> GELU -> pwconv2 in ONE pass (see linear_mlp_fwd_k96_kernel): x [M][96], w1 [384][96], b1 [384], w2 [96][384], b2 [96] -> y1, a [M][384],
 * z [M][96], all bf16.  y1 and a: the bits of slak_linear_nt(.., gelu_out); z: those of slak_linear_nt(a, w2, b2) up to fp32 summation order. */
int slak_linear_mlp_fwd_supported(int M, int C, int C4) {
    static const bool on = [] { const char* e = getenv("SLAK_LINEAR_MLP_FWD"); return !(e && e[0] == '0'); }();
    return (on && M > 0 && C == 96 && C4 == 64 * LM_NP && (long long)M * C4 * 2 < (1LL << 32)) ? 1 : 0;
}
int slak_linear_mlp_fwd(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* y1, void* a, void* z, int M, int C, int C4,
                        void* stream) {
    if (!x || !w1 || !w2 || !y1 || !a || !z) return SLAK_ERR_INVALID_ARG;
    if (!slak_linear_mlp_fwd_supported(M, C, C4)) return SLAK_ERR_UNSUPPORTED;
    const int tiles = (M + 31) / 32;
    int wk = mfma_cu_count(); if (wk * LM_WAVES > tiles) wk = (tiles + LM_WAVES - 1) / LM_WAVES;
    const uint16_t* lut = gelu_table_device();
    if (!lut) return SLAK_ERR_LAUNCH;
    const size_t lds = (size_t)C4 * LS_XP + 1024 + (size_t)LM_WAVES * (LS_XBUF + 2 * LK_OBUF) + GL_BYTES;
    if (!slak_set_max_lds((const void*)linear_mlp_fwd_k96_kernel, lds)) return SLAK_ERR_LAUNCH;
    hipLaunchKernelGGL(linear_mlp_fwd_k96_kernel, dim3(wk), dim3(LM_THREADS), lds, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)w1,
                       (const uint16_t*)b1, (const uint16_t*)w2, (const uint16_t*)b2, (uint16_t*)y1, (uint16_t*)a, (uint16_t*)z, M,
                       (unsigned)((size_t)M * C * 2), lut);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

/* dy1 = round(dz . W2) * gelu'(y1) and dbias = column sums of dy1 in ONE pass (see linear_nt_k96_gbwd_kernel): x = dz [M][K], wt = W2^T [N][K],
 * y1, dy1 [M][N] bf16, dbias [N] fp32.  Round 4: K = 96, N = 384 (stage 1 of SLaK-T/S).  The same dy1 bits as slak_linear_nt followed by
 * slak_gelu_backward_bias; dbias is the same sum of the rounded dy1 in another (fixed) order. */
int slak_linear_nt_gelu_bwd_supported(int M, int N, int K) {
    static const bool on = [] { const char* e = getenv("SLAK_LINEAR_GELU_BWD"); return !(e && e[0] == '0'); }();
    return (on && M >= 32 && M % 32 == 0 && K == 96 && N == 64 * LG_NP && (long long)M * N * 2 < (1LL << 31)) ? 1 : 0;
}
size_t slak_linear_nt_gelu_bwd_workspace_bytes(int M, int N, int K) {
    if (!slak_linear_nt_gelu_bwd_supported(M, N, K)) return 0;
    return align_up((size_t)1024 * LG_WAVES * N * sizeof(float), 256);            // one partial row per wave, at most 1024 workgroups
}
int slak_linear_nt_gelu_bwd(const void* x, const void* wt, const void* y1, void* dy1, float* dbias, int M, int N, int K,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !wt || !y1 || !dy1 || !dbias) return SLAK_ERR_INVALID_ARG;
    if (!slak_linear_nt_gelu_bwd_supported(M, N, K)) return SLAK_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < slak_linear_nt_gelu_bwd_workspace_bytes(M, N, K)) return SLAK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = M / 32;
    int wk = mfma_cu_count(); if (wk > 1024) wk = 1024; if (wk * LG_WAVES > tiles) wk = (tiles + LG_WAVES - 1) / LG_WAVES;
    const float* table = gelu_grad_table_device();
    if (!table) return SLAK_ERR_LAUNCH;
    const size_t lds = (size_t)N * LS_XP + (size_t)LG_WAVES * (LS_XBUF + LK_OBUF) + GD_BYTES;
    if (!slak_set_max_lds((const void*)linear_nt_k96_gbwd_kernel, lds)) return SLAK_ERR_LAUNCH;
    hipLaunchKernelGGL(linear_nt_k96_gbwd_kernel, dim3(wk), dim3(LG_THREADS), lds, st, (const uint16_t*)x, (const uint16_t*)wt, (const uint16_t*)y1,
                       (uint16_t*)dy1, (float*)workspace, M, (unsigned)((size_t)M * K * 2), (unsigned)((size_t)M * N * 2), table);
    SLAK_LAUNCH_CHECK();
    return tail_reduce_columns((const float*)workspace, dbias, wk * LG_WAVES, N, st);
}

/* W1^T [K = 96][N = 384] bf16 (row-major) -> the fragment-major copy slak_linear_nt_gelu_bwd_dt reads.  Element (k, n) with k = 32 j + l31,
 * n = 64 pr + 16 u + 8 lhi + e goes to [pr][u][j][lhi][l31][e] (1 KB per load, every byte used): one 16-byte chunk per thread. */
int slak_pack_w1t_fragments(const void* w1t, void* w1p, int N, int K, void* stream) {
    if (!w1t || !w1p) return SLAK_ERR_INVALID_ARG;
    if (N != 384 || K != 96) return SLAK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_w1t_fragments_kernel, dim3((96 * 48 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const u32x4*)w1t, (u32x4*)w1p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

/* slak_linear_nt_gelu_bwd with the next product of the block's backward in the same launch: dt [M][K] = dy1 . W1 (pwconv1's data gradient), see
 * linear_nt_k96_gbwd_dt_kernel.  w1p = W1^T [K][N] bf16 in FRAGMENT-MAJOR order: viewed (3, 32, 6, 4, 2, 8) = [row tile j][row l31][pair pr][k-step u]
 * [lane half lhi][8 k] and permuted to [pr][u][j][lhi][l31][8] (slak_pack_w1t_fragments above makes it).  dy1 and dbias: the bits of
 * slak_linear_nt_gelu_bwd; dt: bf16 of the same fp32 sums added in another order than slak_linear_nt's. */
int slak_linear_nt_gelu_bwd_dt_supported(int M, int N, int K) {
    static const bool on = [] { const char* e = getenv("SLAK_LINEAR_GELU_BWD_DT"); return !(e && e[0] == '0'); }();
    return (on && slak_linear_nt_gelu_bwd_supported(M, N, K)) ? 1 : 0;
}
int slak_linear_nt_gelu_bwd_dt(const void* x, const void* wt, const void* y1, const void* w1p, void* dy1, void* dt, float* dbias, int M, int N, int K,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !wt || !y1 || !w1p || !dy1 || !dt || !dbias) return SLAK_ERR_INVALID_ARG;
    if (!slak_linear_nt_gelu_bwd_dt_supported(M, N, K)) return SLAK_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < slak_linear_nt_gelu_bwd_workspace_bytes(M, N, K)) return SLAK_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = M / 32;
    int wk = mfma_cu_count(); if (wk > 1024) wk = 1024; if (wk * LG_WAVES > tiles) wk = (tiles + LG_WAVES - 1) / LG_WAVES;
    const float* table = gelu_grad_table_device();
    if (!table) return SLAK_ERR_LAUNCH;
    const size_t lds = (size_t)N * LS_XP + (size_t)LG_WAVES * (LS_XBUF + LK_OBUF) + GD_BYTES;
    if (!slak_set_max_lds((const void*)linear_nt_k96_gbwd_dt_kernel, lds)) return SLAK_ERR_LAUNCH;
    hipLaunchKernelGGL(linear_nt_k96_gbwd_dt_kernel, dim3(wk), dim3(LG_THREADS), lds, st, (const uint16_t*)x, (const uint16_t*)wt, (const uint16_t*)y1,
                       (const uint16_t*)w1p, (uint16_t*)dy1, (uint16_t*)dt, (float*)workspace, M, (unsigned)((size_t)M * K * 2),
                       (unsigned)((size_t)M * N * 2), table);
    SLAK_LAUNCH_CHECK();
    return tail_reduce_columns((const float*)workspace, dbias, wk * LG_WAVES, N, st);
}

}  // extern "C"
