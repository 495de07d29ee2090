// slak_amd/csrc/dwconv_mfma_tri_wgrad_rows.hip -- the THREE weight gradients of a decomposed block (K x 5, 5 x K, 5 x 5: models/SLaK.py:82-100;
// the reference runs backward_filter_fp16.cu:181-243 once per branch) in ONE launch on planes of 2 x 2 MFMA tiles (32 < H, W <= 64, W % 8 == 0:
// the 56 x 56 stage, 48 x 48, 64 x 64).  x is fetched from HBM once for the three correlations (4 plane reads per block instead of 6).
//
//   vertical   (K x 5)  G^v_r[o, i] = sum_{n,u} dYv[o, u] X[i, u + r - 2]   o, i = image ROWS,    contraction along a row     dw_v[tau, r] = sum_o G^v_r[o, o+tau-padL]
//   small      (5 x 5)  G^s_r[o, i] = sum_{n,u} dYs[o, u] X[i, u + r - 2]   (the vertical correlation with its own dY: same B operands) dw_s[tau, r], |tau-2| <= 2
//   horizontal (5 x K)  G^h_r[o, i] = sum_{n,y} dYh[y, o] X[y + r - 2, i]   o, i = image COLUMNS, contraction along a column  dw_h[r, tau] = sum_o G^h_r[o, o+tau-padL]
//
// One LDS image per tensor and plane, rows of CPR 16-byte chunks (CPR odd, >= W/8 + 1: the pad chunks are never written and read as the
// zero padding), landed by LDS-DMA.  The vertical / small operands are 16-byte row reads, the five column shifts of X formed in registers
// from a six-dword window (dwconv_mfma_wgrad_vrows.hip); the horizontal operands are transposing reads (ds_read_b64_tr_b16) of the SAME
// images, and its five ROW shifts are the same six-dword window arithmetic on a twelve-row column segment (three transposing reads)
// -- so every k-step of either kind costs five LDS reads, eight v_alignbit and a few moves beside its ten / five MFMAs.
//
// A workgroup is four waves, ONE PER SIMD (15 accumulators of 16 registers: the kernel is compiled for one wave per SIMD and 512
// registers); wave (mt, nt) owns tile (mt, nt) of all three correlations over the workgroup's batch slice.  With nobody to share a SIMD with,
// the MFMA pipe only stays busy if the wave's own stream does: every k-step is software-pipelined by hand -- the LDS reads and the
// shift arithmetic of step j + 1 and one LDS-DMA piece of plane i + 2 are placed behind the MFMAs of step j and pinned with
// sched_barrier (hipcc otherwise gathers them in front of the MFMAs: measured 73 cycles per MFMA for the compiler-ordered loop of the
// two-branch kernel when it runs alone on a SIMD, against the pipe's 32).  Three ring slots; one workgroup barrier per plane.
// Epilogue (skewed per-wave tile for the diagonal sums, write-through partials, last-arriver reduction in slice order): as in the
// other MFMA weight-gradient kernels -- deterministic, no atomics on data.
#include "tri_wgrad_common.h"
#include <stdlib.h>

namespace slak {

constexpr int TW_NB = 3;                // ring slots: plane i is consumed while i + 1 has landed / is landing and i + 2 is requested
constexpr int TW_J = 3;                 // DMA instructions per wave and plane copy (upper bound: ceil(ipc / 4), ipc <= 12)
constexpr int TW_MAXWG = 320;           // workgroups at most (one per CU: 256 on MI355X; the range table travels in the kernel argument)
constexpr unsigned TW_OOB = 0x80000000u;   // source offset of a lane with nothing to fetch: out of range -> the DMA writes zeros (into padding)

struct TriRowsParams {
    const void* dy[3];                  // dY of the K x 5, 5 x K, 5 x 5 branch
    const void* x; float* partial; float* dw[3]; unsigned* counters;
    int N, C, H, W, K, padL;
    int CPR;               // 16-byte chunks per LDS row (odd, >= 2 * KS + 1)
    int ipc;               // DMA instructions per plane copy: ceil(H * CPR / 64)
    int RS;                // rows per plane copy in LDS (>= 16 * KS + 4: the rows behind the image stay zero)
    int grid;              // workgroups; workgroup b owns planes [qs[b], qs[b + 1]) of the C * N planes in (channel, image) order
    int maxspan;           // channels a range touches at most (partial records per workgroup)
    int qs[TW_MAXWG + 1];  // the ranges, balanced by COST on the host: a range that crosses a channel boundary pays for summing up in mid-stream with fewer planes
    unsigned tensor_bytes;
    int dbg;               // SLAK_TRIROWS_DBG (timing experiments): 1 = no k-loops, 2 = no DMA, 4 = no epilogue, 8 = no wait / barrier, 32 = clock probe (with 4), 64 = every plane from the first MiB  -- only read in dev builds (SLAK_BUILD_DEFS=-DSLAK_TRIROWS_DEV); the shipped kernel compiles the experiments out
};
#ifdef SLAK_TRIROWS_DEV
#define TW_DBG(bit) (p.dbg & (bit))
#else
#define TW_DBG(bit) 0
#endif

// buffer_load_dwordx4 ... lds without saving M0 around it (lds_dma16 of mfma_common.h does)
__device__ __forceinline__ void tw_dma16(unsigned voff, v4i_t rsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

// KS: 16-deep k-steps of both contractions (ceil(max(H, W) / 16): 3 or 4); CPRC: chunks per LDS row as a compile-time constant (every
// fragment offset of a plane is then an instruction immediate)
// SD (round 6, "small branch on the diagonal"): the 5 x 5 correlation only has the five diagonals |i - o| <= 2, so of its 2 x 2 tiles only
// the two DIAGONAL ones are multiplied -- 40 MFMAs per plane instead of 80, 200 instead of 240 for the three branches (the launch runs at the
// board's power cap: energy per plane is the lever).  Three things make that exact and keep the four waves' instruction streams
// identical (no wave-dependent branch in the pinned stream):
//   * the second tile of X's rows starts at row H - 32 instead of 32 (rows H-32 .. H-1: no padding rows; needs H <= 62): the diagonal tile
//     pair {rows [0,32)} x {rows [0,32)}, {rows [H-32,H)} x {rows [H-32,H)} then holds EVERY pair with |i - o| <= 2 (a pair that straddles row
//     32 has both members >= 30 >= H - 32); the block both tiles hold (o, i in [H-32, 32)) is dropped from the second one in the diagonal sums,
//     and so are the columns i < 32 of the vertical branch's tiles over the second X tile;
//   * both waves that hold X tile nt (mt = 0, 1) work on the SAME small tile (nt, nt): at loop step j a wave multiplies taps {0,1,2} (j even) or
//     {3,4} (j odd) -- and wave mt = 1 walks the k-steps ROTATED by one (1, 2, 3, 0: a constant in its operand addresses), so that k-step k gets
//     taps T(k) from wave 0 and T(k - 1) = the other taps from wave 1.  Needs an even KS (KS = 4: the 56 x 56 class);
//   * the four waves' partial sums are added in the epilogue as before.  50 MFMAs per wave and plane instead of 60.
template <typename T, int KS, int CPRC, bool SD>
__global__ __launch_bounds__(MF_THREADS, 1) void dwconv_mfma_tri_wgrad_rows_kernel(const TriRowsParams p) {
    static_assert(!SD || (KS % 2 == 0), "the tap split of the small branch alternates with the k-step's parity");
    constexpr int NG = MF_TAPS;
    constexpr unsigned PB = (unsigned)CPRC * 16;                  // row pitch (bytes)
    constexpr unsigned RSC = KS == 3 ? 56 : 68;                   // rows per copy: >= 16 KS + 4, and a copy's DMA instructions (whole KiB) end inside it
    constexpr unsigned copy_b = RSC * PB;                         // [dYv][X][dYs][dYh]: X's rows -2, -1 are the (zero) tail of dYv's copy
    constexpr unsigned slot_b = 4 * copy_b;
    constexpr unsigned ring_b = 64;                               // zero bytes in front ("chunk -1" of the first row)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const LB = (char*)lds;
    const int HW = p.H * p.W, ntl = p.K * MF_TAPS, ntot = 2 * ntl + MF_TAPS * MF_TAPS;
    float* scratch = (float*)(LB + ring_b + TW_NB * slot_b);      // [MF_WAVES][32][64] skewed tiles of the diagonal sums (NOT in the ring: a workgroup whose
                                                                  // range crosses a channel boundary sums up in mid-stream, with planes in flight)
    float* dwl = scratch + MF_WAVES * 32 * 64;                    // [MF_WAVES][ntot]
    int* flags = (int*)(dwl + MF_WAVES * ntot);                   // [8]
    float* segres = (float*)(flags + 8);                          // [ntot]: the first channel's sums of a workgroup whose range crosses ONE boundary

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = wave & 1, nt = wave >> 1;
    // Work decomposition: the C * N planes in (channel, image) order are cut into `grid` equal ranges, one per workgroup = one per CU (a
    // whole number of slices per channel would leave a quarter of the CUs idle at C = 96: 192 workgroups of 64 planes instead of 256 of 48).
    // A range may cross channel boundaries: the accumulators are summed up, stored as the workgroup's partial record number `kseg` and cleared
    // at each boundary; channel c is reduced by the last of the workgroups b_lo(c) .. b_hi(c) to arrive, in workgroup order.
    const int bwg = blockIdx.x;
    const int q0 = p.qs[bwg], q1 = p.qs[bwg + 1];
    const int iters = q1 > q0 ? q1 - q0 : 0;
    const int c_first = q0 / p.N;

    const unsigned long long clk0 = TW_DBG(32) ? __builtin_amdgcn_s_memtime() : 0ull, rt0 = TW_DBG(32) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    for (unsigned o = tid * 16; o < ring_b + TW_NB * slot_b + (unsigned)(MF_WAVES * 32 * 64 + MF_WAVES * ntot) * 4 + 32; o += MF_THREADS * 16) *(u32x4*)(LB + o) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    // ---- DMA plan: instruction ii of a copy fetches chunks g = 64 ii + lane -> (row g / CPR, piece g % CPR); wave w issues ii = w + 4 j ---------
    v4i_t rs[4];
    {
        const void* src[4] = {p.dy[0], p.x, p.dy[2], p.dy[1]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint64_t a = (uint64_t)src[t];
            rs[t][0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs[t][1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
            rs[t][2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs[t][3] = 0x00020000;
        }
    }
    unsigned src_off[TW_J]; bool j_live[TW_J];
    int my_instr = 0;
#pragma unroll
    for (int j = 0; j < TW_J; ++j) {
        const int ii = wave + MF_WAVES * j;
        j_live[j] = ii < p.ipc;                                   // wave-uniform
        const int g = ii * 64 + lane, row = g / CPRC, piece = g - row * CPRC;
        src_off[j] = (j_live[j] && row < p.H && piece < p.W / 8) ? (unsigned)(row * p.W * 2 + piece * 16) : TW_OOB;
        my_instr += j_live[j] ? 4 : 0;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds) + ring_b;
    const unsigned plane_b = (unsigned)HW * 2, img_b = (unsigned)p.C * plane_b;
    // gb: byte offset of a plane (n, c) in the NCHW tensors (the same in all four); the issue pointer runs two planes ahead of the compute pointer.
    // A piece in the MFMA stream is five instructions (M0, the wait state, the load, one scalar test): the lane offsets of the plane being
    // requested (voff) and whether it exists (iss_on) are formed once per plane, M0 is not saved (the compiler's own code never reads it here:
    // checked in the ISA).
    unsigned voff[TW_J]; bool iss_on = false;
    auto aim = [&](int g, unsigned gb) {
        iss_on = g < iters && !TW_DBG(2);
#pragma unroll
        for (int j = 0; j < TW_J; ++j) voff[j] = src_off[j] == TW_OOB ? TW_OOB : (TW_DBG(64) ? (gb & 0xfffffu) : gb) + src_off[j];      // (64: timing experiment, every plane from the first MiB: cache hits)
    };
    auto issue_piece = [&](int g, int k) __attribute__((always_inline)) {      // k = t * TW_J + j, compile-time at every call site
        const int t = k / TW_J, j = k % TW_J;
        if (iss_on && j_live[j])
            tw_dma16(voff[j], rs[t], __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(g % TW_NB) * slot_b + (unsigned)t * copy_b + (unsigned)(wave + MF_WAVES * j) * 1024));
    };
    int c_iss = c_first, n_iss = q0 - c_first * p.N;               // plane the issue pointer is at
    unsigned gb_iss = (unsigned)n_iss * img_b + (unsigned)c_iss * plane_b;
    auto advance_issue = [&]() {
        ++n_iss; gb_iss += img_b;
        if (n_iss == p.N) { n_iss = 0; ++c_iss; gb_iss = (unsigned)c_iss * plane_b; }
    };
    aim(0, gb_iss);
#pragma unroll
    for (int k = 0; k < 4 * TW_J; ++k) issue_piece(0, k);
    advance_issue();
    aim(1, gb_iss);
#pragma unroll
    for (int k = 0; k < 4 * TW_J; ++k) issue_piece(1, k);
    advance_issue();                                              // -> plane 2

    tw_acc_claim();
    tw_acc_zero<0, 240>();

    // ---- fragment addresses (relative to a slot) ---------------------------------------------------------------------------
    // vertical / small: lane -> image row (o = mt * 32 + l31 resp. i = nt * 32 + l31), 8 consecutive k = columns 16 ks + 8 lhi .. +7
    const int sI = SD ? (nt ? p.H - 32 : 0) : nt * 32;                // first image row of this wave's X tile
    const unsigned rotb = SD ? (unsigned)mt * 32u : 0u;               // SD: wave (1, nt) starts one k-step further along ...
    const unsigned wrapb = SD ? (unsigned)mt * (unsigned)(KS * 32) : 0u;   // ... and its last loop step wraps around to k-step 0
    const unsigned av_off = ring_b + (unsigned)(mt * 32 + l31) * PB + lhi * 16 + rotb;              // dYv (copy 0)
    const unsigned xv_off = ring_b + copy_b + (unsigned)(sI + l31) * PB + lhi * 16 + rotb;         // X   (copy 1)
    const unsigned as_off = ring_b + 2 * copy_b + (unsigned)((SD ? sI : mt * 32) + l31) * PB + lhi * 16 + rotb;   // dYs (copy 2): SD -> the DIAGONAL tile (nt, nt)
    // horizontal: lane -> image column (o = mt * 32 + l31 resp. i = nt * 32 + l31), 8 consecutive k = rows 16 ks + 8 lhi .. +7: a 16-lane
    // group of a transposing read covers 4 rows x 16 columns (lane i16 supplies row i16 / 4, columns 4 (i16 % 4) .. +3, receives column i16)
    const int i16 = lane & 15, gq = lane >> 4;
    const unsigned trl = (unsigned)(8 * lhi + (i16 >> 2)) * PB + (unsigned)(16 * (gq & 1) + 4 * (i16 & 3)) * 2;
    const unsigned ah_off = ring_b + 3 * copy_b + trl + (unsigned)mt * 64;                          // dYh (copy 3), columns of tile mt
    const unsigned xh_off = ring_b + copy_b + trl + (unsigned)nt * 64 - 2 * PB;                     // X, columns of tile nt, the window starts two rows up
    auto rdq = [&](unsigned addr) -> u32x4 { return *(const u32x4*)(LB + addr); };
    auto rdd = [&](unsigned addr) -> unsigned { return *(const unsigned*)(LB + addr); };
    auto rdt = [&](unsigned addr) -> u32x2 { return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, LB + addr))); };

    float* mine = dwl + wave * ntot;
    float* tile = scratch + wave * (32 * 64);
    float* wr = tile + (4 * lhi) * 64 + (l31 - 4 * lhi + 31);
    const int dtau = (nt - mt) * 32 - 31;
    int it = 0, kseg = 0;
    int n_cur = q0 - c_first * p.N;                               // image index of plane `it` inside its channel
    while (it < iters) {
    int seg_end = it + (p.N - n_cur); if (seg_end > iters) seg_end = iters;      // planes [it, seg_end) belong to channel c_first + kseg
    for (; it < seg_end; ++it) {
        if (!TW_DBG(8)) {
        wait_vmcnt_dyn((it + 1 < iters ? 1 : 0) * my_instr);      // my pieces of plane `it` have landed (loads retire in order); plane it + 1's may be in flight
        wg_barrier();                                             // everyone's have; everyone is done with plane it - 1, whose slot plane it + 2 takes
        }
        const unsigned sb = (unsigned)(it % TW_NB) * slot_b;
        aim(it + 2, gb_iss);
        if (TW_DBG(1)) {
#pragma unroll
            for (int k = 0; k < 4 * TW_J; ++k) issue_piece(it + 2, k);
            advance_issue();
            continue;
        }
        const unsigned av = sb + av_off, as = sb + as_off, xv = sb + xv_off, ah = sb + ah_off, xh = sb + xh_off;
        const unsigned avw = av - wrapb, asw = as - wrapb, xvw = xv - wrapb;      // (SD) the bases of the LAST loop step's operands
        s16x8 a[2], a2[2], b[2][NG];
        // operands of the first vertical k-step (nothing to hide them behind: the plane has only just been released)
        {
            a[0] = __builtin_bit_cast(s16x8, rdq(av)); a2[0] = __builtin_bit_cast(s16x8, rdq(as));
            const u32x4 C = rdq(xv); const unsigned P3 = rdd(xv - 4), N0 = rdd(xv + 16);
            tw_taps(b[0], P3, C[0], C[1], C[2], C[3], N0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- vertical + small: KS k-steps of ten MFMAs (SD: eight / seven); behind them the operands of the next step (the last one: of the first horizontal step)
        if constexpr (SD) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cu = ks & 1, nx = cu ^ 1;
            const bool last = ks + 1 == KS, wrap = ks + 2 == KS;                      // wrap: the NEXT step is the loop's last one
            const unsigned on = (unsigned)(ks + 1) * 32u;
            const unsigned avn = (wrap ? avw : av) + on, asn = (wrap ? asw : as) + on, xvn = (wrap ? xvw : xv) + on;
            u32x4 Cn; unsigned Pn = 0, Nn = 0; u32x2 w0, w1, w2, h0, h1;
            if ((ks & 1) == 0) {                                                      // taps 0, 1, 2 of the small branch
                tw_mfma<T, TW_ACC_V + 16 * 0>(a[cu], b[cu][0]);
                if (!last) a[nx] = __builtin_bit_cast(s16x8, rdq(avn)); else h0 = rdt(ah);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_S + 16 * 0>(a2[cu], b[cu][0]);
                if (!last) a2[nx] = __builtin_bit_cast(s16x8, rdq(asn)); else h1 = rdt(ah + 4 * PB);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_V + 16 * 1>(a[cu], b[cu][1]);
                if (!last) Cn = rdq(xvn); else w0 = rdt(xh);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_S + 16 * 1>(a2[cu], b[cu][1]);
                if (!last) Pn = rdd(xvn - 4); else w1 = rdt(xh + 4 * PB);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_V + 16 * 2>(a[cu], b[cu][2]);
                if (!last) Nn = rdd(xvn + 16); else w2 = rdt(xh + 8 * PB);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_S + 16 * 2>(a2[cu], b[cu][2]);
                issue_piece(it + 2, 2 * ks);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_V + 16 * 3>(a[cu], b[cu][3]);
                issue_piece(it + 2, 2 * ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_V + 16 * 4>(a[cu], b[cu][4]);
            } else {                                                                  // taps 3, 4
                tw_mfma<T, TW_ACC_V + 16 * 0>(a[cu], b[cu][0]);
                if (!last) a[nx] = __builtin_bit_cast(s16x8, rdq(avn)); else h0 = rdt(ah);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_V + 16 * 1>(a[cu], b[cu][1]);
                if (!last) a2[nx] = __builtin_bit_cast(s16x8, rdq(asn)); else h1 = rdt(ah + 4 * PB);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_V + 16 * 2>(a[cu], b[cu][2]);
                if (!last) Cn = rdq(xvn); else w0 = rdt(xh);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_V + 16 * 3>(a[cu], b[cu][3]);
                if (!last) { Pn = rdd(xvn - 4); Nn = rdd(xvn + 16); } else { w1 = rdt(xh + 4 * PB); w2 = rdt(xh + 8 * PB); }
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_S + 16 * 3>(a2[cu], b[cu][3]);
                issue_piece(it + 2, 2 * ks);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_V + 16 * 4>(a[cu], b[cu][4]);
                issue_piece(it + 2, 2 * ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                tw_mfma<T, TW_ACC_S + 16 * 4>(a2[cu], b[cu][4]);
            }
            if (!last) tw_taps(b[nx], Pn, Cn[0], Cn[1], Cn[2], Cn[3], Nn);
            else { a[nx] = __builtin_bit_cast(s16x8, u32x4{h0[0], h0[1], h1[0], h1[1]}); tw_taps(b[nx], w0[0], w0[1], w1[0], w1[1], w2[0], w2[1]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cu = ks & 1, nx = cu ^ 1;
            const bool last = ks + 1 == KS;
            u32x4 Cn; unsigned Pn = 0, Nn = 0; u32x2 w0, w1, w2, h0, h1;
            tw_mfma<T, TW_ACC_V + 16 * 0>(a[cu], b[cu][0]);
            if (!last) a[nx] = __builtin_bit_cast(s16x8, rdq(av + (ks + 1) * 32)); else h0 = rdt(ah);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_S + 16 * 0>(a2[cu], b[cu][0]);
            if (!last) a2[nx] = __builtin_bit_cast(s16x8, rdq(as + (ks + 1) * 32)); else h1 = rdt(ah + 4 * PB);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_V + 16 * 1>(a[cu], b[cu][1]);
            if (!last) Cn = rdq(xv + (ks + 1) * 32); else w0 = rdt(xh);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_S + 16 * 1>(a2[cu], b[cu][1]);
            if (!last) Pn = rdd(xv + (ks + 1) * 32 - 4); else w1 = rdt(xh + 4 * PB);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_V + 16 * 2>(a[cu], b[cu][2]);
            if (!last) Nn = rdd(xv + (ks + 1) * 32 + 16); else w2 = rdt(xh + 8 * PB);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_S + 16 * 2>(a2[cu], b[cu][2]);
            issue_piece(it + 2, 2 * ks);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_V + 16 * 3>(a[cu], b[cu][3]);
            issue_piece(it + 2, 2 * ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_S + 16 * 3>(a2[cu], b[cu][3]);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_V + 16 * 4>(a[cu], b[cu][4]);
            if (!last) tw_taps(b[nx], Pn, Cn[0], Cn[1], Cn[2], Cn[3], Nn);
            else { a[nx] = __builtin_bit_cast(s16x8, u32x4{h0[0], h0[1], h1[0], h1[1]}); tw_taps(b[nx], w0[0], w0[1], w1[0], w1[1], w2[0], w2[1]); }
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_S + 16 * 4>(a2[cu], b[cu][4]);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        // ---- horizontal: KS k-steps of five MFMAs (a / b buffers continue to alternate: step ks uses buffer (KS + ks) & 1) ------------------------
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cu = (KS + ks) & 1, nx = cu ^ 1;
            const bool last = ks + 1 == KS;
            const unsigned ro = (unsigned)(ks + 1) * 16 * PB;
            u32x2 w0, w1, w2, h0, h1;
            tw_mfma<T, TW_ACC_H + 16 * 0>(a[cu], b[cu][0]);
            if (!last) { h0 = rdt(ah + ro); h1 = rdt(ah + ro + 4 * PB); }
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_H + 16 * 1>(a[cu], b[cu][1]);
            if (!last) { w0 = rdt(xh + ro); w1 = rdt(xh + ro + 4 * PB); w2 = rdt(xh + ro + 8 * PB); }
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_H + 16 * 2>(a[cu], b[cu][2]);
            issue_piece(it + 2, 2 * KS + ks);
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_H + 16 * 3>(a[cu], b[cu][3]);
            if (!last) { a[nx] = __builtin_bit_cast(s16x8, u32x4{h0[0], h0[1], h1[0], h1[1]}); tw_taps(b[nx], w0[0], w0[1], w1[0], w1[1], w2[0], w2[1]); }
            __builtin_amdgcn_sched_barrier(0);
            tw_mfma<T, TW_ACC_H + 16 * 4>(a[cu], b[cu][4]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (KS = 3: the 4 * TW_J = 12 piece slots of a plane are 2 * KS + KS = 9 above; the rest go out here)
#pragma unroll
        for (int k = 3 * KS; k < 4 * TW_J; ++k) issue_piece(it + 2, k);
        advance_issue();
    }
    // ---- end of a channel segment: diagonal sums of the fifteen accumulators through the wave's skewed tile (G[o][i] -> row o, column i - o + 31), the
    //      workgroup's partial record `kseg`, accumulators cleared.  In mid-stream (planes of the next channel in flight) the stores below and the
    //      DMA pieces share the wave's vmcnt: everything is drained once here (one boundary per workgroup at most when per <= N).
    if (!TW_DBG(4)) {
        {   // (every tap writes all of the wave's 32 x 32 entries of the skewed tile: nothing to clear between taps or extents)
            int o_max = p.H - mt * 32; if (o_max > 32) o_max = 32;
            if constexpr (SD) {
                const int ov = 64 - p.H;                                    // rows [H - 32, 32): in both X tiles -> counted with the first
                const bool dup = nt == 1 && l31 < ov;
                tw_diag5<TW_ACC_V>(tile, wr, dup ? 0 : o_max - 4 * lhi, lane, sI - mt * 32 - 31 + p.padL, p.K, mine, MF_TAPS, 1);               // dw_v[tau][r = g]
                // the small branch's tile is (nt, nt) for both waves of an X tile (every row of it is an image row); of the second tile the block o, i < 32 is the first tile's
                tw_diag5<TW_ACC_S>(tile, wr, 64, lane, -31 + MF_TAPS / 2, MF_TAPS, mine + 2 * ntl, MF_TAPS, 1, dup ? ov - 4 * lhi : -64);
            } else {
            const int lim = nt * 32 + l31 < p.H ? o_max - 4 * lhi : 0;
            tw_diag5<TW_ACC_V>(tile, wr, lim, lane, dtau + p.padL, p.K, mine, MF_TAPS, 1);                          // dw_v[tau][r = g]
            tw_diag5<TW_ACC_S>(tile, wr, lim, lane, dtau + MF_TAPS / 2, MF_TAPS, mine + 2 * ntl, MF_TAPS, 1);       // dw_s[tau][r = g]
            }
        }
        {
            int o_max = p.W - mt * 32; if (o_max > 32) o_max = 32;
            const int lim = nt * 32 + l31 < p.W ? o_max - 4 * lhi : 0;
            tw_diag5<TW_ACC_H>(tile, wr, lim, lane, dtau + p.padL, p.K, mine + ntl, 1, p.K);                          // dw_h[r = g][tau]
        }
        __syncthreads();
        // In mid-stream nothing goes to memory if it can wait: stores would share the wave's vmcnt with the DMA pieces in flight (counted waits) and
        // draining it costs a plane's time.  With at most one boundary per workgroup (maxspan == 2) the first channel's sums stay in LDS and leave
        // with the last record; more boundaries than that (per > N: tiny batches) store and drain.
        const bool more = it < iters, defer = more && p.maxspan == 2;
        float* out = p.partial + ((size_t)bwg * p.maxspan + kseg) * ntot;
        for (int t = tid; t < ntot; t += MF_THREADS) {
            float s = dwl[t];
#pragma unroll
            for (int w = 1; w < MF_WAVES; ++w) s += dwl[w * ntot + t];
            if (defer) segres[t] = s;
            else {
                wgrad_store_partial(&out[t], s);
                if (p.maxspan == 2 && kseg == 1) wgrad_store_partial(&out[t - ntot], segres[t]);      // record 0: the deferred first channel
            }
#pragma unroll
            for (int w = 0; w < MF_WAVES; ++w) dwl[w * ntot + t] = 0.f;      // (taps no diagonal reaches are never written: they must read 0 next time too)
        }
        if (more && !defer) { wait_vmcnt<0>(); __syncthreads(); }
    }
    ++kseg; n_cur = 0;
    if (it < iters) tw_acc_zero<0, 240>();                        // more planes (of the next channel) follow
    }   // while (it < iters)
    wait_vmcnt<0>();
    __syncthreads();
    if (TW_DBG(32) && tid == 0 && bwg < 8) {                    // (dev, tools/clk_tri_rows.py: shader cycles and 100 MHz ticks this workgroup ran -> the effective clock)
        p.dw[2][bwg * 2] = (float)(__builtin_amdgcn_s_memtime() - clk0); p.dw[2][bwg * 2 + 1] = (float)(__builtin_amdgcn_s_memrealtime() - rt0);
    }
    if (TW_DBG(4)) return;
    // ---- arrive at every channel this workgroup holds a record of; the last arriver of a channel adds the records of workgroups b_lo .. b_hi in
    //      workgroup order (bitwise reproducible) and scatters into the three dw tensors (hand-off form: see wgrad_finish in slak_common.h) --------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nseg = kseg;
    auto owner = [&](int q) -> int {                              // the workgroup whose range holds plane q: the last b with qs[b] <= q
        int lo = 0, hi = p.grid - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (p.qs[mid] <= q) lo = mid; else hi = mid - 1; }
        return lo;
    };
    if (tid < nseg) {
        const int ch = c_first + tid;
        const int b_lo = owner(ch * p.N), b_hi = owner(ch * p.N + p.N - 1);
        int last = 1;
        if (b_hi > b_lo) {
            const unsigned old = __hip_atomic_fetch_add(p.counters + ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == (unsigned)(b_hi - b_lo);
            if (last) __hip_atomic_store(p.counters + ch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        flags[tid] = last;
    }
    __syncthreads();
    for (int sg = 0; sg < nseg; ++sg) {
        if (!flags[sg]) continue;
        const int ch = c_first + sg;
        const int b_lo = owner(ch * p.N), b_hi = owner(ch * p.N + p.N - 1);
        for (int t = tid; t < ntot; t += MF_THREADS) {
            float s = 0.f;
            for (int b0 = b_lo; b0 <= b_hi; b0 += 8) {            // 8 loads in flight, added in workgroup order; agent-scope loads read at the coherence point
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int bb = b0 + j;
                    const int kk = bb <= b_hi ? ch - p.qs[bb] / p.N : 0;      // the record of workgroup bb that belongs to channel ch
                    v[j] = bb <= b_hi ? __hip_atomic_load(p.partial + ((size_t)bb * p.maxspan + kk) * ntot + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
            if (t < ntl) p.dw[0][(size_t)ch * ntl + t] = s;
            else if (t < 2 * ntl) p.dw[1][(size_t)ch * ntl + (t - ntl)] = s;
            else p.dw[2][(size_t)ch * (MF_TAPS * MF_TAPS) + (t - 2 * ntl)] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
static bool tri_rows_enabled() {               // SLAK_TRI_ROWS=0: the 2 x 2-tile planes keep the two-launch weight gradient (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_TRI_ROWS"); return !(e && e[0] == '0'); }();
    return v;
}

static bool fill_tri_rows_params(TriRowsParams& p, int N, int C, int H, int W, int K, int wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.padL = K / 2;
    if (N <= 0 || C <= 0 || K <= MF_TAPS || !(K & 1) || K > 63) return false;
    if (H <= 32 || H > 64 || W <= 32 || W > 64 || (W % 8)) return false;
    const int ks = (H > W ? H : W) > 48 ? 4 : 3;
    p.CPR = 2 * ks + 1;                                               // 9 (maps up to 64) or 7 (up to 48): odd, >= W / 8 + 1, and chunk 2 ks (the "next" of the last k-step) is padding
    if (p.CPR < W / 8 + 1) return false;
    p.ipc = (H * p.CPR + 63) / 64;
    p.RS = ks == 3 ? 56 : 68;
    if ((p.ipc + MF_WAVES - 1) / MF_WAVES > TW_J) return false;
    if ((size_t)p.ipc * 1024 > (size_t)p.RS * p.CPR * 16) return false;      // a copy's DMA instructions stay inside the copy
    if ((size_t)N * C >= 0x40000000ull) return false;
    const int P = N * C;
    if (wgs < 1) wgs = 1;
    if (wgs > TW_MAXWG) wgs = TW_MAXWG;
    // Ranges of equal COST: a plane costs 4 units, summing up at a channel boundary inside a range 29 (measured: 1.5 us per plane, ~11 us per sum),
    // so a workgroup that crosses a boundary gets ~7 planes fewer.  The smallest budget T for which a greedy cut needs <= wgs ranges.
    struct RangeCache { int N, C, wgs, grid, maxspan; int qs[TW_MAXWG + 1]; };
    static thread_local RangeCache rc = {0, 0, 0, 0, 0, {0}};        // (the cut costs ~0.2 ms of host time: once per shape, not per launch)
    if (rc.N == N && rc.C == C && rc.wgs == wgs) {
        p.grid = rc.grid; p.maxspan = rc.maxspan;
        for (int b = 0; b <= rc.grid; ++b) p.qs[b] = rc.qs[b];
    } else {
    auto cut = [&](long long T, int* qs) -> int {                     // -> ranges used (qs filled if not NULL), or wgs + 1 if T is too small
        int b = 0, q = 0;
        while (q < P) {
            if (b >= wgs) return wgs + 1;
            if (qs) qs[b] = q;
            long long cost = 0; int q2 = q;
            while (q2 < P) {
                const long long add = 4 + ((q2 > q && q2 % N == 0) ? 29 : 0);
                if (cost + add > T && q2 > q) break;
                cost += add; ++q2;
            }
            q = q2; ++b;
        }
        if (qs) qs[b] = P;
        return b;
    };
    long long lo = 4, hi = 4ll * P + 29ll * C;
    while (lo < hi) { const long long mid = (lo + hi) / 2; if (cut(mid, nullptr) <= wgs) hi = mid; else lo = mid + 1; }
    p.grid = cut(lo, p.qs);
    p.maxspan = 1;
    for (int b = 0; b < p.grid; ++b) { const int span = (p.qs[b + 1] - 1) / N - p.qs[b] / N + 1; if (span > p.maxspan) p.maxspan = span; }
    rc.N = N; rc.C = C; rc.wgs = wgs; rc.grid = p.grid; rc.maxspan = p.maxspan;
    for (int b = 0; b <= p.grid; ++b) rc.qs[b] = p.qs[b];
    }
    if (p.maxspan > 8) return false;
    p.tensor_bytes = (unsigned)((size_t)N * C * H * W * 2);
#ifdef SLAK_TRIROWS_DEV
    { static const int dbg = [] { const char* e = slak_dev_getenv("SLAK_TRIROWS_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }
#else
    p.dbg = 0;
#endif
    return (size_t)N * C * H * W * 2 < 0x7fffffffull;
}

static size_t tri_rows_lds_bytes(const TriRowsParams& p) {
    const size_t ntot = 2 * p.K * MF_TAPS + MF_TAPS * MF_TAPS;
    return 64 + (size_t)TW_NB * 4 * p.RS * p.CPR * 16 + (size_t)MF_WAVES * 32 * 64 * 4 + (size_t)(MF_WAVES + 1) * ntot * 4 + 32 + 64;
}

static int tri_rows_wgs() {
    static const int wgs = [] { const char* e = slak_dev_getenv("SLAK_TRIROWS_WGS"); return e ? atoi(e) : 0; }();      // (dev: a grid other than one workgroup per CU)
    return wgs > 0 ? wgs : mfma_cu_count();
}

bool dwconv_mfma_tri_wgrad_rows_supported(int N, int C, int H, int W, int K, int dtype) {
    if (!tri_rows_enabled() || (dtype != SLAK_BF16 && dtype != SLAK_F16)) return false;
    TriRowsParams p;
    // the SAME workgroup count as the launch (the device's CU count; 256 without a device): a caller that plans on this answer -- the C++ block runner asks once per
    // shape -- must get what the launch will decide (a fixed 256 here and the real count there could disagree on a part with another CU count)
    return fill_tri_rows_params(p, N, C, H, W, K, tri_rows_wgs()) && tri_rows_lds_bytes(p) <= 160 * 1024 - 256;
}

// records: grid * maxspan <= C + 2 * grid + 2 whatever the CU count turns out to be (grid <= min(C * N, CUs)); sized for up to 1024 CUs
size_t dwconv_mfma_tri_wgrad_rows_workspace(int N, int C, int K) {
    const size_t P = (size_t)N * C, g = P < 1024 ? P : 1024;
    return align_up(((size_t)C + 2 * g + 2 + (P / g + N - 1) / N) * (2 * K * MF_TAPS + MF_TAPS * MF_TAPS) * sizeof(float), 256);
}

template <typename T, int KS, int CPRC, bool SD>
static int launch_tri_rows_t(TriRowsParams& p, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_tri_wgrad_rows_kernel<T, KS, CPRC, SD>;
    const size_t lds = tri_rows_lds_bytes(p);
    if (!slak_set_max_lds((const void*)k, lds)) return SLAK_ERR_UNSUPPORTED;      // (process-wide maximum per kernel and device: slak_common.h)
    if ((size_t)p.grid * p.maxspan * (2 * p.K * MF_TAPS + MF_TAPS * MF_TAPS) * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    hipLaunchKernelGGL(k, dim3((unsigned)p.grid), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_tri_wgrad_rows(const void* const* dy, const void* x, float* const* dw, int dtype,
                                      int N, int C, int H, int W, int K, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_tri_wgrad_rows_supported(N, C, H, W, K, dtype)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    TriRowsParams p;
    if (!fill_tri_rows_params(p, N, C, H, W, K, tri_rows_wgs())) return SLAK_ERR_UNSUPPORTED;   // one wave per SIMD: one four-wave workgroup per CU
                                                                                       // (the CU count decides `per`, and with it how many channels a range can span)
    for (int b = 0; b < 3; ++b) { p.dy[b] = dy[b]; p.dw[b] = dw[b]; }
    p.x = x; p.partial = (float*)ws;
    p.counters = wgrad_arrival_counters(C);
    if (!p.counters) return SLAK_ERR_UNSUPPORTED;
    const bool bf = dtype == SLAK_BF16;
    // small branch on the diagonal tiles (see the kernel): four k-steps and a second X tile that reaches back to row 30 or further
    static const bool sd_on = [] { const char* e = getenv("SLAK_TRI_ROWS_SD"); return !(e && e[0] == '0'); }();      // A/B switch: 0 = all four tiles of the small branch (round 4-5)
    if (p.CPR == 9 && sd_on && H <= 62)
        return bf ? launch_tri_rows_t<bf16_t, 4, 9, true>(p, ws_bytes, st) : launch_tri_rows_t<f16_t, 4, 9, true>(p, ws_bytes, st);
    if (p.CPR == 9) return bf ? launch_tri_rows_t<bf16_t, 4, 9, false>(p, ws_bytes, st) : launch_tri_rows_t<f16_t, 4, 9, false>(p, ws_bytes, st);
    return bf ? launch_tri_rows_t<bf16_t, 3, 7, false>(p, ws_bytes, st) : launch_tri_rows_t<f16_t, 3, 7, false>(p, ws_bytes, st);
}

}  // namespace slak
