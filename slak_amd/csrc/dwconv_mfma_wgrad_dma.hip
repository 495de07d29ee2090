// slak_amd/csrc/dwconv_mfma_wgrad_dma.hip -- the MFMA weight-gradient kernel of dwconv_mfma_wgrad.hip with an LDS-DMA input ring,
// for the large maps (56x56 / 28x28 class; plane bytes a multiple of 16, H and W multiples of 4).
//
// Same arithmetic (per-tap 1-D correlation GEMM over a stacked contraction axis, diagonal sums at the end), but x and dy planes go
// HBM -> LDS with `buffer_load_dwordx4 ... lds`, DW_NB groups deep.  There are no global stores inside the loop, so EVERY wave
// issues its share of the DMAs and waits for them with an exact counted `s_waitcnt vmcnt(N)`.
//   * horizontal kernels (5xK): each plane image is DMA'd straight to its place in the contraction stack (pitch W, two zero
//     rows between planes that the DMA never touches); x sits two rows lower so that "row k + rho" is x[k + rho - 2].
//   * vertical kernels (Kx5): images land compact and are transposed LDS->LDS into the stacks (ds_read_b64_tr_b16 + ds_write_b64).
// Fragment reads past column W of a row fall into the next row: they only feed correlation entries (o or i >= Wt) that the
// diagonal reduction discards.
#include "mfma_common.h"

namespace slak {

extern unsigned long long* g_dma_dbg;

constexpr int DW_NB = 2;                // ring depth: the next group streams in while the current one is consumed
constexpr int DW_MAX_IPW = 8;           // DMA instructions per wave per group (upper bound)
constexpr int DW_NTR = 4;               // transpose blocks of one plane per 16-lane group (upper bound: 64 blocks)

struct WgradDmaParams {
    const void* dy; const void* x; float* partial;
    float* dw; unsigned* counters;   // in-kernel slice reduction (counters == NULL: partials only, reduce kernel follows)
    int N, C, H, W, kh, kw;
    int Wt, Wl, KL, padL;
    int G;                 // planes per group
    int NKS;               // 16-deep k-steps per group
    int P;                 // pitch of the stacks the core reads (horizontal: W; vertical: MT*32)
    int chunks_pp;         // 16-byte chunks per plane
    int img_elems;         // LDS elements of one tensor's part of a ring slot
    int stack_elems;       // vertical: LDS elements of one transposed stack (dy; the x stack has 8 more rows)
    int planes_per_wg, slices;
    unsigned tensor_bytes;
    unsigned long long* dbg;   // dev: per-workgroup 100 MHz stamps (NULL in production)
};
#define WD_STAMP(k) do { if (p.dbg && threadIdx.x == 0) p.dbg[64 + blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// PC: the pitch P of the stacks as a compile-time constant (0: read it from the parameters) -- with it the ten row offsets of a
// k-step's fragments are instruction immediates and a k-step costs two address adds instead of twelve.
template <typename T, int MT, bool VERT, int PC>
__global__ __launch_bounds__(MF_THREADS, 2) void dwconv_mfma_wgrad_dma_kernel(const WgradDmaParams p) {
    constexpr int NG = MF_TAPS;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int HW = p.H * p.W, ntap = p.kh * p.kw;
    uint16_t* ring = lds;                                         // DW_NB slots of [dy part | x part], img_elems each
    const int slot_elems = 2 * p.img_elems;
    // the diagonal-sum scratch ([MF_WAVES][32][64] fp32 = 32 KB) aliases the ring AND the stacks (all dead by then): pad the ring
    // only if the two together are smaller than that
    const int live_elems = DW_NB * slot_elems + (VERT ? 2 * p.stack_elems + 8 * p.P : 0);
    const int ring_elems = DW_NB * slot_elems + (live_elems < MF_WAVES * 32 * 64 * 2 ? MF_WAVES * 32 * 64 * 2 - live_elems : 0);
    uint16_t* stk = lds + ring_elems;                             // vertical: dy stack, then x stack
    float* dwl = (float*)(stk + (VERT ? (2 * p.stack_elems + 8 * p.P) : 0));   // [MF_WAVES][ntap]
    float* scratch = (float*)lds;                                 // [MF_WAVES][32][64] for the diagonal sums: aliases the (dead) ring

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = (MT == 2) ? (wave & 1) : 0, nt = (MT == 2) ? (wave >> 1) : 0;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    const int iters = (n_end > n_begin) ? (n_end - n_begin + p.G - 1) / p.G : 0;

    WD_STAMP(0);
    // ---- zero the ring (gaps between plane images stay zero), the stacks and the per-wave tap arrays --------------
    {
        const int n8 = (ring_elems + (VERT ? 2 * p.stack_elems + 8 * p.P : 0)) / 8;
        for (int i = tid; i < n8; i += MF_THREADS) ((u32x4*)lds)[i] = u32x4{0u, 0u, 0u, 0u};
        for (int i = tid; i < MF_WAVES * ntap; i += MF_THREADS) dwl[i] = 0.f;
    }
    __syncthreads();

    // ---- DMA plan.  Chunk list of a group: tensor t (0 = dy, 1 = x) x plane j x chunk q, split evenly over the 4 waves in whole
    //      (tensor, plane) instruction groups so that every instruction writes lane-linear LDS. ---------------------------
    v4i_t rs_dy, rs_x;
    {
        const uint64_t a = (uint64_t)p.dy, b = (uint64_t)p.x;
        rs_dy[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs_dy[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs_dy[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs_dy[3] = 0x00020000;
        rs_x[0] = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu)); rs_x[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        rs_x[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs_x[3] = 0x00020000;
    }
    // instruction list: for (t, j): ceil(chunks_pp / 64) instructions; instruction id -> wave id round-robin
    const int ipp = (p.chunks_pp + 63) >> 6;                      // instructions per plane image
    const int ninstr = 2 * p.G * ipp;
    int my_ipw = 0;
    for (int id = wave; id < ninstr; id += MF_WAVES) ++my_ipw;    // wave-uniform
    const unsigned ring_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, ring);
    // per-wave instruction descriptors (fixed for every group): source offset relative to the group's first plane (per lane),
    // LDS destination relative to the slot (uniform), plane index, lane validity
    unsigned ins_src[DW_MAX_IPW], ins_dst[DW_MAX_IPW]; int ins_j[DW_MAX_IPW]; bool ins_ok[DW_MAX_IPW];
#pragma unroll
    for (int k = 0; k < DW_MAX_IPW; ++k) {
        const int id = wave + k * MF_WAVES;
        const bool live = id < ninstr;
        const int t = live ? id / (p.G * ipp) : 0, rem = live ? id - t * p.G * ipp : 0;
        const int j = rem / ipp, ii = rem - j * ipp;
        const int q = ii * 64 + lane;
        // horizontal: straight into the stack (x two rows lower); vertical: compact images, transposed later
        const int pdst = VERT ? j * HW : (2 + j * (p.Wl + 2) + (t ? 2 : 0)) * p.W;
        ins_src[k] = (unsigned)(j * p.C * HW * 2) + (unsigned)q * 16u;
        ins_dst[k] = (unsigned)((t * p.img_elems + pdst) * 2 + ii * 1024) | (t ? 0x80000000u : 0u);   // top bit: tensor select
        ins_j[k] = live ? j : (1 << 30);
        ins_ok[k] = live && q < p.chunks_pp;
    }
    auto issue_group = [&](int g) {
        if (g >= iters) return;
        const int n0 = n_begin + g * p.G;
        const unsigned gbase = (unsigned)(((size_t)n0 * p.C + c) * HW * 2);
        const unsigned slot = ring_base + (unsigned)((g % DW_NB) * slot_elems * 2);
#pragma unroll
        for (int k = 0; k < DW_MAX_IPW; ++k) {
            if (k < my_ipw && n0 + ins_j[k] < n_end) {            // wave-uniform: planes past the end of the slice are not staged
                const unsigned dst = slot + (ins_dst[k] & 0x7fffffffu);
                if (ins_ok[k]) {
                    if (ins_dst[k] & 0x80000000u) lds_dma16(gbase + ins_src[k], rs_x, __builtin_amdgcn_readfirstlane(dst));
                    else lds_dma16(gbase + ins_src[k], rs_dy, __builtin_amdgcn_readfirstlane(dst));
                }
            }
        }
    };

    WD_STAMP(3);
    issue_group(0);

    // vertical: transpose map of one plane (as in dwconv_mfma_wgrad.hip)
    int tr_r[DW_NTR], tr_w[DW_NTR];
    if constexpr (VERT) {
        const int g16 = lane >> 4, i16t = lane & 15;
        const int cbs = (p.W + 15) / 16, per_plane = (p.H / 4) * cbs;
#pragma unroll
        for (int k = 0; k < DW_NTR; ++k) {
            const int b = (k * MF_WAVES + wave) * 4 + g16;
            const bool ok = b < per_plane;
            const int kb = ok ? b / cbs : 0, cb = ok ? b - kb * cbs : 0;
            const int col = cb * 16 + i16t;
            const bool rd_ok = (cb * 16 + (i16t & 3) * 4) < p.W;
            tr_r[k] = ok ? (rd_ok ? (kb * 4 + (i16t >> 2)) * p.W + cb * 16 + (i16t & 3) * 4 : 0) : -1;
            tr_w[k] = (ok && col < p.W) ? (2 + col) * p.P + kb * 4 : -1;
        }
    }

    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

    // fragment addresses (byte offsets at k-step 0) -- tr-read: group grp reads a 4(k) x 16 block
    const int P = PC ? PC : p.P;
    const int grp = lane >> 4, i16 = lane & 15;
    const int krow = (grp >> 1) * 8 + (i16 >> 2);
    const unsigned a_off = (unsigned)(krow * P + mt * 32 + (grp & 1) * 16 + (i16 & 3) * 4) * 2;
    const unsigned b_off = (unsigned)(krow * P + nt * 32 + (grp & 1) * 16 + (i16 & 3) * 4) * 2;     // tap g: + g rows
    const unsigned kstep_b = (unsigned)(16 * P) * 2;
    const int ks_first = (MT == 2) ? 0 : wave, ks_stride = (MT == 2) ? 1 : MF_WAVES;
    char* const LB = (char*)lds;
    auto frag = [&](unsigned addr) -> s16x8 {                     // 8 k of one column: two transposing reads, 4 rows apart
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, LB + addr));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, LB + addr + (unsigned)(4 * P) * 2));
        return s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    };

    WD_STAMP(1);
    for (int it = 0; it < iters; ++it) {
        wait_vmcnt<0>();                                          // my DMAs of group `it` (the only ones in flight) have landed
        wg_barrier();                                             // everyone's have; everyone is done with the other slot
        // the other slot: clear stale rows if the next group is a partial one (horizontal: planes not staged must read as zero)
        if constexpr (!VERT) {
            if (it + 1 < iters && n_begin + (it + 1) * p.G + p.G > n_end) {          // (wave-uniform) last group of the slice only
                uint16_t* sl = ring + ((it + 1) % DW_NB) * slot_elems;
                for (int i = tid; i < slot_elems / 8; i += MF_THREADS) ((u32x4*)sl)[i] = u32x4{0u, 0u, 0u, 0u};
                wg_barrier();
            }
        }
        issue_group(it + 1);                                      // streams in while this group is consumed
        const uint16_t* slot = ring + (it % DW_NB) * slot_elems;
        const uint16_t* dys; const uint16_t* xs;
        if constexpr (VERT) {
            const int n0 = n_begin + it * p.G;
            for (int t = 0; t < 2; ++t) {
                for (int j = 0; j < p.G; ++j) {
                    const uint16_t* src = slot + t * p.img_elems + j * HW;
                    uint16_t* dst = stk + (t ? p.stack_elems + 2 * p.P : 0) + j * (p.Wl + 2) * p.P;
                    const bool live = n0 + j < n_end;              // planes that were not staged read as zero
#pragma unroll
                    for (int k = 0; k < DW_NTR; ++k) {
                        if (tr_r[k] >= 0) {
                            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, src + tr_r[k]));
                            if (!live) v = s16x4{0, 0, 0, 0};
                            if (tr_w[k] >= 0) *(s16x4*)(dst + tr_w[k]) = v;
                        }
                    }
                }
            }
            wg_barrier();
            dys = stk; xs = stk + p.stack_elems;
        } else {
            dys = slot; xs = slot + p.img_elems;
        }
        // k-loop, software-pipelined by hand and pinned with sched_barrier (hipcc otherwise emits ds_read; s_waitcnt lgkmcnt(0);
        // v_mfma five times per k-step): tap g's fragment of the NEXT k-step is fetched right after this k-step's MFMA g issued
        if (ks_first < p.NKS) {
            const unsigned ab = (unsigned)((const char*)dys - LB) + a_off, xb = (unsigned)((const char*)xs - LB) + b_off;
            unsigned ko = (unsigned)ks_first * kstep_b;
            s16x8 a = frag(ab + ko), b[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) b[g] = frag(xb + ko + (unsigned)(g * P) * 2);
            __builtin_amdgcn_sched_barrier(0);
            for (int ks = ks_first; ks < p.NKS; ks += ks_stride) {
                const bool more = ks + ks_stride < p.NKS;
                const unsigned kn = more ? ko + (unsigned)ks_stride * kstep_b : ko;     // last k-step: re-read (discarded)
                const s16x8 an = frag(ab + kn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    acc[g] = mfma32<T>(a, b[g], acc[g]);
                    b[g] = frag(xb + kn + (unsigned)(g * P) * 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                a = an; ko = kn;
            }
        }
        if constexpr (VERT) wg_barrier();                         // the stacks are rewritten at the top of the next iteration
    }
    WD_STAMP(2);
    wait_vmcnt<0>();
    __syncthreads();                                              // the ring is dead: its space becomes the diagonal-sum scratch

    // ---- diagonal sums dw[tau] = sum_o G[o, o + tau - padL], per wave, fixed order.  The 32x32 tile of a tap is written SKEWED --
    // G[o][i] to row o, column i - o + 31 -- so that a diagonal becomes a column and lane d just adds the rows of column d: no
    // index arithmetic or selects in the sum (the plain 32x33 tile cost 320 VALU per tap and wave: 9 us of a 47 us kernel).
    // Cells no (o, i) maps to, and the columns of lanes beyond the plane edge (which never write), stay zero from the fill below.
    float* mine = dwl + wave * ntap;
    float* tile = scratch + wave * (32 * 64);
    for (int i = lane; i < 32 * 64 / 4; i += 64) ((u32x4*)tile)[i] = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool col_ok = nt * 32 + l31 < p.Wt;                     // this lane's input position i exists
    int o_max = p.Wt - mt * 32; if (o_max > 32) o_max = 32;       // rows o that exist (wave-uniform)
    float* wr = tile + (4 * lhi) * 64 + (l31 - 4 * lhi + 31);     // register r -> row (r&3) + 8*(r>>2) (+4*lhi), column l31 - row + 31
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (col_ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r)                          // rows beyond the plane edge are not written: they stay zero
                if ((r & 3) + 8 * (r >> 2) + 4 * lhi < o_max) wr[((r & 3) + 8 * (r >> 2)) * 63] = acc[g][r];      // +64 per row, -1 per row
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 63) {
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 32; ++o) part[o & 3] += tile[o * 64 + lane];       // 32 independent reads
            const int tau = lane - 31 + (nt - mt) * 32 + p.padL;
            if (tau >= 0 && tau < p.KL) mine[VERT ? (tau * p.kw + g) : (g * p.kw + tau)] = (part[0] + part[1]) + (part[2] + part[3]);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    WD_STAMP(4);
    __syncthreads();
    for (int t = tid; t < ntap; t += MF_THREADS) {
        float s = dwl[t];
#pragma unroll
        for (int w = 1; w < MF_WAVES; ++w) s += dwl[w * ntap + t];
        wgrad_store_partial(&p.partial[((size_t)slice * p.C + c) * ntap + t], s);
    }
    WD_STAMP(5);
    if (p.counters) wgrad_finish(p.partial, p.dw, p.counters + c, (int*)lds, p.slices, p.C, c, 1, ntap, tid, MF_THREADS);
    WD_STAMP(6);
}

// ------------------------------------------------------------------------------------------------------------
static int wdma_class(const ConvDims& d, bool vert) {
    const int Wt = vert ? d.H : d.W;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return 0;
    if (Wt > 64 || Wt <= 16) return 0;
    return Wt > 32 ? 2 : 1;
}

static bool fill_wdma_params(WgradDmaParams& p, const ConvDims& d, bool vert, int MT, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    const int HW = d.H * d.W;
    if (HW % 8 || d.W % 4 || d.H % 4) return false;
    if (!vert && d.W > MT * 32) return false;
    p.chunks_pp = HW / 8;
    int slices = resident_wgs / d.C; if (slices < 1) slices = 1;
    if (slices > d.N) slices = d.N;
    int per = (d.N + slices - 1) / slices;
    // planes per group: ~12.5 KB per tensor for the horizontal kernels, half that for the vertical ones (they also hold the
    // transposed stacks), at most what a wave can issue
    int G = (vert ? 6272 : 12544) / (HW * 2); if (G < 1) G = 1;
    if (G > per) G = per;
    while (G > 1 && (2 * G * ((p.chunks_pp + 63) / 64) + MF_WAVES - 1) / MF_WAVES > DW_MAX_IPW) --G;
    p.G = G;
    per = (per + G - 1) / G * G;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    const int K = 2 + G * (p.Wl + 2);
    p.NKS = (K + 15) / 16;
    const int Kp = p.NKS * 16;
    if (vert) {
        p.P = MT * 32;
        p.img_elems = G * HW;
        p.stack_elems = Kp * p.P;
        if ((d.H / 4) * ((d.W + 15) / 16) > DW_NTR * MF_WAVES * 4) return false;
    } else {
        p.P = d.W;
        p.img_elems = (Kp + 8 + 2) * d.W;                    // stack rows + x shift + slack for the wrap-around of the last row
        p.stack_elems = 0;
    }
    p.img_elems = (p.img_elems + 7) & ~7;
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * HW * 2);
    return true;
}

static size_t wdma_lds_bytes(const WgradDmaParams& p, bool vert) {
    const size_t stacks = vert ? (size_t)(2 * p.stack_elems + 8 * p.P) * 2 : 0, scratch = (size_t)MF_WAVES * 32 * 64 * 4;
    size_t live = (size_t)DW_NB * 2 * p.img_elems * 2 + stacks;
    if (live < scratch) live = scratch;                              // the diagonal-sum scratch aliases the ring and the stacks
    return live + (size_t)MF_WAVES * p.kh * p.kw * 4 + 32;
}

bool dwconv_mfma_wgrad_dma_supported(const ConvDims& d, int dy_dt, int x_dt) {
    if (dy_dt != x_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16)) return false;
    const bool vert = d.kh > d.kw;
    const int cls = wdma_class(d, vert);
    if (!cls) return false;
    WgradDmaParams p;
    if (!fill_wdma_params(p, d, vert, cls == 2 ? 2 : 1, 512)) return false;
    return wdma_lds_bytes(p, vert) <= 78 * 1024;
}

size_t dwconv_mfma_wgrad_dma_workspace(const ConvDims& d) {
    return align_up((size_t)(d.N < 2048 ? d.N : 2048) * d.C * d.kh * d.kw * sizeof(float), 256);   // slices <= min(N, resident workgroups)
}

template <typename T, int MT, bool VERT, int PC>
static int launch_wdma_tp(WgradDmaParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_wgrad_dma_kernel<T, MT, VERT, PC>;
    fill_wdma_params(p, d, VERT, MT, 512);
    const size_t lds = wdma_lds_bytes(p, VERT);
    static int resident = 0;
    (void)slak_set_max_lds((const void*)k, lds);
    if (resident == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, MF_THREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        resident = per_cu * mfma_cu_count();
    }
    fill_wdma_params(p, d, VERT, MT, resident);
    if ((size_t)p.slices * d.C * d.kh * d.kw * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(MF_THREADS), wdma_lds_bytes(p, VERT), st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, int MT, bool VERT>
static int launch_wdma_t(WgradDmaParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    constexpr int PCOMMON = VERT ? MT * 32 : (MT == 2 ? 56 : 28);  // the SLaK maps; anything else takes the runtime-pitch build
    fill_wdma_params(p, d, VERT, MT, 512);
    if (p.P == PCOMMON) return launch_wdma_tp<T, MT, VERT, PCOMMON>(p, d, ws_bytes, st);
    return launch_wdma_tp<T, MT, VERT, 0>(p, d, ws_bytes, st);
}

int launch_dwconv_mfma_wgrad_dma(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                 const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_wgrad_dma_supported(d, dy_dt, x_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    const bool vert = d.kh > d.kw;
    const int cls = wdma_class(d, vert);
    WgradDmaParams p;
    p.dy = dy; p.x = x; p.partial = (float*)ws;
    p.dw = dw; p.counters = wgrad_arrival_counters(d.C);
    p.dbg = g_dma_dbg;
    int rc;
    if (x_dt == SLAK_BF16) {
        if (cls == 2) rc = vert ? launch_wdma_t<bf16_t, 2, true>(p, d, ws_bytes, st) : launch_wdma_t<bf16_t, 2, false>(p, d, ws_bytes, st);
        else rc = vert ? launch_wdma_t<bf16_t, 1, true>(p, d, ws_bytes, st) : launch_wdma_t<bf16_t, 1, false>(p, d, ws_bytes, st);
    } else {
        if (cls == 2) rc = vert ? launch_wdma_t<f16_t, 2, true>(p, d, ws_bytes, st) : launch_wdma_t<f16_t, 2, false>(p, d, ws_bytes, st);
        else rc = vert ? launch_wdma_t<f16_t, 1, true>(p, d, ws_bytes, st) : launch_wdma_t<f16_t, 1, false>(p, d, ws_bytes, st);
    }
    if (rc != SLAK_OK || p.counters) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, d.C * d.kh * d.kw, p.slices, st);
}

}  // namespace slak
