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
};

template <typename T, int MT, bool VERT>
__global__ __launch_bounds__(MF_THREADS, 2) void dwconv_mfma_wgrad_dma_kernel(const WgradDmaParams p) {
    constexpr int NG = MF_TAPS;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int HW = p.H * p.W, ntap = p.kh * p.kw;
    uint16_t* ring = lds;                                         // DW_NB slots of [dy part | x part], img_elems each
    const int slot_elems = 2 * p.img_elems;
    const int ring_elems = DW_NB * slot_elems > MF_WAVES * 32 * 33 * 2 ? DW_NB * slot_elems : MF_WAVES * 32 * 33 * 2;   // >= the scratch that aliases it
    uint16_t* stk = lds + ring_elems;                             // vertical: dy stack, then x stack
    float* dwl = (float*)(stk + (VERT ? (2 * p.stack_elems + 8 * p.P) : 0));   // [MF_WAVES][ntap]
    float* scratch = (float*)lds;                                 // [MF_WAVES][32*33] for the diagonal sums: aliases the (dead) ring

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = (MT == 2) ? (wave & 1) : 0, nt = (MT == 2) ? (wave >> 1) : 0;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    const int iters = (n_end > n_begin) ? (n_end - n_begin + p.G - 1) / p.G : 0;

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

    // fragment addresses (element offsets at k-step 0) -- tr-read: group grp reads a 4(k) x 16 block
    const int grp = lane >> 4, i16 = lane & 15;
    const int krow = (grp >> 1) * 8 + (i16 >> 2);
    const int a_off = krow * p.P + mt * 32 + (grp & 1) * 16 + (i16 & 3) * 4;
    int b_off[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) b_off[g] = (krow + g) * p.P + nt * 32 + (grp & 1) * 16 + (i16 & 3) * 4;
    const int kstep_elems = 16 * p.P;
    const int ks_first = (MT == 2) ? 0 : wave, ks_stride = (MT == 2) ? 1 : MF_WAVES;

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
        for (int ks = ks_first; ks < p.NKS; ks += ks_stride) {
            const uint16_t* ap = dys + a_off + ks * kstep_elems;
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, ap));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, ap + 4 * p.P));
            const s16x8 a = s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const uint16_t* bp = xs + b_off[g] + ks * kstep_elems;
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, bp));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, bp + 4 * p.P));
                acc[g] = mfma32<T>(a, s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]}, acc[g]);
            }
        }
        if constexpr (VERT) wg_barrier();                         // the stacks are rewritten at the top of the next iteration
    }
    wait_vmcnt<0>();
    __syncthreads();                                              // the ring is dead: its space becomes the diagonal-sum scratch

    // ---- diagonal sums (as in dwconv_mfma_wgrad.hip): per-wave 32x33 fp32 tile, fixed order -------------------------------
    float* mine = dwl + wave * ntap;
    float* tile = scratch + wave * (32 * 33);
    const int dd = lane;                                          // diagonal i - o = dd - 31
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 33 + l31] = acc[g][r];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (dd < 63) {
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 32; ++o) {
                const int il = o + dd - 31;
                const bool ok = il >= 0 && il < 32 && mt * 32 + o < p.Wt && nt * 32 + il < p.Wt;
                const float v = tile[o * 33 + (ok ? il : 0)];
                part[o & 3] += ok ? v : 0.f;
            }
            const int tau = dd - 31 + (nt - mt) * 32 + p.padL;
            if (tau >= 0 && tau < p.KL) mine[VERT ? (tau * p.kw + g) : (g * p.kw + tau)] = (part[0] + part[1]) + (part[2] + part[3]);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int t = tid; t < ntap; t += MF_THREADS) {
        float s = dwl[t];
#pragma unroll
        for (int w = 1; w < MF_WAVES; ++w) s += dwl[w * ntap + t];
        wgrad_store_partial(&p.partial[((size_t)slice * p.C + c) * ntap + t], s);
    }
    if (p.counters) wgrad_finish(p.partial, p.dw, p.counters + c, (int*)lds, p.slices, p.C, c, 1, ntap, tid, MF_THREADS);
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
    size_t ring = (size_t)DW_NB * 2 * p.img_elems * 2, scratch = (size_t)MF_WAVES * 32 * 33 * 4;
    if (ring < scratch) ring = scratch;                              // the diagonal-sum scratch aliases the ring
    return ring + (vert ? (size_t)(2 * p.stack_elems + 8 * p.P) * 2 : 0) + (size_t)MF_WAVES * p.kh * p.kw * 4 + 32;
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

template <typename T, int MT, bool VERT>
static int launch_wdma_t(WgradDmaParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_wgrad_dma_kernel<T, MT, VERT>;
    fill_wdma_params(p, d, VERT, MT, 512);
    const size_t lds = wdma_lds_bytes(p, VERT);
    static int resident = 0;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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

int launch_dwconv_mfma_wgrad_dma(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                 const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_wgrad_dma_supported(d, dy_dt, x_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    const bool vert = d.kh > d.kw;
    const int cls = wdma_class(d, vert);
    WgradDmaParams p;
    p.dy = dy; p.x = x; p.partial = (float*)ws;
    p.dw = dw; p.counters = wgrad_arrival_counters(d.C);
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
