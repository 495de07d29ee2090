// slak_amd/csrc/dwconv_mfma_dma.hip -- MFMA depthwise-conv forward / data-grad for the large maps (56x56, 28x28 class;
// plane bytes a multiple of 16): LDS-DMA input ring + single-accumulator Toeplitz GEMM.
//
//   long axis t (extent Wt, KL taps)    short axis l (extent Wl, 5 taps)      per plane, per channel:
//   Y[o, u] = sum_r sum_i T_r[o, i] * X[i, u + r - 2]          A = T_r (dense 1-D Toeplitz, packed once per call by
//                                                              toeplitz_pack_kernel), B = rows u-2..u+2 of the plane image:
// the five short taps are five B fragments read at five row addresses and accumulated into ONE 32x32 accumulator, so
// there is no cross-lane shift at all and only 16 accumulator registers (the register-staged kernel of dwconv_mfma.hip
// keeps one accumulator per tap and combines them with DPP lane shifts, ~11 cycles each on gfx950: tools/dpp_probe.hip).
//   * horizontal kernels (5xK): the contraction runs along W, contiguous in the DMA image: B = ds_read_b128 of image rows.
//   * vertical kernels (Kx5): the contraction runs along H.  Each group is transposed once LDS->LDS (ds_read_b64_tr_b16 +
//     ds_write_b64, 13 instruction pairs per 56x56 plane) and then runs the SAME core on x^T; results are written
//     back column-wise.
// Input: planes go HBM -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip), DMA_NB groups deep, so every
// workgroup keeps several planes in flight.  LDS-DMA writes are lane-linear (destination = M0 base + lane*16:
// tools/dma_probe.hip), so a plane lands as a plain row-major image of pitch W; the zero padding the algorithm needs is
// applied to the B fragments (v_cndmask on rows outside the plane and on k >= Wt), which also keeps NaN/Inf of a
// neighbouring row from leaking in.
// Synchronisation: waves 0,1 issue the DMAs (inline asm, invisible to hipcc's waitcnt pass) and wait for them with a
// counted `s_waitcnt vmcnt(N)` -- exact, because those waves issue no other vector-memory operation in the loop;
// waves 2,3 copy finished planes from the LDS out-buffer to HBM.  Barriers are raw s_barrier + lgkmcnt(0), never
// __syncthreads() (which would drain the DMA queue).
#include "mfma_common.h"

namespace slak {

constexpr int DMA_NB = 4;               // ring depth (groups)

unsigned long long* g_dma_dbg = nullptr;   // dev hook: slak_debug_set_phase_buffer()

struct MfmaDmaParams {
    const void* x; const float* w; const uint16_t* frags; void* y;
    int N, C, H, W, kh, kw, flip;
    int Wt, Wl, KL, padL;
    int G;                 // planes per group (iteration)
    int tpp;               // 32-lane tiles per plane
    int ntiles;            // G * tpp
    int chunks_pp;         // 16-byte chunks per plane (HW/8)
    int group_elems;       // LDS elements per ring slot (G*HW)
    int PT;                // pitch of the transposed image (vertical kernels)
    int xt_rows;           // rows of one transposed plane image (W rounded up to 16)
    int planes_per_wg, slices;
    unsigned tensor_bytes;
    unsigned long long* dbg;   // optional phase timers (s_memtime), [wave][8]; NULL in production
};

constexpr int DMA_WAVES = 2;            // waves 0..1 issue DMAs, waves 2..3 store results
constexpr int DMA_NCO = 4;              // 16-byte copy-out chunks per storing thread per group (upper bound)
constexpr int DMA_MAX_IPW = 4;          // DMA instructions per issuing wave per group (upper bound)
constexpr int DMA_NTR = 4;              // transpose blocks (4 rows x 16 cols) per lane group per group of planes (upper bound)

// MT: 32-row tiles along the Toeplitz axis (wave w owns tile w % MT); KS: 16-deep k-steps; VERT: long axis = H;
// BAND: the filter is much shorter than the map (5x5 branch) -> skip all-zero Toeplitz blocks (wave-uniform branches).
// PACKED: Toeplitz fragments come from the workspace (toeplitz_pack_kernel; pays for MT=2: 20 fragments per wave), else they are
// built in the kernel from the fp32 filter (MT=1: 10 fragments per wave, cheaper than a second launch).
// R16: image rows are 16-byte aligned (W % 8 == 0); otherwise B fragments are read as two 8-byte halves.
template <typename T, int MT, int KS, bool VERT, bool BAND, bool R16, bool PACKED>
__global__ __launch_bounds__(MF_THREADS, 3) void dwconv_mfma_dma_kernel(const MfmaDmaParams p) {
    constexpr int NG = MF_TAPS;
    constexpr int WL = MF_WAVES / MT;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int HW = p.H * p.W;
    uint16_t* ring = lds;                                            // DMA_NB slots of group_elems (+ slack behind the last)
    uint16_t* lout = lds + DMA_NB * p.group_elems + 64;              // 2 x [G][HW]
    float* lw = (float*)(lout + 2 * p.G * HW);                       // this channel's kh*kw filter (fp32)
    uint16_t* xt = (uint16_t*)(lw + ((p.kh * p.kw + 3) & ~3));       // vertical only: [G][xt_rows][PT]

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = wave % MT, wl = wave / MT;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    uint16_t* __restrict__ y = (uint16_t*)p.y;

    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    if (n_begin >= n_end) return;
    const int iters = (n_end - n_begin + p.G - 1) / p.G;

    // ---- DMA descriptor and this wave's share of a group's chunks -------------------------------------
    v4i_t rsrc;
    {
        const uint64_t a = (uint64_t)p.x;
        rsrc[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        rsrc[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rsrc[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes);
        rsrc[3] = 0x00020000;
    }
    const int TC = p.G * p.chunks_pp;                                // chunks per group
    const int CPW = (TC + DMA_WAVES - 1) / DMA_WAVES;                // chunks per issuing wave
    int my_chunks = 0;
    if (wave < DMA_WAVES) { my_chunks = TC - wave * CPW; if (my_chunks > CPW) my_chunks = CPW; if (my_chunks < 0) my_chunks = 0; }
    const int my_ipw = (my_chunks + 63) >> 6;                        // wave-uniform
    const unsigned ring_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, ring);
    // chunk ci of a group = (plane j, chunk q): source = plane (n0+j, c) + 16q bytes; LDS = slot + 16*ci (lane-linear)
    unsigned src_rel[DMA_MAX_IPW]; bool act[DMA_MAX_IPW];
#pragma unroll
    for (int i = 0; i < DMA_MAX_IPW; ++i) {
        const int loc = i * 64 + lane;
        const int ci = wave * CPW + loc;
        act[i] = (i < my_ipw) && (loc < my_chunks);
        const int j = act[i] ? ci / p.chunks_pp : 0, q = act[i] ? ci - j * p.chunks_pp : 0;
        src_rel[i] = (unsigned)(j * p.C * HW * 2 + q * 16);
    }
    auto issue_group = [&](int g) {
        if (g >= iters) return;
        const int n0 = n_begin + g * p.G;
        const unsigned base_off = (unsigned)(((size_t)n0 * p.C + c) * HW * 2);
        const unsigned slot = ring_base + (unsigned)((g % DMA_NB) * p.group_elems * 2) + (unsigned)(wave * CPW * 16);
#pragma unroll
        for (int i = 0; i < DMA_MAX_IPW; ++i) {
            if (i < my_ipw) {                                         // wave-uniform: the instruction count per group is exact
                if (act[i]) lds_dma16(base_off + src_rel[i], rsrc, __builtin_amdgcn_readfirstlane(slot + i * 1024));
            }
        }
    };

    // ---- prologue ---------------------------------------------------------------------------------------
    for (int g = 0; g < DMA_NB - 1; ++g) issue_group(g);
    s16x8 afrag[NG][KS];
    bool ks_active[KS];
    if constexpr (PACKED) {
        // (the vmcnt(0) in front of the first use also drains the prologue DMAs: once)
        load_toeplitz_frags<NG, KS>(afrag, ks_active, p.frags, c, MT, mt, lane, 32, p.Wt, p.KL, p.padL);
    } else {
        const int ntap = p.kh * p.kw;
        for (int i = tid; i < ntap; i += MF_THREADS) lw[i] = p.w[(size_t)c * ntap + i];
        __syncthreads();
        build_toeplitz_frags_lds<T, NG, KS, VERT>(afrag, ks_active, lw, mt, lane, p.Wt, p.KL, p.padL, p.kw, p.flip);
    }

    // copy-out map of the storing waves (fixed per thread): 16-byte chunk idx -> out-buffer offset == idx*8, global offset
    int co_g[DMA_NCO], co_j[DMA_NCO];
    {
        const int cpp = HW / 8, total = p.G * cpp;
#pragma unroll
        for (int k = 0; k < DMA_NCO; ++k) {
            const int idx = (tid - DMA_WAVES * 64) + k * (MF_WAVES - DMA_WAVES) * 64;
            const bool ok = wave >= DMA_WAVES && idx < total;
            const int j = ok ? idx / cpp : 0, rem = ok ? idx - j * cpp : 0;
            co_j[k] = ok ? j : -1;
            co_g[k] = j * p.C * HW + rem * 8;
        }
    }
    // vertical: transpose map (fixed per thread).  Block b of a group = (plane j, 4 image rows kb, 16 image columns cb);
    // the 16 lanes of a group read it with one ds_read_b64_tr_b16 (lane i16 supplies row kb*4 + i16/4, columns cb*16 + 4*(i16%4)
    // and receives column cb*16 + i16, rows kb*4..+3) and write 8 bytes of x^T.
    int tr_r[DMA_NTR], tr_w[DMA_NTR];
    if constexpr (VERT) {
        const int grp = lane >> 4, i16 = lane & 15;
        const int kbs = p.H / 4, cbs = p.xt_rows / 16, per_plane = kbs * cbs, total = p.G * per_plane;
        for (int i = tid; i < p.G * p.xt_rows * p.PT / 2; i += MF_THREADS) ((unsigned*)xt)[i] = 0u;   // pads of x^T stay zero
#pragma unroll
        for (int k = 0; k < DMA_NTR; ++k) {
            const int b = (k * MF_WAVES + wave) * 4 + grp;
            const bool ok = b < total;
            const int j = ok ? b / per_plane : 0, rem = ok ? b - j * per_plane : 0;
            const int kb = rem / cbs, cb = rem - kb * cbs;
            tr_r[k] = ok ? j * HW + (kb * 4 + (i16 >> 2)) * p.W + cb * 16 + (i16 & 3) * 4 : -1;
            tr_w[k] = (j * p.xt_rows + cb * 16 + i16) * p.PT + kb * 4;
        }
    }

    // ---- per-lane constants of the compute core ------------------------------------------------------------
    const int pitch = VERT ? p.PT : p.W;                              // row pitch of the image the core reads
    const int plane_stride = VERT ? p.xt_rows * p.PT : HW;

    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = p.dbg != nullptr && blockIdx.x == 0;
#define PH_T0() unsigned long long t__ = prof ? __builtin_readcyclecounter() : 0
#define PH_ADD(k) do { if (prof) { unsigned long long n__ = __builtin_readcyclecounter(); tph[k] += n__ - t__; t__ = n__; } } while (0)
    for (int it = 0; it < iters; ++it) {
        const int n0 = n_begin + it * p.G;
        PH_T0();
        if (wave < DMA_WAVES) {                                        // my part of group `it` has landed: only the DMAs of the
            int younger = iters - 1 - it; if (younger > DMA_NB - 2) younger = DMA_NB - 2;   // younger groups may still be in flight
            wait_vmcnt_dyn(younger * my_ipw);
        }
        PH_ADD(0);
        wg_barrier();                                                 // B1: everyone's part has landed
        PH_ADD(1);

        const uint16_t* img = ring + (it % DMA_NB) * p.group_elems;
        if constexpr (VERT) {
#pragma unroll
            for (int k = 0; k < DMA_NTR; ++k) {
                if (tr_r[k] >= 0) {
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, img + tr_r[k]));
                    *(s16x4*)(xt + tr_w[k]) = v;
                }
            }
            wg_barrier();                                             // B1b: x^T complete
            img = xt;
        }
        uint16_t* outb = lout + (it & 1) * p.G * HW;
        for (int tile = wl; tile < p.ntiles; tile += WL) {
            const int j = tile / p.tpp, sub = tile - j * p.tpp;
            const uint16_t* pim = img + j * plane_stride;
            const int pos = sub * 32 + l31;                           // lane -> position along the short (lane) axis
            // B fragment of tap r, k-step ks: 8 consecutive k of image row pos + r - 2 (zero outside the plane / beyond Wt)
            const uint16_t* rp[MF_TAPS]; bool inb[MF_TAPS];
#pragma unroll
            for (int r = 0; r < MF_TAPS; ++r) {
                const int row = pos + r - 2;
                inb[r] = (unsigned)row < (unsigned)p.Wl;
                rp[r] = pim + (inb[r] ? row : 0) * pitch + lhi * 8;
            }
            auto load_b = [&](int r, int ks) -> s16x8 {
                u32x4 b;
                if constexpr (VERT || R16) b = *(const u32x4*)(rp[r] + ks * 16);            // 16-byte aligned rows
                else {                                                                    // W % 8 == 4: rows are 8-byte aligned
                    const u32x2 lo = *(const u32x2*)(rp[r] + ks * 16), hi = *(const u32x2*)(rp[r] + ks * 16 + 4);
                    b = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
                const int k0 = ks * 16 + lhi * 8;
                const bool lo_ok = inb[r] && (VERT || k0 < p.Wt), hi_ok = inb[r] && (VERT || k0 + 4 < p.Wt);
                b[0] = lo_ok ? b[0] : 0u; b[1] = lo_ok ? b[1] : 0u; b[2] = hi_ok ? b[2] : 0u; b[3] = hi_ok ? b[3] : 0u;
                return __builtin_bit_cast(s16x8, b);
            };
            f32x16 acc0, acc1;                                        // two chains (even / odd taps) keep the MFMA pipe busy
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
            if constexpr (!BAND) {
                s16x8 bcur[MF_TAPS], bnxt[MF_TAPS];
#pragma unroll
                for (int r = 0; r < MF_TAPS; ++r) bcur[r] = load_b(r, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) {
#pragma unroll
                        for (int r = 0; r < MF_TAPS; ++r) bnxt[r] = load_b(r, ks + 1);
                    }
#pragma unroll
                    for (int r = 0; r < MF_TAPS; ++r) {
                        if (r & 1) acc1 = mfma32<T>(afrag[r][ks], bcur[r], acc1);
                        else acc0 = mfma32<T>(afrag[r][ks], bcur[r], acc0);
                    }
#pragma unroll
                    for (int r = 0; r < MF_TAPS; ++r) bcur[r] = bnxt[r];
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (!ks_active[ks]) continue;
                    s16x8 b[MF_TAPS];
#pragma unroll
                    for (int r = 0; r < MF_TAPS; ++r) b[r] = load_b(r, ks);
#pragma unroll
                    for (int r = 0; r < MF_TAPS; ++r) {
                        if (r & 1) acc1 = mfma32<T>(afrag[r][ks], b[r], acc1);
                        else acc0 = mfma32<T>(afrag[r][ks], b[r], acc0);
                    }
                }
            }
            PH_ADD(2);
            if (pos < p.Wl && (n0 + j) < n_end) {
                uint16_t* op = outb + j * HW;
                if constexpr (!VERT) {
                    // lane = output row oh, register quad = 4 consecutive ow -> one 8-byte LDS store
                    uint16_t* orow = op + pos * p.W + mt * 32 + 4 * lhi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (mt * 32 + 8 * q + 4 * lhi < p.Wt) {
                            u32x2 v;
                            v[0] = pack2<T>(acc0[4 * q + 0] + acc1[4 * q + 0], acc0[4 * q + 1] + acc1[4 * q + 1]);
                            v[1] = pack2<T>(acc0[4 * q + 2] + acc1[4 * q + 2], acc0[4 * q + 3] + acc1[4 * q + 3]);
                            *(u32x2*)(orow + 8 * q) = v;
                        }
                    }
                } else {
                    // lane = output column ow, registers = rows oh -> column-wise 2-byte stores into the row-major plane
                    uint16_t* ocol = op + (mt * 32 + 4 * lhi) * p.W + pos;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (mt * 32 + 8 * q + 4 * lhi < p.Wt) {
                            const unsigned p0 = pack2<T>(acc0[4 * q + 0] + acc1[4 * q + 0], acc0[4 * q + 1] + acc1[4 * q + 1]);
                            const unsigned p1 = pack2<T>(acc0[4 * q + 2] + acc1[4 * q + 2], acc0[4 * q + 3] + acc1[4 * q + 3]);
                            ocol[(8 * q + 0) * p.W] = (uint16_t)(p0 & 0xffffu);
                            ocol[(8 * q + 1) * p.W] = (uint16_t)(p0 >> 16);
                            ocol[(8 * q + 2) * p.W] = (uint16_t)(p1 & 0xffffu);
                            ocol[(8 * q + 3) * p.W] = (uint16_t)(p1 >> 16);
                        }
                    }
                }
            }
        }
        PH_ADD(3);
        wg_barrier();                                                 // B2: out-buffer complete, ring slot `it` free
        PH_ADD(4);
        if (wave < DMA_WAVES) {
            issue_group(it + DMA_NB - 1);                             // into the slot that group it-1 used
        } else {
            // waves 2,3: finished planes -> HBM, 16 bytes per lane
            uint16_t* base = y + ((size_t)n0 * p.C + c) * HW;
#pragma unroll
            for (int k = 0; k < DMA_NCO; ++k) {
                const int idx = (tid - DMA_WAVES * 64) + k * (MF_WAVES - DMA_WAVES) * 64;
                if (co_j[k] >= 0 && n0 + co_j[k] < n_end) *(u32x4*)(base + co_g[k]) = *(const u32x4*)(outb + idx * 8);
            }
        }
        PH_ADD(5);
    }
    if (prof && lane == 0) { for (int k = 0; k < 6; ++k) p.dbg[wave * 8 + k] = tph[k]; p.dbg[wave * 8 + 6] = (unsigned long long)iters; }
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_dma_params(MfmaDmaParams& p, const ConvDims& d, bool vert, int MT, int KS, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    const int HW = d.H * d.W;
    if (HW % 8 || d.W % 4 || d.H % 4) return false;
    p.tpp = (p.Wl + 31) / 32;
    const int WLW = MF_WAVES / MT;
    p.G = p.tpp >= WLW ? 1 : WLW / p.tpp;
    if (p.G > d.N) p.G = d.N;
    p.ntiles = p.G * p.tpp;
    p.chunks_pp = HW / 8;
    p.group_elems = p.G * HW;
    p.PT = KS * 16 + 8;
    p.xt_rows = (d.W + 15) & ~15;
    const int TC = p.G * p.chunks_pp, CPW = (TC + DMA_WAVES - 1) / DMA_WAVES;
    if ((CPW + 63) / 64 > DMA_MAX_IPW) return false;
    if ((DMA_NB - 2) * ((CPW + 63) / 64) > 16) return false;          // wait_vmcnt_dyn covers 0..16
    if (TC > DMA_NCO * (MF_WAVES - DMA_WAVES) * 64) return false;
    if (vert && p.G * (d.H / 4) * (p.xt_rows / 16) > DMA_NTR * MF_WAVES * 4) return false;
    int slices = resident_wgs / d.C; if (slices < 1) slices = 1;      // one resident round: never more workgroups than fit at once
    int per = (d.N + slices - 1) / slices; per = (per + p.G - 1) / p.G * p.G; if (per < p.G) per = p.G;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * HW * 2);
    return true;
}

static size_t dma_lds_bytes(const MfmaDmaParams& p, bool vert) {
    return (size_t)(DMA_NB * p.group_elems + 64) * 2 + (size_t)2 * p.G * p.H * p.W * 2 + (size_t)((p.kh * p.kw + 3) & ~3) * 4 +
           (vert ? (size_t)p.G * p.xt_rows * p.PT * 2 : 0) + 16;
}

static int dma_class(const ConvDims& d, bool vert) {              // 2: MT=2/KS=4, 1: MT=1/KS=2, 0: not covered
    const int Wt = vert ? d.H : d.W;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return 0;
    if (Wt > 64 || Wt <= 16) return 0;
    return Wt > 32 ? 2 : 1;
}

bool dwconv_mfma_dma_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt) {
    if (x_dt != y_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16) || w_dt != SLAK_F32) return false;
    const bool vert = d.kh > d.kw;
    const int cls = dma_class(d, vert);
    if (!cls) return false;
    MfmaDmaParams p;
    if (!fill_dma_params(p, d, vert, cls == 2 ? 2 : 1, cls == 2 ? 4 : 2, 512)) return false;
    return dma_lds_bytes(p, vert) <= 64 * 1024;
}

template <typename K>
static int resident_workgroups(K kernel, size_t lds) {          // workgroups the chip holds at once for this kernel
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, MF_THREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    return per_cu * mfma_cu_count();
}

template <typename T, int MT, int KS, bool VERT, bool BAND, bool R16>
static int launch_dma_tv(MfmaDmaParams& p, const ConvDims& d, hipStream_t st) {
    auto k = dwconv_mfma_dma_kernel<T, MT, KS, VERT, BAND, R16, (MT == 2)>;
    fill_dma_params(p, d, VERT, MT, KS, 512);
    const size_t lds = dma_lds_bytes(p, VERT);                   // does not depend on the slice count
    static int resident = 0;                                      // per instantiation; LDS size varies little within a class
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (resident == 0) resident = resident_workgroups(k, lds);
    fill_dma_params(p, d, VERT, MT, KS, resident);
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, int MT, int KS>
static int launch_dma_t(MfmaDmaParams& p, const ConvDims& d, bool vert, bool band, hipStream_t st) {
    if (vert) return band ? launch_dma_tv<T, MT, KS, true, true, true>(p, d, st) : launch_dma_tv<T, MT, KS, true, false, true>(p, d, st);
    if (d.W % 8 == 0) return band ? launch_dma_tv<T, MT, KS, false, true, true>(p, d, st) : launch_dma_tv<T, MT, KS, false, false, true>(p, d, st);
    return band ? launch_dma_tv<T, MT, KS, false, true, false>(p, d, st) : launch_dma_tv<T, MT, KS, false, false, false>(p, d, st);
}

int launch_dwconv_mfma_dma(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                           const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_dma_supported(d, x_dt, w_dt, y_dt)) return SLAK_ERR_UNSUPPORTED;
    const bool vert = d.kh > d.kw;
    const int cls = dma_class(d, vert);
    const int MT = cls == 2 ? 2 : 1, KS = cls == 2 ? 4 : 2;
    MfmaDmaParams p;
    fill_dma_params(p, d, vert, MT, KS, 512);
    p.x = x; p.w = (const float*)w; p.frags = (const uint16_t*)ws; p.y = y; p.flip = flip_filter ? 1 : 0;
    if (MT == 2) {                                            // fragments packed once per call (20 per wave: cheaper than in-kernel)
        if (ws == nullptr || ws_bytes < toeplitz_pack_bytes(d.C, MT, MF_TAPS, KS)) return SLAK_ERR_WORKSPACE;
        ToeplitzPackParams tp{(const float*)w, (uint16_t*)ws, d.C, d.kh, d.kw, MT, MF_TAPS, KS, 1,
                              vert ? 1 : 0, flip_filter ? 1 : 0, p.Wt, p.KL, p.padL, x_dt == SLAK_BF16 ? 1 : 0};
        launch_toeplitz_pack(tp, st);
        SLAK_LAUNCH_CHECK();
    }
    p.dbg = g_dma_dbg;
    // band skipping pays when some (mt, ks) Toeplitz block is empty: filter half-width + 32 < 16*(KS-1)
    const bool band = (MT == 2) && (p.padL + 31 < 16 * (KS - 1));
    if (x_dt == SLAK_BF16) return cls == 2 ? launch_dma_t<bf16_t, 2, 4>(p, d, vert, band, st) : launch_dma_t<bf16_t, 1, 2>(p, d, vert, false, st);
    return cls == 2 ? launch_dma_t<f16_t, 2, 4>(p, d, vert, band, st) : launch_dma_t<f16_t, 1, 2>(p, d, vert, false, st);
}

}  // namespace slak
