// slak_amd/csrc/dwconv_mfma_wide_wgrad.hip -- MFMA depthwise-conv weight gradient for maps wider than 64 along the filter's long
// axis (the maps of dwconv_mfma_wide.hip: 96x96 of SLaK at 384 px, 128x128 of the 512 px segmentation crops).
//
// G_r[o, i] = sum_{n,u} dY[o, u] X[i, u + r - 2] restricted to the band |i - o| <= KL/2: the 32x32 tiles (mo, mi) with
// |mo - mi| <= 1 (7 at Wt = 96, 10 at 128), ONE WAVE PER TILE holding its five tap accumulators for the whole batch slice (the
// workgroup has as many waves as there are tiles); diagonal sums through the skewed per-wave tile and the last-arriver slice
// reduction exactly as in dwconv_mfma_wgrad_dma.hip.  Whole (dy, x) plane pairs are DMA'd with a padded pitch (destination
// chunk q of an image takes source chunk (q / cd) * cs + q % cd; pad chunks are skipped lanes); both MFMA operands are
// ds_read_b64_tr_b16 reads (the contraction runs over image rows); x sits two rows lower than dy so that "row k + r" is
// x[k + r - 2].  Vertical kernels land the pair compact and transpose it LDS -> LDS into the two stacks, then start the next
// pair's DMA into the same slot.
#include "mfma_common.h"

namespace slak {

struct WideWgradParams {
    const void* dy; const void* x; float* partial; float* dw; unsigned* counters;
    int N, C, H, W, kh, kw;
    int Wt, Wl, KL, padL;
    int MT, ntiles, NKS;
    int cs, cd, ninstr;    // source chunks per row, chunk pitch of the DMA images, DMA instructions per image
    int P;                 // pitch (elements) of the images the core reads
    int dyimg_bytes;       // slot = [dy image | x image]
    int slot_bytes, NB;
    int stack_dy_bytes, stack_x_bytes;     // vertical
    int tr_cbs, tr_q, tr_r;                // vertical: blocks per 4-row band; (4 * waves) / tr_cbs, (4 * waves) % tr_cbs
    int planes_per_wg, slices;
    unsigned tensor_bytes;
};

constexpr int WW_MAXW = 10;             // waves per workgroup (tiles), upper bound

template <typename T, bool VERT>
__global__ __launch_bounds__(WW_MAXW * 64) void dwconv_mfma_wide_wgrad_kernel(const WideWgradParams p) {
    constexpr int NG = MF_TAPS;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const LB = (char*)lds;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform(), nwaves = p.ntiles, nthreads = nwaves * 64;
    const int HW = p.H * p.W, ntap = p.kh * p.kw;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    const int nplanes = n_end > n_begin ? n_end - n_begin : 0;
    // tile of this wave: row mo of the band has tiles mi = mo-1 .. mo+1 clipped to [0, MT)
    int mo = 0, mi = 0;
    {
        int t = wave;
        for (int m = 0; m < p.MT; ++m) {
            const int lo = m > 0 ? m - 1 : 0, hi = m + 1 < p.MT ? m + 1 : p.MT - 1, cnt = hi - lo + 1;
            if (t >= 0 && t < cnt) { mo = m; mi = lo + t; }
            t -= cnt;
        }
    }
    const unsigned ring_b = 0;
    const unsigned stk_b = (unsigned)(p.NB * p.slot_bytes);          // vertical: dy stack, then x stack
    const unsigned live_bytes = stk_b + (VERT ? (unsigned)(p.stack_dy_bytes + p.stack_x_bytes) : 0u) + 8u * (unsigned)p.P * 2u;

    // ---- zero everything once: gaps, guard rows and the rows below the images stay zero ---------------------------------
    for (unsigned o = tid * 16; o < live_bytes; o += nthreads * 16) *(u32x4*)(LB + o) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    v4i_t rs_dy, rs_x;
    {
        const uint64_t a = (uint64_t)p.dy, b = (uint64_t)p.x;
        rs_dy[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs_dy[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs_dy[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs_dy[3] = 0x00020000;
        rs_x[0] = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu)); rs_x[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        rs_x[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs_x[3] = 0x00020000;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    const unsigned pitch = (unsigned)p.cd * 16;
    // DMA: instruction k of an image covers destination chunks [64 k, 64 k + 64); wave w issues k = w, w + nwaves, ... of BOTH images
    // and waits for them with vmcnt(0) (there are no global stores in the loop)
    const int q0 = wave * 64 + lane, qs = nwaves * 64;
    const int row0 = q0 / p.cd, cc0 = q0 - row0 * p.cd;
    const int inc_r = qs / p.cd, inc_c = qs - inc_r * p.cd;           // wave-uniform
    auto issue_pair = [&](int pl) {
        if (pl >= nplanes) return;
        const unsigned src0 = (unsigned)(((size_t)(n_begin + pl) * p.C + c) * HW * 2);
        const unsigned slot = lds_base + ring_b + (unsigned)(pl % p.NB) * (unsigned)p.slot_bytes;
        const unsigned d_dy = slot, d_x = slot + (unsigned)p.dyimg_bytes + (VERT ? 0u : 2u * pitch);
        int row = row0, cc = cc0;
        for (int k = wave; k < p.ninstr; k += nwaves) {
            if (cc < p.cs && row < p.H) {
                const unsigned so = src0 + (unsigned)(row * p.cs + cc) * 16u;
                lds_dma16(so, rs_dy, __builtin_amdgcn_readfirstlane(d_dy + (unsigned)k * 1024u));
                lds_dma16(so, rs_x, __builtin_amdgcn_readfirstlane(d_x + (unsigned)k * 1024u));
            }
            cc += inc_c; row += inc_r;
            if (cc >= p.cd) { cc -= p.cd; ++row; }
        }
    };
    issue_pair(0);

    // vertical: transpose both landed images into the stacks (block = 4 image rows x 16 image columns, see the forward kernel)
    auto transpose_pair = [&]() {
        const int grp = lane >> 4, i16 = lane & 15;
        const int total = (p.H >> 2) * p.tr_cbs;
        int b = wave * 4 + grp;
        int kb = b / p.tr_cbs, cb = b - kb * p.tr_cbs;
        for (; b < total; b += 4 * nwaves) {
            const unsigned src = ring_b + (unsigned)(kb * 4 + (i16 >> 2)) * pitch + (unsigned)(cb * 32 + (i16 & 3) * 8);
            const s16x4 vd = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, LB + src));
            const s16x4 vx = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, LB + src + (unsigned)p.dyimg_bytes));
            const int col = cb * 16 + i16;
            if (col < p.W) {
                *(s16x4*)(LB + stk_b + (unsigned)(col * p.P + kb * 4) * 2u) = vd;
                *(s16x4*)(LB + stk_b + (unsigned)p.stack_dy_bytes + (unsigned)((col + 2) * p.P + kb * 4) * 2u) = vx;
            }
            cb += p.tr_r; kb += p.tr_q;
            if (cb >= p.tr_cbs) { cb -= p.tr_cbs; ++kb; }
        }
    };

    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

    // fragment addresses (byte offsets at k-step 0) -- tr-read: the 16 lanes of group grp read a 4(k) x 16 block
    const int P = p.P;
    const int grp = lane >> 4, i16 = lane & 15;
    const int krow = (grp >> 1) * 8 + (i16 >> 2);
    const unsigned a_off = (unsigned)(krow * P + mo * 32 + (grp & 1) * 16 + (i16 & 3) * 4) * 2;
    const unsigned b_off = (unsigned)(krow * P + mi * 32 + (grp & 1) * 16 + (i16 & 3) * 4) * 2;     // tap g: + g rows
    const unsigned kstep_b = (unsigned)(16 * P) * 2, p4 = (unsigned)(4 * P) * 2, prow = (unsigned)P * 2;
    auto frag = [&](unsigned addr) -> s16x8 {                     // 8 k of one column: two transposing reads, 4 rows apart
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, LB + addr));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, LB + addr + p4));
        return s16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    };

    for (int pl = 0; pl < nplanes; ++pl) {
        wait_vmcnt<0>();                                          // my DMAs of pair `pl` (the only ones in flight) have landed
        wg_barrier();                                             // everyone's have; everyone is done with the previous pair
        unsigned dyb, xb;
        if constexpr (VERT) {
            transpose_pair();
            wg_barrier();
            issue_pair(pl + 1);
            dyb = stk_b; xb = stk_b + (unsigned)p.stack_dy_bytes;
        } else {
            issue_pair(pl + 1);
            dyb = ring_b + (unsigned)(pl % p.NB) * (unsigned)p.slot_bytes; xb = dyb + (unsigned)p.dyimg_bytes;
        }
        // k-loop over the image rows, software-pipelined by hand: tap g's fragment of the NEXT k-step is fetched right after
        // this k-step's MFMA g has issued
        const unsigned ab = dyb + a_off, xbb = xb + b_off;
        unsigned ko = 0;
        s16x8 a = frag(ab), b[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) b[g] = frag(xbb + (unsigned)g * prow);
        __builtin_amdgcn_sched_barrier(0);
        for (int ks = 0; ks < p.NKS; ++ks) {
            const unsigned kn = ks + 1 < p.NKS ? ko + kstep_b : ko;     // last k-step: re-read (discarded)
            const s16x8 an = frag(ab + kn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                acc[g] = mfma32<T>(a, b[g], acc[g]);
                b[g] = frag(xbb + kn + (unsigned)g * prow);
                __builtin_amdgcn_sched_barrier(0);
            }
            a = an; ko = kn;
        }
    }
    wait_vmcnt<0>();
    __syncthreads();                                              // ring and stacks are dead: their space becomes the scratch

    // ---- diagonal sums dw[tau] = sum_o G[o, o + tau - padL], per wave, through the SKEWED tile (G[o][i] -> row o, column
    // i - o + 31: a diagonal becomes a column), fixed order
    float* scratch = (float*)lds;                                 // [nwaves][32][64]
    float* dwl = scratch + nwaves * (32 * 64);                    // [nwaves][ntap]
    float* mine = dwl + wave * ntap;
    float* tile = scratch + wave * (32 * 64);
    for (int i = lane; i < 32 * 64 / 4; i += 64) ((u32x4*)tile)[i] = u32x4{0u, 0u, 0u, 0u};
    for (int i = lane; i < ntap; i += 64) mine[i] = 0.f;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool col_ok = mi * 32 + l31 < p.Wt;                     // this lane's input position i exists
    int o_max = p.Wt - mo * 32; if (o_max > 32) o_max = 32;       // rows o that exist (wave-uniform)
    float* wr = tile + (4 * lhi) * 64 + (l31 - 4 * lhi + 31);     // register r -> row (r&3) + 8*(r>>2) (+4*lhi), column l31 - row + 31
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (col_ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) + 4 * lhi < o_max) wr[((r & 3) + 8 * (r >> 2)) * 63] = acc[g][r];      // +64 per row, -1 per row
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 63) {
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 32; ++o) part[o & 3] += tile[o * 64 + lane];
            const int tau = lane - 31 + (mi - mo) * 32 + p.padL;
            if (tau >= 0 && tau < p.KL) mine[VERT ? (tau * p.kw + g) : (g * p.kw + tau)] = (part[0] + part[1]) + (part[2] + part[3]);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int t = tid; t < ntap; t += nthreads) {
        float s = dwl[t];
        for (int w = 1; w < nwaves; ++w) s += dwl[w * ntap + t];
        wgrad_store_partial(&p.partial[((size_t)slice * p.C + c) * ntap + t], s);
    }
    __syncthreads();                                              // dwl has been read: lds[0] may become the arrival flag
    if (p.counters) wgrad_finish(p.partial, p.dw, p.counters + c, (int*)lds, p.slices, p.C, c, 1, ntap, tid, nthreads);
}

static bool fill_wide_wgrad_params(WideWgradParams& p, const ConvDims& d, bool vert, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return false;
    if (p.Wt <= 64 || p.Wt > 128 || p.Wt % 16) return false;
    if (d.W % 8 || d.H % 4 || p.Wl % 4 || p.Wl > 128) return false;
    if (p.KL > 63) return false;
    p.MT = (p.Wt + 31) / 32; p.ntiles = 3 * p.MT - 2;
    if (p.ntiles > WW_MAXW) return false;
    p.NKS = (p.Wl + 15) / 16;
    p.cs = d.W / 8;
    const int rows16 = p.NKS * 16;
    // tr-reads want a pitch of +-64 bytes mod 256 (chunk pitch 4 or 12 mod 16); fall back to the smallest pitch if LDS runs out
    for (int attempt = 0; attempt < 2; ++attempt) {
        int cdw = p.cs; if (attempt == 0) while (cdw % 16 != 4 && cdw % 16 != 12) ++cdw;
        if (vert) {
            p.cd = p.cs;                                           // compact landing images; the stacks carry the good pitch
            int pw = p.Wt / 8; if (attempt == 0) while (pw % 16 != 4 && pw % 16 != 12) ++pw;
            p.P = pw * 8;
            p.dyimg_bytes = d.H * p.cd * 16;
            p.slot_bytes = 2 * p.dyimg_bytes; p.NB = 1;
            p.stack_dy_bytes = rows16 * p.P * 2; p.stack_x_bytes = (rows16 + 4) * p.P * 2;
        } else {
            p.cd = cdw; p.P = cdw * 8;
            p.dyimg_bytes = rows16 * p.cd * 16;
            p.slot_bytes = p.dyimg_bytes + (rows16 + 4) * p.cd * 16; p.NB = 2;
            p.stack_dy_bytes = p.stack_x_bytes = 0;
        }
        const size_t live = (size_t)p.NB * p.slot_bytes + p.stack_dy_bytes + p.stack_x_bytes;
        if (live + 8 * (size_t)p.P * 2 + 64 <= 160 * 1024) break;
        if (attempt == 1) return false;
    }
    p.ninstr = (d.H * p.cd + 63) / 64;
    p.tr_cbs = (d.W + 15) / 16; p.tr_q = (4 * p.ntiles) / p.tr_cbs; p.tr_r = (4 * p.ntiles) % p.tr_cbs;
    int slices = resident_wgs / d.C; if (slices < 1) slices = 1;
    if (slices > d.N) slices = d.N;
    const int per = (d.N + slices - 1) / slices;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * d.H * d.W * 2);
    return true;
}

static size_t wide_wgrad_lds_bytes(const WideWgradParams& p) {
    // + 8 rows: the last k-step's fragment reads of tap 4 / the second 4-row half reach past the x image
    size_t live = (size_t)p.NB * p.slot_bytes + p.stack_dy_bytes + p.stack_x_bytes + 8 * (size_t)p.P * 2 + 64;
    const size_t epi = (size_t)p.ntiles * (32 * 64 * 4 + p.kh * p.kw * 4) + 64;
    return live > epi ? live : epi;
}

bool dwconv_mfma_wide_wgrad_supported(const ConvDims& d, int dy_dt, int x_dt) {
    if (dy_dt != x_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16)) return false;
    WideWgradParams p;
    if (!fill_wide_wgrad_params(p, d, d.kh > d.kw, 256)) return false;
    return wide_wgrad_lds_bytes(p) <= 160 * 1024;
}

size_t dwconv_mfma_wide_wgrad_workspace(const ConvDims& d) {
    return align_up((size_t)(d.N < 2048 ? d.N : 2048) * d.C * d.kh * d.kw * sizeof(float), 256);   // slices <= min(N, resident workgroups)
}

template <typename T, bool VERT>
static int launch_wide_wgrad_tv(WideWgradParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_wide_wgrad_kernel<T, VERT>;
    const size_t lds = wide_wgrad_lds_bytes(p);
    static thread_local size_t cached_key = 0; static thread_local int cached_per_cu = 0;     // per instantiation; queried once per (LDS size, block size)
    const size_t key = ((size_t)(slak_current_device() + 1) << 40) | (lds * 16 + (size_t)p.ntiles);      // (the attribute is per device)
    if (cached_key != key) {
        (void)slak_set_max_lds((const void*)k, lds);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, p.ntiles * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        cached_per_cu = per_cu > 8 ? 8 : per_cu; cached_key = key;
    }
    fill_wide_wgrad_params(p, d, VERT, cached_per_cu * mfma_cu_count());
    if ((size_t)p.slices * d.C * d.kh * d.kw * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3((unsigned)(p.ntiles * 64)), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_wide_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                  const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_wide_wgrad_supported(d, dy_dt, x_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    const bool vert = d.kh > d.kw;
    WideWgradParams p;
    fill_wide_wgrad_params(p, d, vert, 256);
    p.dy = dy; p.x = x; p.partial = (float*)ws;
    p.dw = dw; p.counters = wgrad_arrival_counters(d.C);
    int rc;
    if (x_dt == SLAK_BF16) rc = vert ? launch_wide_wgrad_tv<bf16_t, true>(p, d, ws_bytes, st) : launch_wide_wgrad_tv<bf16_t, false>(p, d, ws_bytes, st);
    else rc = vert ? launch_wide_wgrad_tv<f16_t, true>(p, d, ws_bytes, st) : launch_wide_wgrad_tv<f16_t, false>(p, d, ws_bytes, st);
    if (rc != SLAK_OK || p.counters) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, d.C * d.kh * d.kw, p.slices, st);
}

}  // namespace slak
