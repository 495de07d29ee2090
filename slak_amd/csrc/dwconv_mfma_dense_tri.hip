// slak_amd/csrc/dwconv_mfma_dense_tri.hip -- the three branches of a decomposed large-kernel block (K x 5, 5 x K, 5 x 5:
// models/SLaK.py:82-100) on planes of at most 64 pixels (the 7 x 7 stage of SLaK: 98-byte planes), forward and data gradient, one
// launch each, 16-bit activations.
//
// On a 7 x 7 plane a 13 x 5 filter reaches every pixel from every pixel: the convolution of one channel is a DENSE 49 x 49 matrix
// M_c, and the batch is the other GEMM dimension -- the implicit-GEMM formulation of the reference (one HW x PQ Toeplitz operand
// per channel: SURVEY 2a), which is wasteful on large maps and exact on this one:
//     Y_c[po, n] = sum_pi M_c[po, pi] X_c[pi, n]        M_c[po, pi] = w[ri - ro + kh/2][ci - co + kw/2]   (0 outside the filter)
// v_mfma_f32_32x32x16: M = 32 output pixels, N = 32 IMAGES, K = 16 input pixels: 8 MFMAs per branch for 32 planes (the
// wave-independent per-plane kernels of dwconv_mfma_small_tri.hip spend 9 MFMAs, 18 fragment reads and 12-24 stores PER PLANE
// and are bound by instructions per plane: 0.2 of the HBM roofline).  Per wave (one channel, a slice of the batch), no workgroup
// barrier anywhere:
//   * the 32 planes of a unit arrive by LDS-DMA, one 16-byte piece per lane at a 2-byte aligned source (lane -> (image, piece)),
//     lane-linear in LDS = plane pitch 112 bytes: the B fragment of a lane (its image, 8 consecutive pixels) is one conflict-free
//     ds_read_b128.  What the last piece of a plane drags in from the next plane is cleared in the registers of the last k-step
//     (nothing foreign, NaN or not, reaches an MFMA); the tensor's very last piece is fetched early and shifted into place;
//   * the four fragments go to registers and the DMA of the unit after next is issued at once (two slots); the wave waits with a
//     counted vmcnt that knows how many stores it has issued since;
//   * the operator fragments (3 branches x 2 row tiles x 4 k-steps) are built once per wave from the filter in LDS;
//   * results leave straight from the accumulator: a lane holds 4 consecutive pixels of its image = one 8-byte store at a 2-byte
//     aligned address (tools/unaligned_probe.hip: gfx950 takes it); forward 3 x 7 stores per 32 planes, data gradient 7 (the three
//     branches accumulate in ONE accumulator over three units: dy_v, dy_h, dy_s).
#include <stdlib.h>
#include <type_traits>

#include "mfma_common.h"

namespace slak {

constexpr int DN_IMG = 32;              // planes (images of one channel) per unit
constexpr int DN_NS = 2;                // ring slots per wave
constexpr int DN_MAXW = 2 * 63 * MF_TAPS + 25;     // filter elements of the three branches
constexpr int DN_WBYTES = (DN_MAXW * 4 + 15) & ~15;

struct DenseTriParams {
    const void* in[3]; void* out[3]; const float* w[3];       // branch order: vertical (K x 5), horizontal (5 x K), small (5 x 5)
    int N, C, H, W, K, dgrad;
    int PE;                // pixels per plane (H * W <= 64)
    int NP, NPP;           // 16-byte pieces per plane, piece pitch (odd) of a plane in LDS
    int KS;                // 16-deep k-steps (ceil(PE / 16))
    int images_per_slice, slices;
    unsigned tensor_bytes;
    int dbg;               // dev (SLAK_DENSE_DBG): 1 skip the operator-fragment build, 2 skip the stores, 4 skip the loop
};

__device__ __forceinline__ unsigned dn_funnel(const unsigned (&oo)[6], int k, int wsh, int bsh) {
    unsigned lo = 0u, hi = 0u;
#pragma unroll
    for (int j = 0; j < 6; ++j) { if (j == k + wsh) lo = oo[j]; if (j == k + wsh + 1) hi = oo[j]; }
    return bsh ? ((lo >> bsh) | (hi << (32 - bsh))) : lo;
}

template <typename T, bool DGRAD>
__global__ __launch_bounds__(MF_THREADS, 2) void dwconv_mfma_dense_tri_kernel(const DenseTriParams p) {
    constexpr int NT = DGRAD ? 3 : 1;                             // input tensors
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + 3) >> 2;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c = cb * 4 + wave;
    const int n_begin = slice * p.images_per_slice;
    int n_end = n_begin + p.images_per_slice; if (n_end > p.N) n_end = p.N;
    if (c >= p.C || n_begin >= n_end) return;                     // no workgroup barrier anywhere: waves may leave
    const int PE = p.PE, pitch = p.NPP * 16;
    const int slot_bytes = DN_IMG * pitch;
    const int wave_bytes = DN_NS * slot_bytes + 64 + DN_WBYTES;
    char* const L = (char*)lds + wave * wave_bytes;               // this wave's private region: [ring][rc table 64 B][filters fp32]
    unsigned char* const rc = (unsigned char*)(L + DN_NS * slot_bytes);        // pixel -> (row << 4 | col)
    float* const lw = (float*)(L + DN_NS * slot_bytes + 64);
    const int iters = (n_end - n_begin + DN_IMG - 1) / DN_IMG;
    const int nunits = iters * NT;

    // ---- DMA plan: destination piece q = 64 k + lane -> (image q / NPP, piece q % NPP) ------------------------------------
    v4i_t rs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint64_t a = (uint64_t)p.in[t];
        rs[t][0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs[t][1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs[t][2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs[t][3] = 0x00020000;
    }
    const unsigned plane_b = (unsigned)PE * 2;
    const unsigned gplane_b = (unsigned)p.C * plane_b;            // HBM bytes from image n to image n+1 of this channel
    const unsigned chan_b = (unsigned)c * plane_b;
    const int ndma = (DN_IMG * p.NPP + 63) >> 6;                  // instructions per unit (<= 5)
    int d_img[5]; unsigned d_off[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int q = 64 * k + lane, im = q / p.NPP, pc = q - im * p.NPP;
        const bool ok = k < ndma && im < DN_IMG && pc < p.NP;
        d_img[k] = ok ? im : (1 << 20);
        d_off[k] = (unsigned)im * gplane_b + (unsigned)pc * 16u;
    }
    const unsigned lds_wave = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds) + wave * wave_bytes;
    const unsigned tail_src = p.tensor_bytes - 16u;               // where the tensor's last piece is fetched from instead
    const unsigned last_piece = p.tensor_bytes - plane_b + (unsigned)(p.NP - 1) * 16u;   // its nominal source (ends behind the tensor)
    const bool tail_short = (unsigned)p.NP * 16u > plane_b;
    int issued = 0;                                               // vector-memory operations this wave has issued so far
    int mark[DN_NS];                                              // `issued` right after the DMA of the unit in slot s
    auto issue_unit = [&](int u) {
        if (u >= nunits) return;
        const int it = DGRAD ? u / 3 : u, t = DGRAD ? u - it * 3 : 0;
        const int n0 = n_begin + it * DN_IMG;
        const unsigned gb = (unsigned)n0 * gplane_b + chan_b;
        const unsigned dst = lds_wave + (unsigned)(u % DN_NS) * (unsigned)slot_bytes;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (k < ndma) {                                       // (wave-uniform)
                unsigned so = gb + d_off[k];
                if (tail_short && so == last_piece) so = tail_src;
                const bool ok = n0 + d_img[k] < n_end;
                if constexpr (DGRAD) {
                    if (ok) { if (t == 0) lds_dma16(so, rs[0], __builtin_amdgcn_readfirstlane(dst + k * 1024));
                              else if (t == 1) lds_dma16(so, rs[NT > 1 ? 1 : 0], __builtin_amdgcn_readfirstlane(dst + k * 1024));
                              else lds_dma16(so, rs[NT > 2 ? 2 : 0], __builtin_amdgcn_readfirstlane(dst + k * 1024)); }
                } else {
                    if (ok) lds_dma16(so, rs[0], __builtin_amdgcn_readfirstlane(dst + k * 1024));
                }
            }
        }
        // every instruction has a live lane unless the unit's first image of that instruction is past the slice: count what was issued
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) if (k < ndma && n0 + (64 * k) / p.NPP < n_end) ++cnt;
        issued += cnt;
        mark[u % DN_NS] = issued;
    };
    issue_unit(0);
    issue_unit(1);

    // ---- filters -> LDS (fp32), pixel table --------------------------------------------------------------------------
    const int ntl = p.K * MF_TAPS;
    for (int e = lane; e < ntl; e += 64) { lw[e] = p.w[0][(size_t)c * ntl + e]; lw[ntl + e] = p.w[1][(size_t)c * ntl + e]; }
    if (lane < 25) lw[2 * ntl + lane] = p.w[2][(size_t)c * 25 + lane];
    if (lane < 64) { const int r = lane / p.W; rc[lane] = lane < PE ? (unsigned char)((r << 4) | (lane - r * p.W)) : (unsigned char)0xff; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (also lands the first two units: prologue only)
    __builtin_amdgcn_wave_barrier();

    // ---- operator fragments: A[b][mt][ks], lane (m = l31 -> pixel mt*32 + l31, k = ks*16 + 8 lhi + e) --------------------
    // forward: M[po = m][pi = k] = w[ri - ro + kh/2][ci - co + kw/2]; data gradient: dx[pi] = sum_po M[po][pi] dy[po], so m = pi, k = po
    s16x8 mf[3][2][4];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int kh = b == 0 ? p.K : MF_TAPS, kw = b == 1 ? p.K : MF_TAPS;
        const float* wb = lw + (b == 0 ? 0 : (b == 1 ? ntl : 2 * ntl));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int pm = mt * 32 + l31;
            const unsigned char rcm = rc[pm < 64 ? pm : 63];
            const int rm = rcm >> 4, cm = rcm & 15;
            const bool m_ok = pm < PE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 d = {0u, 0u, 0u, 0u};
                if (ks < p.KS && !(p.dbg & 1)) {                  // (wave-uniform)
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int pk = ks * 16 + lhi * 8 + e;
                        const unsigned char rck = rc[pk];
                        const int rk = rck >> 4, ck = rck & 15;
                        const int a = (DGRAD ? rm - rk : rk - rm) + (kh >> 1), bb = (DGRAD ? cm - ck : ck - cm) + (kw >> 1);
                        const bool ok = m_ok && pk < PE && a >= 0 && a < kh && bb >= 0 && bb < kw;
                        v[e] = ok ? wb[ok ? a * kw + bb : 0] : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 8; e += 2) d[e >> 1] = pack2<T>(v[e], v[e + 1]);
                }
                mf[b][mt][ks] = __builtin_bit_cast(s16x8, d);
            }
        }
    }
    // masks of the image fragment of the LAST k-step: pixels >= PE hold the next plane's data
    unsigned xm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p0 = (p.KS - 1) * 16 + lhi * 8 + 2 * k;
        xm[k] = p0 + 1 < PE ? 0xffffffffu : (p0 < PE ? 0xffffu : 0u);
    }
    // stores: lane = image l31; quad (mt, q): pixels px0 = mt*32 + 8q + 4 lhi .. +3.  Number of store instructions of one tile set
    // (wave-uniform: the predicates depend on lhi only, and both halves are present in every instruction)
    const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc(p.out[0], 0, (int)p.tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc(p.out[DGRAD ? 0 : 1], 0, (int)p.tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(p.out[DGRAD ? 0 : 2], 0, (int)p.tensor_bytes, 0x00020000);
    int nst = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a0 = mt * 32 + 8 * q, a1 = a0 + 4;          // px0 of the two lane halves
            if (a0 + 3 < PE || a1 + 3 < PE) ++nst;                // the 8-byte store
#pragma unroll
            for (int j = 0; j < 3; ++j)                           // 2-byte stores of a partial quad
                if ((a0 + 3 >= PE && a0 + j < PE) || (a1 + 3 >= PE && a1 + j < PE)) ++nst;
        }
    auto store_tiles = [&](const f32x16 (&acc)[2], const __amdgpu_buffer_rsrc_t& r, unsigned img_off) {
        if (p.dbg & 2) img_off = 0xffffff00u;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a0 = mt * 32 + 8 * q, a1 = a0 + 4;
                const int px0 = a0 + 4 * lhi;
                const unsigned off = img_off + (unsigned)px0 * 2u;    // img_off = 0xffffff00 for images past the slice: dropped by the range check
                if (a0 + 3 < PE || a1 + 3 < PE) {
                    if (px0 + 3 < PE)
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2<T>(acc[mt][4 * q + 0], acc[mt][4 * q + 1]), pack2<T>(acc[mt][4 * q + 2], acc[mt][4 * q + 3])}, r, off, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if ((a0 + 3 >= PE && a0 + j < PE) || (a1 + 3 >= PE && a1 + j < PE)) {
                        if (px0 + 3 >= PE && px0 + j < PE)
                            __builtin_amdgcn_raw_buffer_store_b16((short)(pack2<T>(acc[mt][4 * q + j], 0.f) & 0xffffu), r, off + 2u * j, 0, 0);
                    }
            }
    };

    f32x16 acc[2];
    // one unit: wait for its planes, fragments -> registers, next-but-one unit's DMA, then the MFMAs of branch(es) TB
    auto unit = [&](int u, int it, auto TB) {
        constexpr int tb = decltype(TB)::value;                   // data gradient: the branch of this unit (0..2); forward: unused
        {
            int allowed = issued - mark[u % DN_NS];               // operations issued after this unit's DMA may still be outstanding
            wait_vmcnt_dyn(allowed > 63 ? 63 : allowed);
        }
        __builtin_amdgcn_wave_barrier();
        const int n0 = n_begin + it * DN_IMG;
        char* const slot = L + (u % DN_NS) * slot_bytes;
        if (tail_short && c == p.C - 1 && n0 + DN_IMG > p.N - 1 && n0 <= p.N - 1) {   // (wave-uniform) the unit holds the tensor's last plane
            // its last piece was fetched 16*NP - 2*PE bytes early: shift the tail elements to the front of the piece
            if (lane == 0) {
                char* pp = slot + (p.N - 1 - n0) * pitch + (p.NP - 1) * 16;
                const u32x4 o = *(const u32x4*)pp;
                const unsigned oo[6] = {o[0], o[1], o[2], o[3], 0u, 0u};
                const int shb = (p.NP * 16 - PE * 2) * 8, wsh = shb >> 5, bsh = shb & 31;
                *(u32x4*)pp = u32x4{dn_funnel(oo, 0, wsh, bsh), dn_funnel(oo, 1, wsh, bsh), dn_funnel(oo, 2, wsh, bsh), dn_funnel(oo, 3, wsh, bsh)};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        s16x8 xf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ks < p.KS) v = *(const u32x4*)(slot + l31 * pitch + ks * 32 + lhi * 16);
            if (ks == p.KS - 1) { v[0] &= xm[0]; v[1] &= xm[1]; v[2] &= xm[2]; v[3] &= xm[3]; }
            xf[ks] = __builtin_bit_cast(s16x8, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();                          // every lane has its fragments: the slot is free
        issue_unit(u + DN_NS);
        const int img = n0 + l31;
        const unsigned img_off = img < n_end ? (unsigned)img * gplane_b + chan_b : 0xffffff00u;
        if constexpr (!DGRAD) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        if (ks < p.KS) acc[mt] = mfma32<T>(mf[b][mt][ks], xf[ks], acc[mt]);
                }
                store_tiles(acc, b == 0 ? ro0 : (b == 1 ? ro1 : ro2), img_off);
                issued += nst;
            }
        } else {
            if constexpr (tb == 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (ks < p.KS) acc[mt] = mfma32<T>(mf[tb][mt][ks], xf[ks], acc[mt]);
            if constexpr (tb == 2) { store_tiles(acc, ro0, img_off); issued += nst; }
        }
    };
    for (int it = 0; it < ((p.dbg & 4) ? 0 : iters); ++it) {
        if constexpr (DGRAD) {
            unit(3 * it + 0, it, std::integral_constant<int, 0>{});
            unit(3 * it + 1, it, std::integral_constant<int, 1>{});
            unit(3 * it + 2, it, std::integral_constant<int, 2>{});
        } else {
            unit(it, it, std::integral_constant<int, 0>{});
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_dense_params(DenseTriParams& p, int N, int C, int H, int W, int K, bool dgrad, int target_wgs) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.dgrad = dgrad ? 1 : 0;
    if (N <= 0 || C <= 0 || K <= MF_TAPS || !(K & 1) || K > 63) return false;
    if (H < 1 || W < 1 || H > 15 || W > 15) return false;                // (row, col) packed in a byte
    p.PE = H * W;
    if (p.PE > 64 || p.PE < 8) return false;
    p.NP = (p.PE * 2 + 15) / 16; p.NPP = p.NP | 1;
    p.KS = (p.PE + 15) / 16;
    if ((DN_IMG * p.NPP + 63) / 64 > 5) return false;
    const int cblocks = (C + 3) / 4;
    int slices = target_wgs / cblocks; if (slices < 1) slices = 1;
    int per = (N + slices - 1) / slices; per = (per + DN_IMG - 1) / DN_IMG * DN_IMG;     // whole units
    if (per > (N + DN_IMG - 1) / DN_IMG * DN_IMG) per = (N + DN_IMG - 1) / DN_IMG * DN_IMG;
    p.images_per_slice = per; p.slices = (N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)N * C * H * W * 2);
    return (size_t)N * C * H * W * 2 < 0xffffff00ull && (size_t)N * C * H * W * 2 >= 32;
}

static size_t dense_wave_bytes(const DenseTriParams& p) { return (size_t)DN_NS * DN_IMG * p.NPP * 16 + 64 + (size_t)DN_WBYTES; }

bool dwconv_mfma_dense_tri_supported(int N, int C, int H, int W, int K, int dtype) {
    if (dtype != SLAK_BF16 && dtype != SLAK_F16) return false;
    DenseTriParams p;
    return fill_dense_params(p, N, C, H, W, K, false, 768);
}

template <typename T, bool DGRAD>
static int launch_dense_t(DenseTriParams& p, hipStream_t st) {
    auto k = dwconv_mfma_dense_tri_kernel<T, DGRAD>;
    fill_dense_params(p, p.N, p.C, p.H, p.W, p.K, DGRAD, 2 * mfma_cu_count());
    const size_t lds = (size_t)MF_WAVES * dense_wave_bytes(p);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(((p.C + 3) / 4) * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_dense_tri(bool dgrad, const void* const* in, void* const* out, const float* const* w, int dtype,
                                 int N, int C, int H, int W, int K, hipStream_t st) {
    if (!dwconv_mfma_dense_tri_supported(N, C, H, W, K, dtype)) return SLAK_ERR_UNSUPPORTED;
    DenseTriParams p;
    fill_dense_params(p, N, C, H, W, K, dgrad, 768);
    for (int i = 0; i < 3; ++i) { p.in[i] = in[dgrad ? i : 0]; p.out[i] = out[dgrad ? 0 : i]; p.w[i] = w[i]; }
    { const char* e = getenv("SLAK_DENSE_DBG"); p.dbg = e ? atoi(e) : 0; }
    if (dtype == SLAK_BF16) return dgrad ? launch_dense_t<bf16_t, true>(p, st) : launch_dense_t<bf16_t, false>(p, st);
    return dgrad ? launch_dense_t<f16_t, true>(p, st) : launch_dense_t<f16_t, false>(p, st);
}

}  // namespace slak
