// slak_amd/csrc/dwconv_mfma_small_wgrad_dma.hip -- MFMA weight gradient for the 14x14-class planes (W even, 8 <= W <= 16,
// H <= 14, W <= 14), 16-bit activations: the wave-independent LDS-DMA streaming of dwconv_mfma_small_dma.hip applied to
//   G_r[o, i] = sum_{n,u} dY[o, u] * X[i, u + r - 2]        dw[tau, r] = sum_o G_r[o, o + tau - padL]
// (o, i along the long axis, u along the short axis).  A wave owns one channel and a slice of the batch:
//   * ONE `buffer_load_dwordx4 ... lds` per tensor per PAIR of planes lands two row-major pitch-16 images (two zero guard rows
//     behind each plane, never written);
//   * v_mfma_f32_16x16x32 with K = 32 = the 2 x 16 image rows of the pair: one MFMA per tap and pair, five 4-register
//     accumulators that live in registers over the whole slice.  Both operands are column reads of the row-major images =
//     ds_read_b64_tr_b16 (the pitch-16 image has the 8-byte alignment that read needs); the tap shift r-2 is a row offset;
//   * vertical kernels (Kx5) transpose both planes LDS->LDS first (one transposing read + one 8-byte write per plane) and
//     run the same core on dY^T, X^T;
//   * no global stores in the loop: the counted `s_waitcnt vmcnt(N)` only sees this wave's own DMAs;
//   * diagonal sums through a skewed per-wave 16x32 fp32 tile in fixed order, partial per slice, last-arriver reduction
//     (wgrad_finish): bitwise reproducible.
#include "mfma_common.h"

namespace slak {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ring slots (plane pairs of both tensors) per wave: template parameter NS (4: 47 KB per workgroup, 3 workgroups per CU; 3: 39 KB, 4 per CU)
constexpr int SWD_SLOT = 2048;          // bytes per slot: [dY pair 1024][X pair 1024]
// per-wave LDS region (bytes): [64 zero pad][ring][64 zero pad][dY^T pair][X^T pair][64 pad][res: 256 floats]
constexpr int SWD_RING = 64;
constexpr int swd_t(int ns) { return SWD_RING + ns * SWD_SLOT + 64; }
constexpr int swd_res(int ns) { return swd_t(ns) + 2048 + 64; }
constexpr int swd_wave_bytes(int ns) { return swd_res(ns) + 320 * 4; }
static_assert(3 * SWD_SLOT >= 16 * 32 * 4, "the diagonal-sum tile aliases the ring");

struct SmallWgradDmaParams {
    const void* dy; const void* x; float* partial; float* dw; unsigned* counters;
    int N, C, H, W, kh, kw, KL, padL, Wt, Wl;
    int images_per_slice, slices;
    unsigned tensor_bytes;
};

template <typename T> __device__ __forceinline__ f32x4_t swd_mfma16(s16x8 a, s16x8 b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t swd_mfma16<bf16_t>(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t swd_mfma16<f16_t>(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <typename T, bool VERT, int SWD_NS>
__global__ __launch_bounds__(MF_THREADS) void dwconv_mfma_small_wgrad_dma_kernel(const SmallWgradDmaParams p) {
    constexpr int SWD_T = swd_t(SWD_NS), SWD_RES = swd_res(SWD_NS), SWD_WAVE_BYTES = swd_wave_bytes(SWD_NS);
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + 3) >> 2;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c0 = cb * 4, c = c0 + wave;
    const int nch = p.C - c0 < 4 ? p.C - c0 : 4;
    const int n_begin = slice * p.images_per_slice;
    int n_end = n_begin + p.images_per_slice; if (n_end > p.N) n_end = p.N;
    const bool live = c < p.C && n_begin < n_end;                // (every wave reaches wgrad_finish: it has workgroup barriers)
    const int npairs = live ? (n_end - n_begin + 1) >> 1 : 0;
    char* const L = (char*)lds + wave * SWD_WAVE_BYTES;          // this wave's private region
    const int HW = p.H * p.W, ntap = p.kh * p.kw;

    for (int o = lane * 16; o < SWD_WAVE_BYTES; o += 64 * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // zeros are in place before any DMA can land on them

    // ---- DMA: lane -> (plane of the pair, image row, half of the row: columns 0..7 / W-8..W-1) ------------------------
    v4i_t rs_dy, rs_x;
    {
        const uint64_t a = (uint64_t)p.dy, b = (uint64_t)p.x;
        rs_dy[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs_dy[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs_dy[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs_dy[3] = 0x00020000;
        rs_x[0] = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu)); rs_x[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        rs_x[2] = rs_dy[2]; rs_x[3] = 0x00020000;
    }
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;
    const int d_pp = lane >> 5, d_row = (lane >> 1) & 15, d_half = lane & 1;
    const unsigned d_src = (unsigned)d_pp * gplane_b + (unsigned)(d_row * p.W) * 2 + (d_half ? (unsigned)(p.W - 8) * 2 : 0u);
    const bool d_rowok = d_row < p.H;
    const unsigned lds_wave = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds) + wave * SWD_WAVE_BYTES;
    const unsigned chan_b = (unsigned)(live ? c : 0) * (unsigned)HW * 2;
    auto issue_pair = [&](int q) {                                // two DMA instructions (dY pair, X pair) -> slot q % SWD_NS
        const int n0 = n_begin + 2 * q;
        const unsigned gb = (unsigned)n0 * gplane_b + chan_b;
        const unsigned dst = lds_wave + SWD_RING + (unsigned)(q % SWD_NS) * SWD_SLOT;
        if (d_rowok && n0 + d_pp < n_end) {
            lds_dma16(gb + d_src, rs_dy, __builtin_amdgcn_readfirstlane(dst));
            lds_dma16(gb + d_src, rs_x, __builtin_amdgcn_readfirstlane(dst + 1024));
        }
    };
    for (int q = 0; q < SWD_NS - 1 && q < npairs; ++q) issue_pair(q);

    // ---- lane constants.  Column reads: the 16 lanes of group g4 fetch a 4-row x 16-column block (lane i16 supplies row
    // i16/4 and the 8-byte piece i16%4, receives column i16, rows +0..3).  MFMA operand lane l: column l&15, k = (l>>4)*8 + e,
    // k = 16*plane + image row  ->  plane g4>>1, rows (g4&1)*8 + {0..3}, {4..7}: two reads.
    const int g4 = lane >> 4, i16 = lane & 15;
    const unsigned rd = (unsigned)((g4 >> 1) * 512 + ((g4 & 1) * 8 + (i16 >> 2)) * 32 + (i16 & 3) * 8);
    // vertical: transposing copy of plane pp: group g4 takes image rows 4*g4..+3, lane receives column slot i16 -> row of the
    // transposed image = true column (slots 8.. hold columns W-8..: duplicates of columns < 8 are not written)
    const unsigned trd = (unsigned)((4 * g4 + (i16 >> 2)) * 32 + (i16 & 3) * 8);
    const int t_row = i16 < 8 ? i16 : i16 - (16 - p.W);
    const bool twr_ok = i16 < 8 || t_row >= 8;
    const unsigned twr = (unsigned)(t_row * 32 + g4 * 8);

    f32x4_t acc[MF_TAPS];
#pragma unroll
    for (int r = 0; r < MF_TAPS; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int q = 0; q < npairs; ++q) {
        {
            int dm = npairs - 1 - q; if (dm > SWD_NS - 2) dm = SWD_NS - 2;
            wait_vmcnt_dyn(2 * dm);                               // only the DMAs of the pairs behind this one may be outstanding
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned slot = (unsigned)SWD_RING + (unsigned)(q % SWD_NS) * SWD_SLOT;
        if (q == npairs - 1 && ((n_end - n_begin) & 1)) {
            // odd slice: the second plane of the last pair was not fetched and its rows still hold an older plane: clear them
            // (lanes 0..31: dY plane, 32..63: X plane)
            *(u32x4*)(L + slot + 512 + (lane >> 5) * 1024 + (lane & 31) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        unsigned ab, xb;                                          // dY image pair; X image pair minus two rows (the tap shift is +r rows)
        if constexpr (VERT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                         // dY plane 0,1, X plane 0,1
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + slot + k * 512 + trd));
                if (twr_ok) *(s16x4*)(L + SWD_T + k * 512 + twr) = v;
            }
            ab = (unsigned)SWD_T + rd; xb = (unsigned)SWD_T + 1024 - 64 + rd;
        } else {
            ab = slot + rd; xb = slot + 1024 - 64 + rd;
        }
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + ab));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + ab + 128));
        const s16x8 a = s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + xb + r * 32));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + xb + r * 32 + 128));
            acc[r] = swd_mfma16<T>(a, s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]}, acc[r]);
        }
        if (q + SWD_NS - 1 < npairs) issue_pair(q + SWD_NS - 1);    // into the slot pair q-1 used
    }

    // ---- diagonal sums: D of tap r: column i-slot = lane & 15, rows o-slot = 4*(lane>>4) + reg; slot s holds long-axis
    // position s (s < 8) or s - (16 - Wt) (horizontal kernels: duplicated columns, not valid).  The 16x16 tile of a tap is written
    // SKEWED -- G[o][i] to row o-slot, column pos(i) - pos(o) + 15 -- so that a diagonal becomes a column and lane dd just adds the
    // 16 rows of column dd (unconditional reads at immediate offsets; invalid slots are never written and stay zero).  The first
    // version resolved slots and validity per read: 130 VALU per tap and wave, more than the whole streaming loop of a slice.
    if (live) {
        float* tile = (float*)(L + SWD_RING);                     // [16][32] fp32: the ring is dead
        float* res = (float*)(L + SWD_RES);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *(u32x4*)((char*)tile + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        *(u32x4*)((char*)tile + 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        const int dup = VERT ? 0 : 16 - p.Wt;
        auto pos = [&](int sl) { return sl < 8 ? sl : sl - dup; };
        auto valid = [&](int sl) { const int q = pos(sl); return sl < 8 ? sl < p.Wt : (q >= 8 && q < p.Wt); };
        const int pi = pos(i16);
        const bool vi = valid(i16);
        int wofs[4]; bool wok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int so = 4 * g4 + e;
            wok[e] = vi && valid(so);
            wofs[e] = so * 32 + (pi - pos(so) + 15);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < MF_TAPS; ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (wok[e]) tile[wofs[e]] = acc[r][e];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 31) {
                float v[16];
#pragma unroll
                for (int o = 0; o < 16; ++o) v[o] = tile[o * 32 + lane];       // 16 independent reads, added in order below
                float sum = 0.f;
#pragma unroll
                for (int o = 0; o < 16; ++o) sum += v[o];
                const int tau = lane - 15 + p.padL;
                if (tau >= 0 && tau < p.KL) res[VERT ? (tau * p.kw + r) : (r * p.kw + tau)] = sum;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        float* out = p.partial + ((size_t)slice * p.C + c) * ntap;
        for (int t = lane; t < ntap; t += 64) wgrad_store_partial(&out[t], res[t]);     // taps no diagonal reaches stay 0
    } else if (c < p.C) {                                         // empty slice: its partial must still be zero
        float* out = p.partial + ((size_t)slice * p.C + c) * ntap;
        for (int t = lane; t < ntap; t += 64) wgrad_store_partial(&out[t], 0.f);
    }
    wgrad_finish(p.partial, p.dw, p.counters + cb, (int*)lds, p.slices, p.C, c0, nch, ntap, tid, MF_THREADS);
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_swd_params(SmallWgradDmaParams& p, const ConvDims& d, bool vert, int target_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return false;
    if (d.W < 8 || d.W > 16 || (d.W & 1)) return false;             // 16-byte row halves at 4-byte aligned addresses
    if (d.H > 14 || (vert && d.W > 14)) return false;               // two zero guard rows behind every plane (and every transposed plane)
    if (d.kh * d.kw > 320) return false;
    const int cblocks = (d.C + 3) / 4;
    int slices = target_wgs / cblocks; if (slices < 1) slices = 1;
    int per = (d.N + slices - 1) / slices; per = (per + 1) & ~1;    // whole pairs
    if (per < 8) per = 8;
    if (per > ((d.N + 1) & ~1)) per = (d.N + 1) & ~1;
    p.images_per_slice = per; p.slices = (d.N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * d.H * d.W * 2);
    return (size_t)d.N * d.C * d.H * d.W * 2 < 0xffffffffull;
}

bool dwconv_mfma_small_wgrad_dma_supported(const ConvDims& d, int dy_dt, int x_dt) {
    if (dy_dt != x_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16)) return false;
    SmallWgradDmaParams p;
    return fill_swd_params(p, d, d.kh > d.kw, 768);
}

size_t dwconv_mfma_small_wgrad_dma_workspace(const ConvDims& d) {
    return align_up((size_t)((d.N + 7) / 8 + 1) * d.C * d.kh * d.kw * sizeof(float), 256);   // slices <= ceil(N / 8)
}

template <typename T, bool VERT, int NS>
static int launch_swd_ns(SmallWgradDmaParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_small_wgrad_dma_kernel<T, VERT, NS>;
    fill_swd_params(p, d, VERT, (NS == 3 ? 4 : 3) * mfma_cu_count());
    if ((size_t)p.slices * d.C * d.kh * d.kw * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    const int cblocks = (d.C + 3) / 4;
    const size_t lds = (size_t)MF_WAVES * swd_wave_bytes(NS);
    (void)slak_set_max_lds((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(cblocks * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

template <typename T, bool VERT>
static int launch_swd_t(SmallWgradDmaParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    static const int ns = [] { const char* e = slak_dev_getenv("SLAK_SWD_NS"); return e && atoi(e) == 3 ? 3 : 4; }();
    return ns == 3 ? launch_swd_ns<T, VERT, 3>(p, d, ws_bytes, st) : launch_swd_ns<T, VERT, 4>(p, d, ws_bytes, st);
}

int launch_dwconv_mfma_small_wgrad_dma(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                       const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dwconv_mfma_small_wgrad_dma_supported(d, dy_dt, x_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    const bool vert = d.kh > d.kw;
    SmallWgradDmaParams p;
    p.dy = dy; p.x = x; p.partial = (float*)ws; p.dw = dw;
    p.counters = wgrad_arrival_counters((d.C + 3) / 4);
    if (!p.counters) return SLAK_ERR_UNSUPPORTED;                 // (the caller falls through to the kernels with a reduce pass)
    if (x_dt == SLAK_BF16) return vert ? launch_swd_t<bf16_t, true>(p, d, ws_bytes, st) : launch_swd_t<bf16_t, false>(p, d, ws_bytes, st);
    return vert ? launch_swd_t<f16_t, true>(p, d, ws_bytes, st) : launch_swd_t<f16_t, false>(p, d, ws_bytes, st);
}

}  // namespace slak
