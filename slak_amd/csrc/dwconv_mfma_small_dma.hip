// slak_amd/csrc/dwconv_mfma_small_dma.hip -- MFMA depthwise-conv forward / data-grad for the 14x14-class planes (W even,
// 8 <= W <= 16, <= 14 rows along the short axis), 16-bit activations: WAVE-INDEPENDENT streaming, no workgroup barrier.
//
// The channel-blocked kernel of dwconv_mfma_small.hip is bound by its instruction count (element-wise staging through
// registers, scattered 2-byte epilogue stores, two barriers per group: ~110 instructions per plane and wave).  Here a wave
// owns one channel and a slice of the batch and does everything for its planes itself, in ~20 instructions per plane:
//   * input: ONE `buffer_load_dwordx4 ... lds` per PAIR of planes.  Lane L fetches 16 bytes of image row (L/2)%16 of plane
//     L/32 -- half 0: columns 0..7, half 1: columns W-8..W-1 (always inside the row: nothing is ever read past a plane) --
//     and the lane-linear LDS destination turns that into a row-major image of pitch 16 elements (32 B).  Lanes of rows
//     >= H stay inactive, so each plane keeps two all-zero guard rows behind it (written once): short-axis taps that leave
//     the plane read zeros, no select.
//   * v_mfma_f32_16x16x32: M = 16 long-axis outputs, N = 16 short-axis positions, K = 32 = 2 taps x 16 k-slots; the five
//     short taps are 3 MFMAs into ONE 4-register accumulator.  k-slot j holds column j (j < 8) or W-16+j (j >= 8); the
//     Toeplitz fragment carries zeros for the duplicated columns, so the plane fragment needs no mask.  Fragments are
//     16-byte reads at (lane constant) + (immediate): no address arithmetic per plane.
//   * vertical kernels (Kx5): the plane is transposed LDS->LDS with one ds_read_b64_tr_b16 + one ds_write_b64 per plane
//     (the pitch-16 image satisfies the 8-byte alignment the transposing read needs) and the same core runs on x^T with
//     the MFMA operands swapped, so in both cases a lane ends up with 4 consecutive ow of one output row;
//   * output: two `buffer_store_dword` per plane straight from the accumulator (a wave-level store covers one contiguous
//     392-byte plane); no LDS staging, no copy-out pass.
//   * the wave waits for its own DMAs with a counted `s_waitcnt vmcnt(N)`; N counts the younger DMAs AND the stores in
//     between (gfx9 returns vector-memory operations in issue order -- hipcc's own waitcnt insertion relies on it).
//   * Toeplitz fragments from LDS filter windows as in dwconv_mfma_dma.hip (per wave: its own channel).
#include "mfma_common.h"

namespace slak {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int SD_NS = 6;                // ring slots (plane pairs) per wave
constexpr int SD_SLOT = 1024;           // bytes per slot: 2 planes x 16 rows x 32 B
constexpr int SD_WZP = 16;              // zeros in front of a filter row
constexpr int SD_WLEN = 96;             // elements per padded filter row (16 + 63 + 17)
constexpr int SD_WCH = 5;               // filter elements staged per lane (upper bound)
// per-wave LDS region (bytes): [64 zero pad][ring][64 zero pad][x^T: 2 planes][64 zero row][filter windows]
constexpr int SD_RING = 64;
constexpr int SD_XT = SD_RING + SD_NS * SD_SLOT + 64;
constexpr int SD_ZROW = SD_XT + SD_SLOT;
constexpr int SD_WIN = SD_ZROW + 64;
constexpr int SD_WAVE_BYTES = SD_WIN + 2 * MF_TAPS * SD_WLEN * 2;

struct SmallDmaParams {
    const void* x; const float* w; void* y;
    int N, C, H, W, kh, kw, flip, KL, padL, Wt, Wl;
    int images_per_slice, slices;
    unsigned tensor_bytes;
};

template <typename T> __device__ __forceinline__ f32x4_t sd_mfma16(s16x8 a, s16x8 b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t sd_mfma16<bf16_t>(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t sd_mfma16<f16_t>(s16x8 a, s16x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <typename T, bool VERT>
__global__ __launch_bounds__(MF_THREADS) void dwconv_mfma_small_dma_kernel(const SmallDmaParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    const int cblocks = (p.C + 3) >> 2;
    const int cb = blockIdx.x % cblocks, slice = blockIdx.x / cblocks;
    const int c = cb * 4 + wave;
    const int n_begin = slice * p.images_per_slice;
    int n_end = n_begin + p.images_per_slice; if (n_end > p.N) n_end = p.N;
    if (c >= p.C || n_begin >= n_end) return;                     // no workgroup barrier anywhere: waves may leave
    const int npairs = (n_end - n_begin + 1) >> 1;
    char* const L = (char*)lds + wave * SD_WAVE_BYTES;           // this wave's private region
    const int HW = p.H * p.W;

    // ---- zero the region, fetch the filter ------------------------------------------------------------------
    const int ntap = p.kh * p.kw;
    float wreg[SD_WCH];
#pragma unroll
    for (int k = 0; k < SD_WCH; ++k) { const int e = lane + 64 * k; wreg[k] = e < ntap ? p.w[(size_t)c * ntap + e] : 0.f; }
    for (int o = lane * 16; o < SD_WAVE_BYTES; o += 64 * 16) *(u32x4*)(L + o) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // zeros are in place before any DMA can land on them

    // ---- DMA: lane -> (plane of the pair, image row, half of the row) ------------------------------------------
    v4i_t rs_x;
    {
        const uint64_t a = (uint64_t)p.x;
        rs_x[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs_x[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs_x[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs_x[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.tensor_bytes, 0x00020000);
    const unsigned gplane_b = (unsigned)(p.C * HW) * 2;           // HBM bytes from image n to image n+1 of this channel
    const int d_pp = lane >> 5, d_row = (lane >> 1) & 15, d_half = lane & 1;
    const unsigned d_src = (unsigned)d_pp * gplane_b + (unsigned)(d_row * p.W) * 2 + (d_half ? (unsigned)(p.W - 8) * 2 : 0u);
    const bool d_rowok = d_row < p.H;
    const unsigned lds_wave = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds) + wave * SD_WAVE_BYTES;
    const unsigned chan_b = (unsigned)c * (unsigned)HW * 2;
    auto issue_pair = [&](int q) {                                // pair q: images n_begin + 2q, + 2q + 1 -> slot q % SD_NS
        const int n0 = n_begin + 2 * q;
        const unsigned gb = (unsigned)n0 * gplane_b + chan_b;
        if (d_rowok && n0 + d_pp < n_end)
            lds_dma16(gb + d_src, rs_x, __builtin_amdgcn_readfirstlane(lds_wave + SD_RING + (unsigned)(q % SD_NS) * SD_SLOT));
    };
    for (int q = 0; q < SD_NS - 1 && q < npairs; ++q) issue_pair(q);

    // ---- filter windows (bf16/fp16, two copies one element apart) and the three Toeplitz fragments ---------------------
#pragma unroll
    for (int k = 0; k < SD_WCH; ++k) {
        const int e = lane + 64 * k;
        if (e < ntap) {
            int r = VERT ? e % MF_TAPS : e / p.kw, t = VERT ? e / MF_TAPS : e - (e / p.kw) * p.kw;      // short tap r, long tap t
            if (p.flip) { r = MF_TAPS - 1 - r; t = p.KL - 1 - t; }
            const uint16_t v = cvt_to_bits(wreg[k], (T*)nullptr);
            uint16_t* win = (uint16_t*)(L + SD_WIN);
            win[r * SD_WLEN + SD_WZP + t] = v;
            win[MF_TAPS * SD_WLEN + r * SD_WLEN + SD_WZP + t - 1] = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // lane -> (o = long-axis output position, k-group kg -> tap-in-pair rsel, half of the 16 k-slots)
    const int l15 = lane & 15, kg = lane >> 4, rsel = kg >> 1, half = kg & 1;
    s16x8 tfrag[3];
    {
        // first long-axis input position of this lane's 8 k-slots: horizontal k-slots skip the duplicated columns
        const int i0 = half ? (VERT ? 8 : p.W - 8) : 0;
        const int a = SD_WZP + i0 - l15 + p.padL;                 // window start (>= 1: padL >= 0, l15 <= 15)
        const int par = a & 1;
        const unsigned* src = (const unsigned*)(L + SD_WIN + par * MF_TAPS * SD_WLEN * 2) + ((a - par) >> 1);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int r = 2 * m + rsel;
            u32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = r < MF_TAPS ? src[(r < MF_TAPS ? r : 0) * (SD_WLEN / 2) + k] : 0u;
                if (!VERT && half && 2 * k < 16 - p.W) d[k] = 0u;   // columns already covered by the first half (W even)
            }
            tfrag[m] = __builtin_bit_cast(s16x8, d);
        }
    }

    // ---- lane constants of the loop ------------------------------------------------------------------------------
    // plane fragment of MFMA m: 16 bytes at xoff + pp*512 + m*64 (row l15 + 2m + rsel - 2 of the guarded image: the 64-byte
    // pad in front of the image and the -2 rows cancel); MFMA 2's second tap does not exist: those lanes read the zero row
    const unsigned xlane = (unsigned)(l15 * 32 + rsel * 32 + half * 16);
    const unsigned zlane = (unsigned)SD_ZROW + half * 16;
    // vertical: transposing read of plane pp = one instruction: 16-lane group g4 takes image rows 4*g4..+3
    const int g4 = lane >> 4;
    const unsigned trd = (unsigned)((4 * g4 + (l15 >> 2)) * 32 + (l15 & 3) * 8);
    const int xt_row = l15 < 8 ? l15 : l15 - (16 - p.W);          // k-slot -> image column; duplicates (< 8) are not written
    const bool twr_ok = l15 < 8 || xt_row >= 8;
    const unsigned twr = (unsigned)(SD_XT + xt_row * 32 + g4 * 8);
    // output: lane = output row l15, registers = 4 consecutive columns 4*kg..+3 -> two dword stores
    const unsigned ooff = (unsigned)(l15 * p.W + 4 * kg) * 2;
    const int nrow = VERT ? p.Wt : p.Wl, ncol = VERT ? p.Wl : p.Wt;
    const bool st0 = l15 < nrow && 4 * kg < ncol, st1 = l15 < nrow && 4 * kg + 2 < ncol;

    for (int q = 0; q < npairs; ++q) {
        // pair q has landed when at most the operations issued after its DMA are outstanding: 4 stores per completed pair of the
        // last SD_NS-2 iterations + the DMAs of the pairs behind it
        {
            const int st = (q < SD_NS - 2 ? q : SD_NS - 2) * 4;
            int dm = npairs - 1 - q; if (dm > SD_NS - 2) dm = SD_NS - 2;
            wait_vmcnt_dyn(st + dm);
        }
        __builtin_amdgcn_wave_barrier();
        const int n0 = n_begin + 2 * q;
        const unsigned slot = (unsigned)SD_RING + (unsigned)(q % SD_NS) * SD_SLOT;   // where the DMA put the pair
        unsigned xb;                                              // fragment base: 64 bytes (two rows) in front of the image
        if constexpr (VERT) {
            const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + slot + trd));
            const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(SLAK_LDS(s16x4, L + slot + 512 + trd));
            if (twr_ok) { *(s16x4*)(L + twr) = t0; *(s16x4*)(L + twr + 512) = t1; }
            xb = (unsigned)(SD_XT - 64) + xlane;
        } else {
            xb = slot - 64 + xlane;
        }
        const unsigned xz = rsel ? zlane : xb;                    // MFMA 2, second tap: zero row (offsets below stay inside it: see z2)
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const s16x8 b00 = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xb));
        const s16x8 b01 = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xb + 64));
        const s16x8 b02 = __builtin_bit_cast(s16x8, *(const u32x4*)(L + (rsel ? xz : xb + 128)));
        const s16x8 b10 = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xb + 512));
        const s16x8 b11 = __builtin_bit_cast(s16x8, *(const u32x4*)(L + xb + 512 + 64));
        const s16x8 b12 = __builtin_bit_cast(s16x8, *(const u32x4*)(L + (rsel ? xz : xb + 512 + 128)));
        if constexpr (VERT) {                                     // operands swapped: D^T = X^T-tile x T^T
            acc0 = sd_mfma16<T>(b00, tfrag[0], acc0); acc1 = sd_mfma16<T>(b10, tfrag[0], acc1);
            acc0 = sd_mfma16<T>(b01, tfrag[1], acc0); acc1 = sd_mfma16<T>(b11, tfrag[1], acc1);
            acc0 = sd_mfma16<T>(b02, tfrag[2], acc0); acc1 = sd_mfma16<T>(b12, tfrag[2], acc1);
        } else {
            acc0 = sd_mfma16<T>(tfrag[0], b00, acc0); acc1 = sd_mfma16<T>(tfrag[0], b10, acc1);
            acc0 = sd_mfma16<T>(tfrag[1], b01, acc0); acc1 = sd_mfma16<T>(tfrag[1], b11, acc1);
            acc0 = sd_mfma16<T>(tfrag[2], b02, acc0); acc1 = sd_mfma16<T>(tfrag[2], b12, acc1);
        }
        // ---- results -> HBM (always two store instructions per existing plane: the vmcnt arithmetic above counts them) ----
        const unsigned gb = (unsigned)n0 * gplane_b + chan_b;
        {
            const unsigned p0 = pack2<T>(acc0[0], acc0[1]), p1 = pack2<T>(acc0[2], acc0[3]);
            if (st0) __builtin_amdgcn_raw_buffer_store_b32(p0, rs_y, ooff, gb, 0);
            if (st1) __builtin_amdgcn_raw_buffer_store_b32(p1, rs_y, ooff + 4, gb, 0);
        }
        if (n0 + 1 < n_end) {
            const unsigned p0 = pack2<T>(acc1[0], acc1[1]), p1 = pack2<T>(acc1[2], acc1[3]);
            if (st0) __builtin_amdgcn_raw_buffer_store_b32(p0, rs_y, ooff, gb + gplane_b, 0);
            if (st1) __builtin_amdgcn_raw_buffer_store_b32(p1, rs_y, ooff + 4, gb + gplane_b, 0);
        }
        if (q + SD_NS - 1 < npairs) issue_pair(q + SD_NS - 1);      // into the slot pair q-1 used (this wave is done with it)
    }
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_small_dma_params(SmallDmaParams& p, const ConvDims& d, bool vert, int target_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.Wt = vert ? d.H : d.W; p.Wl = vert ? d.W : d.H;
    p.KL = vert ? d.kh : d.kw; p.padL = p.KL / 2;
    if ((vert ? d.kw : d.kh) != MF_TAPS) return false;
    if (d.W < 8 || d.W > 16 || (d.W & 1)) return false;             // 16-byte row halves at 4-byte aligned addresses
    if (p.Wl > 14 || p.Wt > 16 || d.H > 16) return false;           // two guard rows per 16-row block along the short axis
    if (p.KL > 63 || d.kh * d.kw > SD_WCH * 64) return false;
    const int cblocks = (d.C + 3) / 4;
    int slices = target_wgs / cblocks; if (slices < 1) slices = 1;
    int per = (d.N + slices - 1) / slices; per = (per + 1) & ~1;    // whole pairs
    if (per < 8) per = 8;                                            // at least 4 pairs per wave: amortise its prologue
    if (per > ((d.N + 1) & ~1)) per = (d.N + 1) & ~1;
    p.images_per_slice = per; p.slices = (d.N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * d.H * d.W * 2);
    return (size_t)d.N * d.C * d.H * d.W * 2 < 0xffffffffull;
}

bool dwconv_mfma_small_dma_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt) {
    if (x_dt != y_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16) || w_dt != SLAK_F32) return false;
    SmallDmaParams p;
    return fill_small_dma_params(p, d, d.kh > d.kw, 768);
}

template <typename T, bool VERT>
static int launch_small_dma_t(SmallDmaParams& p, const ConvDims& d, hipStream_t st) {
    auto k = dwconv_mfma_small_dma_kernel<T, VERT>;
    fill_small_dma_params(p, d, VERT, 3 * mfma_cu_count());      // ~12 waves per CU
    const int cblocks = (d.C + 3) / 4;
    hipLaunchKernelGGL(k, dim3((unsigned)(cblocks * p.slices)), dim3(MF_THREADS), (size_t)MF_WAVES * SD_WAVE_BYTES, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int launch_dwconv_mfma_small_dma(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                                 const ConvDims& d, bool flip_filter, hipStream_t st) {
    if (!dwconv_mfma_small_dma_supported(d, x_dt, w_dt, y_dt)) return SLAK_ERR_UNSUPPORTED;
    const bool vert = d.kh > d.kw;
    SmallDmaParams p;
    fill_small_dma_params(p, d, vert, 768);
    p.x = x; p.w = (const float*)w; p.y = y; p.flip = flip_filter ? 1 : 0;
    if (x_dt == SLAK_BF16) return vert ? launch_small_dma_t<bf16_t, true>(p, d, st) : launch_small_dma_t<bf16_t, false>(p, d, st);
    return vert ? launch_small_dma_t<f16_t, true>(p, d, st) : launch_small_dma_t<f16_t, false>(p, d, st);
}

}  // namespace slak
