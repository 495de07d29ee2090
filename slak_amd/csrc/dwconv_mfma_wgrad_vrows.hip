// slak_amd/csrc/dwconv_mfma_wgrad_vrows.hip -- MFMA weight gradient of the VERTICAL kernels (Kx5) on the 56x56 class
// (32 < H <= 64, W % 8 == 0, W <= 64) WITHOUT transposing anything.
//
//   G_r[o, i] = sum_{n,u} dY[o, u] * X[i, u + r - 2]      (o, i = image rows, u = image columns)      dw[tau, r] = sum_o G_r[o, o+tau-padL]
// The contraction index u runs along image rows, so both MFMA operands are plain 16-byte reads of 8 consecutive elements of a
// row, and the shifted operand X[i, u + s] of tap shift s = -2..2 is formed IN REGISTERS from the lane's previous / current / next
// aligned 16-byte chunks: even shifts are a selection of four of their dwords, odd shifts four v_alignbit_b32 (a 16-bit funnel shift
// of neighbouring dwords) -- 8 VALU instructions per k-step beside 5 MFMAs.  (4-byte LDS reads at a 16-byte-multiple lane stride are
// four-way bank conflicted and are avoided entirely; round 1 fetched a SECOND, one-element-shifted copy of the plane by DMA for the
// odd taps: half again the L2->LDS volume and LDS space for -4 % of the kernel time.  dwconv_mfma_wgrad_dma.hip transposes both
// planes LDS->LDS for this case and spends twice the horizontal kernel's time doing it.)
// LDS image of a plane: rows of CPR 16-byte chunks (W/8 data chunks + pad, CPR odd: conflict-free row-per-lane reads); the
// pad chunks are never written, so X[i, -2..-1] and X[i, W..W+1] read zeros and dY is zero for the k beyond W.  Each lane
// of a DMA instruction fetches ONE chunk (row = g / CPR, chunk = g % CPR of its global lane number g; pad lanes inactive).
// Everything else -- per-tap accumulators in registers over the slice, diagonal sums through a skewed per-wave tile,
// write-through partials, last-arriver reduction -- is dwconv_mfma_wgrad_dma.hip's.
#include "mfma_common.h"

namespace slak {

constexpr int VR_NB_DEFAULT = 2;        // ring depth: the next nb-1 planes stream in while the current one is consumed (env SLAK_VROWS_NB)
constexpr int VR_MAX_IPW = 8;           // DMA instructions per wave per plane (upper bound)
constexpr int VR_COPIES = 2;            // plane images per slot: dY, X (PAIR: a third, the 5 x 5 branch's dY)

struct WgradRowsParams {
    const void* dy; const void* x; float* partial; float* dw; unsigned* counters;
    const void* dy2; float* dw2;      // PAIR: the 5 x 5 branch of the same block (its dY, its dw): shares x and the five shifted operands
    int N, C, H, W, kh, kw, KL, padL;
    int CPR;               // 16-byte chunks per LDS row (odd, >= W/8 + 1)
    int ipc;               // DMA instructions per plane copy: ceil(H * CPR / 64)
    int KS;                // 16-deep k-steps per plane: ceil(W / 16)
    int planes_per_wg, slices;
    int nb;                // ring depth (slots)
    unsigned tensor_bytes;
    int dbg;               // SLAK_VROWS_DBG (timing experiments): 1 = no k-loop, 2 = no DMA, 4 = no epilogue
};
#ifdef SLAK_DEV_KNOBS                  // dev builds only: the shipped kernel compiles the timing experiments out
#define VR_DBG(bit) (p.dbg & (bit))
#else
#define VR_DBG(bit) 0
#endif

// PAIR: dw (K x 5) and dw2 (5 x 5) of one block in one launch: G_r[o, i] of the small branch is the same correlation with its own dY, so x
// is fetched and shifted once for both -- five more MFMAs per k-step behind the same operand reads, a third plane copy per slot
template <typename T, bool PAIR>
__global__ __launch_bounds__(MF_THREADS, PAIR ? 2 : 3) void dwconv_mfma_wgrad_vrows_kernel(const WgradRowsParams p) {
    constexpr int NG = MF_TAPS;
    constexpr int COPIES = PAIR ? 3 : VR_COPIES;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    char* const LB = (char*)lds;
    const int HW = p.H * p.W, ntap1 = p.kh * p.kw, ntap = ntap1 + (PAIR ? MF_TAPS * MF_TAPS : 0);
    const unsigned PB = (unsigned)p.CPR * 16;                     // row pitch (bytes)
    const unsigned copy_b = (unsigned)p.ipc * 1024;               // one plane copy (whole DMA instructions)
    const unsigned slot_b = COPIES * copy_b;                      // [dY][X]([dY of the 5 x 5 branch])
    const unsigned ring_b = 64;                                   // 64 zero bytes in front: "row -1" of the first plane
    unsigned live_b = (unsigned)p.nb * slot_b; if (live_b < MF_WAVES * 32 * 64 * 4) live_b = MF_WAVES * 32 * 64 * 4;   // >= the epilogue scratch
    float* dwl = (float*)(LB + ring_b + live_b);                  // [MF_WAVES][ntap]
    float* scratch = (float*)(LB + ring_b);                       // [MF_WAVES][32][64]: aliases the (dead) ring

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = wave_id_uniform();
    const int mt = wave & 1, nt = wave >> 1;
    const int c = blockIdx.x % p.C, slice = blockIdx.x / p.C;
    const int n_begin = slice * p.planes_per_wg;
    int n_end = n_begin + p.planes_per_wg; if (n_end > p.N) n_end = p.N;
    const int iters = n_end > n_begin ? n_end - n_begin : 0;

    for (unsigned o = tid * 16; o < ring_b + live_b + (unsigned)(MF_WAVES * ntap) * 4; o += MF_THREADS * 16) *(u32x4*)(LB + o) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    // ---- DMA plan: instruction id -> (copy t, instruction ii of the copy); ids round-robin over the waves -----------------
    v4i_t rs_dy, rs_x, rs_d2;
    {
        const uint64_t a = (uint64_t)p.dy, b = (uint64_t)p.x, a2 = (uint64_t)(PAIR ? p.dy2 : p.dy);
        rs_d2[0] = __builtin_amdgcn_readfirstlane((int)(a2 & 0xffffffffu)); rs_d2[1] = __builtin_amdgcn_readfirstlane((int)((a2 >> 32) & 0xffffu));
        rs_d2[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs_d2[3] = 0x00020000;
        rs_dy[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)); rs_dy[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs_dy[2] = __builtin_amdgcn_readfirstlane((int)p.tensor_bytes); rs_dy[3] = 0x00020000;
        rs_x[0] = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu)); rs_x[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        rs_x[2] = rs_dy[2]; rs_x[3] = 0x00020000;
    }
    const int ninstr = COPIES * p.ipc, DC = p.W / 8;
    int ins_src[VR_MAX_IPW]; unsigned ins_dst[VR_MAX_IPW]; int ins_t[VR_MAX_IPW]; bool ins_ok[VR_MAX_IPW];
#pragma unroll
    for (int k = 0; k < VR_MAX_IPW; ++k) {
        const int id = wave + k * MF_WAVES;
        const bool live = id < ninstr;
        const int t = live ? id / p.ipc : 0, ii = live ? id - t * p.ipc : 0;
        const int g = ii * 64 + lane, row = g / p.CPR, piece = g - row * p.CPR;
        ins_t[k] = live ? t : -1;
        ins_ok[k] = live && row < p.H && piece < DC;
        ins_src[k] = row * p.W * 2 + piece * 16;                               // (bytes from the plane start)
        ins_dst[k] = (unsigned)t * copy_b + (unsigned)ii * 1024;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)SLAK_LDS(uint16_t, lds);
    auto issue_plane = [&](int g) {
        if (g >= iters) return;
        const int n0 = n_begin + g;
        const unsigned gbase = (unsigned)(((size_t)n0 * p.C + c) * HW * 2);
        const unsigned slot = lds_base + ring_b + (unsigned)(g % p.nb) * slot_b;
#pragma unroll
        for (int k = 0; k < VR_MAX_IPW; ++k) {
            if (ins_t[k] >= 0) {                                      // wave-uniform
                const int off = (int)gbase + ins_src[k];
                if (ins_ok[k] && !VR_DBG(2)) {
                    if (ins_t[k] == 0) lds_dma16((unsigned)off, rs_dy, __builtin_amdgcn_readfirstlane(slot + ins_dst[k]));
                    else if (ins_t[k] == 1) lds_dma16((unsigned)off, rs_x, __builtin_amdgcn_readfirstlane(slot + ins_dst[k]));
                    else lds_dma16((unsigned)off, rs_d2, __builtin_amdgcn_readfirstlane(slot + ins_dst[k]));
                }
            }
        }
    };
    int my_instr = 0;                                             // DMA instructions this wave issues per plane (for the counted wait)
#pragma unroll
    for (int k = 0; k < VR_MAX_IPW; ++k) my_instr += (ins_t[k] >= 0 && __builtin_amdgcn_ballot_w64(ins_ok[k]) != 0) ? 1 : 0;
    for (int g = 0; g < p.nb - 1; ++g) issue_plane(g);

    f32x16 acc[NG], acc2[PAIR ? NG : 1];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[g][i] = 0.f; if constexpr (PAIR) acc2[g][i] = 0.f; }

    // ---- fragment addresses: lane -> image row (o resp. i), 8 consecutive k = columns 16*ks + 8*lhi .. +7 -------------------
    const unsigned a_off = (unsigned)(mt * 32 + l31) * PB + lhi * 16;               // dY copy
    const unsigned x_off = copy_b + (unsigned)(nt * 32 + l31) * PB + lhi * 16;      // X copy
    // Fragments are assembled from ALIGNED 16-byte reads only.  A lane's addresses are a multiple of 16 bytes apart from its neighbours'
    // (one image row per lane), so a 4-byte ds_read hits 16 of the 64 banks: four-way conflicts, 8 cycles per wave instruction against
    // 4 for a conflict-free b128 that moves four times the data (the first version fetched the +-2 taps with four b32 reads each and
    // spent 5/6 of its LDS time on them).  A tap shift of 2 elements is exactly one dword, so with the previous / current / next
    // chunk in registers every tap is a selection of four dwords:
    //   s=-2: {P.w, C.x, C.y, C.z}   s=0: C   s=+2: {C.y, C.z, C.w, N.x}        (P, C, N: the lane's previous / own / next chunk of X)
    // an odd shift is the 16-bit funnel shift of neighbouring dwords (v_alignbit_b32 hi, lo, 16 = {lo.hi16, hi.lo16}):
    //   s=-1: {P.w:C.x, C.x:C.y, C.y:C.z, C.z:C.w}                  s=+1: {C.x:C.y, C.y:C.z, C.z:C.w, C.w:N.x}
    // The pad chunk at the end of every row (zeros) is P of a row's first chunk and N of its last.
    auto rdq = [&](unsigned addr) -> u32x4 { return *(const u32x4*)(LB + addr); };
    auto rd16 = [&](unsigned addr) -> s16x8 { return __builtin_bit_cast(s16x8, rdq(addr)); };
    auto frag = [](unsigned d0, unsigned d1, unsigned d2, unsigned d3) -> s16x8 { return __builtin_bit_cast(s16x8, u32x4{d0, d1, d2, d3}); };

    for (int it = 0; it < iters; ++it) {
        int later = iters - 1 - it; if (later > p.nb - 2) later = p.nb - 2;       // planes issued after `it` that may still be in flight
        if (!VR_DBG(8)) {
        wait_vmcnt_dyn(later * my_instr);                         // my DMAs of plane `it` have landed (loads retire in order)
        wg_barrier();                                             // everyone's have; everyone is done with the slot refilled next
        }
        issue_plane(it + p.nb - 1);                               // streams in while this and the following planes are consumed
        const unsigned slot = ring_b + (unsigned)(it % p.nb) * slot_b;
        if (VR_DBG(1)) continue;
        const unsigned ab = slot + a_off, xb = slot + x_off, a2b = ab + 2 * copy_b;
        // k-loop, pinned software pipeline: the five 16-byte reads of the next k-step are issued one behind each of this k-step's MFMAs
        auto sh = [](unsigned hi, unsigned lo) -> unsigned { return __builtin_amdgcn_alignbit(hi, lo, 16); };
        auto taps = [&](s16x8 (&b)[NG], const u32x4& P, const u32x4& C, const u32x4& N) {
            b[0] = frag(P[3], C[0], C[1], C[2]);
            b[1] = frag(sh(C[0], P[3]), sh(C[1], C[0]), sh(C[2], C[1]), sh(C[3], C[2]));
            b[2] = __builtin_bit_cast(s16x8, C);
            b[3] = frag(sh(C[1], C[0]), sh(C[2], C[1]), sh(C[3], C[2]), sh(N[0], C[3]));
            b[4] = frag(C[1], C[2], C[3], N[0]);
        };
        s16x8 a = rd16(ab), b[NG], a2 = a;
        if constexpr (PAIR) a2 = rd16(a2b);
        {
            const u32x4 P = rdq(xb - 16), C = rdq(xb), N = rdq(xb + 16);
            taps(b, P, C, N);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int ks = 0; ks < p.KS; ++ks) {
            const int kn = ks + 1 < p.KS ? ks + 1 : ks;            // last k-step: re-read (discarded)
            const unsigned xo = xb + (unsigned)kn * 32;
            const s16x8 an = rd16(ab + (unsigned)kn * 32);
            s16x8 a2n = an;
            if constexpr (PAIR) a2n = rd16(a2b + (unsigned)kn * 32);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = mfma32<T>(a, b[0], acc[0]);
            if constexpr (PAIR) acc2[0] = mfma32<T>(a2, b[0], acc2[0]);
            const u32x4 Pn = rdq(xo - 16);
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = mfma32<T>(a, b[1], acc[1]);
            if constexpr (PAIR) acc2[1] = mfma32<T>(a2, b[1], acc2[1]);
            const u32x4 Cn = rdq(xo);
            __builtin_amdgcn_sched_barrier(0);
            acc[2] = mfma32<T>(a, b[2], acc[2]);
            if constexpr (PAIR) acc2[2] = mfma32<T>(a2, b[2], acc2[2]);
            const u32x4 Nn = rdq(xo + 16);
            __builtin_amdgcn_sched_barrier(0);
            acc[3] = mfma32<T>(a, b[3], acc[3]);
            if constexpr (PAIR) acc2[3] = mfma32<T>(a2, b[3], acc2[3]);
            __builtin_amdgcn_sched_barrier(0);
            acc[4] = mfma32<T>(a, b[4], acc[4]);
            if constexpr (PAIR) acc2[4] = mfma32<T>(a2, b[4], acc2[4]);
            taps(b, Pn, Cn, Nn);
            a = an; a2 = a2n;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vmcnt<0>();
    __syncthreads();                                              // the ring is dead: its space becomes the diagonal-sum scratch

    if (VR_DBG(4)) return;
    // ---- diagonal sums through a skewed per-wave tile (see dwconv_mfma_wgrad_dma.hip) -----------------------------------------
    float* mine = dwl + wave * ntap;
    float* tile = scratch + wave * (32 * 64);
    for (int i = lane; i < 32 * 64 / 4; i += 64) ((u32x4*)tile)[i] = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool col_ok = nt * 32 + l31 < p.H;
    int o_max = p.H - mt * 32; if (o_max > 32) o_max = 32;
    float* wr = tile + (4 * lhi) * 64 + (l31 - 4 * lhi + 31);
    // (written out twice instead of a lambda over the accumulator array: taking its address cost 30 VGPRs and spills in the k-loop)
#define SLAK_VR_DIAG(AC, KL_, PADL_, OUT_)                                                                                             \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                                                   \
        if (col_ok) {                                                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                             \
                if ((r & 3) + 8 * (r >> 2) + 4 * lhi < o_max) wr[((r & 3) + 8 * (r >> 2)) * 63] = AC[g][r];                            \
        }                                                                                                                              \
        __builtin_amdgcn_wave_barrier();                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                             \
        if (lane < 63) {                                                                                                               \
            float part[4] = {0.f, 0.f, 0.f, 0.f};                                                                                      \
            _Pragma("unroll") for (int o = 0; o < 32; ++o) part[o & 3] += tile[o * 64 + lane];                                         \
            const int tau = lane - 31 + (nt - mt) * 32 + (PADL_);                                                                      \
            if (tau >= 0 && tau < (KL_)) (OUT_)[tau * MF_TAPS + g] = (part[0] + part[1]) + (part[2] + part[3]);                        \
        }                                                                                                                              \
        __builtin_amdgcn_wave_barrier();                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                             \
    }
    SLAK_VR_DIAG(acc, p.KL, p.padL, mine)
    if constexpr (PAIR) { SLAK_VR_DIAG(acc2, MF_TAPS, MF_TAPS / 2, mine + ntap1) }      // (every tile entry is rewritten by each tap: no re-zeroing)
#undef SLAK_VR_DIAG
    __syncthreads();
    for (int t = tid; t < ntap; t += MF_THREADS) {
        float s = dwl[t];
#pragma unroll
        for (int w = 1; w < MF_WAVES; ++w) s += dwl[w * ntap + t];
        wgrad_store_partial(&p.partial[((size_t)slice * p.C + c) * ntap + t], s);
    }
    wgrad_finish(p.partial, p.dw, p.counters + c, (int*)lds, p.slices, p.C, c, 1, ntap, tid, MF_THREADS, PAIR ? p.dw2 : nullptr, ntap1);
}

// ------------------------------------------------------------------------------------------------------------
static bool fill_vrows_params(WgradRowsParams& p, const ConvDims& d, int resident_wgs) {
    p.N = d.N; p.C = d.C; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw;
    p.KL = d.kh; p.padL = p.KL / 2;
    if (d.kw != MF_TAPS || d.kh <= d.kw) return false;
    if (d.H <= 32 || d.H > 64 || d.W % 8 || d.W < 16 || d.W > 64) return false;
    p.CPR = d.W / 8 + 1; if (!(p.CPR & 1)) ++p.CPR;                   // odd number of 16-byte chunks per row, >= 1 pad chunk
    p.ipc = (d.H * p.CPR + 63) / 64;
    p.KS = (d.W + 15) / 16;
    if ((3 * p.ipc + MF_WAVES - 1) / MF_WAVES > VR_MAX_IPW) return false;      // (three copies in the PAIR launch)
    int slices = resident_wgs / d.C; if (slices < 1) slices = 1;
    if (slices > d.N) slices = d.N;
    const int per = (d.N + slices - 1) / slices;
    p.planes_per_wg = per; p.slices = (d.N + per - 1) / per;
    p.tensor_bytes = (unsigned)((size_t)d.N * d.C * d.H * d.W * 2);
    static const int nb_env = [] { const char* e = slak_dev_getenv("SLAK_VROWS_NB"); const int v = e ? atoi(e) : VR_NB_DEFAULT; return v < 2 ? 2 : (v > 4 ? 4 : v); }();
    p.nb = nb_env;
    { static const int dbg = [] { const char* e = slak_dev_getenv("SLAK_VROWS_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }
    return (size_t)d.N * d.C * d.H * d.W * 2 < 0x7fffffffull;         // (signed source offsets in the DMA plan)
}

static size_t vrows_lds_bytes(const WgradRowsParams& p, int copies = VR_COPIES) {
    size_t live = (size_t)p.nb * copies * p.ipc * 1024, scratch = (size_t)MF_WAVES * 32 * 64 * 4;
    if (live < scratch) live = scratch;
    return 64 + live + (size_t)MF_WAVES * (p.kh * p.kw + (copies > VR_COPIES ? MF_TAPS * MF_TAPS : 0)) * 4 + 32;
}

bool dwconv_mfma_wgrad_vrows_supported(const ConvDims& d, int dy_dt, int x_dt) {
    if (dy_dt != x_dt || (x_dt != SLAK_BF16 && x_dt != SLAK_F16)) return false;
    WgradRowsParams p;
    return fill_vrows_params(p, d, 512) && vrows_lds_bytes(p) <= 100 * 1024;
}

size_t dwconv_mfma_wgrad_vrows_workspace(const ConvDims& d) {
    return align_up((size_t)(d.N < 2048 ? d.N : 2048) * d.C * (d.kh * d.kw + MF_TAPS * MF_TAPS) * sizeof(float), 256);      // (PAIR records)
}

template <typename T, bool PAIR>
static int launch_vrows_t(WgradRowsParams& p, const ConvDims& d, size_t ws_bytes, hipStream_t st) {
    auto k = dwconv_mfma_wgrad_vrows_kernel<T, PAIR>;
    constexpr int copies = PAIR ? 3 : VR_COPIES;
    fill_vrows_params(p, d, 512);
    const size_t lds = vrows_lds_bytes(p, copies);
    static int resident = 0;
    (void)slak_set_max_lds((const void*)k, lds);
    if (resident == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, MF_THREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        { const char* e = slak_dev_getenv("SLAK_VROWS_WGS"); if (e && atoi(e) > 0 && atoi(e) < per_cu) per_cu = atoi(e); }      // (dev: fewer workgroups per CU)
        if ((size_t)per_cu * (lds + 512) > 160 * 1024) --per_cu;       // (the query ignores the LDS allocation granule)
        if (per_cu < 1) per_cu = 1;
        resident = per_cu * mfma_cu_count();
    }
    fill_vrows_params(p, d, resident);
    if ((size_t)p.slices * d.C * (d.kh * d.kw + (PAIR ? MF_TAPS * MF_TAPS : 0)) * sizeof(float) > ws_bytes) return SLAK_ERR_WORKSPACE;
    hipLaunchKernelGGL(k, dim3((unsigned)(p.C * p.slices)), dim3(MF_THREADS), lds, st, p);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

// dy2 / dw2 != nullptr: the 5 x 5 branch's weight gradient in the same launch
int launch_dwconv_mfma_wgrad_vrows(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                   const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st, const void* dy2, float* dw2) {
    if (!dwconv_mfma_wgrad_vrows_supported(d, dy_dt, x_dt)) return SLAK_ERR_UNSUPPORTED;
    if (ws == nullptr) return SLAK_ERR_WORKSPACE;
    WgradRowsParams p;
    p.dy = dy; p.x = x; p.partial = (float*)ws; p.dw = dw; p.dy2 = dy2; p.dw2 = dw2;
    p.counters = wgrad_arrival_counters(d.C);
    if (!p.counters) return SLAK_ERR_UNSUPPORTED;
    if (dy2 && dw2) return x_dt == SLAK_BF16 ? launch_vrows_t<bf16_t, true>(p, d, ws_bytes, st) : launch_vrows_t<f16_t, true>(p, d, ws_bytes, st);
    return x_dt == SLAK_BF16 ? launch_vrows_t<bf16_t, false>(p, d, ws_bytes, st) : launch_vrows_t<f16_t, false>(p, d, ws_bytes, st);
}

}  // namespace slak
