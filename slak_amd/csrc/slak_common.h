// slak_amd/csrc/slak_common.h -- shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "../../include/slak_hip.h"

namespace slak {

// The device the calling thread launches on.  Function attributes (hipFuncSetAttribute) are PER DEVICE: launchers that cache "already set"
// key the cache on (device, value), so a thread that moves to a second GPU sets the attribute there too.
static inline int slak_current_device() { int dev = 0; return hipGetDevice(&dev) == hipSuccess ? dev : 0; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) and LDS size that exceeds what the process has set before, instead of on
// every launch (the call takes the runtime's lock and ~2 us of host time).  The attribute is PROCESS-wide per (kernel, device) while
// launches come from several threads (the caller's for a forward, autograd's for a backward) and some kernels' LDS size depends on the shape: the
// value is therefore only ever RAISED -- a mutex-guarded process-wide maximum per (kernel, device) -- so a thread's cached "at least this much is
// set" can never be undercut by another thread setting a smaller size for another shape.  (Per translation unit, which is enough: a
// kernel is launched from the file that defines it.)
static inline bool slak_set_max_lds(const void* kernel, size_t lds) {
    if (lds <= 48 * 1024) return true;
    struct Slot { const void* k; int dev; size_t lds; };
    static thread_local Slot cache[64];                           // direct-mapped on the kernel's address: a collision only costs the locked lookup again
    const int dev = slak_current_device();
    Slot& s = cache[((uintptr_t)kernel >> 4) & 63];
    if (s.k == kernel && s.dev == dev && s.lds >= lds) return true;
    static std::mutex mu;
    static std::vector<Slot> set;                                 // the process-wide maxima
    std::lock_guard<std::mutex> lk(mu);
    Slot* g = nullptr;
    for (Slot& e : set) if (e.k == kernel && e.dev == dev) { g = &e; break; }
    if (!g || g->lds < lds) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (g) g->lds = lds; else { set.push_back(Slot{kernel, dev, lds}); g = &set.back(); }
    }
    s.k = kernel; s.dev = dev; s.lds = g->lds;
    return true;
}


// Tuning and experiment knobs (workgroup counts, ring depths, the timing-experiment modes of a kernel) are read from the environment only by DEV builds
// (SLAK_BUILD_DEFS=-DSLAK_DEV_KNOBS, on top of the per-kernel -DSLAK_*_DEV defines): the shipped library has the measured defaults compiled in.  What stays a
// plain getenv in the shipped library are the A/B switches that choose between two SHIPPED code paths (SLAK_MFMA_DMA, SLAK_TRI_ROWS, SLAK_LINEAR_GEMM, ...: DESIGN 4).
static inline const char* slak_dev_getenv(const char* name) {
#ifdef SLAK_DEV_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- element types -------------------------------------------------------------------------
struct bf16_t { uint16_t v; };          // storage-only bfloat16
typedef _Float16 f16_t;

template <typename T> struct dtype_of;
template <> struct dtype_of<float>  { static constexpr int value = SLAK_F32; };
template <> struct dtype_of<f16_t>  { static constexpr int value = SLAK_F16; };
template <> struct dtype_of<bf16_t> { static constexpr int value = SLAK_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v.v) << 16); }

// round-to-nearest-even, NaN stays (quiet) NaN -- same rule as torch's float->bfloat16 cast
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float  from_f32<float>(float v)  { return v; }
template <> __device__ __forceinline__ f16_t  from_f32<f16_t>(float v)  { return (f16_t)v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { bf16_t r; r.v = f32_to_bf16_bits(v); return r; }

static inline size_t dtype_size(int dt) { return dt == SLAK_F32 ? 4 : 2; }
static inline bool dtype_ok(int dt) { return dt == SLAK_F32 || dt == SLAK_F16 || dt == SLAK_BF16; }

__device__ __forceinline__ int wave_id_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define SLAK_LAUNCH_CHECK()                                                  \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) { slak::set_last_hip_error(e__); return SLAK_ERR_LAUNCH; } \
    } while (0)

void set_last_hip_error(hipError_t e);
// reduce_jobs.hip: inside slak_defer_reductions_begin / _end on this thread the fixed-order column sums are recorded (true) instead of launched
// type 0: block_tail_reduce1(part, out0, out1, split, ntiles, width); 1 / 2: linear_wgrad_reduce(_few)_kernel (width = float4 elements, ntiles = slabs)
bool reduce_defer_push(int type, const float* part, float* out0, float* out1, int split, int ntiles, int width, hipStream_t st);

// ---- launchers implemented in the .hip files (host side, C++ linkage) -----------------------
struct ConvDims { int N, C, H, W, kh, kw; };

int launch_dwconv_direct(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                         const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st);
size_t dwconv_direct_workspace(const ConvDims& d);

bool dwconv_mfma_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt);
size_t dwconv_mfma_workspace(const ConvDims& d);
int launch_dwconv_mfma(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                       const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st);

int launch_dwconv_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                        const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st);
size_t dwconv_wgrad_workspace(const ConvDims& d);
int launch_wgrad_reduce(const float* partial, float* dw, int total, int nslices, hipStream_t st);

bool dwconv_mfma_small_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt);
size_t dwconv_mfma_small_workspace(const ConvDims& d);
int launch_dwconv_mfma_small(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                             const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st);

bool dwconv_mfma_dma_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt);
int launch_dwconv_mfma_dma(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                           const ConvDims& d, bool flip_filter, void* ws, size_t ws_bytes, hipStream_t st, bool accumulate = false,
                           float* stats = nullptr, int stats_capacity_rows = 0, int* stats_rows = nullptr);

bool dwconv_mfma_wide_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt);
int launch_dwconv_mfma_wide(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                            const ConvDims& d, bool flip_filter, hipStream_t st);
bool dwconv_mfma_wide_wgrad_supported(const ConvDims& d, int dy_dt, int x_dt);
size_t dwconv_mfma_wide_wgrad_workspace(const ConvDims& d);
int launch_dwconv_mfma_wide_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                  const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st);

bool dwconv_mfma_small_dma_supported(const ConvDims& d, int x_dt, int w_dt, int y_dt);
int launch_dwconv_mfma_small_dma(const void* x, int x_dt, const void* w, int w_dt, void* y, int y_dt,
                                 const ConvDims& d, bool flip_filter, hipStream_t st);
bool dwconv_mfma_small_tri_supported(int N, int C, int H, int W, int K, int dtype);
int launch_dwconv_mfma_small_tri(bool dgrad, const void* const* in, void* const* out, const float* const* w, int dtype,
                                 int N, int C, int H, int W, int K, hipStream_t st, float* stats = nullptr);
int dwconv_mfma_small_tri_stats_rows(int N, int C, int H, int W, int K, int dtype);
bool dwconv_mfma_small_tri_wgrad_supported(int N, int C, int H, int W, int K, int dtype);
size_t dwconv_mfma_small_tri_wgrad_workspace(int N, int C, int K);
int launch_dwconv_mfma_small_tri_wgrad(const void* const* dy, const void* x, float* const* dw, int dtype,
                                       int N, int C, int H, int W, int K, void* ws, size_t ws_bytes, hipStream_t st);
bool dwconv_mfma_small_tri_bwd_supported(int N, int C, int H, int W, int K, int dtype);
int launch_dwconv_mfma_small_tri_bwd(const void* const* dy, const void* x, const float* const* w, void* dx, float* const* dw, int dtype,
                                     int N, int C, int H, int W, int K, void* ws, size_t ws_bytes, hipStream_t st);
bool dwconv_mfma_tri_wgrad_rows_supported(int N, int C, int H, int W, int K, int dtype);
size_t dwconv_mfma_tri_wgrad_rows_workspace(int N, int C, int K);
int launch_dwconv_mfma_tri_wgrad_rows(const void* const* dy, const void* x, float* const* dw, int dtype,
                                      int N, int C, int H, int W, int K, void* ws, size_t ws_bytes, hipStream_t st);
bool dwconv_mfma_tri_wgrad_wave_supported(int N, int C, int H, int W, int K, int dtype);
size_t dwconv_mfma_tri_wgrad_wave_workspace(int N, int C, int K);
int launch_dwconv_mfma_tri_wgrad_wave(const void* const* dy, const void* x, float* const* dw, int dtype,
                                      int N, int C, int H, int W, int K, void* ws, size_t ws_bytes, hipStream_t st);
bool dwconv_mfma_stream_tri_supported(int N, int C, int H, int W, int K, int dtype);
int dwconv_mfma_stream_tri_stats_rows(int N, int C, int H, int W, int K, int dtype);
int launch_dwconv_mfma_stream_tri(const void* x, void* const* out, const float* const* w, int dtype,
                                  int N, int C, int H, int W, int K, hipStream_t st, float* stats = nullptr);
// dwconv_mfma_wide_tri.hip: the three branches of a block in one launch on maps with 64 < H, W <= 96 (forward, data gradient)
bool dwconv_mfma_wide_tri_supported(int N, int C, int H, int W, int K, int dtype, bool dgrad);
int launch_dwconv_mfma_wide_tri(bool dgrad, const void* const* in, void* const* out, const float* const* w, int dtype,
                                int N, int C, int H, int W, int K, hipStream_t st);
bool dwconv_mfma_team_tri_supported(int N, int C, int H, int W, int K, int dtype, bool dgrad);
int dwconv_mfma_team_tri_stats_rows(int N, int C, int H, int W, int K, int dtype);
int launch_dwconv_mfma_team_tri(bool dgrad, const void* const* in, void* const* out, const float* const* w, int dtype,
                                int N, int C, int H, int W, int K, hipStream_t st, float* stats = nullptr);
bool dwconv_mfma_wgrad_vwave_supported(const ConvDims& d, int dy_dt, int x_dt);
size_t dwconv_mfma_wgrad_vwave_workspace(const ConvDims& d);
int launch_dwconv_mfma_wgrad_vwave(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                   const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st, const void* dy2 = nullptr, float* dw2 = nullptr);
bool dwconv_mfma_wgrad_vrows_supported(const ConvDims& d, int dy_dt, int x_dt);
size_t dwconv_mfma_wgrad_vrows_workspace(const ConvDims& d);
int launch_dwconv_mfma_wgrad_vrows(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                   const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st, const void* dy2 = nullptr, float* dw2 = nullptr);
bool dwconv_mfma_small_wgrad_dma_supported(const ConvDims& d, int dy_dt, int x_dt);
size_t dwconv_mfma_small_wgrad_dma_workspace(const ConvDims& d);
int launch_dwconv_mfma_small_wgrad_dma(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                       const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st);
bool dwconv_mfma_small_wgrad_supported(const ConvDims& d, int dy_dt, int x_dt);
size_t dwconv_mfma_small_wgrad_workspace(const ConvDims& d);
int launch_dwconv_mfma_small_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                   const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st);

bool dwconv_mfma_wgrad_dma_supported(const ConvDims& d, int dy_dt, int x_dt);
size_t dwconv_mfma_wgrad_dma_workspace(const ConvDims& d);
int launch_dwconv_mfma_wgrad_dma(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                                 const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st);

bool dwconv_mfma_wgrad_supported(const ConvDims& d, int dy_dt, int x_dt);
size_t dwconv_mfma_wgrad_workspace(const ConvDims& d);
int launch_dwconv_mfma_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw,
                             const ConvDims& d, void* ws, size_t ws_bytes, hipStream_t st);


// ---- weight-gradient slice reduction without a second launch -----------------------------------------------------------
// Every workgroup stores its partial [ntap] per channel into partial[slice][C][ntap], then ARRIVES at its channel group's
// counter; the last arriver adds the nslices partials in slice order (bitwise reproducible: the same order whichever
// workgroup happens to be last) and writes dw.  Counters come zeroed from wgrad_arrival_counters() and the last arriver
// puts its counter back to zero, so a launch leaves them as it found them.
unsigned* wgrad_arrival_counters(int ngroups);      // host; nullptr -> use launch_wgrad_reduce() after the kernel instead
#if defined(__HIPCC__)
// partial stores go through wgrad_store_partial(): written through to the device coherence point (agent-scope atomic store),
// so "arriving" needs no L2 write-back -- an agent-scope release fence costs a whole-L2 flush per workgroup on gfx942/gfx950
// (measured: 18 -> 107 us for the 14x14 weight gradient).
__device__ __forceinline__ void wgrad_store_partial(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// dw2 != nullptr: a channel's record of ntap floats holds TWO filters back to back (ntap1 taps of dw, then ntap - ntap1 of dw2).
// rec != 0: a channel's record is `rec` floats long in both `partial` and `dw` and this call sums its first ntap (a launch that produces
// some rows of a filter: the pointers arrive offset to the first of them).
__device__ __forceinline__ void wgrad_finish(const float* partial, float* dw, unsigned* counter, int* lds_flag,
                                             int nslices, int C, int c0, int nch, int ntap, int tid, int nthreads,
                                             float* dw2 = nullptr, int ntap1 = 0, int rec = 0) {
    const int R = rec ? rec : ntap;
    // Hand-off form "sc1 payload -> vmcnt(0) -> agent atomic flag; consumer: sc1 (agent-scope) loads" of MI355X_MICROARCH.md
    // (inter-workgroup visibility, valid forms): the partials were written through to the device coherence point (agent-scope atomic
    // stores), the explicit s_waitcnt below makes this wave's stores acknowledged before it can reach the barrier (inline asm: the
    // compiler may drop a fence's own wait when it believes the counter is already zero), the arrival counter is an agent-scope
    // atomic, and the last arriver reads the partials with agent-scope loads that bypass its L1.  No L2 write-back is needed (an
    // agent-scope release fence costs one per workgroup: 18 -> 107 us measured).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int last = 1;
        if (nslices > 1) {
            const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == (unsigned)(nslices - 1);
            if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *lds_flag = last;
    }
    __syncthreads();
    if (!*lds_flag) return;
    for (int t = tid; t < nch * ntap; t += nthreads) {
        float s = 0.f;
        const float* src = partial + (size_t)c0 * R + t;
        const size_t stride = (size_t)C * R;
        for (int k0 = 0; k0 < nslices; k0 += 8) {      // 8 loads in flight, added in slice order; agent-scope loads read at the coherence point
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = k0 + j < nslices ? __hip_atomic_load(src + (size_t)(k0 + j) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        if (dw2 == nullptr) dw[(size_t)c0 * R + t] = s;
        else {
            const int ch = t / ntap, e = t - ch * ntap;
            if (e < ntap1) dw[(size_t)(c0 + ch) * ntap1 + e] = s; else dw2[(size_t)(c0 + ch) * (ntap - ntap1) + (e - ntap1)] = s;
        }
    }
}
#endif

// block_tail_reg.hip: register-tile block-tail kernels (C <= 256 classes); SLAK_ERR_UNSUPPORTED = no instantiation for C
int launch_ln_nchw_to_nhwc_fwd_reg(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int N, int C, int P, float eps, hipStream_t st);
int launch_ln_nchw_to_nhwc_bwd_reg(const void* g, const void* x, const float* w, const float* mean, const float* rstd, void* dx, float* part, int* rows,
                                   int N, int C, int P, hipStream_t st);       // *rows = partial rows written ([rows][2C]); the caller reduces them
int launch_ln_patch_fwd_reg(const float* x, const float* w, const float* b, void* a, float* mean, float* rstd, int N, int C, int H, int W, float eps, hipStream_t st);
int launch_ln_patch_bwd_reg(const void* g, const float* x, const float* w, const float* mean, const float* rstd, float* dx, float* part, int* rows,
                            int N, int C, int H, int W, hipStream_t st);
int launch_scale_residual_fwd_reg(const void* sc, int sc_dtype, const void* z, const float* gamma, const float* scale, float* out, void* out16,
                                  int N, int C, int P, hipStream_t st);
int launch_scale_residual_bwd_reg(const float* dout, const void* dout16, float* dsum, const void* z, const float* gamma, const float* scale, void* dz,
                                  float* part, int* rows, int N, int C, int P, hipStream_t st);
// linear_wgrad.hip: D[N1][N2] (fp32) = X1^T X2 over M rows (the pointwise Linear weight gradient)
bool linear_wgrad_supported(int M, int N1, int N2);
size_t linear_wgrad_workspace(int M, int N1, int N2);
int launch_linear_wgrad(const void* x1, const void* x2, float* d, int M, int N1, int N2, void* ws, size_t ws_bytes, hipStream_t st);
}  // namespace slak
