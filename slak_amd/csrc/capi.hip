// slak_amd/csrc/capi.hip -- extern "C" entry points of libslak_hip.so (see include/slak_hip.h).
// Argument validation lives here; the reference's extension validates almost nothing and exit()s on
// failure (forward_fp32.cu:173-196).  Every function returns a status code instead.
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <string>

#include "slak_common.h"

namespace slak {
static std::mutex g_err_mu;
static std::string g_last_hip_error = "";
static std::atomic<int> g_conv_algo{SLAK_ALGO_AUTO};     // process-wide A/B switch (tests); read once per call
thread_local const char* g_last_kernel = "";           // slak_debug_last_kernel()
#define SLAK_RAN(name, call) (slak::g_last_kernel = (name), (call))
static bool use_small_dma() {                // SLAK_MFMA_SMALL_DMA=0 keeps the channel-blocked small-plane kernel (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_MFMA_SMALL_DMA"); return !(e && e[0] == '0'); }();
    return v;
}
static bool use_vrows_pair() {               // SLAK_VROWS_PAIR=0: the 56 x 56 class keeps separate K x 5 and 5 x 5 weight-gradient launches (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_VROWS_PAIR"); return !(e && e[0] == '0'); }();
    return v;
}
static bool use_vrows() {                    // SLAK_MFMA_VROWS=0 keeps the transposing vertical weight-gradient kernel (A/B testing)
    static const bool v = [] { const char* e = getenv("SLAK_MFMA_VROWS"); return !(e && e[0] == '0'); }();
    return v;
}
static bool use_small() {                    // SLAK_MFMA_SMALL=0 keeps the generic register-staged kernel for H,W <= 16 (A/B testing)
    static int v = -1;
    if (v < 0) { const char* e = getenv("SLAK_MFMA_SMALL"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
static constexpr int MF_TAPS_HOST = 5;        // the short side the MFMA kernels are written for
static bool use_dma() {                      // SLAK_MFMA_DMA=0 keeps the register-staged MFMA kernels (A/B testing)
    static int v = -1;
    if (v < 0) { const char* e = getenv("SLAK_MFMA_DMA"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}

void set_last_hip_error(hipError_t e) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_last_hip_error = hipGetErrorString(e);
}

static int check_conv_args(const void* a, const void* b, const void* c, int dt0, int dt1, int dt2,
                           int N, int C, int H, int W, int kh, int kw) {
    if (!a || !b || !c) return SLAK_ERR_INVALID_ARG;
    if (!dtype_ok(dt0) || !dtype_ok(dt1) || !dtype_ok(dt2)) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0) return SLAK_ERR_INVALID_ARG;
    if ((kh & 1) == 0 || (kw & 1) == 0) return SLAK_ERR_INVALID_ARG;   // output shape == input shape needs odd kernels
    if ((long long)N * C * H * W >= (1LL << 31)) return SLAK_ERR_UNSUPPORTED;
    if (kh > 127 || kw > 127) return SLAK_ERR_UNSUPPORTED;
    return SLAK_OK;
}
}  // namespace slak

using namespace slak;

namespace slak { extern unsigned long long* g_dma_dbg; }

extern "C" {

/* dev hook (not in the public header): device buffer of 4x8 u64 that workgroup 0 of the DMA conv kernel fills with per-phase cycle counts */
void slak_debug_set_phase_buffer(void* p) { slak::g_dma_dbg = (unsigned long long*)p; }
/* dev hook (tests/test_dispatch_gpu.py): the name of the kernel family the calling thread's last conv / mask entry point launched -- a support
 * predicate that regresses lands a shape on a slower kernel with parity intact; this is what the dispatch tests pin */
const char* slak_debug_last_kernel(void) { return slak::g_last_kernel; }

// An empty launch whose GRID SIZE carries an id: profiling tools cut a kernel trace at these (tools/step_breakdown.py takes exactly the
// dispatches between marker 1 and marker 2 of bench.py --markers: the K timed steps, nothing else)
namespace slak { __global__ void marker_kernel(int id) { (void)id; } }
int slak_debug_marker(int id, void* stream) {
    if (id < 0 || id > 65535) return SLAK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(slak::marker_kernel, dim3((unsigned)(id + 1)), dim3(64), 0, (hipStream_t)stream, id);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

const char* slak_status_string(int status) {
    switch (status) {
        case SLAK_OK: return "ok";
        case SLAK_ERR_INVALID_ARG: return "invalid argument";
        case SLAK_ERR_UNSUPPORTED: return "unsupported dtype/shape";
        case SLAK_ERR_WORKSPACE: return "workspace missing or too small";
        case SLAK_ERR_LAUNCH: return "HIP launch/runtime error";
        case SLAK_ERR_NO_DEVICE: return "no HIP device";
    }
    return "unknown status";
}

const char* slak_last_hip_error(void) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    static thread_local std::string copy;
    copy = g_last_hip_error;
    return copy.c_str();
}

int slak_version(void) { return SLAK_ABI_VERSION; }

int slak_device_info(int* cu_count, int* lds_bytes_per_cu, char* arch_name, size_t arch_name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SLAK_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SLAK_ERR_NO_DEVICE;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (arch_name && arch_name_len) {
        size_t i = 0;
        for (; i + 1 < arch_name_len && prop.gcnArchName[i]; ++i) arch_name[i] = prop.gcnArchName[i];
        arch_name[i] = 0;
    }
    return SLAK_OK;
}

// fp32 tensors on the bf16 matrix cores (two-term split, ~2^-16 relative per product) are OPT-IN, like torch.backends.cudnn.allow_tf32:
// the default fp32 path stays the exact VALU one.  SLAK_FP32_MFMA=1 sets the initial value.
static std::atomic<int> g_fp32_mfma{[] { const char* e = getenv("SLAK_FP32_MFMA"); return (e && e[0] == '1') ? 1 : 0; }()};
static thread_local int t_fp32_mfma = -1;                    // per-thread override: -1 follow the process-wide setting, 0 exact, 1 matrix cores
static bool f32_ok(int dt) {
    if (dt != SLAK_F32 || g_conv_algo == SLAK_ALGO_MFMA) return true;
    return t_fp32_mfma >= 0 ? t_fp32_mfma != 0 : g_fp32_mfma.load() != 0;
}
int slak_set_fp32_matrix_cores(int allow) { g_fp32_mfma = allow ? 1 : 0; return SLAK_OK; }
int slak_get_fp32_matrix_cores(void) { return g_fp32_mfma.load(); }                            // the PROCESS-WIDE switch (what a save / restore pair wants)
int slak_get_fp32_matrix_cores_effective(void) { return t_fp32_mfma >= 0 ? t_fp32_mfma : g_fp32_mfma.load(); }   // what a call on THIS thread would do
int slak_set_fp32_matrix_cores_thread(int mode, int* previous) {
    if (mode < -1 || mode > 1) return SLAK_ERR_INVALID_ARG;
    if (previous) *previous = t_fp32_mfma;
    t_fp32_mfma = mode;
    return SLAK_OK;
}

int slak_set_conv_algo(int algo) {
    if (algo != SLAK_ALGO_AUTO && algo != SLAK_ALGO_DIRECT && algo != SLAK_ALGO_MFMA) return SLAK_ERR_INVALID_ARG;
    g_conv_algo = algo;
    return SLAK_OK;
}

size_t slak_dwconv2d_workspace_bytes(int op, int N, int C, int H, int W, int kh, int kw, int dtype) {
    (void)dtype;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0) return 0;
    ConvDims d{N, C, H, W, kh, kw};
    if (op == 0 || op == 1) { size_t a = dwconv_direct_workspace(d), b = dwconv_mfma_workspace(d), c = dwconv_mfma_small_workspace(d); a = a > b ? a : b; return a > c ? a : c; }
    if (op == 2) { size_t a = dwconv_wgrad_workspace(d), b = dwconv_mfma_wgrad_workspace(d), c = dwconv_mfma_wgrad_dma_workspace(d), e = dwconv_mfma_small_wgrad_workspace(d), f = dwconv_mfma_small_wgrad_dma_workspace(d), g = dwconv_mfma_wgrad_vrows_workspace(d), h = dwconv_mfma_wide_wgrad_workspace(d), i = dwconv_mfma_wgrad_vwave_workspace(d); a = a > i ? a : i; a = a > g ? a : g; a = a > h ? a : h; a = a > b ? a : b; a = a > c ? a : c; a = a > e ? a : e; return a > f ? a : f; }
    return 0;
}

int slak_dwconv2d_forward(const void* x, int x_dtype, const void* w, int w_dtype, void* y, int y_dtype,
                          int N, int C, int H, int W, int kh, int kw,
                          void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv_args(x, w, y, x_dtype, w_dtype, y_dtype, N, C, H, W, kh, kw);
    if (rc != SLAK_OK) return rc;
    ConvDims d{N, C, H, W, kh, kw};
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_dma() && dwconv_mfma_dma_supported(d, x_dtype, w_dtype, y_dtype))
        return SLAK_RAN("dwconv_mfma_dma", launch_dwconv_mfma_dma(x, x_dtype, w, w_dtype, y, y_dtype, d, /*flip=*/false, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && dwconv_mfma_wide_supported(d, x_dtype, w_dtype, y_dtype))
        return SLAK_RAN("dwconv_mfma_wide", launch_dwconv_mfma_wide(x, x_dtype, w, w_dtype, y, y_dtype, d, /*flip=*/false, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_small_dma() && dwconv_mfma_small_dma_supported(d, x_dtype, w_dtype, y_dtype))
        return SLAK_RAN("dwconv_mfma_small_dma", launch_dwconv_mfma_small_dma(x, x_dtype, w, w_dtype, y, y_dtype, d, /*flip=*/false, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_small() && dwconv_mfma_small_supported(d, x_dtype, w_dtype, y_dtype))
        return SLAK_RAN("dwconv_mfma_small", launch_dwconv_mfma_small(x, x_dtype, w, w_dtype, y, y_dtype, d, /*flip=*/false, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && f32_ok(x_dtype) && dwconv_mfma_supported(d, x_dtype, w_dtype, y_dtype))
        return SLAK_RAN(x_dtype == SLAK_F32 ? "dwconv_mfma(f32 split)" : "dwconv_mfma",
                        launch_dwconv_mfma(x, x_dtype, w, w_dtype, y, y_dtype, d, /*flip=*/false, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo == SLAK_ALGO_MFMA) return SLAK_ERR_UNSUPPORTED;
    return SLAK_RAN("dwconv_direct", launch_dwconv_direct(x, x_dtype, w, w_dtype, y, y_dtype, d, /*flip=*/false, workspace, workspace_bytes, (hipStream_t)stream));
}

int slak_dwconv2d_backward_data(const void* dy, int dy_dtype, const void* w, int w_dtype, void* dx, int dx_dtype,
                                int N, int C, int H, int W, int kh, int kw,
                                void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv_args(dy, w, dx, dy_dtype, w_dtype, dx_dtype, N, C, H, W, kh, kw);
    if (rc != SLAK_OK) return rc;
    ConvDims d{N, C, H, W, kh, kw};
    // data-grad of a stride-1 "same" cross-correlation with odd kernels == cross-correlation of dy with
    // the filter rotated by 180 degrees (h + kh/2 - r == h - kh/2 + (kh-1-r)).
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_dma() && dwconv_mfma_dma_supported(d, dy_dtype, w_dtype, dx_dtype))
        return SLAK_RAN("dwconv_mfma_dma", launch_dwconv_mfma_dma(dy, dy_dtype, w, w_dtype, dx, dx_dtype, d, /*flip=*/true, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && dwconv_mfma_wide_supported(d, dy_dtype, w_dtype, dx_dtype))
        return SLAK_RAN("dwconv_mfma_wide", launch_dwconv_mfma_wide(dy, dy_dtype, w, w_dtype, dx, dx_dtype, d, /*flip=*/true, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_small_dma() && dwconv_mfma_small_dma_supported(d, dy_dtype, w_dtype, dx_dtype))
        return SLAK_RAN("dwconv_mfma_small_dma", launch_dwconv_mfma_small_dma(dy, dy_dtype, w, w_dtype, dx, dx_dtype, d, /*flip=*/true, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_small() && dwconv_mfma_small_supported(d, dy_dtype, w_dtype, dx_dtype))
        return SLAK_RAN("dwconv_mfma_small", launch_dwconv_mfma_small(dy, dy_dtype, w, w_dtype, dx, dx_dtype, d, /*flip=*/true, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && f32_ok(dy_dtype) && dwconv_mfma_supported(d, dy_dtype, w_dtype, dx_dtype))
        return SLAK_RAN(dy_dtype == SLAK_F32 ? "dwconv_mfma(f32 split)" : "dwconv_mfma",
                        launch_dwconv_mfma(dy, dy_dtype, w, w_dtype, dx, dx_dtype, d, /*flip=*/true, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo == SLAK_ALGO_MFMA) return SLAK_ERR_UNSUPPORTED;
    return SLAK_RAN("dwconv_direct", launch_dwconv_direct(dy, dy_dtype, w, w_dtype, dx, dx_dtype, d, /*flip=*/true, workspace, workspace_bytes, (hipStream_t)stream));
}

int slak_dwconv2d_backward_data_accumulate(const void* dy, int dy_dtype, const void* w, int w_dtype, void* dx, int dx_dtype,
                                           int N, int C, int H, int W, int kh, int kw,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv_args(dy, w, dx, dy_dtype, w_dtype, dx_dtype, N, C, H, W, kh, kw);
    if (rc != SLAK_OK) return rc;
    ConvDims d{N, C, H, W, kh, kw};
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_dma() && dwconv_mfma_dma_supported(d, dy_dtype, w_dtype, dx_dtype))
        return SLAK_RAN("dwconv_mfma_dma", launch_dwconv_mfma_dma(dy, dy_dtype, w, w_dtype, dx, dx_dtype, d, /*flip=*/true, workspace, workspace_bytes, (hipStream_t)stream, /*accumulate=*/true));
    return SLAK_ERR_UNSUPPORTED;            // the caller computes into a temporary and adds
}

int slak_dwconv2d_backward_filter(const void* dy, int dy_dtype, const void* x, int x_dtype, float* dw,
                                  int N, int C, int H, int W, int kh, int kw,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv_args(dy, x, dw, dy_dtype, x_dtype, SLAK_F32, N, C, H, W, kh, kw);
    if (rc != SLAK_OK) return rc;
    ConvDims d{N, C, H, W, kh, kw};
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_small_dma() && dwconv_mfma_small_wgrad_dma_supported(d, dy_dtype, x_dtype)) {
        const int rc = SLAK_RAN("dwconv_mfma_small_wgrad_dma", launch_dwconv_mfma_small_wgrad_dma(dy, dy_dtype, x, x_dtype, dw, d, workspace, workspace_bytes, (hipStream_t)stream));
        if (rc != SLAK_ERR_UNSUPPORTED) return rc;
    }
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_small() && dwconv_mfma_small_wgrad_supported(d, dy_dtype, x_dtype))
        return SLAK_RAN("dwconv_mfma_small_wgrad", launch_dwconv_mfma_small_wgrad(dy, dy_dtype, x, x_dtype, dw, d, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_dma() && use_vrows() && dwconv_mfma_wgrad_vrows_supported(d, dy_dtype, x_dtype)) {
        const int rc = SLAK_RAN("dwconv_mfma_wgrad_vrows", launch_dwconv_mfma_wgrad_vrows(dy, dy_dtype, x, x_dtype, dw, d, workspace, workspace_bytes, (hipStream_t)stream));
        if (rc != SLAK_ERR_UNSUPPORTED) return rc;
    }
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_dma() && dwconv_mfma_wgrad_vwave_supported(d, dy_dtype, x_dtype)) {     // vertical, planes of <= 32 rows
        const int rc = SLAK_RAN("dwconv_mfma_wgrad_vwave", launch_dwconv_mfma_wgrad_vwave(dy, dy_dtype, x, x_dtype, dw, d, workspace, workspace_bytes, (hipStream_t)stream));
        if (rc != SLAK_ERR_UNSUPPORTED) return rc;
    }
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_dma() && dwconv_mfma_wgrad_dma_supported(d, dy_dtype, x_dtype))
        return SLAK_RAN("dwconv_mfma_wgrad_dma", launch_dwconv_mfma_wgrad_dma(dy, dy_dtype, x, x_dtype, dw, d, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && dwconv_mfma_wide_wgrad_supported(d, dy_dtype, x_dtype))
        return SLAK_RAN("dwconv_mfma_wide_wgrad", launch_dwconv_mfma_wide_wgrad(dy, dy_dtype, x, x_dtype, dw, d, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo != SLAK_ALGO_DIRECT && f32_ok(x_dtype) && dwconv_mfma_wgrad_supported(d, dy_dtype, x_dtype))
        return SLAK_RAN(x_dtype == SLAK_F32 ? "dwconv_mfma_wgrad(f32 split)" : "dwconv_mfma_wgrad",
                        launch_dwconv_mfma_wgrad(dy, dy_dtype, x, x_dtype, dw, d, workspace, workspace_bytes, (hipStream_t)stream));
    if (g_conv_algo == SLAK_ALGO_MFMA) return SLAK_ERR_UNSUPPORTED;
    return SLAK_RAN("dwconv_wgrad", launch_dwconv_wgrad(dy, dy_dtype, x, x_dtype, dw, d, workspace, workspace_bytes, (hipStream_t)stream));
}


/* One launch for the three branches of a block: is there a kernel for (op, shape) that the measurements say should be used?
 * op 0 forward, 1 backward_data.  1: yes; 0: run the per-branch entry points.  Policy (MI355X, profiles/r03_*):
 *   planes up to 16 x 16 (14 x 14, 7 x 7 stages): wave-independent three-branch kernels, both ops;
 *   planes of one MFMA tile (17..32: 28 x 28, 24 x 24): four-wave-team kernels, both ops;
 *   planes of 2 x 2 tiles (33..64: 56 x 56, 48 x 48): team kernel for the data gradient (117 vs 134 us for plain + 2 accumulating launches
 *   at 128 x 96 x 56 x 56); forward stays three launches (108 vs 122 us). */
int slak_dwconv2d_tri_supported_op(int dtype, int N, int C, int H, int W, int K, int op) {
    if (op != 0 && op != 1) return 0;
    if (dwconv_mfma_small_tri_supported(N, C, H, W, K, dtype)) return 1;
    if (op == 0 && dwconv_mfma_stream_tri_supported(N, C, H, W, K, dtype)) return 1;                        // planes of 2 x 2 tiles, forward: one MFMA stream per wave
    if (dwconv_mfma_wide_tri_supported(N, C, H, W, K, dtype, op == 1)) return 1;                             // round 6: maps with 64 < H, W <= 96 (96 x 96: SLaK at 384 px)
    if (!dwconv_mfma_team_tri_supported(N, C, H, W, K, dtype, op == 1)) return 0;
    const bool one_tile = H <= 32 && W <= 32;
    static const bool all = [] { const char* e = slak_dev_getenv("SLAK_TEAM_ALL"); return e && e[0] == '1'; }();     // A/B: team kernels wherever they exist
    return (one_tile || op == 1 || all) ? 1 : 0;
}
/* both ops (the merged inference block, callers that want all or nothing) */
int slak_dwconv2d_tri_supported(int dtype, int N, int C, int H, int W, int K) {
    return (slak_dwconv2d_tri_supported_op(dtype, N, C, H, W, K, 0) && slak_dwconv2d_tri_supported_op(dtype, N, C, H, W, K, 1)) ? 1 : 0;
}

int slak_dwconv2d_tri_forward(const void* x, const float* w_v, const float* w_h, const float* w_s, void* y_v, void* y_h, void* y_s,
                              int dtype, int N, int C, int H, int W, int K, void* stream) {
    if (!x || !w_v || !w_h || !w_s || !y_v || !y_h || !y_s) return SLAK_ERR_INVALID_ARG;
    const void* in[3] = {x, x, x}; void* out[3] = {y_v, y_h, y_s}; const float* w[3] = {w_v, w_h, w_s};
    if (dwconv_mfma_small_tri_supported(N, C, H, W, K, dtype))
        return SLAK_RAN("dwconv_mfma_small_tri", launch_dwconv_mfma_small_tri(false, in, out, w, dtype, N, C, H, W, K, (hipStream_t)stream));
    if (dwconv_mfma_stream_tri_supported(N, C, H, W, K, dtype))
        return SLAK_RAN("dwconv_mfma_stream_tri", launch_dwconv_mfma_stream_tri(x, out, w, dtype, N, C, H, W, K, (hipStream_t)stream));
    if (dwconv_mfma_wide_tri_supported(N, C, H, W, K, dtype, false))
        return SLAK_RAN("dwconv_mfma_wide_tri", launch_dwconv_mfma_wide_tri(false, in, out, w, dtype, N, C, H, W, K, (hipStream_t)stream));
    return SLAK_RAN("dwconv_mfma_team_tri", launch_dwconv_mfma_team_tri(false, in, out, w, dtype, N, C, H, W, K, (hipStream_t)stream));
}

/* The forward three-branch launch that also leaves the branch BatchNorms' batch statistics: stats[rows][C][6] = per (row, channel) partial
 * sums (sum y_v, sum y_v^2, sum y_h, sum y_h^2, sum y_s, sum y_s^2) of the stored (rounded) outputs; rows = slak_dwconv2d_tri_stats_rows(...)
 * (0: this shape has no such kernel -- use slak_dwconv2d_tri_forward).  bf16 only. */
int slak_dwconv2d_tri_stats_rows(int dtype, int N, int C, int H, int W, int K) {
    if (dwconv_mfma_small_tri_supported(N, C, H, W, K, dtype)) return dwconv_mfma_small_tri_stats_rows(N, C, H, W, K, dtype);
    if (dwconv_mfma_stream_tri_supported(N, C, H, W, K, dtype)) return dwconv_mfma_stream_tri_stats_rows(N, C, H, W, K, dtype);
    return dwconv_mfma_team_tri_stats_rows(N, C, H, W, K, dtype);
}
int slak_dwconv2d_tri_forward_stats(const void* x, const float* w_v, const float* w_h, const float* w_s, void* y_v, void* y_h, void* y_s,
                                    float* stats, int dtype, int N, int C, int H, int W, int K, void* stream) {
    if (!x || !w_v || !w_h || !w_s || !y_v || !y_h || !y_s || !stats) return SLAK_ERR_INVALID_ARG;
    if (slak_dwconv2d_tri_stats_rows(dtype, N, C, H, W, K) <= 0) return SLAK_ERR_UNSUPPORTED;
    const void* in[3] = {x, x, x}; void* out[3] = {y_v, y_h, y_s}; const float* w[3] = {w_v, w_h, w_s};
    if (dwconv_mfma_small_tri_supported(N, C, H, W, K, dtype))
        return SLAK_RAN("dwconv_mfma_small_tri", launch_dwconv_mfma_small_tri(false, in, out, w, dtype, N, C, H, W, K, (hipStream_t)stream, stats));
    if (dwconv_mfma_stream_tri_supported(N, C, H, W, K, dtype))
        return SLAK_RAN("dwconv_mfma_stream_tri", launch_dwconv_mfma_stream_tri(x, out, w, dtype, N, C, H, W, K, (hipStream_t)stream, stats));
    return SLAK_RAN("dwconv_mfma_team_tri", launch_dwconv_mfma_team_tri(false, in, out, w, dtype, N, C, H, W, K, (hipStream_t)stream, stats));
}

/* slak_dwconv2d_forward that also leaves the batch statistics of the BatchNorm behind the conv (models/SLaK.py:38-47 conv -> bn):
 * stats[rows][C][2] = partial (sum y, sum y^2) of the stored outputs; *stats_rows = rows written (<= 4 N = the capacity the caller must
 * provide).  Only the LDS-DMA ring kernel (56 x 56 / 28 x 28 class) gathers them; SLAK_ERR_UNSUPPORTED otherwise (use slak_dwconv2d_forward). */
int slak_dwconv2d_forward_stats(const void* x, int x_dtype, const void* w, int w_dtype, void* y, int y_dtype, float* stats, int stats_capacity_rows,
                                int* stats_rows, int N, int C, int H, int W, int kh, int kw, void* stream) {
    int rc = check_conv_args(x, w, y, x_dtype, w_dtype, y_dtype, N, C, H, W, kh, kw);
    if (rc != SLAK_OK) return rc;
    if (!stats || !stats_rows) return SLAK_ERR_INVALID_ARG;
    ConvDims d{N, C, H, W, kh, kw};
    if (g_conv_algo == SLAK_ALGO_DIRECT || !use_dma() || x_dtype != SLAK_BF16 || !dwconv_mfma_dma_supported(d, x_dtype, w_dtype, y_dtype)) return SLAK_ERR_UNSUPPORTED;
    return SLAK_RAN("dwconv_mfma_dma", launch_dwconv_mfma_dma(x, x_dtype, w, w_dtype, y, y_dtype, d, /*flip=*/false, nullptr, 0, (hipStream_t)stream, /*accumulate=*/false,
                                                         stats, stats_capacity_rows, stats_rows));
}

int slak_dwconv2d_tri_backward_data(const void* dy_v, const void* dy_h, const void* dy_s, const float* w_v, const float* w_h,
                                    const float* w_s, void* dx, int dtype, int N, int C, int H, int W, int K, void* stream) {
    if (!dy_v || !dy_h || !dy_s || !w_v || !w_h || !w_s || !dx) return SLAK_ERR_INVALID_ARG;
    const void* in[3] = {dy_v, dy_h, dy_s}; void* out[3] = {dx, dx, dx}; const float* w[3] = {w_v, w_h, w_s};
    if (dwconv_mfma_small_tri_supported(N, C, H, W, K, dtype))
        return SLAK_RAN("dwconv_mfma_small_tri", launch_dwconv_mfma_small_tri(true, in, out, w, dtype, N, C, H, W, K, (hipStream_t)stream));
    if (dwconv_mfma_wide_tri_supported(N, C, H, W, K, dtype, true))
        return SLAK_RAN("dwconv_mfma_wide_tri", launch_dwconv_mfma_wide_tri(true, in, out, w, dtype, N, C, H, W, K, (hipStream_t)stream));
    return SLAK_RAN("dwconv_mfma_team_tri", launch_dwconv_mfma_team_tri(true, in, out, w, dtype, N, C, H, W, K, (hipStream_t)stream));
}

size_t slak_dwconv2d_tri_filter_workspace_bytes(int dtype, int N, int C, int H, int W, int K) {
    if (dwconv_mfma_small_tri_wgrad_supported(N, C, H, W, K, dtype)) return dwconv_mfma_small_tri_wgrad_workspace(N, C, K);
    if (dwconv_mfma_tri_wgrad_rows_supported(N, C, H, W, K, dtype)) return dwconv_mfma_tri_wgrad_rows_workspace(N, C, K);     // planes of 2 x 2 tiles (56 x 56 class)
    if (dwconv_mfma_tri_wgrad_wave_supported(N, C, H, W, K, dtype)) return dwconv_mfma_tri_wgrad_wave_workspace(N, C, K);     // planes of one tile (28 x 28 class)
    return 0;
}

int slak_dwconv2d_tri_backward_filter(const void* dy_v, const void* dy_h, const void* dy_s, const void* x, float* dw_v, float* dw_h,
                                      float* dw_s, int dtype, int N, int C, int H, int W, int K,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy_v || !dy_h || !dy_s || !x || !dw_v || !dw_h || !dw_s) return SLAK_ERR_INVALID_ARG;
    const void* dy[3] = {dy_v, dy_h, dy_s}; float* dw[3] = {dw_v, dw_h, dw_s};
    if (!dwconv_mfma_small_tri_wgrad_supported(N, C, H, W, K, dtype) && dwconv_mfma_tri_wgrad_rows_supported(N, C, H, W, K, dtype))
        return SLAK_RAN("dwconv_mfma_tri_wgrad_rows", launch_dwconv_mfma_tri_wgrad_rows(dy, x, dw, dtype, N, C, H, W, K, workspace, workspace_bytes, (hipStream_t)stream));
    if (!dwconv_mfma_small_tri_wgrad_supported(N, C, H, W, K, dtype) && dwconv_mfma_tri_wgrad_wave_supported(N, C, H, W, K, dtype))
        return SLAK_RAN("dwconv_mfma_tri_wgrad_wave", launch_dwconv_mfma_tri_wgrad_wave(dy, x, dw, dtype, N, C, H, W, K, workspace, workspace_bytes, (hipStream_t)stream));
    return SLAK_RAN("dwconv_mfma_small_tri_wgrad", launch_dwconv_mfma_small_tri_wgrad(dy, x, dw, dtype, N, C, H, W, K, workspace, workspace_bytes, (hipStream_t)stream));
}

/* Data gradient AND the three weight gradients of a block's branches in ONE launch (the 14 x 14 class: the dY planes are staged once for
 * both).  Same bits as slak_dwconv2d_tri_backward_data + slak_dwconv2d_tri_backward_filter; workspace: slak_dwconv2d_tri_filter_workspace_bytes. */
int slak_dwconv2d_tri_backward_supported(int dtype, int N, int C, int H, int W, int K) {
    return dwconv_mfma_small_tri_bwd_supported(N, C, H, W, K, dtype) ? 1 : 0;
}

int slak_dwconv2d_tri_backward(const void* dy_v, const void* dy_h, const void* dy_s, const void* x, const float* w_v, const float* w_h,
                               const float* w_s, void* dx, float* dw_v, float* dw_h, float* dw_s, int dtype, int N, int C, int H, int W, int K,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy_v || !dy_h || !dy_s || !x || !w_v || !w_h || !w_s || !dx || !dw_v || !dw_h || !dw_s) return SLAK_ERR_INVALID_ARG;
    const void* dy[3] = {dy_v, dy_h, dy_s}; float* dw[3] = {dw_v, dw_h, dw_s}; const float* w[3] = {w_v, w_h, w_s};
    return SLAK_RAN("dwconv_mfma_small_tri_bwd", launch_dwconv_mfma_small_tri_bwd(dy, x, w, dx, dw, dtype, N, C, H, W, K, workspace, workspace_bytes, (hipStream_t)stream));
}

/* The K x 5 and the 5 x 5 weight gradient of a block in ONE launch where the three-branch launch above does not reach (x and its five
 * column-shifted operands are shared: the 5 x 5 correlation is the K x 5 one with its own dY).  0 / SLAK_ERR_UNSUPPORTED: two calls. */
size_t slak_dwconv2d_pair_filter_workspace_bytes(int dtype, int N, int C, int H, int W, int K) {
    ConvDims d{N, C, H, W, K, MF_TAPS_HOST};
    if (dwconv_mfma_wgrad_vwave_supported(d, dtype, dtype)) return dwconv_mfma_wgrad_vwave_workspace(d);
    if (use_vrows() && use_vrows_pair() && dwconv_mfma_wgrad_vrows_supported(d, dtype, dtype)) return dwconv_mfma_wgrad_vrows_workspace(d);
    return 0;
}
int slak_dwconv2d_pair_backward_filter(const void* dy_v, const void* dy_s, const void* x, float* dw_v, float* dw_s,
                                       int dtype, int N, int C, int H, int W, int K, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy_v || !dy_s || !x || !dw_v || !dw_s) return SLAK_ERR_INVALID_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 5) return SLAK_ERR_INVALID_ARG;
    ConvDims d{N, C, H, W, K, MF_TAPS_HOST};
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_dma() && dwconv_mfma_wgrad_vwave_supported(d, dtype, dtype))
        return SLAK_RAN("dwconv_mfma_wgrad_vwave(pair)", launch_dwconv_mfma_wgrad_vwave(dy_v, dtype, x, dtype, dw_v, d, workspace, workspace_bytes, (hipStream_t)stream, dy_s, dw_s));
    if (g_conv_algo != SLAK_ALGO_DIRECT && use_dma() && use_vrows() && use_vrows_pair() && dwconv_mfma_wgrad_vrows_supported(d, dtype, dtype))
        return SLAK_RAN("dwconv_mfma_wgrad_vrows(pair)", launch_dwconv_mfma_wgrad_vrows(dy_v, dtype, x, dtype, dw_v, d, workspace, workspace_bytes, (hipStream_t)stream, dy_s, dw_s));
    return SLAK_ERR_UNSUPPORTED;
}

}  // extern "C"
