/*
 * include/slak_hip.h -- C ABI of libslak_hip.so, the MI355X (gfx950) drop-in for the two native
 * boundaries of VITA-Group/SLaK's training hot path.  Plain pointers and sizes only: no torch
 * types, no C++ in the signatures.  All pointers are DEVICE pointers unless a name ends in _host.
 * Every entry point returns a slak_status_t (0 == ok) and never calls exit()
 * (the reference exits the process on failure: forward_fp32.cu:173-192).
 * Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream).  The
 * reference launches on the null stream (cutlass/include/cutlass/convolution/device/convolution.h:243);
 * callers here pass PyTorch's current stream.
 *
 * Boundary 1 -- depthwise conv, replaces pybind module `_depthwise_conv2d_implicit_gemm_C`
 *   (cutlass/examples/19_large_depthwise_conv2d_torch_extension/frontend.h:3-10, frontend.cpp:3-16):
 *     forward_fp32/fp16(x, w)            -> slak_dwconv2d_forward
 *     backward_data_fp32/fp16(dy, w)     -> slak_dwconv2d_backward_data
 *     backward_filter_fp32/fp16(dy,x,w)  -> slak_dwconv2d_backward_filter
 *   Semantics fixed by the reference kernels (forward_fp32.cu:135-144, :227, :235): NCHW contiguous,
 *   weight (C,1,kh,kw), stride 1, dilation 1, padding (kh/2, kw/2), groups == C, cross-correlation,
 *   output shape == input shape, so kh and kw must be odd (the reference silently mis-sizes even
 *   kernels; here that is SLAK_ERR_INVALID_ARG).  bf16 is added (the reference raises TypeError:
 *   depthwise_conv2d_implicit_gemm.py:63).  The filter gradient is always fp32
 *   (backward_filter_fp16.cu:187).
 *
 * Boundary 2 -- dynamic-sparsity step, replaces the per-tensor torch loops of
 *   sparse_core.Masking.apply_mask (sparse_core.py:316-333), .truncate_weights (:335-357),
 *   funcs.magnitude_prune (funcs.py:107-114) and funcs.gradient_growth (funcs.py:196-205).
 *   Masks stay fp32 0/1 tensors owned by the caller (Masking.masks, read by model_sema.py:83-89).
 *   Ties at the cut are broken by LOWEST FLAT INDEX (== torch.sort(stable=True)); the reference's
 *   unstable torch.sort is arbitrary there.
 */
#ifndef SLAK_HIP_H
#define SLAK_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SLAK_OK = 0,
    SLAK_ERR_INVALID_ARG = 1,   /* null pointer, non-positive dim, even kernel, bad dtype enum */
    SLAK_ERR_UNSUPPORTED = 2,   /* dtype combination / size outside what the kernels cover     */
    SLAK_ERR_WORKSPACE = 3,     /* workspace missing or smaller than *_workspace_bytes()       */
    SLAK_ERR_LAUNCH = 4,        /* HIP launch/runtime error; see slak_last_hip_error()         */
    SLAK_ERR_NO_DEVICE = 5
} slak_status_t;

typedef enum { SLAK_F32 = 0, SLAK_F16 = 1, SLAK_BF16 = 2, SLAK_I64 = 3 /* slak_ema_update only */ } slak_dtype_t;

/* conv algorithm selector: AUTO picks per dtype/shape; DIRECT = fp32-exact VALU kernels (any dtype);
 * MFMA = banded-Toeplitz matrix-core kernels (f16/bf16 inputs; fp32 inputs through the two-term split below).  AUTO runs 16-bit tensors on
 * the matrix cores for every kernel with a 5-tap side and, with the rows taken five at a time, for every odd kh <= kw <= 63 kernel on maps
 * up to 64 wide (the square 3 .. 31 kernels of test_correctness.py:16-20, the 51 x 51 of --Decom False); everything else (maps beyond 128,
 * kernels with more rows than columns and no 5-tap side, fp32 unless allowed below) runs the exact VALU kernels. */
typedef enum { SLAK_ALGO_AUTO = 0, SLAK_ALGO_DIRECT = 1, SLAK_ALGO_MFMA = 2 } slak_algo_t;

const char* slak_status_string(int status);
const char* slak_last_hip_error(void);      /* text of the last HIP error seen by this library */
/* ABI version: bumped whenever an entry point's argument list or meaning changes (round 5: 5).  A host module compiled against this header
 * (slak_amd/pybind, *.cpp) records the value it saw and refuses to load on a library that reports another one: a stale module would call raw-pointer
 * entry points with a changed argument list -- silent corruption, not an error. */
#define SLAK_ABI_VERSION 8
int slak_version(void);                     /* == SLAK_ABI_VERSION of the header the library was built from */
int slak_device_info(int* cu_count, int* lds_bytes_per_cu, char* arch_name, size_t arch_name_len);
int slak_set_conv_algo(int algo);           /* process-wide override of the AUTO choice */
/* fp32 tensors on the bf16 matrix cores: x = bf16(x) + bf16(x - bf16(x)) (and the same for the filter / dy), three MFMAs per product,
 * fp32 accumulation: 16 significand bits per operand, ~2e-5 of sum|x||w| per output -- 50x inside the 1e-3 the reference's own test allows
 * (test_correctness.py), not bit-compatible with the exact path.  OFF by default (the fp32 entry points then run the exact VALU kernels), the
 * way torch.backends.cudnn.allow_tf32 gates TF32; SLAK_FP32_MFMA=1 in the environment sets the initial value.  Covers kernels with a 5-tap
 * side on maps up to 64 along the long axis (the SLaK stages at 224 px); everything else stays exact. */
int slak_set_fp32_matrix_cores(int allow);
int slak_get_fp32_matrix_cores(void);            /* the process-wide switch (never the per-thread override: safe for save / restore) */
int slak_get_fp32_matrix_cores_effective(void);  /* what a call on THIS thread would do: the per-thread override if one is set, else the switch */
/* Per-thread override for the calls that follow on this thread: -1 follow the process-wide setting, 0 exact, 1 matrix cores; the mode it
 * replaces is stored to *previous (may be NULL).  Nothing in the library or the Python op module sets it by default: the op module does so
 * only where the user opted in (DepthWiseConv2dImplicitGEMM.fp32_matrix_cores_under_autocast = True / SLAK_FP32_AUTOCAST_SPLIT=1: fp32
 * activations that reach the op under torch.autocast -- the reference's default AMP flow, depthwise_conv2d_implicit_gemm.py:16). */
int slak_set_fp32_matrix_cores_thread(int mode, int* previous);

/* ---------------------------------------------------------------- boundary 1: depthwise conv */

/* Bytes of scratch each op needs for these dims (0 is possible).  `op`: 0 fwd, 1 bwd-data, 2 bwd-filter. */
size_t slak_dwconv2d_workspace_bytes(int op, int N, int C, int H, int W, int kh, int kw, int dtype);

/* y[n,c,p,q] = sum_{r,s} x[n,c,p-kh/2+r,q-kw/2+s] * w[c,0,r,s]      (forward_fp32.cu:199-263) */
int slak_dwconv2d_forward(const void* x, int x_dtype, const void* w, int w_dtype, void* y, int y_dtype,
                          int N, int C, int H, int W, int kh, int kw,
                          void* workspace, size_t workspace_bytes, void* stream);

/* dx[n,c,h,w] = sum_{r,s} dy[n,c,h+kh/2-r,w+kw/2-s] * w[c,0,r,s]    (backward_data_fp32.cu:199-263) */
int slak_dwconv2d_backward_data(const void* dy, int dy_dtype, const void* w, int w_dtype, void* dx, int dx_dtype,
                                int N, int C, int H, int W, int kh, int kw,
                                void* workspace, size_t workspace_bytes, void* stream);

/* dx += the data gradient (same arguments as slak_dwconv2d_backward_data; dx is read and written).  The three branches of a SLaK
 * block (models/SLaK.py:92-100) share their input, so autograd adds the three branch gradients with two elementwise passes
 * (read 2 + write 1 each); accumulating inside the second and third branch's kernel costs one extra read each instead.  The sum is
 * rounded to the tensor dtype after every accumulation, exactly as a bf16 / fp16 tensor add does.  Kernels that cannot accumulate
 * return SLAK_ERR_UNSUPPORTED (nothing has been launched): compute into a temporary and add. */
int slak_dwconv2d_backward_data_accumulate(const void* dy, int dy_dtype, const void* w, int w_dtype, void* dx, int dx_dtype,
                                           int N, int C, int H, int W, int kh, int kw,
                                           void* workspace, size_t workspace_bytes, void* stream);

/* dw[c,0,r,s] = sum_{n,p,q} dy[n,c,p,q] * x[n,c,p-kh/2+r,q-kw/2+s], fp32 out, deterministic
 * (no atomics; the reference atomically adds: dwconv2d_direct_epilogue_simt.h:180)
 *                                                                   (backward_filter_fp32.cu:199-263) */
int slak_dwconv2d_backward_filter(const void* dy, int dy_dtype, const void* x, int x_dtype, float* dw,
                                  int N, int C, int H, int W, int kh, int kw,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- boundary 2: mask step       */

typedef struct {
    float* weight;        /* fp32 parameter, updated in place                                   */
    float* mask;          /* fp32 0/1 mask, same numel                                          */
    const float* grad;    /* fp32 gradient (only read by slak_mask_prune_and_grow; may be NULL) */
    float* momentum;      /* optional SGD momentum buffer masked too (sparse_core.py:327-328)   */
    long long numel;
} slak_mask_segment_t;

typedef struct slak_mask_plan slak_mask_plan_t;   /* opaque: device tables + select scratch      */

/* Build a plan over `nseg` masked tensors (descriptor array in HOST memory, device pointers inside). */
int slak_mask_plan_create(const slak_mask_segment_t* segs_host, int nseg, slak_mask_plan_t** plan_out);
/* Re-point grad / momentum (they are re-allocated by optimizers); arrays of nseg device pointers, in HOST memory. */
int slak_mask_plan_set_grads(slak_mask_plan_t* plan, const void* const* grads_host, void* stream);
int slak_mask_plan_set_momentum(slak_mask_plan_t* plan, void* const* momentum_host, void* stream);
int slak_mask_plan_destroy(slak_mask_plan_t* plan);

/* w *= mask (and momentum *= mask) for every segment, one launch            (sparse_core.py:316-333) */
int slak_mask_apply(slak_mask_plan_t* plan, void* stream);

/* One truncate_weights(): per segment, magnitude-prune then gradient-regrow then apply, all on device.
 * `prune_rate` is the host scheduler's fp64 value; k = ceil(zeros + ceil(rate*nonzeros)) is evaluated
 * on device in fp64 exactly as Python does (funcs.py:107-109).       (sparse_core.py:335-357)       */
int slak_mask_prune_and_grow(slak_mask_plan_t* plan, double prune_rate, void* stream);

/* The prune half alone (magnitude prune, sparse_core.py:337-347): masks hold the pruned masks afterwards, weights are untouched, the
 * statistics (nonzeros before, zeros before, removed, nonzeros after prune) are ready for slak_mask_read_stats.  For the growth
 * modes whose random numbers come from the host generator (funcs.random_growth, funcs.py:170-175: `torch.rand(shape).cuda() < p`):
 * the caller grows on the pruned masks and finishes with slak_mask_apply. */
int slak_mask_prune(slak_mask_plan_t* plan, double prune_rate, void* stream);

/* Copy per-segment statistics of the last prune_and_grow to the host (synchronises `stream`):
 * out_host[4*i + {0,1,2,3}] = nonzeros before, zeros before, removed by prune, nonzeros after. */
int slak_mask_read_stats(slak_mask_plan_t* plan, double* out_host, void* stream);

/* 64-bit order-independent checksum of all masks (for cross-rank agreement checks); synchronises. */
int slak_mask_checksum(slak_mask_plan_t* plan, unsigned long long* out_host, void* stream);

/* ---- next row: the three branches of one decomposed large-kernel block in ONE launch ----------------------------
 * ReparamLargeKernelConv.forward runs LoRA1 (K x 5), LoRA2 (5 x K) and small_conv (5 x 5) on the same input (models/SLaK.py:82-100).
 * forward: x is read once for the three outputs; backward_data: the three partial input gradients are summed inside the kernel
 * (autograd would add them with two elementwise passes).  16-bit tensors (dtype = SLAK_BF16 / SLAK_F16), fp32 filters
 * (C,1,K,5), (C,1,5,K), (C,1,5,5).  slak_dwconv2d_tri_supported_op(.., op) -- op 0 forward, 1 backward_data -- returns 1 where a
 * one-launch kernel exists AND measured faster than the per-branch entry points on MI355X: planes up to 16 x 16 (wave-independent
 * kernels; the three gradient contributions are summed in the fp32 accumulator and rounded once) and planes of one MFMA tile (17..32:
 * four-wave-team kernels, same rounding) for both ops; planes of 2 x 2 tiles (33..64, H % 4 == 0, W % 8 == 0) for the data gradient only
 * (two rounded partial planes -- vertical [+ small], horizontal [+ small] -- added like a tensor add: one rounding fewer than autograd's
 * two adds).  0: the caller issues the three slak_dwconv2d_* calls.  slak_dwconv2d_tri_supported = both ops.  The launches themselves
 * run wherever a kernel exists (SLAK_ERR_UNSUPPORTED otherwise).
 * slak_debug_last_kernel(): name of the kernel family the calling thread's last conv entry point launched (dispatch tests). */
int slak_dwconv2d_tri_supported_op(int dtype, int N, int C, int H, int W, int K, int op);
const char* slak_debug_last_kernel(void);
/* An empty kernel `slak::marker_kernel` with grid = id + 1 workgroups on `stream` (0 <= id <= 65535): a cut mark in a kernel trace
 * (bench.py --markers brackets its K timed steps with ids 1 and 2; tools/step_breakdown.py reads them). */
int slak_debug_marker(int id, void* stream);
int slak_dwconv2d_tri_supported(int dtype, int N, int C, int H, int W, int K);
int slak_dwconv2d_tri_forward(const void* x, const float* w_v, const float* w_h, const float* w_s, void* y_v, void* y_h, void* y_s,
                              int dtype, int N, int C, int H, int W, int K, void* stream);
/* slak_dwconv2d_forward that also leaves the batch statistics of the BatchNorm behind the conv (models/SLaK.py:38-47: conv -> bn):
 * stats[rows][C][2] = partial (sum y, sum y^2) of the stored outputs per (row, channel); *stats_rows = rows written; the caller provides room
 * for 4 N rows.  Only the LDS-DMA ring kernel (56 x 56 / 28 x 28 class, bf16) gathers them: SLAK_ERR_UNSUPPORTED otherwise. */
int slak_dwconv2d_forward_stats(const void* x, int x_dtype, const void* w, int w_dtype, void* y, int y_dtype, float* stats, int stats_capacity_rows,
                                int* stats_rows, int N, int C, int H, int W, int kh, int kw, void* stream);
/* slak_dwconv2d_tri_forward that also leaves the batch statistics of the three branch BatchNorms (models/SLaK.py:92-95: conv -> bn per
 * branch): stats[rows][C][6] = partial sums (sum y_v, sum y_v^2, sum y_h, sum y_h^2, sum y_s, sum y_s^2) of the STORED (rounded) outputs per
 * (row, channel); rows = slak_dwconv2d_tri_stats_rows(...) (0: no such kernel for the shape; bf16 only).  slak_bn3_forward_local takes them in
 * place of its own pass over the three tensors. */
int slak_dwconv2d_tri_stats_rows(int dtype, int N, int C, int H, int W, int K);
int slak_dwconv2d_tri_forward_stats(const void* x, const float* w_v, const float* w_h, const float* w_s, void* y_v, void* y_h, void* y_s,
                                    float* stats, int dtype, int N, int C, int H, int W, int K, void* stream);
int slak_dwconv2d_tri_backward_data(const void* dy_v, const void* dy_h, const void* dy_s, const float* w_v, const float* w_h,
                                    const float* w_s, void* dx, int dtype, int N, int C, int H, int W, int K, void* stream);

/* The three weight gradients of a block in one launch (x fetched from HBM once for the three correlations): dw_v (C,1,K,5), dw_h (C,1,5,K),
 * dw_s (C,1,5,5), fp32 (backward_filter_fp16.cu:187), bitwise reproducible (fixed-order reductions, no atomics on data).  Covered: H <= 14
 * and W even 8..14 or 4..7 (the 14x14 and 7x7 stages); round 4: planes of one MFMA tile (15 <= H <= 32, W even 16..32: the 28x28 stage,
 * C <= 4 x CUs) and planes of 2 x 2 tiles (32 < H, W <= 64, W % 8 == 0: the 56x56 stage).  slak_dwconv2d_tri_filter_workspace_bytes
 * returns 0 for anything else and the call SLAK_ERR_UNSUPPORTED (slak_dwconv2d_pair_backward_filter + one, or three,
 * slak_dwconv2d_backward_filter calls instead; a caller must be ready for SLAK_ERR_UNSUPPORTED from the call even after a non-zero
 * workspace answer: the device's CU count, unknown to the query, decides the work split). */
size_t slak_dwconv2d_tri_filter_workspace_bytes(int dtype, int N, int C, int H, int W, int K);
int slak_dwconv2d_tri_backward_filter(const void* dy_v, const void* dy_h, const void* dy_s, const void* x, float* dw_v, float* dw_h,
                                      float* dw_s, int dtype, int N, int C, int H, int W, int K,
                                      void* workspace, size_t workspace_bytes, void* stream);

/* The whole backward of a block's three branch convs in ONE launch -- dx (what slak_dwconv2d_tri_backward_data writes) and dw_v, dw_h, dw_s
 * (what slak_dwconv2d_tri_backward_filter writes), bit for bit -- where the planes are small enough that a launch's fixed cost and the second
 * staging of the three dY tensors dominate (round 4: H <= 14, W even 8..14 -- the 14 x 14 stage: backward_data_fp16.cu:181-243 and
 * backward_filter_fp16.cu:181-243 per branch in the reference, six launches).  slak_dwconv2d_tri_backward_supported returns 1 where the
 * launch exists; workspace: slak_dwconv2d_tri_filter_workspace_bytes.  Anything else: SLAK_ERR_UNSUPPORTED (the two calls above). */
int slak_dwconv2d_tri_backward_supported(int dtype, int N, int C, int H, int W, int K);
int slak_dwconv2d_tri_backward(const void* dy_v, const void* dy_h, const void* dy_s, const void* x, const float* w_v, const float* w_h,
                               const float* w_s, void* dx, float* dw_v, float* dw_h, float* dw_s, int dtype, int N, int C, int H, int W, int K,
                               void* workspace, size_t workspace_bytes, void* stream);

/* The K x 5 and the 5 x 5 weight gradient of a block in one launch, for the planes the three-branch launch does not cover (round 2:
 * 15 <= H <= 32, W even 16..32 -- the 28 x 28 stage): the 5 x 5 correlation is the K x 5 one with its own dY, so x is fetched and
 * column-shifted once for both.  dw_v (C,1,K,5), dw_s (C,1,5,5), fp32, bitwise reproducible.  slak_dwconv2d_pair_filter_workspace_bytes
 * returns 0 and the call SLAK_ERR_UNSUPPORTED elsewhere (two slak_dwconv2d_backward_filter calls instead). */
size_t slak_dwconv2d_pair_filter_workspace_bytes(int dtype, int N, int C, int H, int W, int K);
int slak_dwconv2d_pair_backward_filter(const void* dy_v, const void* dy_s, const void* x, float* dw_v, float* dw_s,
                                       int dtype, int N, int C, int H, int W, int K, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- next row: block tail glue
 * The layout / normalisation / residual steps around the two pointwise GEMMs of a SLaK block
 * (models/SLaK.py:153-166: permute -> LayerNorm -> [pwconv1, GELU, pwconv2] -> gamma -> permute -> shortcut + drop_path),
 * one HBM-bound kernel per arrow and direction.  Activations bf16, statistics / residual stream / parameters fp32.
 * P = H*W pixels per image; x, dx, shortcut, out are NCHW; y, g, z, dz are NHWC (N,H,W,C) contiguous; C even, <= 1024. */
size_t slak_block_tail_workspace_bytes(int N, int C, int P);

/* y[n,p,:] = LayerNorm_C(x[n,:,p]) * weight + bias   (F.layer_norm over the last dim of x.permute(0,2,3,1): models/SLaK.py:156-157, :253-255) */
int slak_ln_nchw_to_nhwc_forward(const void* x_bf16, const float* weight, const float* bias, void* y_bf16, float* mean, float* rstd,
                                 int N, int C, int P, float eps, void* stream);
int slak_ln_nchw_to_nhwc_backward(const void* g_bf16, const void* x_bf16, const float* weight, const float* mean, const float* rstd,
                                  void* dx_bf16, float* dweight, float* dbias, int N, int C, int P,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* out[n,c,p] = shortcut[n,c,p] + sample_scale[n] * gamma[c] * z[n,p,c]        (models/SLaK.py:161-165; sample_scale = the
 * DropPath mask/keep_prob per sample or NULL) ; backward: dz = sample_scale*gamma*dout (bf16 NHWC), dgamma = sum sample_scale*dout*z */
int slak_scale_residual_forward(const void* shortcut, int shortcut_dtype, const void* z_bf16, const float* gamma, const float* sample_scale,
                                float* out, void* out_bf16 /* NULL, or a second copy of `out` rounded to bf16: the next block's conv input */,
                                int N, int C, int P, void* stream);
/* dout_bf16 (NULL or the gradient that arrived through out_bf16) is added to dout; the sum -- the gradient of `shortcut` -- is
 * written to dout_sum (required iff dout_bf16 is given; otherwise the shortcut gradient is dout itself).
 * dz_colsum[c] = sum over n,p of the (unrounded) dz: the bias gradient of the Linear that produced z (models/SLaK.py:160). */
int slak_scale_residual_backward(const float* dout, const void* dout_bf16, float* dout_sum, const void* z_bf16, const float* gamma,
                                 const float* sample_scale, void* dz_bf16, float* dgamma, float* dz_colsum, int N, int C, int P,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* Downsample layers (models/SLaK.py:285-311: LayerNorm(C, data_format="channels_first") -> Conv2d(C, C', kernel_size=2, stride=2)).  The
 * kernel/stride-2 convolution is a GEMM on non-overlapping 2x2 patches; slak_ln_patch_forward normalises the fp32 NCHW input over C and
 * writes the result (rounded to bf16, as autocast's cast in front of the conv does) directly as that GEMM's operand
 *     a[n][ho * (W/2) + wo][(kh * 2 + kw) * C + c] = LN(x)[n, c, 2 ho + kh, 2 wo + kw],
 * so the conv is Y[n] = Wp . a[n]^T (batched library GEMM, NCHW result, Wp[co][(kh*2+kw)*C + c] = weight[co, c, kh, kw]) with no im2col,
 * layout transposes or casts; _backward takes dL/da in the same layout.  C in {64, 96, 128, 192, 256, 384, 512}, H and W even. */
int slak_ln_patch_supported(int N, int C, int H, int W);
int slak_ln_patch_forward(const float* x, const float* weight, const float* bias, void* a_bf16, float* mean, float* rstd,
                          int N, int C, int H, int W, float eps, void* stream);
int slak_ln_patch_backward(const void* g_bf16, const float* x, const float* weight, const float* mean, const float* rstd,
                           float* dx, float* dweight, float* dbias, int N, int C, int H, int W,
                           void* workspace /* slak_block_tail_workspace_bytes(N, C, H*W) */, size_t workspace_bytes, void* stream);

/* Stem (models/SLaK.py:276-279: Conv2d(in_chans, C, kernel_size=4, stride=4) on the fp32 image): the 4x4 patches as the bf16 GEMM operand
 * a[n][ho * (W/4) + wo][(c * 4 + kh) * 4 + kw] = x[n, c, 4 ho + kh, 4 wo + kw]; the conv is Y[n] = weight.view(C, in_chans*16) . a[n]^T (NCHW).
 * H and W multiples of 4. */
int slak_stem_patchify(const float* x, void* a_bf16, int N, int Cin, int H, int W, void* stream);

/* dst (N, P, C) = src (N, C, P) transposed per image, bf16: the NCHW output gradient of a downsample convolution (models/SLaK.py:195-199) as the
 * row operand [N*P][C] of its weight-gradient GEMM (slak_linear_wgrad) -- torch's `grad.transpose(1, 2).reshape(...)` copy.  C % 8 == 0. */
int slak_nchw_to_pixel_major_bf16(const void* src, void* dst, int N, int C, int P, void* stream);

/* y (N, C, P) bf16 = bf16(bias[c]) broadcast: the accumulator the batched GEMM of a downsample convolution (models/SLaK.py:195-199) adds its
 * products to -- the bias of nn.Conv2d inside the GEMM (fp32 accumulate, one rounding) without torch's strided broadcast copy. */
int slak_fill_channel_bias_bf16(const float* bias, void* y_bf16, int N, int C, int P, void* stream);

/* Bias gradient of the stem / downsample convolutions: out[c] = sum over n and p of the bf16 NCHW gradient x[n][c][p], fp32, fixed summation
 * order.  Replaces `grad_output.sum((0, 2, 3))` of torch's Conv2d backward (reference: models/SLaK.py:188-199, the stem and downsample
 * nn.Conv2d layers).  workspace: slak_channel_sums_workspace_bytes(C) bytes. */
/* Forward of the stem convolution (models/SLaK.py:189-193: nn.Conv2d(3, dims[0], kernel_size=4, stride=4)) under bf16 autocast, from the fp32
 * NCHW image x (N, 3, H, W): y (N, Co, H/4, W/4) bf16 = bf16(bf16(bias) + sum_k bf16(weight[co][k]) * bf16(patch[k])) with fp32 accumulation --
 * the arithmetic of the autocast conv -- and the bf16 patch matrix a (N, P, 48) of slak_stem_patchify written in the same pass (the operand of
 * slak_stem_wgrad).  weight: fp32 (Co, 3, 4, 4) contiguous; bias: fp32 (Co) or NULL.  Supported: Cin = 3, Co % 32 == 0, Co <= 128, H % 4 == 0,
 * W % 4 == 0, (H/4)*(W/4) % 64 == 0; otherwise SLAK_ERR_UNSUPPORTED and the caller keeps slak_stem_patchify + its GEMM. */
int slak_stem_conv_forward_supported(int N, int Cin, int H, int W, int Co);
int slak_stem_conv_forward(const float* x, const float* weight, const float* bias, void* a_bf16, void* y_bf16, int N, int Cin, int H, int W, int Co,
                           void* stream);

/* Weight and bias gradient of the stem convolution (models/SLaK.py:189-193: nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4)) from the bf16
 * NCHW output gradient dy (N, Co, P) and the forward's patch matrix a (N, P, K) of slak_stem_patchify (K = Cin*16): dw (Co, K) fp32 =
 * sum_{n,p} dy[n][co][p] * a[n][p][k] -- reshaped (Co, Cin, 4, 4) it is Conv2d's weight gradient -- and db (Co) = sum_{n,p} dy (NULL: not wanted).
 * One pass over dy and a, fp32 accumulation, fixed summation order.  Supported: Co % 32 == 0, Co <= 128, P % 64 == 0, K % 8 == 0, 32 <= K <= 56
 * (Cin = 3: K = 48); otherwise SLAK_ERR_UNSUPPORTED and the caller keeps its GEMM. */
int slak_stem_wgrad_supported(int N, int Co, int P, int K);
size_t slak_stem_wgrad_workspace_bytes(int N, int Co, int P, int K);
int slak_stem_wgrad(const void* dy_bf16, const void* a_bf16, float* dw, float* db, int N, int Co, int P, int K,
                    void* workspace, size_t workspace_bytes, void* stream);

size_t slak_channel_sums_workspace_bytes(int C);
int slak_channel_sums_bf16(const void* x_bf16, float* out, int N, int C, int P, void* workspace, size_t workspace_bytes, void* stream);

/* The pointwise convolutions on the large maps (stage 1-2: M = N*H*W rows of C <= 192 or 4C <= 768 channels against a weight of a few
 * hundred KB) are HBM streams, not GEMMs: Y[M,N] = X[M,K] . Wt[N,K]^T (+ bias[N]) with both operands K-contiguous ("NT": pwconv1 /
 * pwconv2 forward take the nn.Linear weight as it is stored, models/SLaK.py:158-160; the data gradients take its transpose), bf16 in
 * and out, fp32 accumulate; gelu_out (or NULL) receives nn.GELU() of the rounded Y in the same pass.  Covered: K in {96, 192} with
 * N % 32 == 0, or N in {96, 192} with K % 16 == 0 (no gelu_out); anything else returns SLAK_ERR_UNSUPPORTED (library GEMM). */
int slak_linear_nt_supported(int M, int N, int K, int gelu);
int slak_linear_nt(const void* x_bf16, const void* wt_bf16, const void* bias_bf16 /* or NULL */, void* y_bf16, void* gelu_out_bf16 /* or NULL */,
                   int M, int N, int K, void* stream);

/* pwconv1 -> GELU -> pwconv2 in ONE pass (models/SLaK.py:158-160) where the whole second weight fits a wave's registers: x [M][C], w1 [C4][C],
 * b1 [C4], w2 [C][C4], b2 [C] (the nn.Linear parameters as stored; biases may be NULL), y1 = x w1^T + b1 and a = GELU(y1) [M][C4] (the backward
 * needs both: same bits as slak_linear_nt with gelu_out), z = a w2^T + b2 [M][C], all bf16, fp32 accumulate.  a is not read back from HBM and
 * pwconv2 is not a launch.  Round 4: C = 96, C4 = 384 (stage 1 of SLaK-T / SLaK-S); anything else: SLAK_ERR_UNSUPPORTED (two slak_linear_nt). */
int slak_linear_mlp_fwd_supported(int M, int C, int C4);
int slak_linear_mlp_fwd(const void* x_bf16, const void* w1_bf16, const void* b1_bf16, const void* w2_bf16, const void* b2_bf16,
                        void* y1_bf16, void* a_bf16, void* z_bf16, int M, int C, int C4, void* stream);

/* The fixed-order column sums that end slak_scale_residual_backward, slak_ln_nchw_to_nhwc_backward, slak_gelu_backward_bias,
 * slak_linear_nt_gelu_bwd and slak_linear_wgrad (the parameter gradients of a block's tail: results nothing else of the block's backward
 * reads) in ONE launch instead of one each: between _begin and _end ON THE CALLING THREAD those calls record their reduction instead of
 * launching it -- the partial rows stay in the workspace each call was given, so the caller must hand every call in between its OWN
 * workspace -- and _end launches one kernel that performs all of them, on the stream the calls were given, with the same additions in
 * the same order (the same bits).  Not nestable; a ninth pending reduction launches the eight before it; a call on ANOTHER stream than the
 * first recorded one is not deferred (its reduction launches at once on its own stream). */
int slak_defer_reductions_begin(void);
int slak_defer_reductions_end(void);

/* pwconv2's data gradient WITH nn.GELU()'s backward and pwconv1's bias gradient in the same pass (models/SLaK.py:159-160 backwards):
 * dy1[M,N] = round(dz[M,K] . W2[K,N]) * gelu'(y1[M,N]) (bf16, the same bits as slak_linear_nt followed by slak_gelu_backward_bias),
 * dbias[N] = column sums of the rounded dy1 (fp32, fixed order).  wt = W2^T stored [N][K].  The intermediate dact never reaches HBM.
 * Round 4: K = 96, N = 384, M % 32 == 0 (stage 1 of SLaK-T / SLaK-S); anything else: SLAK_ERR_UNSUPPORTED (the two calls). */
int slak_linear_nt_gelu_bwd_supported(int M, int N, int K);
size_t slak_linear_nt_gelu_bwd_workspace_bytes(int M, int N, int K);
int slak_linear_nt_gelu_bwd(const void* dz_bf16, const void* wt_bf16, const void* y1_bf16, void* dy1_bf16, float* dbias, int M, int N, int K,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same launch with the next product of the block's backward in it: dt[M,K] = dy1 . W1 (pwconv1's data gradient), taken from the dy1 tiles while
 * they are on chip -- the separate slak_linear_nt launch and its re-read of dy1 (308 MB per stage-1 block of SLaK-T) are gone.  w1p = W1^T [K][N]
 * bf16 (pwconv1's nn.Linear weight transposed) in fragment-major order: viewed (3, 32, 6, 4, 2, 8) and permuted (2, 3, 0, 4, 1, 5), contiguous.
 * dy1 and dbias: the bits of slak_linear_nt_gelu_bwd; dt: bf16 of the same fp32 sums added in another order.  Shapes and workspace as above. */
int slak_pack_w1t_fragments(const void* w1t_bf16 /*[K][N] row-major*/, void* w1p_bf16 /*[K * N] out*/, int N, int K, void* stream);   /* the packer of w1p (N = 384, K = 96) */
int slak_linear_nt_gelu_bwd_dt_supported(int M, int N, int K);
int slak_linear_nt_gelu_bwd_dt(const void* dz_bf16, const void* wt_bf16, const void* y1_bf16, const void* w1p_bf16, void* dy1_bf16, void* dt_bf16,
                               float* dbias, int M, int N, int K, void* workspace, size_t workspace_bytes, void* stream);

/* Round 5 -- the pointwise Linear layers of stages 2-4 with their elementwise neighbours in the GEMM's epilogue (models/SLaK.py:156-165:
 * pwconv1 -> nn.GELU() -> pwconv2, and their data gradients): out[M][N] = a[M][K] . b[N][K]^T, both operands K-contiguous (the nn.Linear weight as it is
 * stored; the data gradients take its transpose), bf16 in and out, fp32 accumulate.  epilogue:
 *   SLAK_EPI_BIAS   out = bf16(acc + bias)                                    bias [N] bf16 or NULL
 *   SLAK_EPI_GELU   out = bf16(acc + bias), out2 = GELU(out)                  (exact erf form on the ROUNDED pre-activation, as F.gelu of a bf16 tensor; both
 *                                                                              tensors are kept because nn.GELU's backward reads the pre-activation)
 *   SLAK_EPI_DGELU  out = bf16(bf16(acc) * gelu'(y1)), dbias[N] = column sums of out   (y1 [M][N] bf16: the same dy1 bits as the GEMM followed by
 *                                                                              slak_gelu_backward_bias; dbias fp32, fixed summation order; needs the workspace)
 * Replaces on these stages: at::linear + at::gelu (two passes over [M][4C]) resp. at::mm + slak_gelu_backward_bias (the [M][4C] intermediate `dact`
 * written and read back).  Covered: epilogue GELU / DGELU with K in {192, 256, 384} and N % 256 == 0, or K in {512, 768} and N % 128 == 0 (stages 2-4 of
 * SLaK-T / -S, stages 2-3 of SLaK-B), M >= 1, tensors below 4 GiB; anything else -- SLAK_EPI_BIAS included: reserved -- SLAK_ERR_UNSUPPORTED (library GEMM +
 * elementwise pass). */
enum { SLAK_EPI_BIAS = 0, SLAK_EPI_GELU = 1, SLAK_EPI_DGELU = 2 };
int slak_linear_gemm_supported(int M, int N, int K, int epilogue);
size_t slak_linear_gemm_workspace_bytes(int M, int N, int K, int epilogue);
int slak_linear_gemm(const void* a_bf16, const void* b_bf16, const void* bias_bf16, void* out_bf16, void* out2_bf16, const void* y1_bf16, float* dbias,
                     int M, int N, int K, int epilogue, void* workspace, size_t workspace_bytes, void* stream);

/* Transposed bf16 copies dst[i] [cols][rows] of src[i] [rows][cols] (rows, cols multiples of 8) for n matrices in ONE launch (host arrays of device
 * pointers): the pointwise weights as the data-gradient GEMMs read them (dz . W2, dy1 . W1: models/SLaK.py:158-160 backwards).  Replaces one strided copy
 * kernel per weight, block and step (25 launches per SLaK-T step) by one launch per optimizer step (slak_amd/block_ops.lowp_param_t caches by version). */
int slak_transpose_bf16_batch(const void* const* src, void* const* dst, const int* rows, const int* cols, int n, void* stream);

/* Weight gradient of the pointwise Linear layers (models/SLaK.py:117-118 pwconv1 / pwconv2; autograd's dW = dY^T X):
 * d[N1][N2] (fp32) = x1^T x2 over the M rows of x1 [M][N1] and x2 [M][N2] (bf16, row-major), fp32 accumulate, summed in a fixed
 * order (deterministic).  Covered: N1 and N2 multiples of 192, or one of them 96 and the other a multiple of 384 (the ConvNeXt widths
 * 96 * 2^k and their 4x expansions); M >= 32; anything else returns SLAK_ERR_UNSUPPORTED (library GEMM). */
int slak_linear_wgrad_supported(int M, int N1, int N2);
size_t slak_linear_wgrad_workspace_bytes(int M, int N1, int N2);
int slak_linear_wgrad(const void* x1_bf16, const void* x2_bf16, float* d, int M, int N1, int N2, void* workspace, size_t workspace_bytes, void* stream);

/* GELU backward (exact erf form, nn.GELU()) fused with the bias gradient of the Linear in front of it (models/SLaK.py:158-160):
 * dy1 = dact * gelu'(y1); dbias[col] = sum_rows dy1.  [rows][cols] bf16 contiguous, cols % 8 == 0. */
size_t slak_gelu_bwd_workspace_bytes(int rows, int cols);
int slak_gelu_backward_bias(const void* dact_bf16, const void* y1_bf16, void* dy1_bf16, float* dbias, int rows, int cols,
                            void* workspace, size_t workspace_bytes, void* stream);

/* channels_first LayerNorm of the stem / downsample layers (models/SLaK.py:192-203, :256-261): y[n,c,p] = LN_C(x[n,:,p])*w + b, NCHW in
 * and out; x/y/g/dx fp32 or bf16 (dx has the dtype of x). */
size_t slak_ln_cf_workspace_bytes(int N, int C, int P);
int slak_ln_channels_first_forward(const void* x, int x_dtype, const float* weight, const float* bias, void* y, int y_dtype,
                                   float* mean, float* rstd, int N, int C, int P, float eps, void* stream);
int slak_ln_channels_first_backward(const void* g, int g_dtype, const void* x, int x_dtype, const float* weight, const float* mean,
                                    const float* rstd, void* dx, float* dweight, float* dbias, int N, int C, int P,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* The stem's LayerNorm handing the first block a bf16 copy of its output (what the block's depthwise convs read under autocast), and taking
 * that copy's gradient back: y_bf16 (NULL: none) = bf16(y); g2_bf16 (NULL: none) is ADDED to g on load -- `.to(bfloat16)` in front of the
 * first block and autograd's `grad + grad_lowp.float()` behind it are no separate passes.  _backward_pair with g2 covers what
 * slak_ln_channels_first_backward_pair_supported says (x bf16, g fp32, C = 96 / 128 / 192: one pass over g and x, the channels of a 64-pixel tile
 * in the registers of four waves); without g2 it is slak_ln_channels_first_backward. */
int slak_ln_channels_first_forward_pair(const void* x, int x_dtype, const float* weight, const float* bias, void* y, int y_dtype, void* y_bf16,
                                        float* mean, float* rstd, int N, int C, int P, float eps, void* stream);
int slak_ln_channels_first_backward_pair_supported(int g_dtype, int x_dtype, int N, int C, int P);
int slak_ln_channels_first_backward_pair(const void* g, int g_dtype, const void* g2_bf16, const void* x, int x_dtype, const float* weight,
                                         const float* mean, const float* rstd, void* dx, float* dweight, float* dbias, int N, int C, int P,
                                         void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- next row: branch BatchNorms + adds
 * out = BN1(y1) + BN2(y2) + BN3(y3) of ReparamLargeKernelConv (models/SLaK.py:38-47, :92-95) as one statistics pass, a per-channel
 * finalise and one apply pass; backward likewise (dy_b is affine in (dout, y_b) per channel).  y_b, out, dout, dy_b: bf16 NCHW,
 * P = H*W.  The *_sums calls return THIS RANK's per-channel sums; under SyncBatchNorm the caller all-reduces them (6C doubles / 4C
 * floats, one collective per block and direction instead of three gathers) and passes the global sums and the global element count to the
 * *_apply calls (`count`, or `count_dev` -- a device double, e.g. the all-reduced count -- when non-NULL: no host synchronisation).
 * Round 3: the forward statistics are CENTRED -- slice sums of (y - k) with k the slice's first element, combined as doubles; raw fp32
 * rows from the conv launches (pre_sums) are added in double and a channel with mean^2 > 1024 var is re-measured by a two-pass read --
 * so the variance survives |mean| / std of 1e3 and more (nn.BatchNorm2d's Welford statistics do); the backward sums are the centred
 * products sum dout * (y_b - mean_b). */
size_t slak_bn3_workspace_bytes(int N, int C);
int slak_bn3_forward_sums(const void* y1, const void* y2, const void* y3, double* local_sums /*[C][6]: sum y_b, sum y_b^2*/, int N, int C, int P,
                          void* workspace, size_t workspace_bytes, void* stream,
                          const float* const* pre_sums /* NULL or as in slak_bn3_forward_local */, const int* pre_rows, int pre_stride);
int slak_bn3_forward_apply(const void* y1, const void* y2, const void* y3, const double* global_sums, double count, const double* count_dev,
                           const float* const* gamma_host3, const float* const* beta_host3, float* const* running_mean_host3,
                           float* const* running_var_host3, float eps, float momentum, int training, int update_running,
                           float* coef /*[C][4]*/, float* stats /*[C][6]*/, void* out, int N, int C, int P, void* stream);
int slak_bn3_backward_sums(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats /*[C][6] of the forward*/,
                           float* local_sums /*[C][4]: sum dout, sum dout*(y_b - mean_b)*/,
                           int N, int C, int P, void* workspace, size_t workspace_bytes, void* stream);
int slak_bn3_backward_apply(const void* dout, const void* y1, const void* y2, const void* y3, const float* global_sums,
                            const float* local_sums, double count, const double* count_dev /* device count overrides `count` */,
                            const float* stats, const float* const* gamma_host3,
                            float* bcoef /*[C][9]*/, float* dgamma /*[3][C]*/, float* dbeta /*[3][C]*/,
                            void* dy1, void* dy2, void* dy3, int N, int C, int P, void* stream);
/* Single-process training step of the same op (no all-reduce between the statistics and the apply pass: the slice reduction and the
 * finalise step share a launch).  Same arithmetic and side effects as _forward_sums + _forward_apply(training = 1) resp.
 * _backward_sums + _backward_apply with global_sums = local_sums, count = N * P. */
int slak_bn3_forward_local(const void* y1, const void* y2, const void* y3, const float* const* gamma, const float* const* beta,
                           float* const* running_mean, float* const* running_var, float eps, float momentum, int update_running,
                           float* coef, float* stats, void* out, int N, int C, int P, void* workspace, size_t workspace_bytes, void* stream,
                           const float* const* pre_sums /* NULL, or 3 device pointers: branch b's partial sums, element (row, c, k) at
                                                           pre_sums[b][(row * C + c) * pre_stride + k], k = 0 (sum), 1 (sum of squares) */,
                           const int* pre_rows /* rows per branch */, int pre_stride /* 6: one slak_dwconv2d_tri_forward_stats array, pointers offset
                                                           by 0, 2, 4 floats; 2: three slak_dwconv2d_forward_stats arrays */);
int slak_bn3_backward_local(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats, const float* const* gamma,
                            float* bcoef, float* dgamma, float* dbeta, void* dy1, void* dy2, void* dy3, int N, int C, int P,
                            void* workspace, size_t workspace_bytes, void* stream);
/* Round 6: the same two calls with ONE destination per parameter gradient (dgamma3 / dbeta3: host arrays of three device pointers, [C]
 * floats each) instead of two [3][C] arrays.  Under DistributedDataParallel(gradient_as_bucket_view=True) a parameter's .grad is a view
 * of an all-reduce bucket (main.py:374-376 builds the wrapper): with these the six BatchNorm parameter gradients of a block
 * (models/SLaK.py:38-47) are written where the reducer wants them and its per-parameter copy launch disappears.  The [3][C]
 * forms above are wrappers over these (same kernels, same bits). */
/* Round 6, the SyncBatchNorm exchange with two launches fewer per block and step: _forward_sums_counted also writes this rank's element count
 * N * P into local_sums[6C] (the exchange buffer is [6C + 1] doubles: the count travels with the sums, models/SLaK.py:24-28's SyncBatchNorm
 * needs the global count); _backward_sums_dup writes the sums twice (local_sums stays for the local parameter gradients, sums_copy is what the
 * all-reduce overwrites in place). */
int slak_bn3_forward_sums_counted(const void* y1, const void* y2, const void* y3, double* local_sums /*[6C + 1]*/, int N, int C, int P,
                                  void* workspace, size_t workspace_bytes, void* stream,
                                  const float* const* pre_sums, const int* pre_rows, int pre_stride);
int slak_bn3_backward_sums_dup(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats, float* local_sums /*[C][4]*/,
                               float* sums_copy /*[C][4]*/, int N, int C, int P, void* workspace, size_t workspace_bytes, void* stream);
int slak_bn3_backward_apply_to(const void* dout, const void* y1, const void* y2, const void* y3, const float* global_sums,
                               const float* local_sums, double count, const double* count_dev, const float* stats, const float* const* gamma_host3,
                               float* bcoef /*[C][9]*/, float* const* dgamma3, float* const* dbeta3,
                               void* dy1, void* dy2, void* dy3, int N, int C, int P, void* stream);
int slak_bn3_backward_local_to(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats, const float* const* gamma,
                               float* bcoef, float* const* dgamma3, float* const* dbeta3, void* dy1, void* dy2, void* dy3, int N, int C, int P,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- next row: mask-aware optimizer step and EMA
 * One launch over ALL tensors each.  Descriptor arrays live in HOST memory at plan creation (device pointers inside); plans own their
 * device tables.
 *
 * slak_adamw_step = torch.optim.AdamW's update (optim_factory.py:149-150; decoupled weight decay, bias corrections evaluated in double
 * from the tensor's own step count) followed by Masking.apply_mask's w *= mask (sparse_core.py:316-333) and, optionally, the bf16 copy
 * of the updated weight, in one pass.  `step` points at a device float holding the tensor's step count AFTER this step's increment
 * (the caller adds 1 to its step buffer first: one tiny launch).  Gradients are re-allocated by autograd / DDP every step, so their
 * pointers arrive separately: `grads_dev` is a DEVICE array of nseg device pointers (NULL entry = no gradient: the tensor is skipped,
 * as torch skips p.grad is None).  Hyper-parameters are per group (the reference builds one group per layer x {decay, no_decay}:
 * optim_factory.py:73-112) and are passed by value at every call (lr follows a per-iteration schedule: engine.py:41-46). */
#define SLAK_ADAMW_MAX_GROUPS 64
typedef struct {
    float* param;            /* fp32, updated in place                                        */
    float* exp_avg;          /* fp32 first moment                                             */
    float* exp_avg_sq;       /* fp32 second moment                                            */
    const float* mask;       /* fp32 0/1 mask (Masking.masks[name]) or NULL                   */
    void* param_bf16;        /* optional bf16 copy of the updated parameter, or NULL          */
    const float* step;       /* device scalar: step count of this tensor, already incremented */
    long long numel;
    int group;               /* index into the groups array of slak_adamw_step                */
} slak_adamw_segment_t;
typedef struct { double lr, beta1, beta2, eps, weight_decay; } slak_adamw_group_t;
typedef struct slak_adamw_plan slak_adamw_plan_t;
int slak_adamw_plan_create(const slak_adamw_segment_t* segs_host, int nseg, slak_adamw_plan_t** plan_out);
int slak_adamw_step(slak_adamw_plan_t* plan, const void* const* grads_dev, const slak_adamw_group_t* groups_host, int ngroups, void* stream);
int slak_adamw_plan_destroy(slak_adamw_plan_t* plan);

/* slak_ema_update = ModelEma.update(model, mask) (model_sema.py:67-91) over every state-dict entry:
 *   dense  : ema = ema*decay + (1-decay)*model
 *   masked : ema = (ema*decay + model*(1-decay))*mask + (diff*decay)*model,  diff = ((ema != 0) ^ mask) & mask  (newly grown weights)
 * each product and sum rounded separately in fp32, so the result is bit-identical to the reference's chain of torch kernels.
 * dtype SLAK_F32, or SLAK_I64 for BatchNorm's num_batches_tracked (float32 arithmetic, truncating copy back, as torch does). */
typedef struct {
    void* ema;               /* EMA copy of the entry, updated in place   */
    const void* model;       /* the live model's entry                    */
    const float* mask;       /* fp32 0/1 mask of this entry or NULL       */
    long long numel;
    int dtype;               /* SLAK_F32 or SLAK_I64                      */
} slak_ema_segment_t;
typedef struct slak_ema_plan slak_ema_plan_t;
int slak_ema_plan_create(const slak_ema_segment_t* segs_host, int nseg, slak_ema_plan_t** plan_out);
int slak_ema_update(slak_ema_plan_t* plan, double decay, void* stream);
int slak_ema_plan_destroy(slak_ema_plan_t* plan);

#ifdef __cplusplus
}
#endif
#endif /* SLAK_HIP_H */
