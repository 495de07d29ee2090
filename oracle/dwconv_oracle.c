/*
 * oracle/dwconv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar loops, fp64 accumulation) of the depthwise
 * convolution that the reference's native extension computes:
 *   stride 1, dilation 1, "same" padding (kh/2, kw/2), groups == C,
 *   cross-correlation (no filter flip), NCHW contiguous tensors,
 *   weight (C,1,kh,kw)                      -- forward_fp32.cu:135-144, :227, :235
 * The scalar definition followed here is CUTLASS's own host reference:
 *   fprop : Depsep_Fprop   cutlass/tools/util/include/cutlass/util/reference/host/convolution.h:160-235
 *   dgrad : Depsep_Dgrad   same file :327  (generic form Conv2dDgrad :498)
 *   wgrad : Depsep_Wgrad   same file :420  (generic form Conv2dWgrad :586)
 * and, at the Python boundary, F.conv2d(x, w, padding=k//2, groups=C)
 *   (cutlass/examples/19_large_depthwise_conv2d_torch_extension/test_correctness.py:8-9).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Pinned by tests/test_oracle.py against tests/golden/dwconv_*.npz, which were generated
 * in the build container from torch CPU F.conv2d (+autograd) by tests/golden/make_golden.py.
 *
 * All functions: inputs float32, outputs float64 (callers round to the dtype under test).
 * OpenMP is used over independent (n,c) planes / channels only, so results are
 * run-to-run deterministic.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define IDX4(n, c, h, w, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (h)) * (W) + (w))

/* y[n,c,p,q] = sum_{r,s} x[n,c,p-ph+r,q-pw+s] * w[c,r,s]   (Depsep_Fprop, cross-correlation) */
void slak_oracle_dwconv2d_fwd(const float* x, const float* w, double* y,
                              int N, int C, int H, int W, int kh, int kw) {
    const int ph = kh / 2, pw = kw / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int p = 0; p < H; ++p)
                for (int q = 0; q < W; ++q) {
                    double acc = 0.0;
                    for (int r = 0; r < kh; ++r) {
                        const int ih = p - ph + r;
                        if (ih < 0 || ih >= H) continue;
                        for (int s = 0; s < kw; ++s) {
                            const int iw = q - pw + s;
                            if (iw < 0 || iw >= W) continue;
                            acc += (double)x[IDX4(n, c, ih, iw, C, H, W)] *
                                   (double)w[((size_t)c * kh + r) * kw + s];
                        }
                    }
                    y[IDX4(n, c, p, q, C, H, W)] = acc;
                }
}

/* dx[n,c,h,w] = sum_{r,s} dy[n,c,h+ph-r,w+pw-s] * w[c,r,s]   (Depsep_Dgrad / Conv2dDgrad) */
void slak_oracle_dwconv2d_bwd_data(const float* dy, const float* w, double* dx,
                                   int N, int C, int H, int W, int kh, int kw) {
    const int ph = kh / 2, pw = kw / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h)
                for (int x_ = 0; x_ < W; ++x_) {
                    double acc = 0.0;
                    for (int r = 0; r < kh; ++r) {
                        const int p = h + ph - r;
                        if (p < 0 || p >= H) continue;
                        for (int s = 0; s < kw; ++s) {
                            const int q = x_ + pw - s;
                            if (q < 0 || q >= W) continue;
                            acc += (double)dy[IDX4(n, c, p, q, C, H, W)] *
                                   (double)w[((size_t)c * kh + r) * kw + s];
                        }
                    }
                    dx[IDX4(n, c, h, x_, C, H, W)] = acc;
                }
}

/* dw[c,r,s] = sum_{n,p,q} dy[n,c,p,q] * x[n,c,p-ph+r,q-pw+s]   (Depsep_Wgrad / Conv2dWgrad) */
void slak_oracle_dwconv2d_bwd_filter(const float* dy, const float* x, double* dw,
                                     int N, int C, int H, int W, int kh, int kw) {
    const int ph = kh / 2, pw = kw / 2;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c)
        for (int r = 0; r < kh; ++r)
            for (int s = 0; s < kw; ++s) {
                double acc = 0.0;
                for (int n = 0; n < N; ++n)
                    for (int p = 0; p < H; ++p) {
                        const int ih = p - ph + r;
                        if (ih < 0 || ih >= H) continue;
                        for (int q = 0; q < W; ++q) {
                            const int iw = q - pw + s;
                            if (iw < 0 || iw >= W) continue;
                            acc += (double)dy[IDX4(n, c, p, q, C, H, W)] *
                                   (double)x[IDX4(n, c, ih, iw, C, H, W)];
                        }
                    }
                dw[((size_t)c * kh + r) * kw + s] = acc;
            }
}

/* round-to-nearest-even float -> bfloat16 -> float (what a bf16 tensor holds); NaN kept quiet */
float slak_oracle_bf16_round(float v) {
    uint32_t u; memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) { u |= 0x00400000u; u &= 0xffff0000u; }
    else { u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; }
    float r; memcpy(&r, &u, 4); return r;
}
void slak_oracle_bf16_round_array(const float* in, float* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = slak_oracle_bf16_round(in[i]);
}
