// slak_amd/csrc/branch_bn.hip -- the three branch BatchNorms + the two adds of ReparamLargeKernelConv as ONE op
// (reference models/SLaK.py:38-47, :92-95):
//
//     out = BN1(y1) + BN2(y2) + BN3(y3)          y_b = LoRA1 / LoRA2 / small_conv output, (N,C,H,W) bf16
//
// PyTorch runs this as 3 BatchNorm kernels (each reads its input twice) and 2 adds per direction: ~18 passes over the
// activation per block.  Here: one statistics pass over the three inputs, a per-channel finalise, one apply pass.
//   forward : rowsums (sum y_b, sum y_b^2 per (n,c) row)  ->  [SyncBN: all-reduce of 6C+1 floats]  ->  finalize (mean, invstd,
//             running stats, fused scale_b / shift)  ->  apply:  out = sum_b scale_b[c]*y_b + shift[c]
//   backward: rowsums (sum dout, sum dout*y_b)  ->  [all-reduce 4C]  ->  finalize (dgamma_b, dbeta_b, per-channel A,B,C)  ->
//             apply:  dy_b = A_b[c]*dout + B_b[c]*y_b + C_b[c]     (the BatchNorm backward is affine in (dout, y_b) per channel)
// The tensors are treated as [N*C rows][P = H*W columns]: one wavefront per row with vector loads and a wavefront-wide
// reduction; per-channel sums add the N rows of a channel in a fixed order (deterministic).  All HBM-bound.
#include <stdlib.h>
#include "slak_common.h"

namespace slak {

constexpr int BN_THREADS = 256;

__device__ __forceinline__ float bnf(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __attribute__((ext_vector_type(2))) float bn_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bn_bf16x2;
__device__ __forceinline__ unsigned bn_pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(bn_f32x2{a, b}, bn_bf16x2)); }

__device__ __forceinline__ float bn_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// sum over the LPR (power of two) consecutive lanes that share a row
__device__ __forceinline__ float bn_seg_sum(float v, int lpr) {
    for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// lanes per row: enough lanes to cover a row with 8 elements each, at most a wavefront
static inline int bn_lpr(int P) { int n = (P + 7) / 8, l = 1; while (l < n && l < 64) l <<= 1; return l; }

// 8 consecutive bf16 of a row; zeros beyond P.  mode 2: rows 16-byte aligned (P % 8 == 0), 1: 8-byte aligned (P % 4 == 0), 0: scalar
struct __attribute__((packed, aligned(2))) bn_u4_a2 { unsigned x, y, z, w; };      // sixteen bytes at a 2-byte aligned address: one (split) access
__device__ __forceinline__ void unpack2(unsigned u, float& a, float& b) { a = bnf((uint16_t)(u & 0xffff)); b = bnf((uint16_t)(u >> 16)); }
__device__ __forceinline__ void load8(const uint16_t* __restrict__ row, int p, int P, int mode, float (&v)[8]) {
    if (mode == 2 && p + 8 <= P) {
        const uint4 u = *(const uint4*)(row + p);
        unpack2(u.x, v[0], v[1]); unpack2(u.y, v[2], v[3]); unpack2(u.z, v[4], v[5]); unpack2(u.w, v[6], v[7]);
    } else if (mode == 1) {
        uint2 lo = uint2{0u, 0u}, hi = uint2{0u, 0u};
        if (p + 4 <= P) lo = *(const uint2*)(row + p);
        if (p + 8 <= P) hi = *(const uint2*)(row + p + 4);
        unpack2(lo.x, v[0], v[1]); unpack2(lo.y, v[2], v[3]); unpack2(hi.x, v[4], v[5]); unpack2(hi.y, v[6], v[7]);
    } else if (p + 8 <= P) {
        // rows at 2-byte alignment (P odd: the 7 x 7 stage): ONE 16-byte access at a 2-byte aligned address (the hardware splits it: ~3x the cost of
        // an aligned one) instead of eight 2-byte ones (1.75 TB/s; the whole bn3 backward of a 7 x 7 block took 60 us for 106 MB)
        const bn_u4_a2 u = *(const bn_u4_a2*)(row + p);
        unpack2(u.x, v[0], v[1]); unpack2(u.y, v[2], v[3]); unpack2(u.z, v[4], v[5]); unpack2(u.w, v[6], v[7]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (p + e < P) ? bnf(row[p + e]) : 0.f;
    }
}
__device__ __forceinline__ void store8(uint16_t* __restrict__ row, int p, int P, int mode, const float (&v)[8]) {
    if (mode == 2 && p + 8 <= P) {
        uint4 u; u.x = bn_pack2(v[0], v[1]); u.y = bn_pack2(v[2], v[3]); u.z = bn_pack2(v[4], v[5]); u.w = bn_pack2(v[6], v[7]);
        *(uint4*)(row + p) = u;
    } else if (mode == 1) {
        if (p + 4 <= P) *(uint2*)(row + p) = uint2{bn_pack2(v[0], v[1]), bn_pack2(v[2], v[3])};
        if (p + 8 <= P) *(uint2*)(row + p + 4) = uint2{bn_pack2(v[4], v[5]), bn_pack2(v[6], v[7])};
    } else if (p + 8 <= P) {
        bn_u4_a2 u; u.x = bn_pack2(v[0], v[1]); u.y = bn_pack2(v[2], v[3]); u.z = bn_pack2(v[4], v[5]); u.w = bn_pack2(v[6], v[7]);
        *(bn_u4_a2*)(row + p) = u;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (p + e < P) row[p + e] = (uint16_t)(bn_pack2(v[e], 0.f) & 0xffffu);
    }
}

// Channel-slice sums (round 2; the round-1 row-sum kernels reduced every (n, c) row on its own -- 36 shuffles per row -- and left N partial
// rows per channel): a wavefront owns channel c and the images [s*per, (s+1)*per) of the batch; its lanes stride over the (image, 16-byte
// chunk) pairs of that slice and keep plain per-lane sums, reduced ONCE at the end.
//   forward  (K = 9): part[(s*C + c)*9 + 3b + {0,1,2}] = k_b, sum (y_b - k_b), sum (y_b - k_b)^2 with the SHIFT k_b = the slice's first
//            element of the channel (round 3): E[y^2] - mean^2 on raw fp32 sums loses (mean/std)^2 * 1e-6 of the variance -- everything at
//            mean/std ~ 1e3 -- while shifted sums lose only ((mean_slice - k)/std)^2 * 1e-6; the slices are combined in double
//            (bn3_local_stats) as sums of y and y^2, which 53 bits carry to mean/std ~ 1e6;
//   backward (K = 4): part[..] = sum dout, sum dout * (y_b - mean_b): the centred product, so dgamma = invstd * sum needs no
//            "sum dout*y - mean * sum dout" cancellation either.
template <bool BWD>
__global__ __launch_bounds__(BN_THREADS) void bn3_chansums(const uint16_t* __restrict__ dout, const uint16_t* __restrict__ y1,
                                                         const uint16_t* __restrict__ y2, const uint16_t* __restrict__ y3,
                                                         const float* __restrict__ stats, float* __restrict__ part, int N, int C, int P, int S, int per) {
    constexpr int K = BWD ? 4 : 9;
    const int lane = threadIdx.x & 63, wv = blockIdx.x * (BN_THREADS / 64) + (threadIdx.x >> 6);
    if (wv >= C * S) return;
    const int c = wv % C, sl = wv / C;
    const int n0 = sl * per, n1 = min(n0 + per, N);
    const int vec = (P & 7) == 0 ? 2 : ((P & 3) == 0 ? 1 : 0);
    const int cpr = (P + 7) / 8;                                   // 16-byte chunks per row
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float k1, k2, k3;                                              // forward: the slice's shift; backward: the batch means
    if constexpr (BWD) { k1 = stats[c * 6]; k2 = stats[c * 6 + 2]; k3 = stats[c * 6 + 4]; }
    else { const size_t b0 = ((size_t)n0 * C + c) * P; k1 = bnf(y1[b0]); k2 = bnf(y2[b0]); k3 = bnf(y3[b0]); }
    int n = n0, ch = lane;
    while (ch >= cpr) { ch -= cpr; ++n; }
    while (n < n1) {
        const size_t base = ((size_t)n * C + c) * P;
        float a[8], b[8], d[8];
        load8(y1 + base, ch * 8, P, vec, a); load8(y2 + base, ch * 8, P, vec, b); load8(y3 + base, ch * 8, P, vec, d);
        const int nv = min(8, P - ch * 8);                         // elements of this chunk inside the row (load8 pads with zeros)
        if constexpr (BWD) {
            float g[8];
            load8(dout + base, ch * 8, P, vec, g);                 // g = 0 beyond the row: the padded products vanish
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[0] += g[e]; s[1] += g[e] * (a[e] - k1); s[2] += g[e] * (b[e] - k2); s[3] += g[e] * (d[e] - k3); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = e < nv;
                const float da = in ? a[e] - k1 : 0.f, db = in ? b[e] - k2 : 0.f, dd = in ? d[e] - k3 : 0.f;
                s[0] += da; s[1] += da * da; s[2] += db; s[3] += db * db; s[4] += dd; s[5] += dd * dd;
            }
        }
        ch += 64;
        while (ch >= cpr) { ch -= cpr; ++n; }
    }
    float* o = part + ((size_t)sl * C + c) * K;
    if constexpr (BWD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float t = bn_wave_sum(s[k]); if (lane == 0) o[k] = t; }
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) s[k] = bn_wave_sum(s[k]);
        if (lane == 0) { o[0] = k1; o[1] = s[0]; o[2] = s[1]; o[3] = k2; o[4] = s[2]; o[5] = s[3]; o[6] = k3; o[7] = s[4]; o[8] = s[5]; }
    }
}
static void bn_slices(int N, int C, int* S, int* per) {            // ~16384 wavefronts (a full machine of eight per SIMD, twice over), whole images per slice
    static const int waves = [] { const char* e = slak_dev_getenv("SLAK_BN_WAVES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 16384; }();
    int s = waves / C; if (s < 1) s = 1; if (s > N) s = N;
    *per = (N + s - 1) / s; *S = (N + *per - 1) / *per;
}

// sums[c][k] = sum_n rows[(n*C + c)][k]: one wavefront per channel, lanes over n, fixed butterfly order (deterministic)
__global__ __launch_bounds__(64) void bn3_colreduce(const float* __restrict__ rows, float* __restrict__ sums, int N, int C, int K, float* __restrict__ sums2 = nullptr) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = lane; n < N; n += 64) {
        const float* r = rows + ((size_t)n * C + c) * K;
        for (int k = 0; k < K; ++k) s[k] += r[k];
    }
    for (int k = 0; k < K; ++k) { const float t = bn_wave_sum(s[k]); if (lane == 0) { sums[c * K + k] = t; if (sums2) sums2[c * K + k] = t; } }
}

__device__ __forceinline__ double bn_wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Forward finalise of ONE channel from sums in double (sd[2b] = sum y_b, sd[2b+1] = sum y_b^2 over `count` elements, this rank's or
// -- after the SyncBatchNorm all-reduce -- everybody's): coef[c][0..3] = scale_1, scale_2, scale_3, shift;  stats[c][0..5] = mean_b,
// invstd_b (saved for backward); running stats updated in place like nn.BatchNorm2d (momentum, unbiased variance).
struct Bn3Params {
    const float* gamma[3]; const float* beta[3];
    float* running_mean[3]; float* running_var[3];
};
struct Bn3Grads { float* dgamma[3]; float* dbeta[3]; };         // the six parameter gradients of a block's branch BatchNorms, one pointer each

__device__ __forceinline__ void bn3_finalize_channel(const double (&sd)[6], double count, int c, const Bn3Params& bp, float* __restrict__ coef,
                                                     float* __restrict__ stats, float eps, float momentum, int update_running) {
    float shift = 0.f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const double meand = sd[2 * b] / count;
        double vard = sd[2 * b + 1] / count - meand * meand;
        vard = vard > 0.0 ? vard : 0.0;
        const float mean = (float)meand, var = (float)vard;
        const float inv = 1.0f / sqrtf(var + eps);
        const float sc = bp.gamma[b][c] * inv;
        coef[c * 4 + b] = sc;
        shift += bp.beta[b][c] - mean * sc;
        stats[c * 6 + 2 * b] = mean; stats[c * 6 + 2 * b + 1] = inv;
        if (update_running) {
            const float unb = count > 1.0 ? (float)(vard * count / (count - 1.0)) : var;
            bp.running_mean[b][c] = (1.f - momentum) * bp.running_mean[b][c] + momentum * mean;
            bp.running_var[b][c] = (1.f - momentum) * bp.running_var[b][c] + momentum * unb;
        }
    }
    coef[c * 4 + 3] = shift;
}
// sums[C][6] doubles: GLOBAL sums (after the SyncBatchNorm all-reduce); count_dev: the all-reduced element count (a device double)
__global__ void bn3_finalize_fwd(const double* __restrict__ sums, Bn3Params bp, float* __restrict__ coef, float* __restrict__ stats,
                                 int C, double count, const double* __restrict__ count_dev, float eps, float momentum, int update_running) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (count_dev) count = *count_dev;                             // stays on the device: no host synchronisation
    double sd[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) sd[k] = sums[c * 6 + k];
    bn3_finalize_channel(sd, count, c, bp, coef, stats, eps, momentum, update_running);
}
// Eval mode: coefficients from the running statistics
__global__ void bn3_finalize_eval(Bn3Params bp, float* __restrict__ coef, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float shift = 0.f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float sc = bp.gamma[b][c] / sqrtf(bp.running_var[b][c] + eps);
        coef[c * 4 + b] = sc;
        shift += bp.beta[b][c] - bp.running_mean[b][c] * sc;
    }
    coef[c * 4 + 3] = shift;
}

// out[r][p] = coef[c][0]*y1 + coef[c][1]*y2 + coef[c][2]*y3 + coef[c][3],  c = r % C
__global__ __launch_bounds__(BN_THREADS) void bn3_apply_fwd(const uint16_t* __restrict__ y1, const uint16_t* __restrict__ y2,
                                                          const uint16_t* __restrict__ y3, const float* __restrict__ coef,
                                                          uint16_t* __restrict__ out, int R, int C, int P, int LPR) {
    const int lane0 = threadIdx.x & 63, wv = blockIdx.x * (BN_THREADS / 64) + (threadIdx.x >> 6), nw = gridDim.x * (BN_THREADS / 64);
    const int vec = (P & 7) == 0 ? 2 : ((P & 3) == 0 ? 1 : 0);
    const int rpw = 64 / LPR, sub = lane0 / LPR, lane = lane0 - sub * LPR;
    for (int r0 = wv * rpw; r0 < R; r0 += nw * rpw) {
        const int r = r0 + sub;
        if (r >= R) continue;
        const int c = r % C;
        const float k1 = coef[c * 4], k2 = coef[c * 4 + 1], k3 = coef[c * 4 + 2], k0 = coef[c * 4 + 3];
        const size_t base = (size_t)r * P;
        for (int p = lane * 8; p < P; p += LPR * 8) {
            float a[8], b[8], cc[8], o[8];
            load8(y1 + base, p, P, vec, a); load8(y2 + base, p, P, vec, b); load8(y3 + base, p, P, vec, cc);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = k1 * a[e] + k2 * b[e] + k3 * cc[e] + k0;
            store8(out + base, p, P, vec, o);
        }
    }
}

// Backward finalise.  sums[c][4] GLOBAL: sum dout, sum dout*(y_b - mean_b).  stats[c][6] = mean_b, invstd_b.
// dgamma_b = invstd_b * sum dout*(y_b - mean_b), dbeta_b = sum dout (LOCAL sums give the local parameter gradients that DDP
// then all-reduces, exactly like SyncBatchNorm: the caller passes local sums for the gradients and global sums for the coefficients).
// bcoef[c][b][0..2] = A, B, C0 with dy_b = A*dout + B*y_b + C0.
__global__ void bn3_finalize_bwd(const float* __restrict__ gsums, const float* __restrict__ lsums, const float* __restrict__ stats,
                                 Bn3Params bp, float* __restrict__ bcoef, Bn3Grads out,
                                 int C, float count, const double* __restrict__ count_dev) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (count_dev) count = (float)*count_dev;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float mean = stats[c * 6 + 2 * b], inv = stats[c * 6 + 2 * b + 1], g = bp.gamma[b][c];
        const float gd = gsums[c * 4], gdy = gsums[c * 4 + 1 + b];      // gdy = sum dout * (y_b - mean_b): centred in bn3_chansums<true>
        const float dgam_g = inv * gdy;                                  // global dgamma (for the input gradient)
        const float A = g * inv;
        const float B = -g * inv * inv * dgam_g / count;
        bcoef[(c * 3 + b) * 3 + 0] = A;
        bcoef[(c * 3 + b) * 3 + 1] = B;
        bcoef[(c * 3 + b) * 3 + 2] = -A * gd / count - B * mean;
        out.dgamma[b][c] = inv * lsums[c * 4 + 1 + b];
        out.dbeta[b][c] = lsums[c * 4];
    }
}

// This rank's per-channel sums in double from the partial records -- one wavefront per channel:
//   shifted records (bn3_chansums<false>: k, sum (y-k), sum (y-k)^2 per slice): exact to double via mean_p = k + S'/n_p,
//       M2_p = Q' - S'^2/n_p (centred: no cancellation), sum y = n_p mean_p, sum y^2 = M2_p + n_p mean_p^2;
//   raw records (sum y, sum y^2 in fp32, gathered by the conv launches in their copy-out): added in double; fp32 accumulation inside a
//       record has already lost (mean/std)^2 * ~1e-6 of the variance, so when mean^2 > 1024 var (never on a normalised network's
//       branch outputs; harmless below) the channel is RE-MEASURED here: sum (y - mean), sum (y - mean)^2 in a second read of its
//       planes -- the two-pass algorithm, slow (one wavefront, N*P elements) but exact, instead of a wrong variance.
// FINAL: single process -- finalise in the same launch (no all-reduce in between); otherwise lsum[c][6] doubles for the exchange.
struct Bn3Pre { const float* rows[3]; int S[3]; int stride; int shifted; int per; };
template <bool FINAL>
__global__ __launch_bounds__(64) void bn3_local_stats(const Bn3Pre pre, const uint16_t* __restrict__ y1, const uint16_t* __restrict__ y2,
                                                    const uint16_t* __restrict__ y3, int N, int C, int P, double* __restrict__ lsum,
                                                    Bn3Params bp, float* __restrict__ coef, float* __restrict__ stats, float eps, float momentum,
                                                    int update_running) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const double count = (double)N * P;
    double sd[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        double S = 0.0, Q = 0.0;
        for (int n = lane; n < pre.S[b]; n += 64) {
            const float* r = pre.rows[b] + ((size_t)n * C + c) * pre.stride;
            if (pre.shifted) {
                const int n0 = n * pre.per, n1 = min(n0 + pre.per, N);
                const double np = (double)(n1 - n0) * P, k = r[0], s1 = r[1], q1 = r[2];
                const double mean_p = k + s1 / np, m2 = q1 - s1 * s1 / np;
                S += np * mean_p; Q += (m2 > 0.0 ? m2 : 0.0) + np * mean_p * mean_p;
            } else { S += (double)r[0]; Q += (double)r[1]; }
        }
        S = bn_wave_sum_d(S); Q = bn_wave_sum_d(Q);
        if (!pre.shifted) {
            const double mean = S / count, var = Q / count - mean * mean;
            if (mean * mean > 1024.0 * var) {                        // wave-uniform (all lanes hold the reduced sums)
                const uint16_t* y = b == 0 ? y1 : (b == 1 ? y2 : y3);
                const float mf = (float)mean;
                const int vec = (P & 7) == 0 ? 2 : ((P & 3) == 0 ? 1 : 0), cpr = (P + 7) / 8;
                double D1 = 0.0, D2 = 0.0;
                for (int n = 0; n < N; ++n) {
                    const size_t base = ((size_t)n * C + c) * P;
                    float d1 = 0.f, d2 = 0.f;
                    for (int ch = lane; ch < cpr; ch += 64) {
                        float v[8];
                        load8(y + base, ch * 8, P, vec, v);
                        const int nv = min(8, P - ch * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = e < nv ? v[e] - mf : 0.f; d1 += d; d2 += d * d; }
                    }
                    D1 += d1; D2 += d2;
                }
                D1 = bn_wave_sum_d(D1); D2 = bn_wave_sum_d(D2);
                const double mean2 = (double)mf + D1 / count, m2 = D2 - D1 * D1 / count;
                S = count * mean2; Q = (m2 > 0.0 ? m2 : 0.0) + count * mean2 * mean2;
            }
        }
        sd[2 * b] = S; sd[2 * b + 1] = Q;
    }
    if (lane != 0) return;
    if constexpr (FINAL) bn3_finalize_channel(sd, count, c, bp, coef, stats, eps, momentum, update_running);
    else {
#pragma unroll
        for (int k = 0; k < 6; ++k) lsum[c * 6 + k] = sd[k];
        if (update_running && c == 0) lsum[(size_t)C * 6] = count;    // (!FINAL: the flag means "element [6C] takes this rank's element count": slak_bn3_forward_sums_counted)
    }
}
__global__ __launch_bounds__(64) void bn3_colreduce_finalize_bwd(const float* __restrict__ rows, int S, const float* __restrict__ stats, Bn3Params bp,
                                                               float* __restrict__ bcoef, Bn3Grads out,
                                                               int C, float count) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = lane; n < S; n += 64) {
        const float* r = rows + ((size_t)n * C + c) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += r[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = bn_wave_sum(s[k]);
    if (lane != 0) return;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float mean = stats[c * 6 + 2 * b], inv = stats[c * 6 + 2 * b + 1], g = bp.gamma[b][c];
        const float gd = s[0], gdy = s[1 + b];                          // sum dout * (y_b - mean_b)
        const float dgam = inv * gdy;
        const float A = g * inv;
        const float B = -g * inv * inv * dgam / count;
        bcoef[(c * 3 + b) * 3 + 0] = A;
        bcoef[(c * 3 + b) * 3 + 1] = B;
        bcoef[(c * 3 + b) * 3 + 2] = -A * gd / count - B * mean;
        out.dgamma[b][c] = dgam;
        out.dbeta[b][c] = gd;
    }
}

// dy_b[r][p] = A_b*dout + B_b*y_b + C_b
__global__ __launch_bounds__(BN_THREADS) void bn3_apply_bwd(const uint16_t* __restrict__ dout, const uint16_t* __restrict__ y1,
                                                          const uint16_t* __restrict__ y2, const uint16_t* __restrict__ y3,
                                                          const float* __restrict__ bcoef, uint16_t* __restrict__ d1,
                                                          uint16_t* __restrict__ d2, uint16_t* __restrict__ d3, int R, int C, int P, int LPR) {
    const int lane0 = threadIdx.x & 63, wv = blockIdx.x * (BN_THREADS / 64) + (threadIdx.x >> 6), nw = gridDim.x * (BN_THREADS / 64);
    const int vec = (P & 7) == 0 ? 2 : ((P & 3) == 0 ? 1 : 0);
    const int rpw = 64 / LPR, sub = lane0 / LPR, lane = lane0 - sub * LPR;
    for (int r0 = wv * rpw; r0 < R; r0 += nw * rpw) {
        const int r = r0 + sub;
        if (r >= R) continue;
        const int c = r % C;
        float k[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) k[i] = bcoef[c * 9 + i];
        const size_t base = (size_t)r * P;
        for (int p = lane * 8; p < P; p += LPR * 8) {
            float g[8], a[8], o[8];
            load8(dout + base, p, P, vec, g);
            load8(y1 + base, p, P, vec, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = k[0] * g[e] + k[1] * a[e] + k[2];
            store8(d1 + base, p, P, vec, o);
            load8(y2 + base, p, P, vec, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = k[3] * g[e] + k[4] * a[e] + k[5];
            store8(d2 + base, p, P, vec, o);
            load8(y3 + base, p, P, vec, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = k[6] * g[e] + k[7] * a[e] + k[8];
            store8(d3 + base, p, P, vec, o);
        }
    }
}

static int bn_grid(int R, int P) {
    const int rows_per_block = (BN_THREADS / 64) * (64 / bn_lpr(P));
    long long g = ((long long)R + rows_per_block - 1) / rows_per_block;
    const long long cap = 256LL * 16;
    return (int)(g < cap ? g : cap);
}

}  // namespace slak

using namespace slak;

extern "C" {

/* scratch: slice records rows[N*C][9] (forward: shift, shifted sum, shifted sum of squares per branch; backward uses [N*C][4]) */
size_t slak_bn3_workspace_bytes(int N, int C) { return (N <= 0 || C <= 0) ? 0 : align_up(((size_t)N * C * 9 + (size_t)C * 6) * sizeof(float), 256); }

static int bn_args_ok(int N, int C, int P) {
    if (N <= 0 || C <= 0 || P <= 0) return SLAK_ERR_INVALID_ARG;
    if ((long long)N * C * P >= (1LL << 31)) return SLAK_ERR_UNSUPPORTED;
    return SLAK_OK;
}

static bool bn_pre_ok(const float* const* pre_sums, const int* pre_rows, int pre_stride) {
    return !pre_sums || (pre_rows && pre_stride >= 2 && pre_sums[0] && pre_sums[1] && pre_sums[2] && pre_rows[0] > 0 && pre_rows[1] > 0 && pre_rows[2] > 0);
}
/* the partial records bn3_local_stats reads: the conv launches' raw rows, or (NULL) a read pass that leaves shifted slice records */
static void bn_make_pre(Bn3Pre& pre, const float* const* pre_sums, const int* pre_rows, int pre_stride, const void* y1, const void* y2, const void* y3,
                        int N, int C, int P, void* workspace, hipStream_t st) {
    if (pre_sums) {
        for (int b = 0; b < 3; ++b) { pre.rows[b] = pre_sums[b]; pre.S[b] = pre_rows[b]; }
        pre.stride = pre_stride; pre.shifted = 0; pre.per = 0;
    } else {
        int S, per; bn_slices(N, C, &S, &per);
        hipLaunchKernelGGL(bn3_chansums<false>, dim3((unsigned)((C * S + BN_THREADS / 64 - 1) / (BN_THREADS / 64))), dim3(BN_THREADS), 0, st,
                           (const uint16_t*)nullptr, (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3, (const float*)nullptr, (float*)workspace,
                           N, C, P, S, per);
        for (int b = 0; b < 3; ++b) { pre.rows[b] = (const float*)workspace + 3 * b; pre.S[b] = S; }
        pre.stride = 9; pre.shifted = 1; pre.per = per;
    }
}

/* local_sums[C][6] DOUBLES = per-channel sum y_b, sum y_b^2 over this rank's batch (exact to double: shifted slice sums, or the conv
 * launches' rows `pre_sums` with a two-pass re-measurement of channels whose raw sums cannot carry the variance). */
static int bn3_forward_sums_impl(const void* y1, const void* y2, const void* y3, double* local_sums, int N, int C, int P,
                                 void* workspace, size_t workspace_bytes, void* stream, const float* const* pre_sums, const int* pre_rows, int pre_stride, int with_count) {
    if (!y1 || !y2 || !y3 || !local_sums || !bn_pre_ok(pre_sums, pre_rows, pre_stride)) return SLAK_ERR_INVALID_ARG;
    int rc = bn_args_ok(N, C, P); if (rc) return rc;
    if (!workspace || workspace_bytes < slak_bn3_workspace_bytes(N, C)) return SLAK_ERR_WORKSPACE;
    Bn3Pre pre;
    bn_make_pre(pre, pre_sums, pre_rows, pre_stride, y1, y2, y3, N, C, P, workspace, (hipStream_t)stream);
    Bn3Params bp{};
    hipLaunchKernelGGL(bn3_local_stats<false>, dim3(C), dim3(64), 0, (hipStream_t)stream, pre, (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3,
                       N, C, P, local_sums, bp, (float*)nullptr, (float*)nullptr, 0.f, 0.f, with_count);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_bn3_forward_sums(const void* y1, const void* y2, const void* y3, double* local_sums, int N, int C, int P,
                          void* workspace, size_t workspace_bytes, void* stream, const float* const* pre_sums, const int* pre_rows, int pre_stride) {
    return bn3_forward_sums_impl(y1, y2, y3, local_sums, N, C, P, workspace, workspace_bytes, stream, pre_sums, pre_rows, pre_stride, 0);
}
/* local_sums: [6C + 1] -- element 6C takes this rank's element count N * P (the exchange buffer of the SyncBatchNorm path as one launch writes it) */
int slak_bn3_forward_sums_counted(const void* y1, const void* y2, const void* y3, double* local_sums, int N, int C, int P,
                                  void* workspace, size_t workspace_bytes, void* stream, const float* const* pre_sums, const int* pre_rows, int pre_stride) {
    return bn3_forward_sums_impl(y1, y2, y3, local_sums, N, C, P, workspace, workspace_bytes, stream, pre_sums, pre_rows, pre_stride, 1);
}

/* Single-process training forward (no statistics exchange between the sums and the apply pass): three launches. */
int slak_bn3_forward_local(const void* y1, const void* y2, const void* y3, const float* const* gamma, const float* const* beta,
                           float* const* running_mean, float* const* running_var, float eps, float momentum, int update_running,
                           float* coef, float* stats, void* out, int N, int C, int P, void* workspace, size_t workspace_bytes, void* stream,
                           const float* const* pre_sums, const int* pre_rows, int pre_stride) {
    if (!y1 || !y2 || !y3 || !gamma || !beta || !running_mean || !running_var || !coef || !stats || !out) return SLAK_ERR_INVALID_ARG;
    if (!bn_pre_ok(pre_sums, pre_rows, pre_stride)) return SLAK_ERR_INVALID_ARG;
    int rc = bn_args_ok(N, C, P); if (rc) return rc;
    if (!workspace || workspace_bytes < slak_bn3_workspace_bytes(N, C)) return SLAK_ERR_WORKSPACE;
    Bn3Params bp;
    for (int b = 0; b < 3; ++b) { bp.gamma[b] = gamma[b]; bp.beta[b] = beta[b]; bp.running_mean[b] = running_mean[b]; bp.running_var[b] = running_var[b]; }
    Bn3Pre pre;
    bn_make_pre(pre, pre_sums, pre_rows, pre_stride, y1, y2, y3, N, C, P, workspace, (hipStream_t)stream);
    hipLaunchKernelGGL(bn3_local_stats<true>, dim3(C), dim3(64), 0, (hipStream_t)stream, pre, (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3,
                       N, C, P, (double*)nullptr, bp, coef, stats, eps, momentum, update_running);
    const int R = N * C;
    hipLaunchKernelGGL(bn3_apply_fwd, dim3(bn_grid(R, P)), dim3(BN_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3, (const float*)coef, (uint16_t*)out, R, C, P, bn_lpr(P));
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

/* gamma/beta/running_*: arrays of 3 device pointers (host memory).  training != 0: statistics from global_sums ([C][6] DOUBLES) / count,
 * running stats updated; training == 0: running statistics (global_sums ignored).  Writes coef[C][4], stats[C][6] and out. */
int slak_bn3_forward_apply(const void* y1, const void* y2, const void* y3, const double* global_sums, double count, const double* count_dev,
                           const float* const* gamma, const float* const* beta, float* const* running_mean, float* const* running_var,
                           float eps, float momentum, int training, int update_running,
                           float* coef, float* stats, void* out, int N, int C, int P, void* stream) {
    if (!y1 || !y2 || !y3 || !gamma || !beta || !running_mean || !running_var || !coef || !out) return SLAK_ERR_INVALID_ARG;
    if (training && (!global_sums || !stats)) return SLAK_ERR_INVALID_ARG;
    int rc = bn_args_ok(N, C, P); if (rc) return rc;
    Bn3Params bp;
    for (int b = 0; b < 3; ++b) { bp.gamma[b] = gamma[b]; bp.beta[b] = beta[b]; bp.running_mean[b] = running_mean[b]; bp.running_var[b] = running_var[b]; }
    if (training)
        hipLaunchKernelGGL(bn3_finalize_fwd, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, global_sums, bp, coef, stats, C,
                           count, count_dev, eps, momentum, update_running);
    else
        hipLaunchKernelGGL(bn3_finalize_eval, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, bp, coef, C, eps);
    const int R = N * C;
    hipLaunchKernelGGL(bn3_apply_fwd, dim3(bn_grid(R, P)), dim3(BN_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3, (const float*)coef, (uint16_t*)out, R, C, P, bn_lpr(P));
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

/* local_sums[C][4] = sum dout, sum dout*(y_b - mean_b) over this rank's batch; stats = the forward's [C][6] (mean_b, invstd_b) */
static int bn3_backward_sums_impl(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats, float* local_sums, float* sums_copy,
                                  int N, int C, int P, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dout || !y1 || !y2 || !y3 || !stats || !local_sums) return SLAK_ERR_INVALID_ARG;
    int rc = bn_args_ok(N, C, P); if (rc) return rc;
    if (!workspace || workspace_bytes < slak_bn3_workspace_bytes(N, C)) return SLAK_ERR_WORKSPACE;
    float* rows = (float*)workspace;
    int S, per; bn_slices(N, C, &S, &per);
    hipLaunchKernelGGL(bn3_chansums<true>, dim3((unsigned)((C * S + BN_THREADS / 64 - 1) / (BN_THREADS / 64))), dim3(BN_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)dout, (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3, stats, rows, N, C, P, S, per);
    hipLaunchKernelGGL(bn3_colreduce, dim3(C), dim3(64), 0, (hipStream_t)stream, (const float*)rows, local_sums, S, C, 4, sums_copy);
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}
int slak_bn3_backward_sums(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats, float* local_sums, int N, int C, int P,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return bn3_backward_sums_impl(dout, y1, y2, y3, stats, local_sums, nullptr, N, C, P, workspace, workspace_bytes, stream);
}
/* the same with a second copy of the sums (the buffer the all-reduce then overwrites in place: the local sums are still needed for the local
 * parameter gradients) -- the copy launch between the sums and the exchange disappears */
int slak_bn3_backward_sums_dup(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats, float* local_sums, float* sums_copy,
                               int N, int C, int P, void* workspace, size_t workspace_bytes, void* stream) {
    if (!sums_copy) return SLAK_ERR_INVALID_ARG;
    return bn3_backward_sums_impl(dout, y1, y2, y3, stats, local_sums, sums_copy, N, C, P, workspace, workspace_bytes, stream);
}

/* Single-process backward: three launches.  dgamma3 / dbeta3: HOST arrays of three device pointers ([C] floats each). */
int slak_bn3_backward_local_to(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats, const float* const* gamma,
                               float* bcoef, float* const* dgamma3, float* const* dbeta3, void* dy1, void* dy2, void* dy3, int N, int C, int P,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!dout || !y1 || !y2 || !y3 || !stats || !gamma || !bcoef || !dgamma3 || !dbeta3 || !dy1 || !dy2 || !dy3) return SLAK_ERR_INVALID_ARG;
    Bn3Grads go;
    for (int b = 0; b < 3; ++b) { go.dgamma[b] = dgamma3[b]; go.dbeta[b] = dbeta3[b]; if (!go.dgamma[b] || !go.dbeta[b]) return SLAK_ERR_INVALID_ARG; }
    int rc = bn_args_ok(N, C, P); if (rc) return rc;
    if (!workspace || workspace_bytes < slak_bn3_workspace_bytes(N, C)) return SLAK_ERR_WORKSPACE;
    Bn3Params bp;
    for (int b = 0; b < 3; ++b) { bp.gamma[b] = gamma[b]; bp.beta[b] = nullptr; bp.running_mean[b] = nullptr; bp.running_var[b] = nullptr; }
    float* rows = (float*)workspace;
    int S, per; bn_slices(N, C, &S, &per);
    hipLaunchKernelGGL(bn3_chansums<true>, dim3((unsigned)((C * S + BN_THREADS / 64 - 1) / (BN_THREADS / 64))), dim3(BN_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)dout, (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3, stats, rows, N, C, P, S, per);
    hipLaunchKernelGGL(bn3_colreduce_finalize_bwd, dim3(C), dim3(64), 0, (hipStream_t)stream, (const float*)rows, S, stats, bp, bcoef, go, C,
                       (float)((double)N * P));
    const int R = N * C;
    hipLaunchKernelGGL(bn3_apply_bwd, dim3(bn_grid(R, P)), dim3(BN_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)dout, (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3, (const float*)bcoef,
                       (uint16_t*)dy1, (uint16_t*)dy2, (uint16_t*)dy3, R, C, P, bn_lpr(P));
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

int slak_bn3_backward_local(const void* dout, const void* y1, const void* y2, const void* y3, const float* stats, const float* const* gamma,
                            float* bcoef, float* dgamma, float* dbeta, void* dy1, void* dy2, void* dy3, int N, int C, int P,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!dgamma || !dbeta || C <= 0) return SLAK_ERR_INVALID_ARG;
    float* const g3[3] = {dgamma, dgamma + C, dgamma + 2 * (size_t)C};
    float* const b3[3] = {dbeta, dbeta + C, dbeta + 2 * (size_t)C};
    return slak_bn3_backward_local_to(dout, y1, y2, y3, stats, gamma, bcoef, g3, b3, dy1, dy2, dy3, N, C, P, workspace, workspace_bytes, stream);
}

/* dgamma3, dbeta3: HOST arrays of three device pointers, [C] floats each (local gradients); bcoef scratch [C][9]; dy1..3 outputs */
int slak_bn3_backward_apply_to(const void* dout, const void* y1, const void* y2, const void* y3, const float* global_sums,
                               const float* local_sums, double count, const double* count_dev, const float* stats, const float* const* gamma,
                               float* bcoef, float* const* dgamma3, float* const* dbeta3, void* dy1, void* dy2, void* dy3, int N, int C, int P, void* stream) {
    if (!dout || !y1 || !y2 || !y3 || !global_sums || !local_sums || !stats || !gamma || !bcoef || !dgamma3 || !dbeta3 || !dy1 || !dy2 || !dy3)
        return SLAK_ERR_INVALID_ARG;
    Bn3Grads go;
    for (int b = 0; b < 3; ++b) { go.dgamma[b] = dgamma3[b]; go.dbeta[b] = dbeta3[b]; if (!go.dgamma[b] || !go.dbeta[b]) return SLAK_ERR_INVALID_ARG; }
    int rc = bn_args_ok(N, C, P); if (rc) return rc;
    Bn3Params bp;
    for (int b = 0; b < 3; ++b) { bp.gamma[b] = gamma[b]; bp.beta[b] = nullptr; bp.running_mean[b] = nullptr; bp.running_var[b] = nullptr; }
    hipLaunchKernelGGL(bn3_finalize_bwd, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, global_sums, local_sums, stats, bp,
                       bcoef, go, C, (float)count, count_dev);
    const int R = N * C;
    hipLaunchKernelGGL(bn3_apply_bwd, dim3(bn_grid(R, P)), dim3(BN_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)dout, (const uint16_t*)y1, (const uint16_t*)y2, (const uint16_t*)y3, (const float*)bcoef,
                       (uint16_t*)dy1, (uint16_t*)dy2, (uint16_t*)dy3, R, C, P, bn_lpr(P));
    SLAK_LAUNCH_CHECK();
    return SLAK_OK;
}

/* dgamma, dbeta: [3][C] */
int slak_bn3_backward_apply(const void* dout, const void* y1, const void* y2, const void* y3, const float* global_sums,
                            const float* local_sums, double count, const double* count_dev, const float* stats, const float* const* gamma,
                            float* bcoef, float* dgamma, float* dbeta, void* dy1, void* dy2, void* dy3, int N, int C, int P, void* stream) {
    if (!dgamma || !dbeta || C <= 0) return SLAK_ERR_INVALID_ARG;
    float* const g3[3] = {dgamma, dgamma + C, dgamma + 2 * (size_t)C};
    float* const b3[3] = {dbeta, dbeta + C, dbeta + 2 * (size_t)C};
    return slak_bn3_backward_apply_to(dout, y1, y2, y3, global_sums, local_sums, count, count_dev, stats, gamma, bcoef, g3, b3, dy1, dy2, dy3, N, C, P, stream);
}

}  // extern "C"
