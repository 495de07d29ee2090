// slak_amd/csrc/mfma_common.h -- fragment types, MFMA wrappers, DPP lane shifts and staging-chunk helpers shared by
// the matrix-core depthwise-conv kernels (dwconv_mfma.hip, dwconv_mfma_wgrad.hip).  Lane maps verified on MI355X
// with tools/mfma_probe.hip.
#pragma once
#include "slak_common.h"

namespace slak {

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef s16x8 __attribute__((aligned(2))) s16x8_u;       // LDS vector at a 2-byte-granular address (one ds_read_b128 on gfx950)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define SLAK_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int MF_WAVES = 4;
constexpr int MF_THREADS = MF_WAVES * 64;
constexpr int MF_NCH = 4;              // staging chunks per thread per iteration (upper bound)
constexpr int MF_TAPS = 5;             // short-axis taps the lane-shift epilogue is written for


__device__ __forceinline__ uint16_t cvt_to_bits(float v, bf16_t*) { return f32_to_bf16_bits(v); }
__device__ __forceinline__ uint16_t cvt_to_bits(float v, f16_t*) { f16_t h = (f16_t)v; return __builtin_bit_cast(uint16_t, h); }

template <typename T> __device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma32<bf16_t>(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<f16_t>(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// two fp32 -> packed pair of T (round-to-nearest-even; bf16 via v_cvt_pk_bf16_f32), low half = a
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<bf16_t>(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
template <> __device__ __forceinline__ unsigned pack2<f16_t>(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));
}

// 8 packed T + 8 packed T, each pair added in fp32 and rounded once (a bf16 / fp16 tensor add, as torch does it)
template <typename T> __device__ __forceinline__ u32x4 add_packed(u32x4 a, u32x4 b) {
    u32x4 s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float lo, hi;
        if constexpr (dtype_of<T>::value == SLAK_BF16) {
            lo = __uint_as_float(a[q] << 16) + __uint_as_float(b[q] << 16);
            hi = __uint_as_float(a[q] & 0xffff0000u) + __uint_as_float(b[q] & 0xffff0000u);
        } else {
            lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(a[q] & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (uint16_t)(b[q] & 0xffffu));
            hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(a[q] >> 16)) + (float)__builtin_bit_cast(_Float16, (uint16_t)(b[q] >> 16));
        }
        s[q] = pack2<T>(lo, hi);
    }
    return s;
}

// lane i <- lane i-1 (0 shifted in) / lane i <- lane i+1
__device__ __forceinline__ float wave_shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ float wave_shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false)); }

template <int V> struct chunk_t;
template <> struct chunk_t<8> { u32x4 v; };
template <> struct chunk_t<4> { u32x2 v; };
template <> struct chunk_t<2> { unsigned v; };
template <> struct chunk_t<1> { uint16_t v; };

template <int V> __device__ __forceinline__ chunk_t<V> chunk_zero() { chunk_t<V> c; c.v = {}; return c; }
template <int V> __device__ __forceinline__ chunk_t<V> chunk_load(const uint16_t* p) {
    chunk_t<V> c;
    if constexpr (V == 8) c.v = *(const u32x4*)p;
    else if constexpr (V == 4) c.v = *(const u32x2*)p;
    else if constexpr (V == 2) c.v = *(const unsigned*)p;
    else c.v = *p;
    return c;
}
template <int V> __device__ __forceinline__ void chunk_store(uint16_t* p, const chunk_t<V>& c) {
    if constexpr (V == 8) *(u32x4*)p = c.v;
    else if constexpr (V == 4) *(u32x2*)p = c.v;
    else if constexpr (V == 2) *(unsigned*)p = c.v;
    else *p = c.v;
}
// element access without taking the address of a register value
template <int V> __device__ __forceinline__ uint16_t chunk_get(const chunk_t<V>& c, int i) {
    if constexpr (V == 8) return (uint16_t)(c.v[i >> 1] >> (16 * (i & 1)));
    else if constexpr (V == 4) return (uint16_t)(c.v[i >> 1] >> (16 * (i & 1)));
    else if constexpr (V == 2) return (uint16_t)(c.v >> (16 * (i & 1)));
    else return c.v;
}
template <int V> __device__ __forceinline__ void chunk_set(chunk_t<V>& c, int i, uint16_t val) {
    const unsigned sh = 16 * (i & 1), m = 0xffffu << sh;
    if constexpr (V == 8 || V == 4) {
#pragma unroll
        for (int d = 0; d < V / 2; ++d) if (d == (i >> 1)) c.v[d] = (c.v[d] & ~m) | ((unsigned)val << sh);
    } else if constexpr (V == 2) c.v = (c.v & ~m) | ((unsigned)val << sh);
    else c.v = val;
}
// fp32 operands on the matrix cores (two-term bf16 split): a chunk of V fp32 values and its split x = x_hi + x_lo + O(2^-17 |x|),
// x_hi = bf16(x) (round to nearest even), x_lo = bf16(x - x_hi) (the subtraction is exact).
template <int V> struct fchunk_t { float v[V]; };
template <int V> __device__ __forceinline__ fchunk_t<V> fchunk_zero() { fchunk_t<V> c; for (int i = 0; i < V; ++i) c.v[i] = 0.f; return c; }
template <int V> __device__ __forceinline__ fchunk_t<V> fchunk_load(const float* p) {
    fchunk_t<V> c;
    if constexpr (V == 8) { const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4); for (int i = 0; i < 4; ++i) { c.v[i] = a[i]; c.v[4 + i] = b[i]; } }
    else if constexpr (V == 4) { const f32x4 a = *(const f32x4*)p; for (int i = 0; i < 4; ++i) c.v[i] = a[i]; }
    else if constexpr (V == 2) { const f32x2 a = *(const f32x2*)p; c.v[0] = a[0]; c.v[1] = a[1]; }
    else c.v[0] = *p;
    return c;
}
template <int V> __device__ __forceinline__ void fchunk_store(float* p, const fchunk_t<V>& c) {
    if constexpr (V == 8) { *(f32x4*)p = f32x4{c.v[0], c.v[1], c.v[2], c.v[3]}; *(f32x4*)(p + 4) = f32x4{c.v[4], c.v[5], c.v[6], c.v[7]}; }
    else if constexpr (V == 4) *(f32x4*)p = f32x4{c.v[0], c.v[1], c.v[2], c.v[3]};
    else if constexpr (V == 2) *(f32x2*)p = f32x2{c.v[0], c.v[1]};
    else *p = c.v[0];
}
__device__ __forceinline__ void split_bf16(float x, uint16_t& hi, uint16_t& lo) {
    hi = f32_to_bf16_bits(x);
    lo = ((hi & 0x7f80u) == 0x7f80u) ? (uint16_t)0 : f32_to_bf16_bits(x - __uint_as_float((unsigned)hi << 16));   // inf / NaN: carried by the first term alone
}
template <int V> __device__ __forceinline__ void fchunk_split(const fchunk_t<V>& f, chunk_t<V>& hi, chunk_t<V>& lo) {
    hi = chunk_zero<V>(); lo = chunk_zero<V>();
#pragma unroll
    for (int i = 0; i < V; ++i) { uint16_t h, l; split_bf16(f.v[i], h, l); chunk_set<V>(hi, i, h); chunk_set<V>(lo, i, l); }
}

// LDS store of a chunk whose address is only 4-byte aligned (vertical kernels place planes at lane offset 2)
template <int V> __device__ __forceinline__ void chunk_store_lds_a4(uint16_t* p, const chunk_t<V>& c) {
    if constexpr (V == 8) { unsigned* q = (unsigned*)p; q[0] = c.v[0]; q[1] = c.v[1]; q[2] = c.v[2]; q[3] = c.v[3]; }
    else if constexpr (V == 4) { unsigned* q = (unsigned*)p; q[0] = c.v[0]; q[1] = c.v[1]; }
    else if constexpr (V == 2) *(unsigned*)p = c.v;
    else *p = c.v;
}


// Toeplitz fragments.  A[g][ks] holds, for MFMA row m = lane&31 -> (tap r = g*RPM + m/MPAD, o = mt*32 + m%MPAD) and
// k = ks*16 + (lane>>5)*8 + e -> i, the weight w[r][i - o + padL] (0 outside the filter or the plane), rounded to the
// activation dtype.  They are the same for every workgroup of a channel, so a tiny pre-kernel (toeplitz_pack_kernel) writes
// them to the workspace in register layout -- frag[((c*MT + mt)*NG + g)*KS + ks][lane][8] -- and the conv kernels fetch
// their NG*KS fragments with one 16-byte load each (building them in the conv kernel cost ~40 us per workgroup).
struct ToeplitzPackParams {
    const float* w; uint16_t* frags;
    int C, kh, kw, MT, NG, KS, RPM, vert, flip, Wt, KL, padL, is_bf16;
    uint16_t* frags_lo;    // fp32 operands on the matrix cores: second bf16 term of every filter value (w - bf16(w)), same layout; or NULL
    int NCH;               // short-axis tap chunks of MF_TAPS (1: the K x 5 / 5 x K kernels; > 1: kernels with more than five rows, layout
                           // frag[(((c*MT + mt)*NCH + q)*NG + g)*KS + ks])
};
void launch_toeplitz_pack(const ToeplitzPackParams& p, hipStream_t st);
static inline size_t toeplitz_pack_bytes(int C, int MT, int NG, int KS) { return (size_t)C * MT * NG * KS * 64 * 8 * 2; }

// In-kernel, branch-free construction of the same fragments from this channel's fp32 filter staged in LDS (lw[kh*kw]):
// one clamped ds_read_b32 + select per element, packed with v_cvt_pk.  Used by the kernels whose workgroups need few enough
// fragments that this beats a separate pack launch (~5 us of launch + latency per conv call).
template <typename T, int NG, int KS, bool VERT>
__device__ __forceinline__ void build_toeplitz_frags_lds(s16x8 (&afrag)[NG][KS], bool (&ks_active)[KS], const float* lw,
                                                         int mt, int lane, int Wt, int KL, int padL, int kw, int flip) {
    const int l31 = lane & 31, lhi = lane >> 5, o_abs = mt * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int i_lo = ks * 16, i_hi = ks * 16 + 15, o_lo = mt * 32, o_hi = mt * 32 + 31;
        ks_active[ks] = (i_lo < Wt) && (o_lo < Wt) && (i_lo - o_hi <= KL - 1 - padL) && (o_lo - i_hi <= padL);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i_abs = ks * 16 + lhi * 8 + e;
                int t = i_abs - o_abs + padL;
                const bool ok = o_abs < Wt && i_abs < Wt && t >= 0 && t < KL;
                t = ok ? t : 0;
                int rr = g;
                if (flip) { t = KL - 1 - t; rr = MF_TAPS - 1 - g; }
                const float wv = VERT ? lw[t * kw + rr] : lw[rr * kw + t];
                v[e] = ok ? wv : 0.f;
            }
            u32x4 a;
#pragma unroll
            for (int e = 0; e < 8; e += 2) a[e >> 1] = pack2<T>(v[e], v[e + 1]);
            afrag[g][ks] = __builtin_bit_cast(s16x8, a);
        }
    }
}

template <int NG, int KS>
__device__ __forceinline__ void load_toeplitz_frags_at(s16x8 (&afrag)[NG][KS], bool (&ks_active)[KS], const uint16_t* frags,
                                                       size_t block, int mt, int lane, int MPAD, int Wt, int KL, int padL) {
    const s16x8* base = (const s16x8*)frags + (block * NG * KS) * 64 + lane;       // block = (c*MT + mt) [* NCH + q]
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afrag[g][ks] = base[(g * KS + ks) * 64];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {        // block (mt, ks) intersects the band |i - o| <= padL ?  (wave-uniform)
        const int i_lo = ks * 16, i_hi = ks * 16 + 15, o_lo = mt * 32, o_hi = mt * 32 + MPAD - 1;
        ks_active[ks] = (i_lo < Wt) && (o_lo < Wt) && (i_lo - o_hi <= KL - 1 - padL) && (o_lo - i_hi <= padL);
    }
}

template <int NG, int KS>
__device__ __forceinline__ void load_toeplitz_frags(s16x8 (&afrag)[NG][KS], bool (&ks_active)[KS], const uint16_t* frags,
                                                    int c, int MT, int mt, int lane, int MPAD, int Wt, int KL, int padL) {
    load_toeplitz_frags_at<NG, KS>(afrag, ks_active, frags, (size_t)(c * MT + mt), mt, lane, MPAD, Wt, KL, padL);
}

// ---- LDS-DMA (buffer_load_dwordx4 ... lds) and explicit synchronisation (see dwconv_mfma_dma.hip for the protocol) ----
typedef __attribute__((ext_vector_type(4))) int v4i_t;
__device__ __forceinline__ void lds_dma16(unsigned voff, v4i_t rsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base) : "memory");
}
// NI consecutive 1-KiB pieces (instruction offsets OFF0 + i*1024 advance the HBM source AND the LDS destination) under ONE M0 setup
template <int NI, int OFF0>
__device__ __forceinline__ void lds_dma_run(unsigned voff, v4i_t rsrc, unsigned lds_base) {
    static_assert(NI >= 1 && NI <= 4 && OFF0 + (NI - 1) * 1024 < 4096, "12-bit instruction offset");
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    if constexpr (NI == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%4 lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base), "n"(OFF0) : "memory");
    else if constexpr (NI == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%4 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, 0 offen offset:%5 lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base), "n"(OFF0), "n"(OFF0 + 1024) : "memory");
    else if constexpr (NI == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%4 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, 0 offen offset:%5 lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%6 lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base), "n"(OFF0), "n"(OFF0 + 1024), "n"(OFF0 + 2048) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%4 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, 0 offen offset:%5 lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%6 lds\n\t"
                     "buffer_load_dwordx4 %1, %2, 0 offen offset:%7 lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base), "n"(OFF0), "n"(OFF0 + 1024), "n"(OFF0 + 2048), "n"(OFF0 + 3072) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {       // n is wave-uniform, 0..63 (anything else waits for everything)
#define SLAK_VMC(k) case k: wait_vmcnt<k>(); break;
#define SLAK_VMC8(k) SLAK_VMC(k) SLAK_VMC(k + 1) SLAK_VMC(k + 2) SLAK_VMC(k + 3) SLAK_VMC(k + 4) SLAK_VMC(k + 5) SLAK_VMC(k + 6) SLAK_VMC(k + 7)
    switch (n) {
        SLAK_VMC8(0) SLAK_VMC8(8) SLAK_VMC8(16) SLAK_VMC8(24) SLAK_VMC8(32) SLAK_VMC8(40) SLAK_VMC8(48) SLAK_VMC8(56)
        default: wait_vmcnt<0>(); break;
    }
#undef SLAK_VMC8
#undef SLAK_VMC
}


int mfma_cu_count();

}  // namespace slak
