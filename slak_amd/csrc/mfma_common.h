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
// LDS store of a chunk whose address is only 4-byte aligned (vertical kernels place planes at lane offset 2)
template <int V> __device__ __forceinline__ void chunk_store_lds_a4(uint16_t* p, const chunk_t<V>& c) {
    if constexpr (V == 8) { unsigned* q = (unsigned*)p; q[0] = c.v[0]; q[1] = c.v[1]; q[2] = c.v[2]; q[3] = c.v[3]; }
    else if constexpr (V == 4) { unsigned* q = (unsigned*)p; q[0] = c.v[0]; q[1] = c.v[1]; }
    else if constexpr (V == 2) *(unsigned*)p = c.v;
    else *p = c.v;
}


int mfma_cu_count();

}  // namespace slak
