// slak_amd/csrc/tri_wgrad_common.h -- shared by the one-launch three-branch weight-gradient kernels that run ONE WAVE PER SIMD with fifteen named
// accumulators (dwconv_mfma_tri_wgrad_rows.hip: planes of 2 x 2 MFMA tiles; dwconv_mfma_tri_wgrad_wave.hip: planes of one tile).
#pragma once
#include "mfma_common.h"

namespace slak {

// the five shifted operands from a six-dword window: tap r = the eight 16-bit elements that start r elements behind the window's first
__device__ __forceinline__ void tw_taps(s16x8 (&b)[MF_TAPS], unsigned d0, unsigned d1, unsigned d2, unsigned d3, unsigned d4, unsigned d5) {
    auto sh = [](unsigned hi, unsigned lo) -> unsigned { return __builtin_amdgcn_alignbit(hi, lo, 16); };
    b[0] = __builtin_bit_cast(s16x8, u32x4{d0, d1, d2, d3});
    b[1] = __builtin_bit_cast(s16x8, u32x4{sh(d1, d0), sh(d2, d1), sh(d3, d2), sh(d4, d3)});
    b[2] = __builtin_bit_cast(s16x8, u32x4{d1, d2, d3, d4});
    b[3] = __builtin_bit_cast(s16x8, u32x4{sh(d2, d1), sh(d3, d2), sh(d4, d3), sh(d5, d4)});
    b[4] = __builtin_bit_cast(s16x8, u32x4{d2, d3, d4, d5});
}

// The fifteen accumulators are the accumulator registers a[0:239], NAMED in the instruction text: a[0:79] the vertical branch's five taps,
// a[80:159] the small branch's, a[160:239] the horizontal one's.  (As C++ values -- builtin MFMAs, or inline asm with "+a" operands -- hipcc
// carries them over the loop's back edge in VGPRs and copies sixteen registers in and out around every MFMA: 480 v_accvgpr moves per
// plane.)  Registers written literally belong to the kernel only because tw_acc_claim() lists them as clobbers (that also makes the kernel
// descriptor allocate them); the compiler's own code stays below 256 VGPRs and never touches the accumulator file -- audited in the ISA
// after every edit: no v_accvgpr_* outside ASMSTART / ASMEND, no scratch.  Each string carries its own wait states (hipcc pads nothing
// inside an asm statement): a VALU-written A / B operand -> MFMA needs two; an MFMA's D is only read by the next MFMA that takes it whole as
// C (none) and, after the loop and a barrier, by tw_acc_read (16 states in front of the first read).
constexpr int TW_ACC_V = 0, TW_ACC_S = 80, TW_ACC_H = 160;
__device__ __forceinline__ void tw_acc_claim() {
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239");
}
template <int LO, int HI> __device__ __forceinline__ void tw_acc_zero() {
    if constexpr (LO < HI) {
        asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0"
                     :: "i"(LO), "i"(LO + 1), "i"(LO + 2), "i"(LO + 3));
        tw_acc_zero<LO + 4, HI>();
    }
}
// (s_nop 1 in EVERY string: hipcc is free to sink a v_perm / v_mov that forms a B operand down to right in front of the MFMA that reads it --
// it did, and without the two wait states that MFMA read the register's previous content: wrong sums that changed from run to run)
template <typename T, int BASE> __device__ __forceinline__ void tw_mfma(s16x8 a, s16x8 b) {
    if constexpr (dtype_of<T>::value == SLAK_BF16)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(BASE), "i"(BASE + 15));
    else
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(BASE), "i"(BASE + 15));
}
template <int BASE> __device__ __forceinline__ void tw_acc_read(float (&v)[16]) {
    asm volatile("s_nop 15\n\t"
                 "v_accvgpr_read_b32 %0, a[%c16]\n\tv_accvgpr_read_b32 %1, a[%c17]\n\tv_accvgpr_read_b32 %2, a[%c18]\n\tv_accvgpr_read_b32 %3, a[%c19]\n\t"
                 "v_accvgpr_read_b32 %4, a[%c20]\n\tv_accvgpr_read_b32 %5, a[%c21]\n\tv_accvgpr_read_b32 %6, a[%c22]\n\tv_accvgpr_read_b32 %7, a[%c23]\n\t"
                 "v_accvgpr_read_b32 %8, a[%c24]\n\tv_accvgpr_read_b32 %9, a[%c25]\n\tv_accvgpr_read_b32 %10, a[%c26]\n\tv_accvgpr_read_b32 %11, a[%c27]\n\t"
                 "v_accvgpr_read_b32 %12, a[%c28]\n\tv_accvgpr_read_b32 %13, a[%c29]\n\tv_accvgpr_read_b32 %14, a[%c30]\n\tv_accvgpr_read_b32 %15, a[%c31]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]),
                   "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15])
                 : "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3), "i"(BASE + 4), "i"(BASE + 5), "i"(BASE + 6), "i"(BASE + 7),
                   "i"(BASE + 8), "i"(BASE + 9), "i"(BASE + 10), "i"(BASE + 11), "i"(BASE + 12), "i"(BASE + 13), "i"(BASE + 14), "i"(BASE + 15));
}

// diagonal sums of the five taps of one branch through the wave's skewed tile (G[o][i] -> row o, column i - o + 31: a diagonal is a column).
// ONE copy of the scatter / column-sum code in a run-time loop over the taps (the register NAMES are compile-time: a switch picks the read).
// lim: per lane, how many of its rows are inside the image (0 for a lane whose column is outside): entries beyond are written as zeros, so
// nothing of a padded row / column -- finite or not -- reaches a sum and the stores need no exec masking.  lo: the lane's rows below it are
// written as zeros too (a block of the tile that another tile counts).
template <int BASE>
__device__ __forceinline__ void tw_diag5(float* tile, float* wr, int lim, int lane, int dtau, int KL, float* out, int s_tau, int s_g, int lo = -64) {
#pragma unroll 1
    for (int g = 0; g < MF_TAPS; ++g) {
        float v[16];
        switch (g) {
            case 0: tw_acc_read<BASE + 0>(v); break;  case 1: tw_acc_read<BASE + 16>(v); break; case 2: tw_acc_read<BASE + 32>(v); break;
            case 3: tw_acc_read<BASE + 48>(v); break; default: tw_acc_read<BASE + 64>(v); break;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * 63] = ((r & 3) + 8 * (r >> 2) < lim && (r & 3) + 8 * (r >> 2) >= lo) ? v[r] : 0.f;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 63) {
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 32; ++o) part[o & 3] += tile[o * 64 + lane];
            const int tau = lane + dtau;
            if (tau >= 0 && tau < KL) out[tau * s_tau + g * s_g] = (part[0] + part[1]) + (part[2] + part[3]);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

}  // namespace slak
