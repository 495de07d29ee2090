// tools/mfma_probe.hip -- hardware facts the MFMA depthwise-conv kernels rely on (run on MI355X):
//   1. operand / accumulator lane maps of v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16
//   2. ds_read_b64_tr_b16 source->destination map, and whether 2/4/6-byte misaligned addresses work
//   3. whether misaligned ds_read_b64 / ds_read_b128 return the right bytes
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDSP(T, p) ((__attribute__((address_space(3))) T*)(p))

__global__ void k_tr(uint16_t* out, int off) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s16x4, (char*)lds + threadIdx.x * 8 + off));
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
__global__ void k_mis(uint16_t* out, int off) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned addr = (unsigned)(uintptr_t)LDSP(char, (char*)lds) + threadIdx.x * 32 + off;
  u32x2 a; u32x4 b;
  asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
  asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
  uint16_t* o = out + threadIdx.x * 12;
  o[0] = a[0] & 0xffff; o[1] = a[0] >> 16; o[2] = a[1] & 0xffff; o[3] = a[1] >> 16;
  for (int j = 0; j < 4; j++) { o[4 + 2 * j] = b[j] & 0xffff; o[5 + 2 * j] = b[j] >> 16; }
}
__global__ void k_mfma(const __bf16* a, const __bf16* b, float* d) {
  bf16x8 av, bv;
  for (int j = 0; j < 8; j++) { av[j] = a[threadIdx.x * 8 + j]; bv[j] = b[threadIdx.x * 8 + j]; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  for (int j = 0; j < 16; j++) d[threadIdx.x * 16 + j] = acc[j];
  f32x4 acc2 = {0};
  acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc2, 0, 0, 0);
  for (int j = 0; j < 4; j++) d[1024 + threadIdx.x * 4 + j] = acc2[j];
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
int main() {
  uint16_t* d_out; hipMalloc(&d_out, 64 * 16 * 2);
  std::vector<uint16_t> h(64 * 12);
  for (int off = 0; off <= 6; off += 2) {
    k_tr<<<1, 64>>>(d_out, off); hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    hipMemcpy(h.data(), d_out, 64 * 4 * 2, hipMemcpyDeviceToHost);
    // expected (guide): lane l elem j == (l&15) + j*16 + (l>>4)*64 (+ off/2)
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) if (h[l * 4 + j] != (l & 15) + j * 16 + (l >> 4) * 64 + off / 2) bad++;
    printf("tr_b16 off=%d err=%s mismatches_vs_guide_map=%d  lane0: %d %d %d %d lane1: %d %d %d %d lane17: %d %d %d %d\n", off, hipGetErrorString(e), bad,
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[68], h[69], h[70], h[71]);
  }
  for (int off = 0; off <= 6; off += 2) {
    k_mis<<<1, 64>>>(d_out, off); hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    hipMemcpy(h.data(), d_out, 64 * 12 * 2, hipMemcpyDeviceToHost);
    int bad64 = 0, bad128 = 0;
    for (int l = 0; l < 64; l++) {
      for (int j = 0; j < 4; j++) if (h[l * 12 + j] != l * 16 + off / 2 + j) bad64++;
      for (int j = 0; j < 8; j++) if (h[l * 12 + 4 + j] != l * 16 + off / 2 + j) bad128++;
    }
    printf("misaligned off=%d err=%s b64_bad=%d b128_bad=%d  lane1 b64: %d %d %d %d  b128: %d %d %d %d %d %d %d %d\n", off, hipGetErrorString(e), bad64, bad128,
           h[12], h[13], h[14], h[15], h[16], h[17], h[18], h[19], h[20], h[21], h[22], h[23]);
  }
  // MFMA maps: A[m][k] = small ints, B[k][n] asymmetric
  {
    std::vector<float> A(32 * 32), B(32 * 32);
    for (int i = 0; i < 32; i++) for (int k = 0; k < 32; k++) { A[i * 32 + k] = (float)((i * 3 + k * 5) % 7 - 3); B[k * 32 + i] = (float)((k * 2 + i * 7) % 5 - 2); }
    std::vector<uint16_t> ha(512), hb(512);
    // 32x32x16 assumed: A lane l: m=l&31, k=(l>>5)*8+j ; B lane l: n=l&31, k=(l>>5)*8+j
    for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) { int k = (l >> 5) * 8 + j; ha[l * 8 + j] = f2bf(A[(l & 31) * 32 + k]); hb[l * 8 + j] = f2bf(B[k * 32 + (l & 31)]); }
    uint16_t *da, *db; float* dd; hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, (1024 + 256) * 4);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>((const __bf16*)da, (const __bf16*)db, dd); hipDeviceSynchronize();
    std::vector<float> hd(1024 + 256); hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
      int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float ref = 0; for (int k = 0; k < 16; k++) ref += A[row * 32 + k] * B[k * 32 + col];
      if (hd[l * 16 + r] != ref) bad++;
    }
    printf("mfma 32x32x16 bf16 map mismatches=%d\n", bad);
    // 16x16x32 assumed: A lane l: m=l&15, k=(l>>4)*8+j ; B lane: n=l&15, k=(l>>4)*8+j ; D: col=l&15,row=(l>>4)*4+r
    // the same registers were fed, so reinterpret: a-reg of lane l holds A32[(l&31)][(l>>5)*8+j]
    bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
      int col = l & 15, row = (l >> 4) * 4 + r;
      float ref = 0;
      for (int kk = 0; kk < 32; kk++) {
        int la = (kk >> 3) * 16 + row, lb = (kk >> 3) * 16 + col, j = kk & 7;   // lanes that hold (row,kk) / (kk,col) under the assumed map
        float av = A[(la & 31) * 32 + (la >> 5) * 8 + j], bv = B[((lb >> 5) * 8 + j) * 32 + (lb & 31)];
        ref += av * bv;
      }
      if (hd[1024 + l * 4 + r] != ref) bad++;
    }
    printf("mfma 16x16x32 bf16 map mismatches=%d\n", bad);
  }
  return 0;
}
