// tools/dma_offset_probe.hip -- does the 12-bit instruction offset of `buffer_load_dwordx4 ... lds` advance BOTH the global
// source and the LDS destination?  (run on MI355X).  Expect: lds[(M0 + imm)/2 + 8*lane + e] = src[(voff + imm)/2 + e].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int v4i;
template <int IMM>
__device__ __forceinline__ void dma16_imm(unsigned voff, v4i rsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%4 lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base), "n"(IMM) : "memory");
}
__global__ void k(const uint16_t* in, uint16_t* out, unsigned nbytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0x7777;
  __syncthreads();
  uint64_t a = (uint64_t)in;
  v4i rsrc = {(int)(a & 0xffffffffu), (int)((a >> 32) & 0xffff), (int)nbytes, 0x00020000};
  rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
  rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]);
  unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  dma16_imm<1024>(threadIdx.x * 16, rsrc, __builtin_amdgcn_readfirstlane(ldsb + 64));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
  const int NEL = 4096;
  std::vector<uint16_t> h(NEL), o(4096);
  for (int i = 0; i < NEL; i++) h[i] = 1000 + i;
  uint16_t *din, *dout; hipMalloc(&din, NEL * 2); hipMalloc(&dout, 8192);
  hipMemcpy(din, h.data(), NEL * 2, hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, dout, NEL * 2); hipDeviceSynchronize();
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 4096; i++) if (o[i] != 0x7777) { if (first < 0) first = i; last = i; }
  printf("written LDS elements [%d, %d]; lds[first]=%d (source element %d)\n", first, last, first >= 0 ? o[first] : -1, first >= 0 ? o[first] - 1000 : -1);
  printf("hypothesis A (imm advances both): first=%d src=%d ; B (imm global only): first=32 src=512 ; C (LDS only): first=544 src=0\n", 32 + 512, 512);
  return 0;
}
