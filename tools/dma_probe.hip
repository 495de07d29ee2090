// tools/dma_probe.hip -- LDS-DMA (buffer_load_dwordx4 ... lds) facts used by the MFMA conv kernels (run on MI355X):
//  destination = M0 base + lane*16; inactive lanes write nothing; out-of-range source (incl. "negative" offsets) gives zeros;
//  4-byte-aligned source offsets are fine.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int v4i;
__device__ __forceinline__ void dma16(unsigned voff, v4i rsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base) : "memory");
}
__global__ void k(const uint16_t* in, uint16_t* out, unsigned nbytes, int shift, int nact) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0x7777;
  __syncthreads();
  uint64_t a = (uint64_t)in;
  v4i rsrc = {(int)(a & 0xffffffffu), (int)((a >> 32) & 0xffff), (int)nbytes, 0x00020000};
  rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
  rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]);
  unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  if ((int)threadIdx.x < nact) dma16(threadIdx.x * 16 - shift * 2, rsrc, __builtin_amdgcn_readfirstlane(ldsb + 64));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
  const int NEL = 1024;             // source elements; buffer range covers only the first 256 (512 bytes)
  std::vector<uint16_t> h(NEL), o(2048);
  for (int i = 0; i < NEL; i++) h[i] = 1000 + i;
  uint16_t *din, *dout; hipMalloc(&din, NEL * 2 + 4096); hipMalloc(&dout, 4096);
  hipMemcpy(din + 1024, h.data(), NEL * 2, hipMemcpyHostToDevice);       // keep slack in front: base = din + 1024
  for (int shift = 0; shift <= 3; shift += 1) {
    k<<<1, 64>>>(din + 1024, dout, 512, shift, 40); hipDeviceSynchronize();
    printf("shift=%d err=%s\n", shift, hipGetErrorString(hipGetLastError()));
    hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
    // expect: lds[32 + 8*l + e] = src[8*l + e - shift] for l < 40 if 0 <= byte offset < 512 else 0 ; untouched = 0x7777
    int bad = 0, firstbad = -1;
    for (int i = 0; i < 2048; i++) {
      int exp = 0x7777;
      if (i >= 32 && i < 32 + 40 * 8) { int s = i - 32 - shift; exp = (s >= 0 && s < 256) ? 1000 + s : 0; }
      if (o[i] != exp) { bad++; if (firstbad < 0) firstbad = i; }
    }
    printf("  mismatches=%d first=%d  lds[30..42]:", bad, firstbad);
    for (int i = 30; i < 43; i++) printf(" %d", o[i]);
    printf("  lds[280..296]:"); for (int i = 280; i < 297; i++) printf(" %d", o[i]);
    printf("  lds[350..356]:"); for (int i = 350; i < 357; i++) printf(" %d", o[i]);
    printf("\n");
  }
  return 0;
}
