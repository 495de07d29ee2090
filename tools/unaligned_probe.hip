// dev probe: do 8-byte / 4-byte global stores and buffer stores work at 2-byte aligned addresses on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__global__ void k(uint16_t* p) {
    const int l = threadIdx.x;
    // lane l writes 4 halfwords {l*4+0..3} at element offset 1 + l*49 (odd element offsets for odd l: 2-byte aligned only)
    u32x2 v; v[0] = (unsigned)(l * 4) | ((unsigned)(l * 4 + 1) << 16); v[1] = (unsigned)(l * 4 + 2) | ((unsigned)(l * 4 + 3) << 16);
    *(u32x2*)(p + 1 + l * 49) = v;
}
__global__ void kb(uint16_t* p, unsigned bytes) {
    const int l = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000);
    u32x2 v; v[0] = (unsigned)(l * 4) | ((unsigned)(l * 4 + 1) << 16); v[1] = (unsigned)(l * 4 + 2) | ((unsigned)(l * 4 + 3) << 16);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (unsigned)(1 + l * 49) * 2, 0, 0);
}
int main() {
    uint16_t* d; const int n = 64 * 49 + 16;
    hipMalloc(&d, n * 2);
    uint16_t* h = (uint16_t*)malloc(n * 2);
    for (int variant = 0; variant < 2; ++variant) {
        hipMemset(d, 0xff, n * 2);
        if (variant == 0) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); else hipLaunchKernelGGL(kb, dim3(1), dim3(64), 0, 0, d, (unsigned)(n * 2));
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h, d, n * 2, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (h[1 + l * 49 + j] != (uint16_t)(l * 4 + j)) ++bad;
        int clobber = 0;
        for (int l = 0; l < 64; ++l) if (h[1 + l * 49 + 4] != 0xffff || h[l * 49] != 0xffff) ++clobber;
        printf("variant %d (%s): err=%d wrong=%d clobbered=%d\n", variant, variant ? "buffer_store_b64" : "global_store_dwordx2", (int)e, bad, clobber);
    }
    return 0;
}
