// Which SIMD does each wavefront of a workgroup land on?  (HW_REG_HW_ID: wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13])
// build: hipcc --offload-arch=gfx950 -O2 -o wave_simd wave_simd.hip     usage: ./wave_simd [threads per workgroup] [dynamic LDS bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned* out) {
    extern __shared__ float lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
    if (threadIdx.x == 9999) lds[0] = 1.f;
}
int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 256, lds = argc > 2 ? atoi(argv[2]) : 0;
    unsigned* d; hipMalloc(&d, 4 * 16 * sizeof(unsigned)); hipMemset(d, 0xff, 4 * 16 * sizeof(unsigned));
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(4), dim3(threads), lds, 0, d);
    unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; b++) {
        printf("workgroup %d:", b);
        for (int w = 0; w < threads / 64; w++) printf("  wave %d -> SIMD %u (CU %u, SE %u, slot %u)", w, (h[b * 16 + w] >> 4) & 3, (h[b * 16 + w] >> 8) & 15, (h[b * 16 + w] >> 13) & 7, h[b * 16 + w] & 15);
        printf("\n");
    }
    return 0;
}
