// How fast does ONE wavefront run on an otherwise idle MI355X?  (single-problem latency work: is the shader clock what it is under load?)
// A chain of dependent v_fma_f32 (8 cycles each under load, tools/probes/valu_rate.hip) timed with HIP events and with s_memtime / wall_clock64.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, int iters, long long* ticks) {
    float a = out[0], b = 1.0000001f, c = 1e-9f;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) a = __builtin_fmaf(a, b, c);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
}
int main() {
    float* d; long long* t; hipMalloc(&d, 1 << 24); hipMalloc(&t, 16); hipMemset(d, 0, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
    for (int blocks : {1, 1, 256, 1024, 1}) {
        for (int rep = 0; rep < 2; rep++) {
            const int iters = 20000;
            hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, iters, t); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            const double n = 64.0 * iters;
            printf("blocks %4d: %.3f ms, %.2f ns per dependent fma, clock64 %.2f ticks/fma, wall_clock64 %.3f ticks/fma (rate %d kHz) -> shader clock ~ %.0f MHz if a dependent fma is 8 cycles\n",
                   blocks, ms, ms * 1e6 / n, h[0] / n, h[1] / n, wcr, 8.0 / (ms * 1e6 / n) * 1e3);
        }
    }
    return 0;
}
