// VALU issue-rate probe for gfx950, double-precision and conversion instructions: ns per wave64 instruction per SIMD at 1, 2, 4 waves per SIMD.
// (the quadrotor plug-in's float build promotes its polynomial arithmetic to double through double literals: ~360 v_mul_f64 / v_add_f64 and ~145 conversions per RK3 step)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/valu_rate_f64 tools/probes/valu_rate_f64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(64) void k(float* out, int iters) {
    double d[16]; float a[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 0.001f + i; d[i] = a[i]; }
    double b = 1.0000001, c = 0.5; float fb = 1.0001f, fc = 0.5f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0)      { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(b));) }
        else if (MODE == 1) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));) }
        else if (MODE == 2) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(b), "v"(c));) }
        else if (MODE == 3) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));) }
        else if (MODE == 4) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));) }
        else if (MODE == 5) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fb));) }
        else if (MODE == 6) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));) }
        else if (MODE == 7) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[0]) : "v"(b));) }          // dependent chain
        else if (MODE == 8) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(fb));) }         // dependent chain
        else if (MODE == 9) { REP8(_Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_cvt_f64_f32 %0, %1\n v_mul_f64 %0, %0, %2\n" : "+v"(d[i]) : "v"(a[i]), "v"(b));) }   // cvt -> mul dependent pairs, 8 independent
        else if (MODE == 10) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(fc));) }
        else if (MODE == 11) { REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(fb));) }
    }
    float s = 0; for (int i = 0; i < 16; i++) s += a[i] + (float)d[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int instr_per_iter) {
    float* out; hipMalloc(&out, 8192 * 64 * 4);
    for (int wps : {1, 2, 4}) {
        int blocks = 1024 * wps, iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<blocks, 64>>>(out, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k<MODE><<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double total_instr = (double)iters * instr_per_iter;
        printf("%-34s waves/SIMD %d: %.3f ms, %.3f ns per wave-instr per SIMD\n", name, wps, ms, ms * 1e6 / (total_instr * wps));
    }
    hipFree(out);
}
int main() {
    run<5>("v_mul_f32 x16 indep", 128); run<8>("v_mul_f32 dependent", 128);
    run<0>("v_mul_f64 x16 indep", 128); run<7>("v_mul_f64 dependent", 128); run<1>("v_add_f64 x16 indep", 128); run<2>("v_fma_f64 x16 indep", 128);
    run<3>("v_cvt_f64_f32 x16 indep", 128); run<4>("v_cvt_f32_f64 x16 indep", 128); run<9>("cvt_f64_f32 -> mul_f64 pairs", 128);
    run<6>("v_rcp_f32 x16 indep", 128); run<10>("v_cndmask_b32 x16 indep", 128); run<11>("v_mad_u64_u32 x16 indep", 128);
    return 0;
}
