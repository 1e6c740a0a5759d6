// VALU issue-rate probe for gfx950: cycles per wave64 instruction for v_fma_f32 / v_pk_fma_f32 / dependent chains, at 1, 2, 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/valu_rate tools/probes/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE> __global__ __launch_bounds__(64) void k(float* out, int iters, long long* cyc) {
    float a[16]; f2 p[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    float b = 1.0001f, c = 0.5f; f2 pb = {b, b}, pc = {c, c};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {        // 16 independent v_fma_f32 x 8
            REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));)
        } else if (MODE == 1) { // 16 independent v_pk_fma_f32 x 8
            REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));)
        } else if (MODE == 2) { // dependent chain of v_fma_f32 (1 accumulator) x 128
            REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));)
        } else if (MODE == 3) { // dependent v_mul_f32 -> v_add_f32 pairs on one accumulator (contract off shape)
            REP8(_Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %1, %2, %3\n v_add_f32 %0, %0, %1" : "+v"(a[0]), "+v"(a[1]) : "v"(b), "v"(c));)
        } else if (MODE == 4) { // 2 interleaved dependent chains
            REP8(_Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a[0]), "+v"(a[1]) : "v"(b), "v"(c));)
        } else if (MODE == 5) { // v_fma with DPP-broadcast-like operand: v_mov_b32_dpp + v_fma
            REP8(_Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_mov_b32_dpp %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_fma_f32 %0, %0, %1, %3" : "+v"(a[i]), "+v"(a[8 + i]) : "v"(b), "v"(c));)
        } else if (MODE == 6) { // independent ds_read_b32 (uniform address) + fma
            extern __shared__ float lds[];
            REP8(_Pragma("unroll") for (int i = 0; i < 8; i++) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(0), "n"(4 * i)); asm volatile("s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v), "v"(c)); })
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int instr_per_iter) {
    float* out; long long* cyc; hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&cyc, 8192 * 8);
    for (int wps : {1, 2, 4}) {
        int blocks = 1024 * wps, iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<blocks, 64, 1024>>>(out, 10, cyc); hipDeviceSynchronize();
        hipEventRecord(e0); k<MODE><<<blocks, 64, 1024>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        double total_instr = (double)iters * instr_per_iter;
        printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (clock64 %.2f ticks/instr/wave)\n", name, wps, ms, ms * 1e6 / (total_instr * wps), (double)h[0] / total_instr);
    }
}
int main() {
    run<0>("v_fma_f32 x16 indep", 128); run<1>("v_pk_fma_f32 x16 indep", 128); run<2>("v_fma_f32 dependent", 128);
    run<3>("v_mul+v_add dependent", 128); run<4>("2 dependent fma chains", 128); run<5>("v_mov_dpp + v_fma", 128); run<6>("ds_read_b32 + waitcnt + fma", 128);
    return 0;
}
