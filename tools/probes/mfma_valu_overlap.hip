// Do f32 matrix-core instructions and vector-ALU instructions of DIFFERENT waves on one SIMD overlap on MI355X?
// (k_bp_mfma's time equals its matrix-core busy time PLUS its vector-ALU issue time -- profiles/r02d_b16384/summary.md -- which says they do not.)
// Three launches over every SIMD of the chip: 1 wave per SIMD issuing only v_mfma_f32_16x16x4_f32 (independent accumulators), 1 wave per SIMD issuing only
// independent v_fma_f32, and 2 waves per SIMD -- one of each.  Overlap: mixed ~ max(mfma, valu); no overlap: mixed ~ mfma + valu.  The same with the bf16
// instruction v_mfma_f32_16x16x16_bf16 in place of the f32 one, as a control.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int KIND, int PAD>   // KIND 0: f32 16x16x4   1: bf16 16x16x32;  PAD: s_nop wait states issued by the SAME wave after every mfma (0: none)
__device__ __forceinline__ void mfma_loop(float* out, int iters) {
    f4 acc[8];
    for (int j = 0; j < 8; j++) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
    const float a = out[threadIdx.x] + 1.f, b = 0.5f;
    bf8 ab, bb;
    for (int j = 0; j < 8; j++) { ab[j] = (__bf16)a; bb[j] = (__bf16)b; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[j], 0, 0, 0);
            if (PAD) {
                __builtin_amdgcn_sched_barrier(0);
                if (PAD >= 16) asm volatile("s_nop 15");
                if (PAD >= 32) asm volatile("s_nop 15");
                if (PAD % 16) asm volatile("s_nop %0" ::"n"(PAD % 16 ? PAD % 16 - 1 : 0));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__device__ __forceinline__ void valu_loop(float* out, int iters) {
    float v[16];
    for (int j = 0; j < 16; j++) v[j] = out[threadIdx.x] + j;
    const float b = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) v[j & 15] = __builtin_fmaf(v[j & 15], b, c);      // 64 fma per trip, 16 independent chains
    }
    float s = 0.f;
    for (int j = 0; j < 16; j++) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// mode 0: every wave mfma; 1: every wave valu; 2: waves 0-3 mfma, 4-7 valu (block 512: one of each per SIMD)
template <int KIND, int PAD>
__global__ void probe(float* out, int mode, int it_m, int it_v) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = mode == 0 || (mode == 2 && wave < 4);
    if (do_m) mfma_loop<KIND, PAD>(out, it_m); else valu_loop(out, it_v);
}
template <int KIND, int PAD = 0>
static float run(float* d, int mode, int threads, int it_m, int it_v) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL((probe<KIND, PAD>), dim3(256), dim3(threads), 0, 0, d, mode, it_m, it_v); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    float* d; hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24);
    const int it_m = 40000;                       // 8 mfma per trip
    for (int it_v : {10000, 20000, 40000}) {      // 64 fma per trip
        const float m0 = run<0>(d, 0, 256, it_m, it_v), v = run<0>(d, 1, 256, it_m, it_v), x0 = run<0>(d, 2, 512, it_m, it_v);
        const float m1 = run<1>(d, 0, 256, it_m, it_v), x1 = run<1>(d, 2, 512, it_m, it_v);
        printf("valu alone %.3f ms (%.2f cycles per fma at 2.4 GHz) | f32 16x16x4: alone %.3f ms (%.1f cycles each), beside the valu wave %.3f ms (max %.3f, sum %.3f) | "
               "bf16 16x16x32: alone %.3f ms (%.1f cycles each), beside the valu wave %.3f ms (max %.3f, sum %.3f)\n",
               v, v * 2.4e6 / (64.0 * it_v), m0, m0 * 2.4e6 / (8.0 * it_m), x0, m0 > v ? m0 : v, m0 + v, m1, m1 * 2.4e6 / (8.0 * it_m), x1, m1 > v ? m1 : v, m1 + v);
    }
    // the same wave pads every f32 mfma with s_nop wait states (scalar issue: the vector port stays free): does the OTHER wave's vector work get in then?
    {
        const int it_v = 40000;
        const float v = run<0>(d, 1, 256, it_m, it_v);
        const float a8 = run<0, 8>(d, 0, 256, it_m, it_v), x8 = run<0, 8>(d, 2, 512, it_m, it_v);
        const float a16 = run<0, 16>(d, 0, 256, it_m, it_v), x16 = run<0, 16>(d, 2, 512, it_m, it_v);
        const float a24 = run<0, 24>(d, 0, 256, it_m, it_v), x24 = run<0, 24>(d, 2, 512, it_m, it_v);
        const float a28 = run<0, 28>(d, 0, 256, it_m, it_v), x28 = run<0, 28>(d, 2, 512, it_m, it_v);
        const float a32 = run<0, 32>(d, 0, 256, it_m, it_v), x32 = run<0, 32>(d, 2, 512, it_m, it_v);
        printf("f32 mfma padded by its own wave (valu alone %.3f ms): pad 8: alone %.3f mixed %.3f | pad 16: alone %.3f mixed %.3f | pad 24: alone %.3f mixed %.3f | pad 28: alone %.3f mixed %.3f | pad 32: alone %.3f mixed %.3f\n",
               v, a8, x8, a16, x16, a24, x24, a28, x28, a32, x32);
    }
    return 0;
}
