// Per-instruction issue cost on gfx950 for the VALU operations the closed-form plant kernels are made of: ns (and cycles at the device's reported clock) per wave64
// instruction per SIMD with 2 waves per SIMD, 16 independent destination registers.  The table goes to profiles/r04_valu_ops.md.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/valu_ops tools/probes/valu_ops.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define REP8(X) X X X X X X X X
#define BODY(ASM, ...) REP8(_Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM __VA_ARGS__);)
template <int MODE> __global__ __launch_bounds__(64) void k(float* out, int iters) {
    float a[16], c[16]; double d[16]; unsigned u[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 0.001f + i; c[i] = a[i] + 1.f; d[i] = a[i]; u[i] = threadIdx.x * 7 + i; }
    float fb = 1.0001f, fc = 0.5f; unsigned ub = 3; double db = 1.0000001, dc = 0.5;
    asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_mov_b64 s[10:11], vcc" :: "v"(a[0]), "v"(fc) : "vcc", "s10", "s11");
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { BODY("v_mul_f32 %0, %0, %1", : "+v"(a[i]) : "v"(fb)) }
        else if (MODE == 1) { BODY("v_fma_f32 %0, %0, %1, %2", : "+v"(a[i]) : "v"(fb), "v"(fc)) }
        else if (MODE == 2) { BODY("v_cndmask_b32 %0, %0, %1, vcc", : "+v"(a[i]) : "v"(fc)) }
        else if (MODE == 3) { BODY("v_cndmask_b32_e64 %0, %0, %1, s[10:11]", : "+v"(a[i]) : "v"(fc)) }
        else if (MODE == 4) { BODY("v_cmp_gt_f32 vcc, %0, %1", :: "v"(a[i]), "v"(fc) : "vcc") }
        else if (MODE == 5) { BODY("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc", : "+v"(a[i]) : "v"(c[i]), "v"(fc) : "vcc") }
        else if (MODE == 6) { BODY("v_cmp_gt_f32_e64 s[12:13], %1, %2\n v_cndmask_b32_e64 %0, %0, %2, s[12:13]", : "+v"(a[i]) : "v"(c[i]), "v"(fc) : "s12", "s13") }
        else if (MODE == 7) { BODY("v_add_u32 %0, %0, %1", : "+v"(u[i]) : "v"(ub)) }
        else if (MODE == 8) { BODY("v_lshlrev_b32 %0, 1, %0", : "+v"(u[i])) }
        else if (MODE == 9) { BODY("v_and_b32 %0, %0, %1", : "+v"(u[i]) : "v"(ub)) }
        else if (MODE == 10) { BODY("v_mul_lo_u32 %0, %0, %1", : "+v"(u[i]) : "v"(ub)) }
        else if (MODE == 11) { BODY("v_mul_hi_u32 %0, %0, %1", : "+v"(u[i]) : "v"(ub)) }
        else if (MODE == 12) { BODY("v_alignbit_b32 %0, %0, %1, 5", : "+v"(u[i]) : "v"(ub)) }
        else if (MODE == 13) { BODY("v_max_f32 %0, %0, %1", : "+v"(a[i]) : "v"(fc)) }
        else if (MODE == 14) { BODY("v_sqrt_f32 %0, %0", : "+v"(a[i])) }
        else if (MODE == 15) { BODY("v_sin_f32 %0, %0", : "+v"(a[i])) }
        else if (MODE == 16) { BODY("v_rcp_f32 %0, %0", : "+v"(a[i])) }
        else if (MODE == 17) { BODY("v_mul_f64 %0, %0, %1", : "+v"(d[i]) : "v"(db)) }
        else if (MODE == 18) { BODY("v_fma_f64 %0, %0, %1, %2", : "+v"(d[i]) : "v"(db), "v"(dc)) }
        else if (MODE == 19) { BODY("v_rcp_f64 %0, %0", : "+v"(d[i])) }
        else if (MODE == 20) { BODY("v_cvt_f64_f32 %0, %1", : "=v"(d[i]) : "v"(a[i])) }
        else if (MODE == 21) { BODY("v_mov_b32 %0, %1", : "=v"(a[i]) : "v"(c[i])) }
        else if (MODE == 22) { BODY("v_pk_fma_f32 %0, %0, %1, %1", : "+v"(d[i]) : "v"(d[(i + 1) & 15])) }
        else if (MODE == 23) { BODY("v_and_b32 %0, 0x7fffffff, %0", : "+v"(a[i])) }     // |x| as the compiler writes it
        else if (MODE == 24) { BODY("v_add_f32 %0, |%0|, %1", : "+v"(a[i]) : "v"(fc)) }  // |x| as a source modifier
        else if (MODE == 25) { BODY("v_bfe_u32 %0, %0, 3, 5", : "+v"(u[i])) }
        else if (MODE == 26) { BODY("v_cndmask_b32 %0, %0, %2, vcc\n v_mul_f32 %1, %1, %3", : "+v"(a[i]), "+v"(c[i]) : "v"(fc), "v"(fb)) }   // alternating with a multiply: 2 instructions
        else if (MODE == 27) { BODY("v_div_scale_f32 %0, vcc, %0, %1, %0", : "+v"(a[i]) : "v"(fb) : "vcc") }
        else if (MODE == 28) { BODY("v_div_fmas_f32 %0, %0, %1, %1", : "+v"(a[i]) : "v"(fb)) }
        else if (MODE == 29) { BODY("v_div_fixup_f32 %0, %0, %1, %1", : "+v"(a[i]) : "v"(fb)) }
        else if (MODE == 30) { BODY("v_rndne_f32 %0, %0", : "+v"(a[i])) }
        else if (MODE == 31) { BODY("v_cvt_i32_f32 %0, %1", : "=v"(u[i]) : "v"(a[i])) }
        else if (MODE == 32) { BODY("v_fract_f32 %0, %0", : "+v"(a[i])) }
        else if (MODE == 33) { BODY("v_xor_b32 %0, %0, %1", : "+v"(u[i]) : "v"(ub)) }
        else if (MODE == 34) { BODY("v_cmp_class_f32 vcc, %0, %1", :: "v"(a[i]), "v"(ub) : "vcc") }
        else if (MODE == 35) { BODY("v_ldexp_f32 %0, %0, %1", : "+v"(a[i]) : "v"(ub)) }
        else if (MODE == 36) { BODY("v_readfirstlane_b32 s12, %0", :: "v"(a[i]) : "s12") }
        else if (MODE == 37) { BODY("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", : "+v"(a[i]) : "v"(c[i])) }
        else if (MODE == 38) { BODY("v_add_f64 %0, %0, %1", : "+v"(d[i]) : "v"(dc)) }
        else if (MODE == 39) { BODY("v_cvt_f32_f64 %0, %1", : "=v"(a[i]) : "v"(d[i])) }
    }
    float s = 0; for (int i = 0; i < 16; i++) s += a[i] + c[i] + (float)d[i] + (float)u[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
static double g_clk_ghz = 0;
template <int MODE> void run(const char* name, int instr_per_iter) {
    float* out; (void)hipMalloc(&out, 8192 * 64 * 4);
    const int wps = 2, blocks = 1024 * wps, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, 10); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) { (void)hipEventRecord(e0); k<MODE><<<blocks, 64>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    const double ns = best * 1e6 / ((double)iters * instr_per_iter * wps);
    printf("%-46s %7.3f ns  %6.2f cycles @%.2f GHz\n", name, ns, ns * g_clk_ghz, g_clk_ghz);
    (void)hipFree(out);
}
int main() {
    int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0); g_clk_ghz = khz / 1e6;
    printf("# ns per wave64 instruction per SIMD, 2 waves per SIMD, 16 independent registers; cycles at the device's reported peak clock\n");
    run<0>("v_mul_f32", 128); run<1>("v_fma_f32", 128); run<22>("v_pk_fma_f32", 128); run<21>("v_mov_b32", 128); run<13>("v_max_f32", 128);
    run<2>("v_cndmask_b32 (vcc)", 128); run<3>("v_cndmask_b32_e64 (sgpr pair)", 128); run<26>("v_cndmask_b32 + v_mul_f32 (per pair)", 128);
    run<4>("v_cmp_gt_f32 vcc", 128); run<34>("v_cmp_class_f32 vcc", 128); run<5>("v_cmp_gt_f32 + v_cndmask (vcc, per pair)", 128); run<6>("v_cmp_e64 + v_cndmask_e64 (sgpr, per pair)", 128);
    run<7>("v_add_u32", 128); run<8>("v_lshlrev_b32", 128); run<9>("v_and_b32", 128); run<33>("v_xor_b32", 128); run<25>("v_bfe_u32", 128); run<12>("v_alignbit_b32", 128);
    run<10>("v_mul_lo_u32", 128); run<11>("v_mul_hi_u32", 128);
    run<23>("v_and_b32 0x7fffffff (abs)", 128); run<24>("v_add_f32 with |src| modifier", 128);
    run<14>("v_sqrt_f32", 128); run<15>("v_sin_f32", 128); run<16>("v_rcp_f32", 128); run<30>("v_rndne_f32", 128); run<32>("v_fract_f32", 128); run<31>("v_cvt_i32_f32", 128); run<35>("v_ldexp_f32", 128);
    run<27>("v_div_scale_f32", 128); run<28>("v_div_fmas_f32", 128); run<29>("v_div_fixup_f32", 128);
    run<17>("v_mul_f64", 128); run<38>("v_add_f64", 128); run<18>("v_fma_f64", 128); run<19>("v_rcp_f64", 128); run<20>("v_cvt_f64_f32", 128); run<39>("v_cvt_f32_f64", 128);
    run<36>("v_readfirstlane_b32", 128); run<37>("v_mov_b32_dpp quad_perm", 128);
    return 0;
}
