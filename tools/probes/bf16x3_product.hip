// Can the float32 products of the matrix-core backward pass (bp_mfma.hpp: C + X'Y on 16 x 16 tiles in accumulator layout) leave the SIMD's float32 lanes?
// v_mfma_f32_16x16x4_f32 executes on them (tools/probes/mfma_valu_overlap.hip); the bf16 matrix instructions have their own pipe.  A float32 value splits EXACTLY into
// three bf16 parts (hi = rne(x), mid = rne(x - hi), lo = x - hi - mid: 8 + 8 + 8 significand bits), and the six leading cross terms of a product
// (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid; the dropped ones are <= 2^-26 of |x||y|) are three bf16 instructions with float32 accumulation:
//   full tiles (K = 16 rows):  v_mfma_f32_16x16x32_bf16 -- its 8 K slots per lane take TWO terms at once
//   half tiles (K = 8: control rows, or the velocity rows of the Euler step): v_mfma_f32_16x16x16_bf16 likewise
// This probe measures (1) the error of that product against float64 next to the float32 instruction's, (2) the issue time of the instructions involved and
// (3) the time of a dependent chain "split both operands, multiply" against the four-instruction float32 chain, at 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk(float a, float b) { f2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2)); }
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk(r0, r1);
    l = cvt_pk(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
// the same split with integer rounding (v_cvt_pk_bf16_f32 turned out to cost ~8 cycles): hi = (x + 0x8000) & 0xffff0000 (nearest, ties away), two levels, the parts packed by v_perm_b32.
// P layout [h m l h], Q layout [h m h h l] of a register PAIR (x0, x1): a half product (K = 8 rows) is then X.P[0:4] . Y.Q[1:5] (hi mid + mid hi + lo hi + hi lo, one K = 32
// instruction) + X.P[0:2] . Y.Q[0:2] (hi hi + mid mid, one K = 16 instruction) without any register copies
typedef unsigned u5 __attribute__((ext_vector_type(5)));
__device__ __forceinline__ unsigned rn_hi(float x) { return (__float_as_uint(x) + 0x8000u) & 0xffff0000u; }
__device__ __forceinline__ unsigned pk_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // {a.hi16, b.hi16}
__device__ __forceinline__ void split2i(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned h0 = rn_hi(x0), h1 = rn_hi(x1);
    const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
    const unsigned m0 = rn_hi(r0), m1 = rn_hi(r1);
    const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
    h = pk_hi(h0, h1); m = pk_hi(m0, m1); l = pk_hi(__float_as_uint(s0), __float_as_uint(s1));
}
__device__ __forceinline__ u4 split_P(float x0, float x1) { unsigned h, m, l; split2i(x0, x1, h, m, l); return u4{h, m, l, h}; }
__device__ __forceinline__ u5 split_Q(float x0, float x1) { unsigned h, m, l; split2i(x0, x1, h, m, l); return u5{h, m, h, h, l}; }
__device__ __forceinline__ f4 mfma32(u4 a, u4 b, f4 c);
__device__ __forceinline__ f4 mfma16(u2 a, u2 b, f4 c);
__device__ __forceinline__ f4 prod_half(u4 X, u5 Y, f4 c) {
    c = mfma32(X, __builtin_shufflevector(Y, Y, 1, 2, 3, 4), c);
    c = mfma16(__builtin_shufflevector(X, X, 0, 1), __builtin_shufflevector(Y, Y, 0, 1), c);
    return c;
}
// full tile: [H0 H1 M0 M1 L0 L1 H0 H1] (H0 = registers 0, 1 of the tile, H1 = registers 2, 3)
__device__ __forceinline__ u8 split_full(f4 x) {
    unsigned h0, m0, l0, h1, m1, l1;
    split2(x[0], x[1], h0, m0, l0); split2(x[2], x[3], h1, m1, l1);
    return u8{h0, h1, m0, m1, l0, l1, h0, h1};
}
__device__ __forceinline__ f4 mfma32(u4 a, u4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0); }
__device__ __forceinline__ f4 mfma16(u2 a, u2 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf4, a), __builtin_bit_cast(bf4, b), c, 0, 0, 0); }
#define W4(v, o) __builtin_shufflevector(v, v, o, o + 1, o + 2, o + 3)
// C + X'Y, six terms: (L H | H M) + (H M | L H) + (H M | H M) -- smallest first
__device__ __forceinline__ f4 prod_full(u8 X, u8 Y, f4 c) {
    c = mfma32(W4(X, 4), W4(Y, 0), c);       // lo hi + hi mid
    c = mfma32(W4(X, 0), W4(Y, 4), c);       // hi lo + mid hi
    c = mfma32(W4(X, 0), W4(Y, 0), c);       // hi hi + mid mid
    return c;
}
__device__ __forceinline__ f4 prod_f32(f4 X, f4 Y, f4 c) {
    for (int r = 0; r < 4; r++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(X[r], Y[r], c, 0, 0, 0);
    return c;
}
// ---- (1) accuracy: tiles in accumulator layout, lane (g, c) register r <-> element [4g + r][c]
__global__ void k_acc(const float* X, const float* Y, float* o32, float* obf, float* ohalf) {
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    f4 x, y;
    for (int r = 0; r < 4; r++) { x[r] = X[(4 * g + r) * 16 + c]; y[r] = Y[(4 * g + r) * 16 + c]; }
    const f4 z = {0, 0, 0, 0};
    const f4 a = prod_f32(x, y, z), b = prod_full(split_full(x), split_full(y), z);
    // half product: registers 0, 1 only (rows 4g, 4g + 1)
    unsigned hx, mx, lx, hy, my, ly;
    split2(x[0], x[1], hx, mx, lx); split2(y[0], y[1], hy, my, ly);
    f4 h = mfma16(u2{lx, hx}, u2{hy, my}, z);
    h = mfma16(u2{hx, mx}, u2{ly, hy}, h);
    h = mfma16(u2{hx, mx}, u2{hy, my}, h);
    for (int r = 0; r < 4; r++) { o32[(4 * g + r) * 16 + c] = a[r]; obf[(4 * g + r) * 16 + c] = b[r]; ohalf[(4 * g + r) * 16 + c] = h[r]; }
}
// ---- (2), (3) timing
template <int KIND>
__global__ __launch_bounds__(256) void k_time(float* out, int iters) {
    f4 acc[4];
    const float s = out[threadIdx.x];
    f4 x = {1.f + s, 0.5f + s, 0.25f - s, 2.f + s};
    for (int j = 0; j < 4; j++) acc[j] = f4{s, s, s, s};
    const u8 xs = split_full(x);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[0], x[1], acc[j], 0, 0, 0);                       // issue: float32
            if (KIND == 1) acc[j] = mfma32(W4(xs, 0), W4(xs, 2), acc[j]);                                                 // issue: bf16 K = 32
            if (KIND == 2) acc[j] = mfma16(u2{xs[0], xs[1]}, u2{xs[2], xs[3]}, acc[j]);                                  // issue: bf16 K = 16
            if (KIND == 3) acc[j] = prod_f32(acc[j], x, f4{0, 0, 0, 0});                                                  // dependent chain: float32 product of the previous result
            if (KIND == 4) acc[j] = prod_full(split_full(acc[j]), xs, f4{0, 0, 0, 0});                                    // the same through the split (one operand split per product)
            if (KIND == 5) acc[j] = prod_full(split_full(acc[j]), split_full(acc[(j + 1) & 3]), f4{0, 0, 0, 0});          // both operands split
            if (KIND == 6) { const u8 t = split_full(acc[j]); for (int r = 0; r < 4; r++) acc[j][r] = __uint_as_float((t[r] ^ t[r + 4]) | 0x3f000000u) ; }   // the vector part of KIND 4 alone (+ 8 logic operations to keep it live)
            if (KIND == 8) { const u4 t = split_P(acc[j][0], acc[j][1]); const u5 q = split_Q(acc[j][2], acc[j][3]); acc[j][0] = __uint_as_float((t[0] ^ t[1]) | 0x3f000000u); acc[j][1] = __uint_as_float((t[2] ^ t[3]) | 0x3f000000u); acc[j][2] = __uint_as_float((q[0] ^ q[1]) | 0x3f000000u); acc[j][3] = __uint_as_float((q[2] ^ q[4]) | 0x3f000000u); }   // integer split of two pairs alone (+ 8 logic operations)
            if (KIND == 9) { acc[j] = prod_half(split_P(acc[j][0], acc[j][1]), split_Q(acc[(j + 1) & 3][2], acc[(j + 1) & 3][3]), f4{0, 0, 0, 0}); }           // half product, both operands split: 2 matrix instructions
            if (KIND == 10) { f4 c = prod_half(split_P(acc[j][0], acc[j][1]), split_Q(x[0], x[1]), f4{0, 0, 0, 0}); acc[j] = prod_half(split_P(acc[j][2], acc[j][3]), split_Q(x[2], x[3]), c); }   // full product as two half products, one operand loop-invariant
            if (KIND == 7) acc[j] = prod_full(xs, xs, acc[j]);                                                            // the matrix part alone: 3 dependent bf16 instructions per product
        }
    }
    float t = 0.f;
    for (int j = 0; j < 4; j++) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
// ---- (4) do the bf16 matrix instructions of one wave overlap with the vector instructions of ANOTHER wave on the same SIMD -- and does it matter whether the matrix
// instructions are issued as dependent chains (the same accumulator back to back, as a product issues them) or independent ones?  512-thread workgroups: waves 0-3
// and 4-7 land on the four SIMDs pairwise.  mode 0: every wave matrix work; 1: every wave vector work; 2: waves 0-3 matrix, 4-7 vector.
template <int DEP>
__global__ __launch_bounds__(512) void k_mix(float* out, int mode, int iters) {
    const int wave = threadIdx.x >> 6;
    const float s = out[threadIdx.x];
    const bool matrix = mode == 0 || (mode == 2 && wave < 4);
    float t = 0.f;
    if (matrix) {
        const u4 xp = split_P(1.f + s, 0.5f - s); const u5 xq = split_Q(0.25f + s, 2.f - s);
        f4 acc[8];
        for (int j = 0; j < 8; j++) acc[j] = f4{s, s, s, s};
        for (int i = 0; i < iters; i++) {
            if (DEP) {
#pragma unroll
                for (int j = 0; j < 4; j++) { acc[j] = prod_half(xp, xq, acc[j]); acc[j] = prod_half(xp, xq, acc[j]); }      // 4 dependent instructions per accumulator, back to back
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = mfma32(xp, __builtin_shufflevector(xq, xq, 1, 2, 3, 4), acc[j]);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = mfma16(__builtin_shufflevector(xp, xp, 0, 1), __builtin_shufflevector(xq, xq, 0, 1), acc[j]);
            }
        }
        for (int j = 0; j < 8; j++) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f4 acc[4];
        for (int j = 0; j < 4; j++) acc[j] = f4{1.f + s, 2.f + s, 3.f + s, 4.f + s};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) { const u4 tt = split_P(acc[j][0], acc[j][1]); const u5 q = split_Q(acc[j][2], acc[j][3]); acc[j][0] = __uint_as_float((tt[0] ^ tt[1]) | 0x3f000000u); acc[j][1] = __uint_as_float((tt[2] ^ tt[3]) | 0x3f000000u); acc[j][2] = __uint_as_float((q[0] ^ q[1]) | 0x3f000000u); acc[j][3] = __uint_as_float((q[2] ^ q[4]) | 0x3f000000u); }
        }
        for (int j = 0; j < 4; j++) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <int DEP> static float run_mix(float* d, int mode, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(d, 0, 1 << 22);
        hipEventRecord(e0); hipLaunchKernelGGL((k_mix<DEP>), dim3(256 * 2), dim3(512), 0, 0, d, mode, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
template <int KIND> static float run(float* d, int threads, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(d, 0, 1 << 22);
        hipEventRecord(e0); hipLaunchKernelGGL((k_time<KIND>), dim3(256 * 4), dim3(threads), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    std::vector<float> X(256), Y(256), o32(256), obf(256), oh(256);
    float *dX, *dY, *d32, *dbf, *dh;
    hipMalloc(&dX, 1024); hipMalloc(&dY, 1024); hipMalloc(&d32, 1024); hipMalloc(&dbf, 1024); hipMalloc(&dh, 1024);
    srand(7);
    double w32 = 0, wbf = 0, wh = 0, s32 = 0, sbf = 0; int cnt = 0;
    for (int trial = 0; trial < 200; trial++) {
        const double scale = std::pow(10.0, (trial % 7) - 3);
        for (int i = 0; i < 256; i++) { X[i] = (float)(scale * (2.0 * rand() / RAND_MAX - 1.0)); Y[i] = (float)((2.0 * rand() / RAND_MAX - 1.0) / scale * (1 + trial % 3)); }
        hipMemcpy(dX, X.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dY, Y.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dX, dY, d32, dbf, dh);
        hipMemcpy(o32.data(), d32, 1024, hipMemcpyDeviceToHost); hipMemcpy(obf.data(), dbf, 1024, hipMemcpyDeviceToHost); hipMemcpy(oh.data(), dh, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            double ref = 0, mag = 0, refh = 0, magh = 0;
            for (int k = 0; k < 16; k++) { const double t = (double)X[k * 16 + i] * Y[k * 16 + j]; ref += t; mag += std::fabs(t); if ((k & 3) < 2) { refh += t; magh += std::fabs(t); } }
            const double e32 = std::fabs(o32[i * 16 + j] - ref) / mag, ebf = std::fabs(obf[i * 16 + j] - ref) / mag, eh = std::fabs(oh[i * 16 + j] - refh) / magh;
            if (e32 > w32) w32 = e32; if (ebf > wbf) wbf = ebf; if (eh > wh) wh = eh;
            s32 += e32; sbf += ebf; cnt++;
        }
    }
    printf("accuracy |C - X'Y| / sum|x y| over %d elements: float32 instruction worst %.3g mean %.3g | bf16 x 3 (six terms, K = 32) worst %.3g mean %.3g | half product (K = 16 form) worst %.3g   [2^-24 = %.3g]\n",
           cnt, w32, s32 / cnt, wbf, sbf / cnt, wh, std::pow(2.0, -24));
    float* d; hipMalloc(&d, 1 << 22);
    const int it = 20000;
    const double cyc = 2.4e6 / (4.0 * it) / 4.0;      // ms -> cycles per (instruction or product) and wave-slot: 4 per trip; 4 waves per SIMD share the pipes, so / 4 gives SIMD cycles per item
    const float t0 = run<0>(d, 256, it), t1 = run<1>(d, 256, it), t2 = run<2>(d, 256, it);
    printf("issue, one wave per SIMD (256-thread workgroups, 4 per CU -> 4 waves per SIMD): f32 16x16x4 %.3f ms = %.1f SIMD cycles each | bf16 16x16x32 %.3f ms = %.1f | bf16 16x16x16 %.3f ms = %.1f\n",
           t0, t0 * cyc, t1, t1 * cyc, t2, t2 * cyc);
    const float t3 = run<3>(d, 256, it), t4 = run<4>(d, 256, it), t5 = run<5>(d, 256, it);
    printf("products at 4 waves per SIMD: float32 mfma4 %.3f ms = %.1f SIMD cycles per product | split one operand + 3 bf16 %.3f ms = %.1f | split both + 3 bf16 %.3f ms = %.1f\n",
           t3, t3 * cyc, t4, t4 * cyc, t5, t5 * cyc);
    const float t6 = run<6>(d, 256, it), t7 = run<7>(d, 256, it);
    printf("parts of the one-operand product: vector part alone %.3f ms = %.1f SIMD cycles | matrix part alone %.3f ms = %.1f | together (above) %.1f: overlap if ~max, none if ~sum\n",
           t6, t6 * cyc, t7, t7 * cyc, t4 * cyc);
    const float t8 = run<8>(d, 256, it), t9 = run<9>(d, 256, it), t10 = run<10>(d, 256, it);
    printf("integer-rounded split: two pairs alone %.3f ms = %.1f SIMD cycles (incl. 8 logic operations) | half product, both operands split (2 x 15 vector + 2 matrix instructions) %.3f ms = %.1f | full product as two half products, one operand split (30 vector + 4 matrix) %.3f ms = %.1f\n",
           t8, t8 * cyc, t9, t9 * cyc, t10, t10 * cyc);
    for (int dep = 0; dep < 2; dep++) {
        const float m = dep ? run_mix<1>(d, 0, it) : run_mix<0>(d, 0, it), v = dep ? run_mix<1>(d, 1, it) : run_mix<0>(d, 1, it), x = dep ? run_mix<1>(d, 2, it) : run_mix<0>(d, 2, it);
        printf("two workgroups of 8 waves per CU, %s matrix instructions (16 per trip): all waves matrix %.3f ms | all waves vector (integer split, 4 x 42 per trip) %.3f ms | half matrix, half vector %.3f ms (perfect overlap: %.3f, none: %.3f)\n",
               dep ? "DEPENDENT (same accumulator back to back)" : "independent", m, v, x, (m > v ? m : v) / 2, (m + v) / 2);
    }
    return 0;
}
