// parallel-ddp_amd: common device/host scaffolding for the wave-cooperative kernels (gfx950).
//
// Execution model.  Every unit of work of the DDP hot path (one backward-pass block, one forward
// shooting segment, one knot's derivatives) is owned by ONE 64-lane wavefront.  The reference assigns
// a 56-thread block with __syncthreads() between micro-stages (DDPWrappers.cuh:27-29); on CDNA4 that
// whole block fits one wave, so stage boundaries become wave-local LDS fences (no s_barrier traffic)
// and a workgroup can hold several independent waves whose loops have different trip counts.
//
// Code is written as a sequence of STAGES.  Inside a stage every lane writes disjoint LDS/registers and
// reads only what earlier stages produced; stages are separated by wsync().  The same source compiles
// for the host with a 1-lane "wave" (lane loops run serially), which is how tests/hostsim checks the
// kernel arithmetic on a machine without a GPU.  The product library never uses that mode.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
// The device code is written for gfx950 and nothing else: workgroups of up to 160 KB of LDS (k_nis_kb, k_fp_tl4, k_nis_tl, the sweep staging: several exceed the 64 KB of
// gfx90a / gfx942), v_mfma_f32_16x16x4_f32 / _f64, LDS-direct buffer loads, DPP row_newbcast.  A build for another target fails HERE, with this message, instead of
// at some kernel's LDS allocation (the Makefile takes ARCH as a parameter).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libpddp's kernels are written for gfx950 (MI355X): build with --offload-arch=gfx950"
#endif
#define PDDP_HD __host__ __device__ __forceinline__
#define PDDP_D __device__ __forceinline__
#else
#define PDDP_HD inline
#define PDDP_D inline
#endif

namespace pddp {

constexpr int kWave = 64;  // CDNA wavefront width

struct Wave {
    int lane;    // first index this lane handles
    int nlanes;  // stride between indices (64 on the GPU, 1 in host emulation; 256 for a whole-workgroup "wave")
    int block;   // 1: the cooperating lanes are a whole workgroup, stage boundaries are s_barrier
};

PDDP_HD Wave this_wave() {
#if defined(__HIP_DEVICE_COMPILE__)
    return Wave{static_cast<int>(threadIdx.x) & (kWave - 1), kWave, 0};
#else
    return Wave{0, 1, 0};
#endif
}
// every thread of the workgroup as one cooperating set (used where a unit of work has more independent outputs per stage than a
// wave has lanes and the latency of ONE unit matters: the backward pass of a single problem)
PDDP_HD Wave this_block() {
#if defined(__HIP_DEVICE_COMPILE__)
    return Wave{static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x), 1};
#else
    return Wave{0, 1, 0};
#endif
}
// Wave-local stage boundary: LDS operations of one wave are issued and retired in order, so only the
// compiler has to be stopped from moving LDS traffic across the boundary.
PDDP_HD void wsync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// stage boundary of the cooperating set `w`
PDDP_HD void wsync(const Wave& w) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (w.block) __syncthreads(); else wsync();
#else
    (void)w;
#endif
}

#define PDDP_FOR(i, n) for (int i = w.lane; i < (n); i += w.nlanes)

template <typename T> PDDP_HD T tsin(T v);
template <typename T> PDDP_HD T tcos(T v);
// float sin/cos: the reference's host instantiation calls glibc sinf/cosf, which return the correctly rounded float
// in all but a vanishing fraction of arguments; ocml's device sinf/cosf are 1-2 ulp functions, and the mass-matrix
// solve behind them amplifies that to ~5e-5 in qdd.  On the device we therefore evaluate in double and round once
// (7 lanes x 2 calls per dynamics evaluation -- not on any critical resource).
template <> PDDP_HD float tsin<float>(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<float>(sin(static_cast<double>(v)));
#else
    return sinf(v);
#endif
}
template <> PDDP_HD float tcos<float>(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<float>(cos(static_cast<double>(v)));
#else
    return cosf(v);
#endif
}
template <> PDDP_HD double tsin<double>(double v) { return sin(v); }
template <> PDDP_HD double tcos<double>(double v) { return cos(v); }
// Sine and cosine of one angle together, for the CLOSED-FORM plants (pendulum, cart-pole, quadrotor: 1 / 2 / 6 angles per dynamics evaluation, three evaluations per RK3
// step -- with tsin / tcos above, i.e. double-precision sin and cos rounded to float, they were ~3/4 of the quadrotor rollout kernel's instructions).  float on the device:
// ocml's single-precision sincosf (<= 2 units in the last place -- what the reference's device code gets from CUDA's sinf / cosf); double: sincos.  The host instantiations
// (CPU entry points, the test tool) keep libm's sinf / cosf / sin / cos.  The arm keeps tsin / tcos: its mass-matrix solve amplifies a last-place difference (above).
template <typename T> PDDP_HD void cf_sincos(T v, T& s, T& c);
// (round 4, late: ocml's sincosf reduces its argument in DOUBLE on this target -- two conversions and five double-precision operations per call on a chip whose
// double rate is a quarter of the float rate, plus a Payne-Hanek branch with a 1 KB scratch table that every wave pays for at launch.  The device form is now the
// branch-free float evaluation the arm's thread-lane kernels use (plant_arm_tl.hpp tl_sincos): two-term Cody-Waite reduction by pi/2 through fused multiply-adds,
// degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4], ~1 unit in the last place for the few radians an attitude or a pole angle takes.)
PDDP_HD void cf_sincos_poly(float q, float& s, float& c) {
    const float k = rintf(q * 0.636619772367581343f);
    float r = fmaf(k, -1.57079637050628662109375f, q);
    r = fmaf(k, 4.37113900018624283e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
    const int n = static_cast<int>(k) & 3;
    const float ss = (n & 1) ? cp : sp, cc = (n & 1) ? sp : cp;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
template <> PDDP_HD void cf_sincos<float>(float v, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PDDP_CF_TRIG_OCML)
    cf_sincos_poly(v, s, c);
#elif defined(__HIP_DEVICE_COMPILE__)
    sincosf(v, &s, &c);
#else
    s = sinf(v); c = cosf(v);
#endif
}
// cos(2 v) for the quadrotor's inertia terms (dynamics_quad.cuh:60): 2 v is exact in either precision, so the float device form is cosf of the same argument
template <typename T> PDDP_HD T cf_cos2(T v) {
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(PDDP_CF_TRIG_OCML)
    if constexpr (sizeof(T) == 4) return cosf(2.0f * v);
#else
    if constexpr (sizeof(T) == 4) { float s2, c2; cf_sincos_poly(2.0f * static_cast<float>(v), s2, c2); return c2; }
#endif
#endif
    return static_cast<T>(cos(2.0 * v));
}
template <> PDDP_HD void cf_sincos<double>(double v, double& s, double& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(v, &s, &c);
#else
    s = sin(v); c = cos(v);
#endif
}
// atan2 / sqrt of the end-effector cost family (ee_cost.hpp): atan2 like sin/cos above, sqrt is correctly rounded on both sides
template <typename T> PDDP_HD T tatan2(T y, T x);
template <> PDDP_HD float tatan2<float>(float y, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<float>(atan2(static_cast<double>(y), static_cast<double>(x)));   // as tsin/tcos: rounded once from double
#else
    return atan2f(y, x);
#endif
}
template <> PDDP_HD double tatan2<double>(double y, double x) { return atan2(y, x); }
template <typename T> PDDP_HD T tsqrt(T v);
template <> PDDP_HD float tsqrt<float>(float v) { return sqrtf(v); }
template <> PDDP_HD double tsqrt<double>(double v) { return sqrt(v); }

template <typename T> PDDP_HD T tabs(T v) { return v < T(0) ? -v : v; }
template <typename T> PDDP_HD T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> PDDP_HD T tmin(T a, T b) { return a < b ? a : b; }

// ---- spatial 6-vector helpers ([angular; linear]) on register arrays -----------------------------
template <typename T> PDDP_HD void cross3(T* o, const T* a, const T* b) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// o = crm(a) b   (spatial motion cross product)
template <typename T> PDDP_HD void crm_mul(T* o, const T* a, const T* b) {
    T t[3];
    cross3(o, a, b);
    cross3(o + 3, a, b + 3);
    cross3(t, a + 3, b);
    o[3] += t[0]; o[4] += t[1]; o[5] += t[2];
}
// o = crf(a) f   (spatial force cross product, crf = -crm^T)
template <typename T> PDDP_HD void crf_mul(T* o, const T* a, const T* f) {
    T t[3];
    cross3(o, a, f);
    cross3(t, a + 3, f + 3);
    o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
    cross3(o + 3, a, f + 3);
}
// o = A v, A column-major 6x6
template <typename T> PDDP_HD void mat6_mul(T* o, const T* A, const T* v) {
    for (int r = 0; r < 6; r++) {
        T s = 0;
        for (int c = 0; c < 6; c++) s += A[r + 6 * c] * v[c];
        o[r] = s;
    }
}
template <typename T> PDDP_HD T dot6(const T* a, const T* b) {
    T s = 0;
    for (int i = 0; i < 6; i++) s += a[i] * b[i];
    return s;
}

// Problem dimensions shared by every kernel.  Runtime values (the reference fixes them at compile time,
// config.cuh:90-94,113-115,133-136).
struct Dims {
    int N;   // knots (NUM_TIME_STEPS)
    int M;   // shooting segments == backward blocks (M_BLOCKS_F == M_BLOCKS_B)
    int A;   // line-search step sizes (NUM_ALPHA)
    int NB;  // knots per block, N / M
    PDDP_HD bool on_defect_boundary(int k) const { return (((k + 1) % NB) == 0) && (k < N - 1); }  // config.cuh:127
};

}  // namespace pddp
