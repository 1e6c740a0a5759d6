// A SERIAL arm rollout as a pipeline over the wavefronts of one workgroup (few problems in flight: one MPC solve, the warm-start rollout of a control cycle).
//
// With few rollouts only the LENGTH of one step's dependent instruction stream counts (a lone wave issues a dependent instruction every ~6-8 cycles).  One Euler step is
//     control law -> (sines / cosines -> Newton-Euler bias | composite bodies -> mass matrix -> L D L') -> two triangular solves -> x_{k+1}
// and k_fp_tl2 ran bias and factors side by side on two waves with two barriers per step (~1100 instructions per step).  But Euler's position update
// q_{k+1} = q_k + dt qd_k does not wait for the dynamics of step k: the FACTORS of step k + 1 -- sines / cosines, composite bodies, mass matrix, L D L' (~1100
// instructions, the longest piece) -- can start as soon as x_k exists, a whole step ahead of the chain that needs them.  The workgroup's waves therefore take ROLES and meet through monotonic step counters in LDS (no barrier; every wave sits on its own SIMD):
//     chain wave     x_k -> bias with the posted sines -> torque with the posted control -> solve with the posted factors -> x_{k+1} -> post             ~650 instructions per step
//     factor waves   two, alternating steps: wait for x_{k-1}, q_k = Euler, post sines / cosines, post L, D^-1                                         (two steps of time each)
//     control wave   (closed-loop rollouts) picks x_k up, evaluates the control law (its operands prefetched a step ahead) and posts u_k long before the chain wave has its
//                    bias; stores the trajectory, adds up the cost
// Same arithmetic as the one-thread rollout (fp_tl.hpp, plant_arm_tl.hpp): the control law is tl_control_law, the Euler update ONE function with an explicit fused
// multiply-add, so every wave that forms q_{k+1} gets the chain's bits.
// Tried and dropped: splitting the bias recursion itself over two waves (outer links | inner links, one hand-over per step) -- a hand-over through LDS costs what ~50
// dependent instructions cost, the step did not get shorter (72 -> 75 us for 32 steps); packed float32 instructions (SLP vectorisation on) make the chain SLOWER (100 us).
// Replaces forwardSimKern's inner loop (fpHelpers.cuh:225-301) for few rollouts and rolloutMPCKern (MPCHelpers.cuh:520-560).
#pragma once

#include "plant_arm_tl.hpp"

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#pragma clang fp contract(fast)

namespace pddp {

// LDS strides per lane in ELEMENTS: 4 x odd, so that 16-byte accesses of 8 consecutive lanes cover the 32 banks exactly once (float; the double instantiation is the parity build)
constexpr int kPipeSX = 20, kPipeSL = 44, kPipeSU = 12;
constexpr int kPipeXbuf = 2 * 64 * kPipeSX, kPipeLbuf = 2 * 64 * kPipeSL, kPipeUbuf = 2 * 64 * kPipeSU;
constexpr int kPipeFlags = 8;                                                      // x, u, (spare), cs[2], l[2]
template <typename T> constexpr int pipe_lds_bytes(bool closed_loop) { return (kPipeXbuf + kPipeLbuf + (closed_loop ? kPipeUbuf : 0)) * (int)sizeof(T) + kPipeFlags * 4; }
constexpr int kPipeLdsOpenLoop = pipe_lds_bytes<float>(false);                      // bytes: chain + factor waves
constexpr int kPipeLdsClosedLoop = pipe_lds_bytes<float>(true);

// the step counters are accessed as LDS (address space 3) explicitly: through a generic volatile pointer the compiler emits FLAT loads / stores with system-coherence bits --
// every poll and post then takes the flat path instead of a ds_read / ds_write (found in the ISA: 73 -> 6x us per 32 steps)
typedef __attribute__((address_space(3))) volatile int tl_pipe_flag;
// T = float: the production kernels.  T = double: the same pipeline as a parity instantiation (PDDP_FP=tl4 on a double handle; tests/test_f64_benched_family.py).
template <typename T>
struct TlPipeLdsT {
    T* xbuf;                // [2][64][20]   x_k at slot k & 1
    T* lbuf;                // [2][64][44]   step k's c[7] s[7] (2 pad) L[21] Dinv[7] at slot k & 1
    T* ubuf;                // [2][64][12]   u_k (closed loop)
    tl_pipe_flag* flag;     // [0] x: v = x_v is in xbuf   [1] u: v = u_{v-1} is in ubuf   [3 + r] cs, [5 + r] l: v = step v-1's are in lbuf[r]
};
using TlPipeLds = TlPipeLdsT<float>;
template <typename T>
__device__ __forceinline__ TlPipeLdsT<T> tl_pipe_lds(T* base, bool closed_loop) {
    TlPipeLdsT<T> p;
    p.xbuf = base; p.lbuf = p.xbuf + kPipeXbuf; p.ubuf = p.lbuf + kPipeLbuf;
    p.flag = (tl_pipe_flag*)(closed_loop ? p.ubuf + kPipeUbuf : p.ubuf);
    return p;
}
template <typename T> struct TlPipeVec { typedef T v4 __attribute__((ext_vector_type(4), aligned(16))); };
// n elements (a multiple of 4 slots are touched) between registers and a 16-byte aligned LDS run
template <int N4, typename T> __device__ __forceinline__ void tl_pipe_ld(T* dst, const T* src) {
    typedef typename TlPipeVec<T>::v4 V4;
#pragma unroll
    for (int i = 0; i < N4; i++) { const V4 v = reinterpret_cast<const V4*>(src)[i]; dst[4 * i] = v[0]; dst[4 * i + 1] = v[1]; dst[4 * i + 2] = v[2]; dst[4 * i + 3] = v[3]; }
}
template <int N4, typename T> __device__ __forceinline__ void tl_pipe_st(T* dst, const T* src) {
    typedef typename TlPipeVec<T>::v4 V4;
#pragma unroll
    for (int i = 0; i < N4; i++) { V4 v; v[0] = src[4 * i]; v[1] = src[4 * i + 1]; v[2] = src[4 * i + 2]; v[3] = src[4 * i + 3]; reinterpret_cast<V4*>(dst)[i] = v; }
}
__device__ __forceinline__ void tl_pipe_wait(tl_pipe_flag* f, int v) {
    while (*f < v) {}
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void tl_pipe_post(tl_pipe_flag* f, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *f = v;
}
__device__ __forceinline__ float tl_pipe_euler(float a, float b, float dt) { return __builtin_fmaf(dt, b, a); }      // Euler (utils/integrators.cuh:24-36)
__device__ __forceinline__ double tl_pipe_euler(double a, double b, double dt) { return __builtin_fma(dt, b, a); }

// factor wave r (0 / 1): the sines / cosines and the factors of M(q_j) for the steps j = r, r + 2, ... < njobs.  x0[14]: the rollout's start state.
// xout != null: lane 0 also stores the states it picks up (x_{j-1}, j >= 2) to xout[14 (j - 1)] -- global stores kept off the chain wave, whose posts would wait for them.
template <int V, typename T>
__device__ __forceinline__ void tl_pipe_factor_wave(const TlPipeLdsT<T>& p, int r, int njobs, const T* x0, T dt, int lane, T* xout = nullptr) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    T* o = p.lbuf + (r * 64 + lane) * kPipeSL;
    for (int j = r; j < njobs; j += 2) {
        T q[7];
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = x0[i];
        } else if (j == 1) {
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = tl_pipe_euler(x0[i], x0[7 + i], dt);
        } else {
            tl_pipe_wait(p.flag + 0, j - 1);
            T xv[16];
            tl_pipe_ld<4>(xv, p.xbuf + ((((j - 1) & 1) * 64) + lane) * kPipeSX);
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = tl_pipe_euler(xv[i], xv[7 + i], dt);
            if (xout && lane == 0) {
#pragma unroll
                for (int i = 0; i < 14; i++) xout[14 * (j - 1) + i] = xv[i];
            }
        }
        ArmTlState<T> st;
        arm_tl_trig<T>(st, q);
        T cs[16], ld[28];
#pragma unroll
        for (int i = 0; i < 7; i++) { cs[i] = st.c[i]; cs[7 + i] = st.s[i]; }
        cs[14] = T(0); cs[15] = T(0);
        tl_pipe_st<4>(o, cs);
        tl_pipe_post(p.flag + 3 + r, j + 1);
        arm_tl_factor<T>(md, st);
#pragma unroll
        for (int e = 0; e < 21; e++) ld[e] = st.L[e];
#pragma unroll
        for (int i = 0; i < 7; i++) ld[21 + i] = st.Dinv[i];
        tl_pipe_st<7>(o + 16, ld);
        tl_pipe_post(p.flag + 5 + r, j + 1);
    }
}

// chain wave, step k: x[14] (in: x_k, out: x_{k+1}), u[7] the control of this step.  Posts x_{k+1}.
// CLOSED: the control of the step is the control wave's (posted through ubuf after it completed the control law on x_k -- while this wave computes the bias); else u[7].
template <int V, bool CLOSED, typename T>
__device__ __forceinline__ void tl_pipe_chain_step(const TlPipeLdsT<T>& p, int k, T* x, const T* u, T dt, T grav, int lane) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    const int r = k & 1;
    const T* o = p.lbuf + (r * 64 + lane) * kPipeSL;
    ArmTlState<T> st;
    tl_pipe_wait(p.flag + 3 + r, k + 1);
    T cs[16], ld[28];
    tl_pipe_ld<4>(cs, o);
#pragma unroll
    for (int i = 0; i < 7; i++) { st.c[i] = cs[i]; st.s[i] = cs[7 + i]; }
    T bias[7], qdd[7];
    arm_tl_bias<T>(md, grav, st, x + 7, bias);
    // the waits below are volatile LDS reads, the bias is register arithmetic: without this the compiler sinks the whole recursion BELOW the waits (cycle stamps: the chain
    // then sat at the control wave's post for 1.1 k cycles and at the factors' with its bias still to do -- 6.9 k cycles per step instead of 4.x k)
#pragma unroll
    for (int i = 0; i < 7; i++) asm volatile("" : "+v"(bias[i]));
    if (CLOSED) {
        T uv[8];
        tl_pipe_wait(p.flag + 1, k + 1);
        tl_pipe_ld<2>(uv, p.ubuf + (((k & 1) * 64) + lane) * kPipeSU);
#pragma unroll
        for (int i = 0; i < 7; i++) qdd[i] = uv[i] - bias[i];
    } else {
#pragma unroll
        for (int i = 0; i < 7; i++) qdd[i] = u[i] - bias[i];
    }
    tl_pipe_wait(p.flag + 5 + r, k + 1);
    tl_pipe_ld<7>(ld, o + 16);
#pragma unroll
    for (int e = 0; e < 21; e++) st.L[e] = ld[e];
#pragma unroll
    for (int i = 0; i < 7; i++) st.Dinv[i] = ld[21 + i];
    tl_ldl_solve(st, qdd);
    T xn[16];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const T qn = tl_pipe_euler(x[i], x[7 + i], dt), vn = tl_pipe_euler(x[7 + i], qdd[i], dt);
        x[i] = qn; x[7 + i] = vn; xn[i] = qn; xn[7 + i] = vn;
    }
    xn[14] = T(0); xn[15] = T(0);
    tl_pipe_st<4>(p.xbuf + ((((k + 1) & 1) * 64) + lane) * kPipeSX, xn);
    tl_pipe_post(p.flag + 0, k + 1);
}

}  // namespace pddp

#pragma clang fp contract(off)
#endif
