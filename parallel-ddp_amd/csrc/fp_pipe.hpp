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

// LDS strides per lane in floats: 4 x odd, so that 16-byte accesses of 8 consecutive lanes cover the 32 banks exactly once
constexpr int kPipeSX = 20, kPipeSL = 44, kPipeSU = 12;
constexpr int kPipeXbuf = 2 * 64 * kPipeSX, kPipeLbuf = 2 * 64 * kPipeSL, kPipeUbuf = 2 * 64 * kPipeSU;
constexpr int kPipeFlags = 8;                                                      // x, u, (spare), cs[2], l[2]
constexpr int kPipeLdsOpenLoop = (kPipeXbuf + kPipeLbuf + kPipeFlags) * 4;          // bytes: chain + factor waves
constexpr int kPipeLdsClosedLoop = (kPipeXbuf + kPipeLbuf + kPipeUbuf + kPipeFlags) * 4;

// the step counters are accessed as LDS (address space 3) explicitly: through a generic volatile pointer the compiler emits FLAT loads / stores with system-coherence bits --
// every poll and post then takes the flat path instead of a ds_read / ds_write (found in the ISA: 73 -> 6x us per 32 steps)
typedef __attribute__((address_space(3))) volatile int tl_pipe_flag;
struct TlPipeLds {
    float* xbuf;            // [2][64][20]   x_k at slot k & 1
    float* lbuf;            // [2][64][44]   step k's c[7] s[7] (2 pad) L[21] Dinv[7] at slot k & 1
    float* ubuf;            // [2][64][12]   u_k (closed loop)
    tl_pipe_flag* flag;     // [0] x: v = x_v is in xbuf   [1] u: v = u_{v-1} is in ubuf   [3 + r] cs, [5 + r] l: v = step v-1's are in lbuf[r]
};
__device__ __forceinline__ TlPipeLds tl_pipe_lds(float* base, bool closed_loop) {
    TlPipeLds p;
    p.xbuf = base; p.lbuf = p.xbuf + kPipeXbuf; p.ubuf = p.lbuf + kPipeLbuf;
    p.flag = (tl_pipe_flag*)(closed_loop ? p.ubuf + kPipeUbuf : p.ubuf);
    return p;
}
typedef float tl_pipe_f4 __attribute__((ext_vector_type(4), aligned(16)));
// n floats (a multiple of 4 slots are touched) between registers and a 16-byte aligned LDS run
template <int N4> __device__ __forceinline__ void tl_pipe_ld(float* dst, const float* src) {
#pragma unroll
    for (int i = 0; i < N4; i++) { const tl_pipe_f4 v = reinterpret_cast<const tl_pipe_f4*>(src)[i]; dst[4 * i] = v[0]; dst[4 * i + 1] = v[1]; dst[4 * i + 2] = v[2]; dst[4 * i + 3] = v[3]; }
}
template <int N4> __device__ __forceinline__ void tl_pipe_st(float* dst, const float* src) {
#pragma unroll
    for (int i = 0; i < N4; i++) { tl_pipe_f4 v; v[0] = src[4 * i]; v[1] = src[4 * i + 1]; v[2] = src[4 * i + 2]; v[3] = src[4 * i + 3]; reinterpret_cast<tl_pipe_f4*>(dst)[i] = v; }
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

// factor wave r (0 / 1): the sines / cosines and the factors of M(q_j) for the steps j = r, r + 2, ... < njobs.  x0[14]: the rollout's start state.
// xout != null: lane 0 also stores the states it picks up (x_{j-1}, j >= 2) to xout[14 (j - 1)] -- global stores kept off the chain wave, whose posts would wait for them.
template <int V>
__device__ __forceinline__ void tl_pipe_factor_wave(const TlPipeLds& p, int r, int njobs, const float* x0, float dt, int lane, float* xout = nullptr) {
    constexpr ArmTlModel<float> md = arm_tl_builtin<float>(V);
    float* o = p.lbuf + (r * 64 + lane) * kPipeSL;
    for (int j = r; j < njobs; j += 2) {
        float q[7];
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = x0[i];
        } else if (j == 1) {
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = tl_pipe_euler(x0[i], x0[7 + i], dt);
        } else {
            tl_pipe_wait(p.flag + 0, j - 1);
            float xv[16];
            tl_pipe_ld<4>(xv, p.xbuf + ((((j - 1) & 1) * 64) + lane) * kPipeSX);
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = tl_pipe_euler(xv[i], xv[7 + i], dt);
            if (xout && lane == 0) {
#pragma unroll
                for (int i = 0; i < 14; i++) xout[14 * (j - 1) + i] = xv[i];
            }
        }
        ArmTlState<float> st;
        arm_tl_trig<float>(st, q);
        float cs[16], ld[28];
#pragma unroll
        for (int i = 0; i < 7; i++) { cs[i] = st.c[i]; cs[7 + i] = st.s[i]; }
        cs[14] = 0.f; cs[15] = 0.f;
        tl_pipe_st<4>(o, cs);
        tl_pipe_post(p.flag + 3 + r, j + 1);
        arm_tl_factor<float>(md, st);
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
template <int V, bool CLOSED>
__device__ __forceinline__ void tl_pipe_chain_step(const TlPipeLds& p, int k, float* x, const float* u, float dt, float grav, int lane) {
    constexpr ArmTlModel<float> md = arm_tl_builtin<float>(V);
    const int r = k & 1;
    const float* o = p.lbuf + (r * 64 + lane) * kPipeSL;
    ArmTlState<float> st;
    tl_pipe_wait(p.flag + 3 + r, k + 1);
    float cs[16], ld[28];
    tl_pipe_ld<4>(cs, o);
#pragma unroll
    for (int i = 0; i < 7; i++) { st.c[i] = cs[i]; st.s[i] = cs[7 + i]; }
    float bias[7], qdd[7];
    arm_tl_bias<float>(md, grav, st, x + 7, bias);
    // the waits below are volatile LDS reads, the bias is register arithmetic: without this the compiler sinks the whole recursion BELOW the waits (cycle stamps: the chain
    // then sat at the control wave's post for 1.1 k cycles and at the factors' with its bias still to do -- 6.9 k cycles per step instead of 4.x k)
#pragma unroll
    for (int i = 0; i < 7; i++) asm volatile("" : "+v"(bias[i]));
    if (CLOSED) {
        float uv[8];
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
    float xn[16];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const float qn = tl_pipe_euler(x[i], x[7 + i], dt), vn = tl_pipe_euler(x[7 + i], qdd[i], dt);
        x[i] = qn; x[7 + i] = vn; xn[i] = qn; xn[7 + i] = vn;
    }
    xn[14] = 0.f; xn[15] = 0.f;
    tl_pipe_st<4>(p.xbuf + ((((k + 1) & 1) * 64) + lane) * kPipeSX, xn);
    tl_pipe_post(p.flag + 0, k + 1);
}

}  // namespace pddp

#pragma clang fp contract(off)
#endif
