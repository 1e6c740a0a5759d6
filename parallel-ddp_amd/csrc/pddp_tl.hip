// Thread-lane kernels of the KUKA arm (fp_tl.hpp, plant_arm_tl.hpp): one thread per rollout / per knot.  gfx950 only.
// Built with -ffp-contract=fast semantics inside the tl headers and -fno-slp-vectorize (Makefile).
#include <hip/hip_runtime.h>

#include "ab_compact.hpp"
#include "bodies.hpp"
#include "fp_lg.hpp"
#include "fp_pipe.hpp"
#include "tl_launch.hpp"

namespace pddp {

// k_fp_tl: grid ceil(B*M*A / 256), block 256.  Thread i rolls out segment (i / A) % M of candidate i % A of problem i / (M*A): the A candidates
// of a (problem, segment) PAIR are adjacent lanes and need the same per-step operands (gain K_k 98 floats, reference state 14, nominal control 7,
// feed-forward 7).  A wave owns 64/A pairs; it fetches their operands of step k+1 cooperatively (every pair's block is contiguous: 8-byte accesses,
// each byte fetched once per wave) while step k computes, parks them in its own double-buffered LDS area (132-float pair stride: the pairs land in
// different banks), and every lane reads its pair's copy from there.  No workgroup barrier: a wave only ever touches its own area.
// ALL = false (the sweep): every candidate writes its (state | control) record of every knot into xw (88 bytes per step; float: TlStagedSink below) and its boundary
// defects into its slot of ds, next to its partial cost / defect sums; the setup kernel adopts the accepted candidate's records (arm_tl_adopt_knot).
// ALL = true (pddp_run_phase(FP), teacher-forced tests): the controls go to us as well, as the reference keeps them.
// The linear sweep (k_sweep_st) runs before it.  Replaces forwardSimKern<<<(M,A),(8,7)>>> + costKern<<<A,N>>> + defectKern<<<A,N>>>
// (fpHelpers.cuh:366,383,388).
// The production instantiation (float, joint-space cost, states only) is held to 168 registers -- three waves per SIMD, no scratch: 0.695 -> 0.654 ms at 16384 problems
// on one box (tools/ab_lib.sh); at 128 registers (four waves) it spills 212 bytes per lane and takes 1.3 ms.
// TlStagedSink (float sweep): the (state | control | pad) records of a wave's 64 rollouts -- 22 floats each, the 8 candidates of a (problem, segment) pair adjacent: one
// 704-byte run of xw per pair and knot -- go through the wave's LDS area and leave as 16-byte pieces: the A lanes of a pair write 16 A consecutive bytes per store
// instruction instead of 8 bytes 88 bytes apart (one memory transaction per LANE and store).  The record of knot k is complete right after the control law of step k
// (the state is the rollout's current one), when the step's operands in LDS have been consumed: the staging area overlays them (both operand buffers are dead between
// the control law and the end-of-step park()).  cur: this lane's record of the current knot; a: its candidate index.
// (every method forced inline: left to the inliner's cost model, a larger caller turns xu() into a CALL, which parks the whole rollout state in scratch -- round 5)
template <typename T> struct TlStagedSink {
    typedef T v4 __attribute__((ext_vector_type(4)));
    T* ds; mutable T* cur; int stride; T* stg; int lane, a, A;
    __device__ __forceinline__ void x(int, const T*) const {}
    __device__ __forceinline__ void u(int, const T*) const {}
    __device__ __forceinline__ void d(int k, const T* v) const { tl_store14(ds + (size_t)k * 14, v); }
    __device__ __forceinline__ void xu(int, const T* xv, const T* uv) const {
        wsync();
        T* rec = stg + lane * 22;
        tl_store14(rec, xv); tl_store7(rec + 14, uv);
        wsync();
        const T* src = stg + (lane - a) * 22 + 4 * a; T* dst = cur - 18 * a;           // the pair's run (A records = 5.5 A pieces of 16 bytes): piece a, then every A-th
        const int step = 4 * A;
#pragma unroll
        for (int j = 0; j < 5; j++) *reinterpret_cast<v4*>(dst + step * j) = *reinterpret_cast<const v4*>(src + step * j);
        if (2 * a < A) *reinterpret_cast<v4*>(dst + step * 5) = *reinterpret_cast<const v4*>(src + step * 5);
        cur += stride;
    }
    __device__ __forceinline__ void xu_last(int, const T* xv, const T* uv) const { tl_store14(cur, xv); tl_store7(cur + 14, uv); }
};
constexpr int kFpTlPS = 132;                 // floats per staged pair: K 98 | xr 14 | uc 7 | du 7 | pad 6
constexpr int kFpTlMaxPairs = 8;             // pairs per wave (A >= 8; smaller A takes the unstaged path)
// EE: the end-effector cost family (fp_tl.hpp tl_rollout_step_ee): every segment runs NB steps, the cost is accumulated in the rollout.
template <typename T, int V, bool ALL, bool EE = false>
#ifndef PDDP_FP_TL_MINBLK
#define PDDP_FP_TL_MINBLK 3
#endif
__global__ __launch_bounds__(256, sizeof(T) == 4 && !EE && !ALL ? PDDP_FP_TL_MINBLK : 2) void k_fp_tl(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, T grav, int batch) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    constexpr int NX = 14, NU = 7, PS = kFpTlPS;
    const int A = dm.A, M = dm.M, N = dm.N, NBk = dm.NB, per_pb = M * A, total = batch * per_pb;
    const int inst = blockIdx.x * 256 + threadIdx.x;
    if (A < 8 || (64 % A) != 0) {            // unstaged: operands straight from global memory
        if (inst >= total) return;
        const int pb = inst / per_pb, rem = inst - pb * per_pb, seg = rem / A, a_idx = rem - seg * A;
        if (!fp_active<T>(b, dm, pb)) return;
        const T* xcur = b.xb + ((size_t)pb * 2 + b.state[pb].cur) * N * NX;
        if (EE) {
            if (ALL) arm_tl_rollout_segment_ee<T>(md, grav, b, dm, cw, dt, pb, a_idx, seg, xcur, tl_candidate_sink<T>(b, dm, pb, a_idx), true);
            else arm_tl_rollout_segment_ee<T>(md, grav, b, dm, cw, dt, pb, a_idx, seg, xcur, tl_state_sink<T>(b, dm, pb, a_idx), true);
        } else {
            if (ALL) arm_tl_rollout_segment<T>(md, grav, b, dm, cw, dt, pb, a_idx, seg, xcur, tl_candidate_sink<T>(b, dm, pb, a_idx), true);
            else arm_tl_rollout_segment<T>(md, grav, b, dm, cw, dt, pb, a_idx, seg, xcur, tl_state_sink<T>(b, dm, pb, a_idx), true);
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) T stage_all[4 * 2 * kFpTlMaxPairs * PS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T* stg = stage_all + wave * (2 * kFpTlMaxPairs * PS);
    const int ppw = 64 / A, npairs = batch * M;
    const int gp0 = (blockIdx.x * 256 + wave * 64) / A;               // first pair of this wave
    if (gp0 >= npairs) return;                                        // whole wave beyond the batch
    // ---- this lane's share of the cooperative fetch: up to 7 eight-byte pieces (K and xr) and 2 four-byte pieces (uc, du) per step
    constexpr int J2 = (kFpTlMaxPairs * 56 + 63) / 64, J1 = (kFpTlMaxPairs * 14 + 63) / 64;
    const T* src2[J2]; int dst2[J2], str2[J2];
    const T* src1[J1]; int dst1[J1];
#pragma unroll
    for (int j = 0; j < J2; j++) {
        const int e = lane + 64 * j, q = e / 56, off = e - q * 56;
        int gq = gp0 + (q < ppw ? q : ppw - 1); gq = gq < npairs ? gq : npairs - 1;
        const int pbq = gq / M, k0 = (gq - pbq * M) * NBk;
        const bool isK = off < 49;
        src2[j] = isK ? b.KT + ((size_t)pbq * N + k0) * (NX * NU) + 2 * off
                      : b.xb + (((size_t)pbq * 2 + b.state[pbq].cur) * N + k0) * NX + 2 * (off - 49);
        str2[j] = isK ? NX * NU : NX;
        dst2[j] = (e < ppw * 56) ? q * PS + (isK ? 2 * off : 98 + 2 * (off - 49)) : -1;
    }
#pragma unroll
    for (int j = 0; j < J1; j++) {
        const int e = lane + 64 * j, q = e / 14, off = e - q * 14;
        int gq = gp0 + (q < ppw ? q : ppw - 1); gq = gq < npairs ? gq : npairs - 1;
        const int pbq = gq / M, k0 = (gq - pbq * M) * NBk;
        src1[j] = (off < 7 ? b.ucur : b.du) + ((size_t)pbq * N + k0) * NU + (off < 7 ? off : off - 7);
        dst1[j] = (e < ppw * 14) ? q * PS + 112 + off : -1;
    }
    TlPair<T> pf2[J2]; T pf1[J1];
    auto fetch = [&]() {                                              // the operands of the next step: running pointers (one 64-bit add each instead of k * stride)
#pragma unroll
        for (int j = 0; j < J2; j++) { pf2[j] = *reinterpret_cast<const TlPair<T>*>(src2[j]); src2[j] += str2[j]; }
#pragma unroll
        for (int j = 0; j < J1; j++) { pf1[j] = src1[j][0]; src1[j] += NU; }
    };
    auto park = [&](int buf) {
        T* d = stg + buf * (kFpTlMaxPairs * PS);
#pragma unroll
        for (int j = 0; j < J2; j++) if (dst2[j] >= 0) *reinterpret_cast<TlPair<T>*>(d + dst2[j]) = pf2[j];
#pragma unroll
        for (int j = 0; j < J1; j++) if (dst1[j] >= 0) d[dst1[j]] = pf1[j];
    };
    // ---- this lane's rollout
    const int p = lane / A, a_idx = lane - p * A, gp = gp0 + p;
    const int pb = (gp < npairs ? gp : npairs - 1) / M, seg = (gp < npairs ? gp : npairs - 1) - pb * M;
    const bool live = gp < npairs && fp_active<T>(b, dm, pb);
    const T* xcur = b.xb + ((size_t)pb * 2 + b.state[pb].cur) * N * NX;
    // joint-space goal, or (EE) the 6-vector tool-point goal in xg[0..5]: one copy per pair in LDS (14 registers the dynamics do not have to carry)
    __shared__ __attribute__((aligned(16))) T xg_all[4 * kFpTlMaxPairs * 16];
    T* xg = xg_all + (wave * kFpTlMaxPairs + p) * 16;
    if (a_idx == 0) { T t[NX]; tl_load14(t, b.xGoal + (size_t)pb * NX); tl_store14(xg, t); }
    T xt[EE ? NX : 1], acc[EE ? NU : 1];                              // EE: nominal-state target, per-joint running cost
    int tshift = 0;
    if (EE) {
        tl_load14(xt, b.xTarget + (size_t)pb * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) acc[i] = T(0);
        tshift = b.tshift[pb];
    }
    TlRollout<T> r;
    r.iters = 0;
    const auto csink = tl_candidate_sink<T>(b, dm, pb, a_idx);
    constexpr bool STG = sizeof(T) == 4 && !ALL;                      // the float sweep: records through LDS (the host API allocates xw with 22-float records for this path)
    const auto rsink = tl_run_sink<T>(b, dm, pb, a_idx, seg * NBk);
    const auto ssink = [&]() { if constexpr (STG) return TlStagedSink<T>{rsink.ds, rsink.cur, rsink.stride, stg, lane, a_idx, A}; else return rsink; }();
    if (live) {
        if (ALL) tl_rollout_begin<T>(r, b, dm, pb, a_idx, seg, xcur, csink); else tl_rollout_begin<T>(r, b, dm, pb, a_idx, seg, xcur, ssink);
        if (EE) r.iters = NBk;
    }
    fetch(); park(0);
    wsync();
    for (int k = 0; k < NBk; k++) {
        if (k + 1 < NBk) fetch();                                     // in flight while this step computes
        if (live && k < r.iters) {
            const T* o = stg + (k & 1) * (kFpTlMaxPairs * PS) + p * PS;
            if (EE) {
                if (ALL) tl_rollout_step_ee<T>(r, acc, md, grav, b, dm, cw, dt, k, o, o + 98, o + 112, o + 119, xg, xt, tshift, csink);
                else tl_rollout_step_ee<T>(r, acc, md, grav, b, dm, cw, dt, k, o, o + 98, o + 112, o + 119, xg, xt, tshift, ssink);
            } else {
                if (ALL) tl_rollout_step<T>(r, md, grav, b, dm, cw, dt, k, o, o + 98, o + 112, o + 119, xg, csink);
                else tl_rollout_step<T>(r, md, grav, b, dm, cw, dt, k, o, o + 98, o + 112, o + 119, xg, ssink);
            }
        }
        if (k + 1 < NBk) park((k + 1) & 1);
        wsync();
    }
    if (!live) return;
    if (EE) tl_rollout_end_ee<T>(r, acc, dm);
    else {
        T ucN[NU];
#pragma unroll
        for (int i = 0; i < NU; i++) ucN[i] = b.ucur[((size_t)pb * N + (N - 1)) * NU + i];
        if (ALL) tl_rollout_end<T>(r, dm, cw, ucN, xg, csink); else tl_rollout_end<T>(r, dm, cw, ucN, xg, ssink);
    }
    const size_t slot = (size_t)pb * A + a_idx;
    b.Jpart[slot * M + seg] = r.J; b.dpart[slot * M + seg] = r.sdef;
    b.parts_fresh[pb] = 1;
}

// k_fp_tl2: grid ceil(B*M*A / 64), block 128.  The rollouts of a handle with FEW problems in flight (one MPC solve: 32 rollouts on a 256-CU device), where only the
// length of one step's instruction stream counts.  Lane = rollout as in k_fp_tl, but a step is split over the workgroup's TWO wavefronts (two SIMDs of one CU),
// which carry the same 64 rollouts:
//     wave 0: control law, running cost, sines / cosines, recursive Newton-Euler -> tau = u - bias        (arm_tl_bias)        ~ 900 instructions
//     wave 1: sines / cosines, composite rigid bodies, mass matrix, L D L'                                 (arm_tl_factor)      ~ 950 instructions
//     barrier; wave 1: tau from LDS, two triangular solves, Euler step, new state to LDS and to xs; barrier; wave 0 picks the new state up
// instead of ~1700 instructions in a row on one wave (a lone wave issues a dependent instruction every ~6 cycles: tools/probes/single_wave_clock.hip).
// The next step's operands (gain, reference state, nominal control, feed-forward) are fetched into registers while the current step computes.
// Candidates are stored like the reference's (x, u, d: the lane-group setup kernel adopts the winner from there); cost / defect leave as per-segment
// partial sums (k_ls adds them in order, as after k_fp_tl).  Same arithmetic as k_fp_tl (fp_tl.hpp, plant_arm_tl.hpp): under the float32 bar.
template <int V>
__global__ __launch_bounds__(128, 1) void k_fp_tl2(Buffers<float> b, Dims dm, CostWeights<float> cw, float dt, float grav, int batch) {
    using T = float;
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    constexpr int NX = 14, NU = 7, ST = 9, SX = 15;                   // odd LDS strides: 64 lanes, 64 banks apart
    __shared__ T s_tau[64 * ST];
    __shared__ T s_x[64 * SX];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int A = dm.A, M = dm.M, N = dm.N, NBk = dm.NB, total = batch * M * A;
    const int inst_raw = blockIdx.x * 64 + lane, inst = inst_raw < total ? inst_raw : total - 1;
    const int pb = inst / (M * A), rem = inst - pb * (M * A), seg = rem / A, a_idx = rem - seg * A;
    const bool live = inst_raw < total && fp_active<T>(b, dm, pb);
    const int kStart = seg * NBk, iters = (seg < M - 1) ? NBk : NBk - 1;
    const T* xcur = b.xb + ((size_t)pb * 2 + b.state[pb].cur) * N * NX;
    const size_t slot = (size_t)pb * A + a_idx;
    T* xs = b.xs + slot * N * NX; T* us = b.us + slot * N * NU; T* ds = b.ds + slot * N * NX;
    T x[NX];
    if (seg == 0) tl_load14(x, xcur); else tl_load14(x, xs + (size_t)kStart * NX);
    if (wave == 0) {
        // ---------------------------------------------------------------- wave 0: control law, cost, bias torque
        const T alpha = b.alpha[a_idx];
        const T* KT = b.KT + (size_t)pb * N * NX * NU; const T* uc = b.ucur + (size_t)pb * N * NU; const T* du = b.du + (size_t)pb * N * NU;
        T xg[NX];
        tl_load14(xg, b.xGoal + (size_t)pb * NX);
        T J = T(0);
        T nK[NX * NU], nxr[NX], nuc[NU], ndu[NU];
        auto fetch = [&](int kn) {
#pragma unroll
            for (int rr = 0; rr < NU; rr++) tl_load14(nK + rr * NX, KT + (size_t)kn * (NX * NU) + rr * NX);
            tl_load14(nxr, xcur + (size_t)kn * NX);
#pragma unroll
            for (int i = 0; i < NU; i++) { nuc[i] = uc[(size_t)kn * NU + i]; ndu[i] = du[(size_t)kn * NU + i]; }
        };
        fetch(kStart);
        for (int k = 0; k < NBk; k++) {
            const int kn = kStart + k;
            const bool act = live && k < iters;
            if (act) {
                T u[NU];
                tl_control_law<T>(u, alpha, ndu, nK, x, nxr, nuc);
                if (k + 1 < NBk) fetch(kn + 1);                         // in flight while this step computes (kn + 1 <= N - 1)
#pragma unroll
                for (int i = 0; i < NU; i++) us[(size_t)kn * NU + i] = u[i];
                J += arm_tl_cost<T>(cw, x, u, xg, false);
                ArmTlState<T> st;
                T bias[NU];
                arm_tl_trig<T>(st, x);
                arm_tl_bias<T>(md, grav, st, x + 7, bias);
#pragma unroll
                for (int i = 0; i < NU; i++) s_tau[lane * ST + i] = u[i] - bias[i];
            }
            __syncthreads();                                            // tau is in LDS
            __syncthreads();                                            // the new state is in LDS
            if (act && k < NBk - 1) {
#pragma unroll
                for (int i = 0; i < NX; i++) x[i] = s_x[lane * SX + i];
            }
        }
        if (live) {
            if (seg == M - 1) {                                         // terminal knot: its (unused) control is carried along (tl_rollout_end)
                T u[NU];
#pragma unroll
                for (int i = 0; i < NU; i++) { u[i] = uc[(size_t)(N - 1) * NU + i]; us[(size_t)(N - 1) * NU + i] = u[i]; }
                J += arm_tl_cost<T>(cw, x, u, xg, true);
            }
            b.Jpart[slot * M + seg] = J;
            b.parts_fresh[pb] = 1;
        }
    } else {
        // ---------------------------------------------------------------- wave 1: factors of the mass matrix, solve, Euler step, trajectory out
        if (live) tl_store14(xs + (size_t)kStart * NX, x);             // (a candidate slot already holds it for seg > 0)
        T sdef = T(0);
        for (int k = 0; k < NBk; k++) {
            const int kn = kStart + k;
            const bool act = live && k < iters;
            ArmTlState<T> st;
            if (act) { arm_tl_trig<T>(st, x); arm_tl_factor<T>(md, st); }
            __syncthreads();                                            // tau is in LDS
            if (act) {
                T qdd[NU], xn[NX];
#pragma unroll
                for (int i = 0; i < NU; i++) qdd[i] = s_tau[lane * ST + i];
                tl_ldl_solve(st, qdd);
#pragma unroll
                for (int i = 0; i < 7; i++) { xn[i] = x[i] + dt * x[7 + i]; xn[7 + i] = x[7 + i] + dt * qdd[i]; }     // Euler (utils/integrators.cuh:24-36)
                if (k < NBk - 1) {
#pragma unroll
                    for (int i = 0; i < NX; i++) { x[i] = xn[i]; s_x[lane * SX + i] = xn[i]; }
                    tl_store14(xs + (size_t)(kn + 1) * NX, xn);
                } else {                                                // last step of a non-final segment: defect against the next segment's start
                    const int ks = (seg + 1) * NBk;
                    T xnext[NX], e[NX];
                    tl_load14(xnext, xs + (size_t)ks * NX);
#pragma unroll
                    for (int i = 0; i < NX; i++) { e[i] = xn[i] - xnext[i]; sdef += tabs(e[i]); }
                    tl_store14(ds + (size_t)(ks - 1) * NX, e);
                }
            }
            __syncthreads();                                            // the new state is in LDS
        }
        if (live) b.dpart[slot * M + seg] = (seg == M - 1) ? T(0) : sdef;
    }
}
void launch_fp_tl2(hipStream_t s, int variant, const Buffers<float>& b, const Dims& dm, const CostWeights<float>& cw, float dt, float grav, int batch) {
    const unsigned inst = (unsigned)batch * dm.M * dm.A;
    if (variant == 0) hipLaunchKernelGGL((k_fp_tl2<0>), dim3((inst + 63) / 64), dim3(128), 0, s, b, dm, cw, dt, grav, batch);
    else hipLaunchKernelGGL((k_fp_tl2<1>), dim3((inst + 63) / 64), dim3(128), 0, s, b, dm, cw, dt, grav, batch);
}

// k_fp_tl4: grid ceil(B*M*A / 64), block 256, dynamic LDS kPipeLdsClosedLoop.  The rollouts of a handle with FEW problems in flight as a pipeline over the workgroup's four
// waves (fp_pipe.hpp): lane = rollout in every wave; wave 0 walks the chain x_k -> x_{k+1}, waves 2 and 3 factor the mass matrix of alternate steps one step AHEAD (Euler's
// q_{k+1} needs no dynamics), wave 1 evaluates the control law on x_k while the chain computes its bias, stores the trajectory and adds up the cost.
// ~650 dependent instructions per step on the chain instead of ~1100 with k_fp_tl2's two barriers per step.  Candidates are stored like the reference's (x, u, d); cost / defect
// leave as per-segment partial sums.  EE: the end-effector cost family (tl_rollout_step_ee's conditions: every segment runs NB steps, the "final" state of a non-final
// segment carries no cost, knot N - 1 no dynamics).  Same arithmetic as k_fp_tl: under the float32 bar.
// ls_mode >= 0 (few problems in flight, every problem's M x A rollouts inside ONE wavefront: 64 % (M A) == 0): the control wave ENDS WITH THE LINE SEARCH of its problems --
// forwardSimGPU's host part + acceptRejectTrajGPU (fpHelpers.cuh:395-408, nisInitHelpers.cuh:489-518; line_search_accept) -- instead of leaving it to a k_ls launch behind
// this kernel: the candidates' totals are formed from the lanes' own partial sums (the order of tl_reduce_parts / k_ls: segments 0..M-1), one kernel boundary and one
// dependent launch (~7 us of a ~120 us iteration) disappear.  ls_mode = freeze_exit of k_ls (1: benchmark mode); -1: no line search here (phase hooks, per-phase timing).
// maps (same condition on M A, and M A >= 16): the kernel BEGINS WITH THE LINEAR FORWARD SWEEP -- forwardSweepKern x A (fpHelpers.cuh:19-63) in the form of k_sweep_maps
// (pddp_mx.hip): the per-segment maps the matrix-core backward pass composed, e <- Phi_s e + gamma_s for the s- and the t-sequence, walked by the first 14 lanes of the
// problem's M A (every wave walks them: the same operations in the same order, no exchange between the waves); a rollout's start state is xcur + (t - alpha s) at its
// segment's boundary, the very expression k_sweep_maps stores.  The defect against the next segment's start state then comes from that rollout's lane, not from memory.
template <typename T> __device__ __forceinline__ T tl4_fma(T a, T b_, T c) { return __builtin_fma(a, b_, c); }
template <> __device__ __forceinline__ float tl4_fma<float>(float a, float b_, float c) { return __builtin_fmaf(a, b_, c); }
template <typename T, int V, bool EE>
__global__ __launch_bounds__(256, 1) void k_fp_tl4(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, T grav, int batch, SolverParams sp, int ls_mode, int maps) {
    constexpr int NX = 14, NU = 7;
    __shared__ T ls_J[64], ls_d[64];
    extern __shared__ __attribute__((aligned(16))) unsigned char pipe_lds_raw[];
    const TlPipeLdsT<T> p = tl_pipe_lds(reinterpret_cast<T*>(pipe_lds_raw), true);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x < kPipeFlags) p.flag[threadIdx.x] = 0;
    const int A = dm.A, M = dm.M, N = dm.N, NBk = dm.NB, total = batch * M * A;
    const int inst_raw = blockIdx.x * 64 + lane, inst = inst_raw < total ? inst_raw : total - 1;
    const int pb = inst / (M * A), rem = inst - pb * (M * A), seg = rem / A, a_idx = rem - seg * A;
    const bool live = inst_raw < total && fp_active<T>(b, dm, pb);
    const int kStart = seg * NBk;
    const T* xcur = b.xb + ((size_t)pb * 2 + b.state[pb].cur) * N * NX;
    const size_t slot = (size_t)pb * A + a_idx;
    T* xs = b.xs + slot * N * NX; T* us = b.us + slot * N * NU; T* ds = b.ds + slot * N * NX;
    const T alpha = b.alpha[a_idx];
    T x[NX];
    if (!maps) { if (seg == 0) tl_load14(x, xcur); else tl_load14(x, xs + (size_t)kStart * NX); }
    else {
        const int W = M * A, l = rem < NX ? rem : NX - 1;                 // W lanes per problem: the shuffles stay inside them
        const T* dcur = b.dcur + (size_t)pb * N * NX;
        tl_load14(x, xcur + (size_t)kStart * NX);                         // the nominal state at the segment's first knot
        T es = T(0), et = T(0);
        auto advance = [&](int sgm, const T* ph, T gam, T dk) {
            T ns = gam, nt = dk;
#pragma unroll
            for (int cc = 0; cc < NX; cc++) { ns = tl4_fma<T>(ph[cc], __shfl(es, cc, W), ns); nt = tl4_fma<T>(ph[cc], __shfl(et, cc, W), nt); }
            es = ns; et = nt;
#pragma unroll
            for (int i = 0; i < NX; i++) {                                 // (every lane shuffles: the source lanes belong to segment 0)
                const T si = __shfl(es, i, W), ti = __shfl(et, i, W);
                if (seg == sgm + 1) x[i] = x[i] + (ti - alpha * si);
            }
        };
        if (M == 4) {                                                    // the usual M: the three maps requested at once (one memory round trip, as in k_sweep_maps)
            T ph[3][NX + 1], dk[3];
#pragma unroll
            for (int sgm = 0; sgm < 3; sgm++) {
                const T* o = b.segmap + ((size_t)pb * 4 + sgm) * 256;
#pragma unroll
                for (int cc = 0; cc <= NX; cc++) ph[sgm][cc] = o[cc * 16 + l];
                dk[sgm] = dcur[(size_t)((sgm + 1) * NBk - 1) * NX + l];
            }
#pragma unroll
            for (int sgm = 0; sgm < 3; sgm++) advance(sgm, ph[sgm], ph[sgm][NX], dk[sgm]);
        } else {
            for (int sgm = 0; sgm < M - 1; sgm++) {
                const T* o = b.segmap + ((size_t)pb * M + sgm) * 256;
                T ph[NX + 1];
#pragma unroll
                for (int cc = 0; cc <= NX; cc++) ph[cc] = o[cc * 16 + l];
                advance(sgm, ph, ph[NX], dcur[(size_t)((sgm + 1) * NBk - 1) * NX + l]);
            }
        }
    }
    __syncthreads();                                                    // the counters are zero
    if (wave >= 2) { tl_pipe_factor_wave<V>(p, wave - 2, NBk, x, dt, lane); return; }
    if (wave == 0) {
        // ---------------------------------------------------------------- chain
        for (int k = 0; k < NBk; k++) tl_pipe_chain_step<V, true>(p, k, x, (const T*)nullptr, dt, grav, lane);
        return;
    }
    // -------------------------------------------------------------------- control wave: operands, trajectory out, cost
    T x0[NX];                                                             // (maps: the rollout's start state, the next-lower segment's defect is taken against it)
#pragma unroll
    for (int i = 0; i < NX; i++) x0[i] = x[i];
    const T* KT = b.KT + (size_t)pb * N * NX * NU; const T* uc = b.ucur + (size_t)pb * N * NU; const T* du = b.du + (size_t)pb * N * NU;
    T xg[NX], goal[6], acc7s[NU];
    int tshift = 0;
    if (EE) {
#pragma unroll
        for (int i = 0; i < 6; i++) goal[i] = b.xGoal[(size_t)pb * NX + i];
        tl_load14(xg, b.xTarget + (size_t)pb * NX);
        tshift = b.tshift[pb];
#pragma unroll
        for (int i = 0; i < NU; i++) acc7s[i] = T(0);
    } else tl_load14(xg, b.xGoal + (size_t)pb * NX);
    T J = T(0), sdef = T(0);
    T nK[NX * NU], nxr[NX], nuc[NU], ndu[NU];
    auto fetch = [&](int kn) {
#pragma unroll
        for (int rr = 0; rr < NU; rr++) tl_load14(nK + rr * NX, KT + (size_t)kn * (NX * NU) + rr * NX);
        tl_load14(nxr, xcur + (size_t)kn * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) { nuc[i] = uc[(size_t)kn * NU + i]; ndu[i] = du[(size_t)kn * NU + i]; }
    };
    fetch(kStart);
    if (live) tl_store14(xs + (size_t)kStart * NX, x);                    // (without maps a candidate slot already holds it for seg > 0)
    const int iters = EE ? NBk : ((seg < M - 1) ? NBk : NBk - 1);
    for (int k = 0; k < NBk; k++) {
        const int kn = kStart + k;
        if (k > 0) {
            T xv[16];
            tl_pipe_wait(p.flag + 0, k);
            tl_pipe_ld<4>(xv, p.xbuf + (((k & 1) * 64) + lane) * kPipeSX);
#pragma unroll
            for (int i = 0; i < NX; i++) x[i] = xv[i];
        }
        T u[8];
        tl_control_law<T>(u, alpha, ndu, nK, x, nxr, nuc);
        u[7] = T(0);
        tl_pipe_st<2>(p.ubuf + (((k & 1) * 64) + lane) * kPipeSU, u);
        tl_pipe_post(p.flag + 1, k + 1);
        if (k + 1 < NBk) fetch(kn + 1);                                   // in flight while the chain walks step k (kn + 1 <= N - 1)
        if (k > 0 && live) tl_store14(xs + (size_t)kn * NX, x);
        if (live && k < iters) {
#pragma unroll
            for (int i = 0; i < NU; i++) us[(size_t)kn * NU + i] = u[i];
            if (!EE) J += arm_tl_cost<T>(cw, x, u, xg, false);
            else if (k < NBk - 1 || seg == M - 1) {
                constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
                ArmTlState<T> st;
                arm_tl_trig<T>(st, x);
                ArmTlFrames<T> fr;
                arm_tl_world_chain<false, T>(md, st.c, st.s, fr);
                T pos[6];
                arm_tl_tool_point<T>(fr, cw.ee_z, tl_ee_rpy_weighted<T>(cw), pos);
#pragma unroll
                for (int ind = 0; ind < NU; ind++) {
                    T cost = T(0);
                    if (ind == 0) cost += ee_term<T>(cw, pos, goal, kn >= N - 1 - tshift);
                    acc7s[ind] += ee_joint_terms<T>(cw, x, u, xg, ind, kn, N, cost);
                }
            }
        }
    }
    T xnext[NX];
    if (maps) {                                                           // (wave-uniform) the start state of the same candidate's next segment: A lanes up
#pragma unroll
        for (int i = 0; i < NX; i++) xnext[i] = __shfl(x0[i], (lane + A) & 63);
    }
    if (seg < M - 1) {                                                    // the step out of the segment's last knot: defect against the next segment's start
        tl_pipe_wait(p.flag + 0, NBk);
        if (live) {
            T xi[16];
            tl_pipe_ld<4>(xi, p.xbuf + ((((NBk) & 1) * 64) + lane) * kPipeSX);
            const int ks = (seg + 1) * NBk;
            T e[NX];
            if (!maps) tl_load14(xnext, xs + (size_t)ks * NX);
#pragma unroll
            for (int i = 0; i < NX; i++) { e[i] = xi[i] - xnext[i]; sdef += tabs(e[i]); }
            tl_store14(ds + (size_t)(ks - 1) * NX, e);
        }
    } else if (!EE && live) {                                             // terminal knot: its (unused) control is carried along (tl_rollout_end)
        T u[NU];
#pragma unroll
        for (int i = 0; i < NU; i++) { u[i] = uc[(size_t)(N - 1) * NU + i]; us[(size_t)(N - 1) * NU + i] = u[i]; }
        J += arm_tl_cost<T>(cw, x, u, xg, true);
    }
    if (EE) J = acc7s[0] + acc7s[1] + acc7s[2] + acc7s[3] + acc7s[4] + acc7s[5] + acc7s[6];
    const T dseg = (seg == M - 1) ? T(0) : sdef;
    if (live) {
        b.Jpart[slot * M + seg] = J;
        b.dpart[slot * M + seg] = dseg;
        if (ls_mode < 0) b.parts_fresh[pb] = 1;
    }
    if (ls_mode >= 0) {                                                   // (wave-uniform) the line search of this wave's problems
        ls_J[lane] = J; ls_d[lane] = dseg;
        wsync();
        const int base = lane - rem;                                      // first lane of this lane's problem
        if (live && seg == 0) {                                           // lane of candidate a: the segments' parts in order
            T Jt = T(0), mx = T(0);
            for (int sgm = 0; sgm < M; sgm++) { Jt += ls_J[base + sgm * A + a_idx]; mx = tmax(mx, ls_d[base + sgm * A + a_idx]); }
            b.J[slot] = Jt; b.dmax[slot] = mx;
            ls_J[lane] = Jt; ls_d[lane] = mx;                             // (lanes base .. base + A - 1: read by the deciding lane; a segment-0 lane only overwrites its own slot,
        }                                                                 //  and every read of the parts above happens before the wave's next instruction)
        wsync();
        if (inst_raw < total && rem == 0) ls_body<T>(b, dm, sp, pb, ls_mode, ls_J + lane, ls_d + lane);
    }
}
template <typename T>
void launch_fp_tl4(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int batch, const SolverParams& sp, int ls_mode, bool maps) {
    const unsigned inst = (unsigned)batch * dm.M * dm.A;
    constexpr int lds = pipe_lds_bytes<T>(true);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fp_tl4<T, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fp_tl4<T, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fp_tl4<T, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fp_tl4<T, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const dim3 g((inst + 63) / 64), t(256);
    if (cw.ee) {
        if (variant == 0) hipLaunchKernelGGL((k_fp_tl4<T, 0, true>), g, t, lds, s, b, dm, cw, dt, grav, batch, sp, ls_mode, maps ? 1 : 0);
        else hipLaunchKernelGGL((k_fp_tl4<T, 1, true>), g, t, lds, s, b, dm, cw, dt, grav, batch, sp, ls_mode, maps ? 1 : 0);
    } else {
        if (variant == 0) hipLaunchKernelGGL((k_fp_tl4<T, 0, false>), g, t, lds, s, b, dm, cw, dt, grav, batch, sp, ls_mode, maps ? 1 : 0);
        else hipLaunchKernelGGL((k_fp_tl4<T, 1, false>), g, t, lds, s, b, dm, cw, dt, grav, batch, sp, ls_mode, maps ? 1 : 0);
    }
}
template void launch_fp_tl4<float>(hipStream_t, int, const Buffers<float>&, const Dims&, const CostWeights<float>&, float, float, int, const SolverParams&, int, bool);
template void launch_fp_tl4<double>(hipStream_t, int, const Buffers<double>&, const Dims&, const CostWeights<double>&, double, double, int, const SolverParams&, int, bool);      // the parity instantiation (PDDP_FP=tl4)

// k_sweep_st: grid ceil(2 B / 8), block 64.  The linear sweep of forwardSweepKern (fpHelpers.cuh:19-63) for ALL candidates of a problem at once.
// The sweep is affine in the step size: with e_k = x_k - xcur_k,  e_{k+1} = F_k e_k - alpha (B du)_k + [boundary] d_k,  e_0 = 0,  so
//     e_k(alpha) = t_k - alpha s_k      with   s_{k+1} = F_k s_k + (B du)_k,   t_{k+1} = F_k t_k + [boundary] d_k,   s_0 = t_0 = 0,
// and two sequences serve every alpha: group 2 pb walks s, group 2 pb + 1 walks t (8-lane groups as in fp_lg.hpp: lane l owns entries l and l + 7,
// F_k's rows stream through registers one step ahead, the running vector is broadcast inside the group by DPP), and at the M - 1 segment
// boundaries the two groups swap their vectors and write x_start(alpha_a) = xcur + (t - alpha_a s) into every candidate's slot -- a quarter of
// the lane-group sweep's work for 8 alphas, and the same numbers up to rounding (float handles only; the float32 bar covers it).
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
__global__ __launch_bounds__(64) void k_sweep_st(Buffers<float> b, Dims dm, int batch) {
    using L = LgDevice<float>;
    constexpr int NX = 14, NP = 7;
    const int gi = blockIdx.x * kLgPerWave + (threadIdx.x >> 3), pb = gi >> 1, which = gi & 1;
    if (pb >= batch || L::lane() == 7 || !fp_active<float>(b, dm, pb)) return;       // lane 7 of every group stays inactive (lanegroup.hpp)
    const unsigned pbN = (unsigned)pb * dm.N, oxc = ((unsigned)pb * 2 + b.state[pb].cur) * dm.N * NX;
    const int k_last = (dm.M - 1) * dm.NB - 1, A = dm.A, a0 = which ? (A + 1) / 2 : 0, a1 = which ? A : (A + 1) / 2;
    float eq = 0.f, ev = 0.f;                                          // entries l, l + 7 of s (which = 0) or t (which = 1)
    // F_k's two rows and the inhomogeneous term of step k, fetched kDepth steps ahead into a ring of register sets (the chain is serial and short on
    // arithmetic: what it waits for is the ~1 us latency of these gathers, so several steps of them have to be in flight)
    constexpr int kDepth = 4;
    float Aq[kDepth][NX], Av[kDepth][NX], cq[kDepth], cv[kDepth];
    const float* add = which ? b.dcur : b.Bdu;                        // the inhomogeneous term of this group's sequence
    auto fetch = [&](int slot, int k) {
        const unsigned oA = (pbN + k) * 196, ob = (pbN + k) * 14;
#pragma unroll
        for (int i = 0; i < 14; i++) { Aq[slot][i] = L::gather_at(b.ApBK, oA, [i](int l) { return l + 14 * i; }); Av[slot][i] = L::gather_at(b.ApBK, oA, [i](int l) { return l + 7 + 14 * i; }); }
        cq[slot] = L::gather_at(add, ob, [](int l) { return l; }); cv[slot] = L::gather_at(add, ob, [](int l) { return l + 7; });
    };
    auto step = [&](int slot, int k) {
        float bc[14];
        lg_bcast14<L>(bc, eq, ev);
        float vq = Aq[slot][0] * bc[0], vv = Av[slot][0] * bc[0];
#pragma unroll
        for (int i = 1; i < NX; i++) { vq += Aq[slot][i] * bc[i]; vv += Av[slot][i] * bc[i]; }
        const bool bnd = dm.on_defect_boundary(k);
        if (which == 0 || bnd) { vq += cq[slot]; vv += cv[slot]; }
        eq = vq; ev = vv;
        if (k + kDepth <= k_last) fetch(slot, k + kDepth);
        if (bnd) {                                                    // segment start states of every candidate
            const float oq = __shfl_xor(eq, 8), ov = __shfl_xor(ev, 8);
            const float sq = which ? oq : eq, sv = which ? ov : ev, tq = which ? eq : oq, tv = which ? ev : ov;
            const float nq = L::gather_at(b.xb, oxc, [k](int l) { return 14 * (k + 1) + l; }), nv = L::gather_at(b.xb, oxc, [k](int l) { return 14 * (k + 1) + l + 7; });
            for (int a = a0; a < a1; a++) {
                const float al = b.alpha[a];
                const unsigned o = (((unsigned)pb * A + a) * dm.N + k + 1) * NX;
                L::scatter_at(b.xs, o, [](int l) { return l; }, nq + (tq - al * sq), true);
                L::scatter_at(b.xs, o, [](int l) { return l + NP; }, nv + (tv - al * sv), true);
            }
        }
    };
#pragma unroll
    for (int j = 0; j < kDepth; j++) if (j <= k_last) fetch(j, j);
    for (int k0 = 0; k0 <= k_last; k0 += kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; j++) if (k0 + j <= k_last) step(j, k0 + j);
    }
}
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
void launch_sweep_st(hipStream_t s, const Buffers<float>& b, const Dims& dm, int batch) {
    hipLaunchKernelGGL(k_sweep_st, dim3((2 * (unsigned)batch + kLgPerWave - 1) / kLgPerWave), dim3(64), 0, s, b, dm, batch);
}

// k_sweep_wg: grid B, block 256, dynamic LDS.  The same two-sequence linear sweep as k_sweep_st for handles with FEW problems in flight, where the
// length of the serial chain is all that counts.  Two things shorten it:
//   * the whole workgroup first copies the chain's operands (A - B K and B du of up to 96 knots: 80 KB) into LDS with 16-byte accesses -- one memory latency for
//     the sweep instead of one per step;
//   * only the states at the M - 1 segment boundaries are wanted, and a segment's effect on (s, t) is an AFFINE MAP  e_end = Phi e_start + gamma.  So every
//     segment is walked by its OWN wavefront, all at once, carrying the 16x16 tile E = [Phi | gamma_s | gamma_t] (14 columns of the composed matrix, column 14
//     the s-part, column 15 the t-part) on the MATRIX CORE: one step E <- F_k E + [0 | B du_k | d_k] is four chained v_mfma_f32_16x16x4_f32 with F_k' as the
//     first operand (lane (g, c) needs F_k(c, 4g + r): four strided LDS words), the inhomogeneous term as the accumulator's initial value, and the result
//     already in the layout the next step consumes -- the extra columns ride for free in the instruction.  The chain is N / M steps long instead of
//     (M - 1) N / M; wave 0 then composes the M - 1 maps (two 14x14 matrix-vector products per boundary) and writes every candidate's segment start state.
// One problem (Kuka N=128, M=4): 66 us (k_sweep_lg, one 8-lane group per alpha) -> 37 us (one chain of 96 steps) -> see profiles/.  float handles only, like
// k_sweep_st; segments longer than the 96-knot staging area use k_sweep_st (launch_sweep_wg).
constexpr int kSweepWgChunk = 96;
typedef float sw4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int sw_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_sweep_wg(Buffers<float> b, Dims dm) {
    extern __shared__ __attribute__((aligned(16))) float sweep_lds[];
    constexpr int NX = 14, SZ = NX * NX;
    const int pb = blockIdx.x;
    if (!fp_active<float>(b, dm, pb)) return;                          // uniform over the workgroup
    const int NBk = dm.NB, nseg = dm.M - 1, N = dm.N;                  // segments 0 .. M - 2 end on a defect boundary
    if (nseg <= 0) return;
    const int seg_per_chunk = kSweepWgChunk / NBk, chunk = seg_per_chunk * NBk;        // whole segments per staging pass (NB <= 96: launch_sweep_wg)
    float* F = sweep_lds; float* cB = F + (size_t)chunk * SZ; float* Et = cB + (size_t)chunk * 16 + 16;   // Et: [M - 1][16][16] composed maps, row-major
    const float* gF = b.ApBK + (size_t)pb * N * SZ; const float* gB = b.Bdu + (size_t)pb * N * NX;
    const float* xcur = b.xb + ((size_t)pb * 2 + b.state[pb].cur) * N * NX; const float* dcur = b.dcur + (size_t)pb * N * NX;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, row0 = 4 * g;
    const int wave = sw_uniform((int)threadIdx.x >> 6);
    // this lane's four words of F_k' (F_k(c, 4g + r) = F[c + 14 (4g + r)]).  Rows 14, 15 of the contraction read finite neighbours and meet the zero rows 14, 15
    // of E; lanes c >= 14 would produce rows 14, 15 of the result, which are cleared after every step instead of masking the operand.
    const float* pF = F + (c < NX ? c : NX - 1) + NX * row0;
    const float* pC = cB + row0;                                        // rows 4g..4g+3 of B du_k as one 16-byte read (lanes of column 14)
    for (int s0 = 0; s0 < nseg; s0 += seg_per_chunk) {
        const int segs = nseg - s0 < seg_per_chunk ? nseg - s0 : seg_per_chunk, k0 = s0 * NBk, cnt = segs * NBk;
        __syncthreads();                                               // the previous staging pass has been consumed
        {
            const float4* src = reinterpret_cast<const float4*>(gF + (size_t)k0 * SZ); float4* dst = reinterpret_cast<float4*>(F);
            for (int i = threadIdx.x; i < cnt * (SZ / 4); i += 256) dst[i] = src[i];
            for (int i = threadIdx.x; i < cnt * 16; i += 256) { const int kk = i >> 4, e = i & 15; cB[i] = e < NX ? gB[((size_t)k0 + kk) * NX + e] : 0.f; }
            // rows 14, 15 of the LAST staged knot's contraction read the 28 words behind its block: the head of cB when the pass fills the staging area, otherwise words
            // nobody has written -- whatever the previous kernel on this compute unit left there, and a NaN or Inf pattern times the zero rows of E is a NaN
            // (found with PDDP_POISON_LDS: N = 64, M = 4 stages 48 of 96 knots)
            if (cnt < chunk && threadIdx.x < 2 * NX) F[(size_t)cnt * SZ + threadIdx.x] = 0.f;
        }
        __syncthreads();
        for (int sl = wave; sl < segs; sl += 4) {                       // this wave's segments of the pass, one after the other
            const int kbase = sl * NBk;
            sw4 E;                                                      // [Phi | gamma_s | gamma_t], starts as [I | 0 | 0]
#pragma unroll
            for (int r = 0; r < 4; r++) E[r] = (row0 + r == c && c < NX) ? 1.f : 0.f;
            for (int j = 0; j < NBk; j++) {
                const int kk = kbase + j;
                sw4 X;
#pragma unroll
                for (int r = 0; r < 4; r++) X[r] = pF[kk * SZ + NX * r];
                const sw4 cv = *reinterpret_cast<const sw4*>(pC + kk * 16);
                sw4 acc;
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = c == 14 ? cv[r] : 0.f;
                if (j == NBk - 1) {                                     // the segment's last step is its defect boundary: column 15 (t) takes the defect
#pragma unroll
                    for (int r = 0; r < 4; r++) if (c == 15 && row0 + r < NX) acc[r] = dcur[(size_t)(k0 + kk) * NX + row0 + r];
                }
#pragma unroll
                for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X[r], E[r], acc, 0, 0, 0);
                E[0] = acc[0]; E[1] = acc[1]; E[2] = g == 3 ? 0.f : acc[2]; E[3] = g == 3 ? 0.f : acc[3];
            }
            float* o = Et + (size_t)(s0 + sl) * 256;
#pragma unroll
            for (int r = 0; r < 4; r++) o[(row0 + r) * 16 + c] = E[r];
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // compose: e <- Phi_s e + gamma_s over the segments, lane l < 14 owns entry l of the s- and of the t-sequence; at every boundary the start states of all candidates
    const int l = lane < NX ? lane : NX - 1;
    float es = 0.f, et = 0.f;
    for (int sgm = 0; sgm < nseg; sgm++) {
        const float* o = Et + (size_t)sgm * 256 + l * 16;
        float ns = o[14], nt = o[15];
#pragma unroll
        for (int cc = 0; cc < NX; cc++) {
            const float ph = o[cc];
            ns = __builtin_fmaf(ph, __shfl(es, cc), ns); nt = __builtin_fmaf(ph, __shfl(et, cc), nt);
        }
        es = ns; et = nt;
        if (lane < NX) {
            const int k = (sgm + 1) * NBk - 1;
            const float base = xcur[(size_t)(k + 1) * NX + lane];
            for (int a = 0; a < dm.A; a++) b.xs[(((size_t)pb * dm.A + a) * N + k + 1) * NX + lane] = base + (et - b.alpha[a] * es);
        }
    }
}
static size_t sweep_wg_lds(const Dims& dm) {
    const int chunk = (kSweepWgChunk / dm.NB) * dm.NB;
    return ((size_t)chunk * (14 * 14 + 16) + 16 + (size_t)(dm.M > 1 ? dm.M - 1 : 1) * 256) * sizeof(float);      // operands + slack (rows 14, 15 of the last knot's contraction read past its block) + composed maps
}
void launch_sweep_wg(hipStream_t s, const Buffers<float>& b, const Dims& dm, int batch) {
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_wg), hipFuncAttributeMaxDynamicSharedMemorySize, (kSweepWgChunk * (14 * 14 + 16) + 16 + 15 * 256) * (int)sizeof(float)); attr_set = true; }
    hipLaunchKernelGGL(k_sweep_wg, dim3((unsigned)batch), dim3(256), sweep_wg_lds(dm), s, b, dm);
}

// k_nis_tl: grid ceil(B*N / 256), block 256.  Thread = knot (global knot index g = pb*N + k).  The Jacobian of a wave's 64 knots is staged in LDS and written out in
// THREE pieces as soon as their columns are complete (arm_tl_gradient's marks: columns {4..6, 11..13} first -- the composite sweep finishes the outer joints first --,
// then {0..3, 7..10}, then the controls {14..20}): 57 staged floats per knot at most, 14.6 KB per wave, so that two workgroups (two waves per SIMD -- the kernel needs
// all 256 registers) share a compute unit.
//   CAB (b.ABc != null, the sweep's default): the piece goes to the compact [A B] (ab_compact.hpp) -- the wave's 64 knots x the piece's columns x 7 dynamic rows are ONE
//                  contiguous 16-byte aligned run of the chunk; the producer stages finished elements knot-major, so the flush is a copy with 16 bytes per lane
//                  (37 groups per lane and knot instead of 147 elements with two divisions each: ~3 k of the kernel's ~13 k instructions per knot were index arithmetic);
//                  the constant rows are not written at all;
//   !CAB:          the reference layout, whole 56-byte columns, adjacent columns of a knot by adjacent lanes (224 / 168 / 392 contiguous bytes per knot and piece).
// double handles (PDDP_FP=tl: test selection) run the same code with two waves per workgroup.  Replaces integratorGradientKern + costGradientHessianKern + memcpyCurrAKern x3 (nisInitHelpers.cuh:247-279).
constexpr int kNisTlStage = 64 * 57;

template <typename T> struct NisTlCfg { static constexpr int kWaves = sizeof(T) == 4 ? 4 : 2, kThreads = 64 * kWaves; };   // double: two waves per workgroup (58 KB of staging)
template <typename T> struct NisTlVec { typedef T v4 __attribute__((ext_vector_type(4), aligned(16))); typedef T v4p __attribute__((ext_vector_type(16 / sizeof(T)), aligned(16))); };   // v4p: one 16-byte piece
template <typename T, int V, bool EE, bool CAB>
__global__ __launch_bounds__(NisTlCfg<T>::kThreads, sizeof(T) == 4 ? 2 : 1) void k_nis_tl(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, T grav, int mode, int batch) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    constexpr int NX = 14, NM = 21;
    const int g = blockIdx.x * NisTlCfg<T>::kThreads + threadIdx.x, total = batch * dm.N;
    const int pb = g / dm.N, k = g - pb * dm.N;
    __shared__ __attribute__((aligned(16))) T stage_all[NisTlCfg<T>::kWaves * kNisTlStage];
    T* stage = stage_all + (threadIdx.x >> 6) * kNisTlStage;
    const int lane = threadIdx.x & 63;
    T x[NX], u[7];
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = T(0);                    // a lane without a knot differentiates the zero state (its columns are never flushed)
#pragma unroll
    for (int i = 0; i < 7; i++) u[i] = T(0);
    bool need = false;
    if (EE) { if (g < total) need = arm_tl_nis_cost_ee<T>(md, b, dm, cw, mode, k, pb, x, u); }
    else {
        // joint-space cost: the wave's 64 gradients g_k (21 elements each, consecutive knots = 84 x 64 contiguous bytes) leave through the staging area as 16-byte pieces
        T gl[NM];
        const int r = (g < total) ? arm_tl_nis_cost_vals<T>(b, dm, cw, mode, k, pb, x, u, gl, false) : 0;
        need = (r == 3);
        constexpr int per16 = 16 / (int)sizeof(T);                        // elements per 16-byte piece
        auto run_out = [&](T* dst, const T* v, auto cnt) {               // element i of lane l -> dst[l * n + i]: through the staging area, whole pieces per lane
            constexpr int n = decltype(cnt)::value;
#pragma unroll
            for (int i = 0; i < n; i++) stage[lane * n + i] = v[i];
            wsync();
            for (int c = lane; c < 64 * n / per16; c += 64)
                *reinterpret_cast<typename NisTlVec<T>::v4p*>(dst + c * per16) = *reinterpret_cast<const typename NisTlVec<T>::v4p*>(stage + c * per16);
            wsync();
        };
        const bool adopting = (mode == 0) && r != 0;                      // (uniform per problem) the accepted state / control are in x, u and still have to go out
        if (__ballot(r != 0) == ~0ull && (dm.N & 63) == 0) {              // the wave's 64 knots belong to ONE problem and all of them moved: contiguous runs
            const size_t k0 = (size_t)(g - lane);                         // global index of the wave's first knot
            if (adopting) {
                run_out(b.xb + ((size_t)pb * 2 + b.state[pb].cur) * dm.N * NX + (size_t)(k - lane) * NX, x, std::integral_constant<int, NX>{});
                run_out(b.ucur + k0 * 7, u, std::integral_constant<int, 7>{});
            }
            if (r & 1) run_out(b.g + k0 * NM, gl, std::integral_constant<int, NM>{});
        } else if (r) {
            if (adopting) {
                tl_store14(b.xb + (((size_t)pb * 2 + b.state[pb].cur) * dm.N + k) * NX, x);
#pragma unroll
                for (int i = 0; i < 7; i++) b.ucur[(size_t)g * 7 + i] = u[i];
            }
            if (r & 1) {
                T* gk = b.g + (size_t)g * NM;
#pragma unroll
                for (int i = 0; i < NM; i++) gk[i] = gl[i];
            }
        }
    }
    const unsigned long long mask = __ballot(need);
    if (!mask) return;                                          // (uniform) nothing to differentiate in this wave
    T* AB0 = b.AB + (size_t)(g - lane) * (NX * NM);            // [A B] of the wave's first knot (reference layout)
    T* ABc0 = CAB ? b.ABc + (size_t)((g - lane) >> 6) * kAbcChunk : nullptr;   // the wave's chunk of the compact array
    constexpr bool compact = CAB;
    // Staging.  Compact: the producer leaves the FINISHED element ([A B] = I + dt dqdd) at knot * P + (column in piece) * 7 + row, P = the piece's elements per knot
    // made odd (57 / 43 / 49: conflict-free writes) -- the flush is then a copy of the piece's run, 16 bytes per lane, whose only arithmetic is the step over the
    // pad at a knot boundary.  Reference layout: raw dqdd entry-major (entry * 65 + knot), finished by the lanes that write whole columns.
    auto emit = [&](int col, int row, T val) {
        const int ent = abc_col_in_piece(col) * 7 + row;
        if (compact) stage[lane * ((abc_piece_cols(abc_piece(col)) * 7) | 1) + ent] = T(col == 7 + row ? 1 : 0) + dt * val;
        else stage[ent * 65 + lane] = val;
    };
    auto flush = [&](auto pc) {
        constexpr int piece = decltype(pc)::value;
        wsync();
        constexpr int ncols = abc_piece_cols(piece);
        if (compact) {
            constexpr int per = ncols * 7, P = per | 1, count = 64 * per;      // elements of this piece per knot / staged stride / per wave (a multiple of 4)
            T* dst = ABc0 + abc_piece_off(piece);
            int kk = (4 * lane) / per, ent = 4 * lane - kk * per;
            const bool whole = (mask == ~0ull);
            // one group of four consecutive elements, every element tested against the mask (ragged waves: knots of several problems, some of them not moving)
            auto group = [&](int e0, int kk, int ent, int limit) {
                T out[4]; bool ok[4], all = true;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool over = ent + j >= per;
                    const int k2 = over ? kk + 1 : kk, en = over ? ent + j - per : ent + j;
                    out[j] = stage[k2 * P + en];
                    ok[j] = e0 + j < limit && (whole || ((mask >> k2) & 1ull)); all = all && ok[j];
                }
                if (all) { typename NisTlVec<T>::v4 v; v[0] = out[0]; v[1] = out[1]; v[2] = out[2]; v[3] = out[3]; *reinterpret_cast<typename NisTlVec<T>::v4*>(dst + e0) = v; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; j++) if (ok[j]) dst[e0 + j] = out[j];
                }
            };
            // The two masks a wave of ONE problem has when it moves: every knot, or every knot but the problem's terminal one (lane 63: it has no [A B]).  Then the wave's
            // first nk knots are one dense run of nk * per elements and the flush is a copy: staged index of element e = e + (knot of e) * (P - per), so a group of four
            // is five consecutive staged words with at most one step over a knot's pad (t = elements left in the knot) -- ~20 instructions per group where the tested
            // form costs ~75 (three 32-bit multiplies and four 64-bit mask shifts per group): a quarter of the kernel's instructions were this loop.
            const int nk = whole ? 64 : (mask == (~0ull >> 1) ? 63 : 0);
            if (nk) {
                const int cnt = nk * per;
                int e0 = 4 * lane, sb = e0 + kk * (P - per), t = per - ent;
#pragma unroll 2
                for (; e0 + 3 < cnt; e0 += 256) {
                    typename NisTlVec<T>::v4 v;
                    if constexpr (P == per) { v[0] = stage[sb]; v[1] = stage[sb + 1]; v[2] = stage[sb + 2]; v[3] = stage[sb + 3]; }
                    else {
                        const T r0 = stage[sb], r1 = stage[sb + 1], r2 = stage[sb + 2], r3 = stage[sb + 3], r4 = stage[sb + 4];
                        v[0] = r0; v[1] = t > 1 ? r1 : r2; v[2] = t > 2 ? r2 : r3; v[3] = t > 3 ? r3 : r4;
                    }
                    *reinterpret_cast<typename NisTlVec<T>::v4*>(dst + e0) = v;
                    sb += 256 + (256 / per) * (P - per); t -= 256 % per; kk += 256 / per;
                    if (t <= 0) { t += per; sb += P - per; kk++; }
                }
                if (e0 < cnt) group(e0, kk, per - t, cnt);                       // the group the run ends in (nk = 63: cnt need not be a multiple of four)
            } else {
                for (int e0 = 4 * lane; e0 < count; e0 += 256) {
                    group(e0, kk, ent, count);
                    ent += 256 % per; kk += 256 / per;
                    if (ent >= per) { ent -= per; kk++; }
                }
            }
        } else {
            for (int it = 0; it * 64 < 64 * ncols; it++) {
                const int pi = it * 64 + lane, kk = pi / ncols, ci = pi - kk * ncols;
                if (kk >= 64 || !((mask >> kk) & 1ull)) continue;
                const int col = abc_piece_col(piece, ci);
                T out[NX];
#pragma unroll
                for (int r = 0; r < 7; r++) {
                    out[r] = tl_AB_const<T>(r, col, dt);
                    out[7 + r] = T(col == 7 + r ? 1 : 0) + dt * stage[(ci * 7 + r) * 65 + kk];
                }
                tl_store14(AB0 + ((size_t)kk * NM + col) * NX, out);
            }
        }
        wsync();
    };
    arm_tl_nis_jac<T>(md, grav, x, u, emit, flush);
}

// k_nis_tl7: grid (ceil(B*N / 64), 7), block 64.  Next-iteration setup of a handle with FEW problems in flight (one MPC solve: 127 knots on a 256-CU device):
// thread = (knot, joint J); blockIdx.y = J, so a wave differentiates ONE joint of 64 knots -- every thread recomputes the knot's forward dynamics and nominal inverse
// dynamics (~2.3 k instructions) and then only joint J's tangent pass (columns J, 7 + J, 14 + J of [A B]): the longest instruction stream is ~4.5 k instead of the
// ~11 k of a whole knot on one thread (k_nis_tl), on 7 x as many waves.  Joint 0's threads also adopt the accepted candidate (a copy of its x, u, d from the
// candidate-major arrays the split rollout kernel k_fp_tl2 wrote) and write the cost gradient (mode 1: the cost Hessian).  Reference layout of [A B].
// Replaces integratorGradientKern + costGradientHessianKern + memcpyCurrAKern x3 (nisInitHelpers.cuh:247-279), like k_nis_lg.
// EE: the end-effector cost family -- an eighth row of workgroups (blockIdx.y == 7) evaluates the tool point, its Jacobian, g_k and the position block of H_k
// (arm_tl_nis_cost_ee_knot), one thread per knot.
template <typename T, int V, bool EE>
__global__ __launch_bounds__(64) void k_nis_tl7(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, T grav, int mode, int batch) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    constexpr int NX = 14, NU = 7, NM = 21;
    const int J = blockIdx.y, g = blockIdx.x * 64 + threadIdx.x, N = dm.N;
    if (g >= batch * N) return;
    const int pb = g / N, k = g - pb * N;
    const SolverState<T>& st = b.state[pb];
    const size_t knot = (size_t)pb * N + k;
    T x[NX], u[NU];
    if (mode == 0) {
        if (!st.win_pending) return;                                      // rejected / failed backward pass: trajectory and derivatives are unchanged
        const size_t src = ((size_t)pb * dm.A + st.alphaIndex) * N + k;   // this knot in the winner's candidate slot
        tl_load14(x, b.xs + src * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) u[i] = b.us[src * NU + i];
        if (J == 0) {
            tl_store14(b.xb + (((size_t)pb * 2 + st.cur) * N + k) * NX, x);
#pragma unroll
            for (int i = 0; i < NU; i++) b.ucur[knot * NU + i] = u[i];
            if (dm.M > 1 && dm.on_defect_boundary(k)) { T d[NX]; tl_load14(d, b.ds + src * NX); tl_store14(b.dcur + knot * NX, d); }
        }
        if (st.done) return;                                              // final accepted step: solution copied, no derivatives needed
    } else {
        tl_load14(x, b.xb + (((size_t)pb * 2 + st.cur) * N + k) * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) u[i] = b.ucur[knot * NU + i];
    }
    const bool fin = (k == N - 1);
    if (EE) {
        if (J == 7) { arm_tl_nis_cost_ee_knot<T>(md, b, dm, cw, mode, k, pb, x, u, true); return; }
    } else if (J == 0) {
        const T w1 = fin ? cw.QF1 : cw.Q1, w2 = fin ? cw.QF2 : cw.Q2, w3 = fin ? T(0) : cw.R;   // ArmPlant::weight
        T xg[NX];
        tl_load14(xg, b.xGoal + (size_t)pb * NX);
        T* gk = b.g + knot * NM;
#pragma unroll
        for (int i = 0; i < 7; i++) { gk[i] = w1 * (x[i] - xg[i]); gk[7 + i] = w2 * (x[7 + i] - xg[7 + i]); gk[14 + i] = w3 * u[i]; }
        if (cw.limits) { const int n = fin ? NX : NM; for (int i = 0; i < n; i++) gk[i] += arm_limit_term<T>(x, u, i, 1); }                     // USE_LIMITS_FLAG (cost_arm.cuh:176-199)
        if (mode == 1) {
            T* H = b.H + knot * (NM * NM);
            for (int e = 0; e < NM * NM; e++) { const int i = e / NM, j = e % NM; H[e] = i != j ? T(0) : (i < 7 ? w1 : (i < NX ? w2 : w3)); }
        }
    }
    if (fin) return;
    ArmTlState<T> ts;
    T qdd[7];
    arm_tl_dynamics<T>(md, grav, ts, qdd, x, x + 7, u);
    T* AB = b.AB + knot * (NX * NM);
    auto emit = [&](int col, int row, T val) { AB[col * NX + 7 + row] = T(col == 7 + row ? 1 : 0) + dt * val; };
    ArmTlNominal<T> nm;
    arm_tl_grad_nominal<T>(md, grav, ts, x + 7, qdd, nm);
    switch (J) {                                                          // uniform over the wave
#define PDDP_TL7_CASE(JJ) case JJ: arm_tl_grad_joint<JJ, T>(md, ts, x + 7, qdd, nm, emit); arm_tl_grad_control<JJ, T>(ts, emit); break;
        PDDP_TL7_CASE(0) PDDP_TL7_CASE(1) PDDP_TL7_CASE(2) PDDP_TL7_CASE(3) PDDP_TL7_CASE(4) PDDP_TL7_CASE(5) PDDP_TL7_CASE(6)
#undef PDDP_TL7_CASE
    }
#pragma unroll
    for (int r = 0; r < 7; r++) {                                         // the constant rows of this thread's three columns
        AB[J * NX + r] = tl_AB_const<T>(r, J, dt); AB[(7 + J) * NX + r] = tl_AB_const<T>(r, 7 + J, dt); AB[(14 + J) * NX + r] = tl_AB_const<T>(r, 14 + J, dt);
    }
}
template <typename T>
void launch_nis_tl7(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int mode, int batch) {
    const dim3 grid(((unsigned)batch * dm.N + 63) / 64, cw.ee ? 8 : 7);
    if (cw.ee) {
        if (variant == 0) hipLaunchKernelGGL((k_nis_tl7<T, 0, true>), grid, dim3(64), 0, s, b, dm, cw, dt, grav, mode, batch);
        else hipLaunchKernelGGL((k_nis_tl7<T, 1, true>), grid, dim3(64), 0, s, b, dm, cw, dt, grav, mode, batch);
    } else if (variant == 0) hipLaunchKernelGGL((k_nis_tl7<T, 0, false>), grid, dim3(64), 0, s, b, dm, cw, dt, grav, mode, batch);
    else hipLaunchKernelGGL((k_nis_tl7<T, 1, false>), grid, dim3(64), 0, s, b, dm, cw, dt, grav, mode, batch);
}
template void launch_nis_tl7<float>(hipStream_t, int, const Buffers<float>&, const Dims&, const CostWeights<float>&, float, float, int, int);
template void launch_nis_tl7<double>(hipStream_t, int, const Buffers<double>&, const Dims&, const CostWeights<double>&, double, double, int, int);     // the parity instantiation (PDDP_FP=tl4)

// API view of the compact [A B]: grid ceil(B*N*21 / 256), block 256, thread = (knot, column).  expand: compact -> the reference layout (pddp_get_array("AB"));
// compact: the reference layout -> compact (pddp_set_array("AB"): teacher-forced tests hand in the oracle's derivatives).
template <typename T>
__global__ __launch_bounds__(256) void k_abc_convert(Buffers<T> b, int knots, int N, T dt, int to_compact) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= knots * 21) return;
    const int G = i / 21, col = i - G * 21;
    if (!to_compact && (G % N) == N - 1) return;                      // the terminal knot has no [A B]: the kernels never write it, the view keeps what it holds
    T* full = b.AB + ((size_t)G * 21 + col) * 14;
    T* cmp = b.ABc + abc_index((size_t)G, col, 0);
    if (to_compact) { for (int r = 0; r < 7; r++) cmp[r] = full[7 + r]; }
    else { for (int r = 0; r < 7; r++) { full[r] = tl_AB_const<T>(r, col, dt); full[7 + r] = cmp[r]; } }
}
template <typename T>
void launch_abc_convert(hipStream_t s, const Buffers<T>& b, int knots, int N, T dt, int to_compact) {
    hipLaunchKernelGGL((k_abc_convert<T>), dim3(((unsigned)knots * 21 + 255) / 256), dim3(256), 0, s, b, knots, N, dt, to_compact);
}
template void launch_abc_convert<float>(hipStream_t, const Buffers<float>&, int, int, float, int);
template void launch_abc_convert<double>(hipStream_t, const Buffers<double>&, int, int, double, int);

// forward dynamics / gradient of `count` (x, u) samples, one thread each (tests, micro-benchmarks)
template <typename T, int V>
__global__ __launch_bounds__(256, 1) void k_plant_eval_tl(T grav, int count, const T* x, const T* u, T* out, int grad) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    T xi[14], ui[7], qdd[7];
    for (int e = 0; e < 14; e++) xi[e] = x[(size_t)i * 14 + e];
    for (int e = 0; e < 7; e++) ui[e] = u[(size_t)i * 7 + e];
    ArmTlState<T> st;
    if (grad == 2) {                                                   // tool point + Jacobian (`grav` carries EE_ON_LINK_Z here): out[i][6 + 42]
        ArmTlFrames<T> fr;
        arm_tl_trig<T>(st, xi); arm_tl_world_chain<true, T>(md, st.c, st.s, fr);
        T pos[6], dpos[42];
        arm_tl_tool_point<T>(fr, grav, true, pos); arm_tl_tool_jacobian<T>(fr, grav, dpos);
        for (int e = 0; e < 6; e++) out[(size_t)i * 48 + e] = pos[e];
        for (int e = 0; e < 42; e++) out[(size_t)i * 48 + 6 + e] = dpos[e];
        return;
    }
    arm_tl_dynamics<T>(md, grav, st, qdd, xi, xi + 7, ui);
    if (!grad) { for (int e = 0; e < 7; e++) out[(size_t)i * 7 + e] = qdd[e]; return; }
    T* o = out + (size_t)i * 147;
    arm_tl_gradient<T>(md, grav, st, xi + 7, qdd, [o](int col, int row, T val) { o[7 * col + row] = val; });
}

template <typename T>
void launch_fp_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int batch, int store_candidates) {
    const unsigned inst = (unsigned)batch * dm.M * dm.A;
    const dim3 g((inst + 255) / 256), t(256);
#define PDDP_FP_TL(VV, AL, EEV) hipLaunchKernelGGL((k_fp_tl<T, VV, AL, EEV>), g, t, 0, s, b, dm, cw, dt, grav, batch)
    if (cw.ee) {
        if (variant == 0) { if (store_candidates) PDDP_FP_TL(0, true, true); else PDDP_FP_TL(0, false, true); }
        else { if (store_candidates) PDDP_FP_TL(1, true, true); else PDDP_FP_TL(1, false, true); }
    } else {
        if (variant == 0) { if (store_candidates) PDDP_FP_TL(0, true, false); else PDDP_FP_TL(0, false, false); }
        else { if (store_candidates) PDDP_FP_TL(1, true, false); else PDDP_FP_TL(1, false, false); }
    }
#undef PDDP_FP_TL
}
template <typename T>
void launch_nis_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int mode, int batch) {
    const unsigned knots = (unsigned)batch * dm.N, th = NisTlCfg<T>::kThreads;
    const dim3 g((knots + th - 1) / th), t(th);
#define PDDP_NIS_TL(VV, EEV) do { if (b.ABc) hipLaunchKernelGGL((k_nis_tl<T, VV, EEV, true>), g, t, 0, s, b, dm, cw, dt, grav, mode, batch); \
                                  else hipLaunchKernelGGL((k_nis_tl<T, VV, EEV, false>), g, t, 0, s, b, dm, cw, dt, grav, mode, batch); } while (0)
    if (cw.ee) { if (variant == 0) PDDP_NIS_TL(0, true); else PDDP_NIS_TL(1, true); }
    else { if (variant == 0) PDDP_NIS_TL(0, false); else PDDP_NIS_TL(1, false); }
#undef PDDP_NIS_TL
}
template <typename T>
void launch_plant_eval_tl(hipStream_t s, int variant, T grav, int count, const T* x, const T* u, T* out, int grad) {
    if (variant == 0) hipLaunchKernelGGL((k_plant_eval_tl<T, 0>), dim3((count + 255) / 256), dim3(256), 0, s, grav, count, x, u, out, grad);
    else hipLaunchKernelGGL((k_plant_eval_tl<T, 1>), dim3((count + 255) / 256), dim3(256), 0, s, grav, count, x, u, out, grad);
}
template void launch_fp_tl<float>(hipStream_t, int, const Buffers<float>&, const Dims&, const CostWeights<float>&, float, float, int, int);
template void launch_fp_tl<double>(hipStream_t, int, const Buffers<double>&, const Dims&, const CostWeights<double>&, double, double, int, int);
template void launch_nis_tl<float>(hipStream_t, int, const Buffers<float>&, const Dims&, const CostWeights<float>&, float, float, int, int);
template void launch_nis_tl<double>(hipStream_t, int, const Buffers<double>&, const Dims&, const CostWeights<double>&, double, double, int, int);
template void launch_plant_eval_tl<float>(hipStream_t, int, float, int, const float*, const float*, float*, int);
template void launch_plant_eval_tl<double>(hipStream_t, int, double, int, const double*, const double*, double*, int);

}  // namespace pddp
