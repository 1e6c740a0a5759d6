// Forward pass and next-iteration setup of the KUKA arm, ONE THREAD PER INSTANCE (plant_arm_tl.hpp): a thread owns one
// (line-search candidate, shooting segment) rollout, or one knot's derivatives; a wave carries 64 of them.
//
// Same functions as the lane-group bodies (fp_lg.hpp, nis_lg.hpp), which restate forwardSimKern / forwardSimInner /
// computeControlKT / costKern / defectKern (DDPHelpers/fpHelpers.cuh:279-301, 225-275, 202-221, 134-152, 96-111) and
// integratorGradientKern + costGradientHessianKern + memcpyCurrAKern + the copies of nextIterationSetupGPU
// (DDPHelpers/nisInitHelpers.cuh:205-221, 46-93, 24-32, 247-279).  What differs:
//   * arithmetic: the body-coordinate dynamics of plant_arm_tl.hpp, fused multiply-adds, running sums in plain serial order
//     (a segment's cost is one register; the reference's pairwise tree over knots becomes: serial within a segment, then the
//     segments in order -- the line-search kernel adds the M partial sums per candidate);
//   * data movement: instance i = ((problem * M + segment) * A + candidate), so the 8 candidates of one (problem, segment) sit in
//     adjacent lanes and read the SAME gains / reference states: one address per 8 lanes, served as broadcasts by the memory
//     pipeline; the per-knot blocks are read with 8-byte accesses straight from global memory (they are 8-byte aligned: 98, 14
//     and 294 floats per knot);
//   * the gradient kernel stages a wave's 64 knots x 147 Jacobian entries through LDS and writes [A B] as ONE contiguous 75 KB
//     run of full cache lines (k_nis_tl in kernels.hpp) -- every line written once.
#pragma once

#include "plant_arm_tl.hpp"
#include "plants.hpp"
#include "solver_state.hpp"

// fused multiply-adds for everything in this header (the rest of the library is built with -ffp-contract=off; see the header comment)
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

namespace pddp {

// Which implementation of the arm's forward pass / next-iteration setup a handle uses.  PDDP_FP=coop|lg|tl overrides (comparison tests).
//   coop: one wave per unit (fp.hpp, nis.hpp)   lg: 8-lane groups (fp_lg.hpp, nis_lg.hpp)   tl: one thread per instance (this file)
// Default: tl for float handles with the joint-space cost and an iiwa-structured model; lg otherwise (float64 handles keep the reference's
// operation order, which the 1e-9 parity tests rely on; the end-effector cost family and the initial rollout exist on lane groups only).
enum FpPath { kFpCoop = 0, kFpLg = 1, kFpTl = 2 };
inline FpPath select_fp_path(const char* env, bool is_float, bool ee_cost, bool tl_model_ok) {
    const bool tl_possible = !ee_cost && tl_model_ok;
    if (env) {
        if (env[0] == 'c') return kFpCoop;
        if (env[0] == 'l') return kFpLg;
        if (env[0] == 't' && tl_possible) return kFpTl;
    }
    return (is_float && tl_possible) ? kFpTl : kFpLg;
}

// 14 floats of one knot as seven 8-byte accesses (every x / d / xGoal knot block is 56 bytes, 8-byte aligned)
template <typename T> struct TlPair { T a, b; };
template <typename T> PDDP_HD void tl_load14(T* o, const T* p) {
#pragma unroll
    for (int i = 0; i < 7; i++) { const TlPair<T> v = *reinterpret_cast<const TlPair<T>*>(p + 2 * i); o[2 * i] = v.a; o[2 * i + 1] = v.b; }
}
template <typename T> PDDP_HD void tl_store14(T* p, const T* v) {
#pragma unroll
    for (int i = 0; i < 7; i++) { TlPair<T> t; t.a = v[2 * i]; t.b = v[2 * i + 1]; *reinterpret_cast<TlPair<T>*>(p + 2 * i) = t; }
}

// running / terminal joint-space cost of one knot (ArmPlant::cost, plants.hpp; plants/cost_arm.cuh:130-153)
template <typename T>
PDDP_HD T arm_tl_cost(const CostWeights<T>& cw, const T* x, const T* u, const T* xg, bool final_knot) {
    T sq = T(0), sv = T(0), su = T(0);
#pragma unroll
    for (int i = 0; i < 7; i++) { const T dq = x[i] - xg[i], dv = x[7 + i] - xg[7 + i]; sq += dq * dq; sv += dv * dv; }
    if (final_knot) return T(0.5) * (cw.QF1 * sq + cw.QF2 * sv);
#pragma unroll
    for (int i = 0; i < 7; i++) su += u[i] * u[i];
    return T(0.5) * (cw.Q1 * sq + cw.Q2 * sv + cw.R * su);
}

// Rollout of shooting segment `seg` of candidate `a_idx` of problem pb.  Writes the candidate's x, u (and the boundary defect) and
// the segment's partial cost / defect norm into b.Jpart / b.dpart [(pb*A + a)*M + seg].
template <typename T>
PDDP_HD void arm_tl_rollout_segment(const ArmTlModel<T>& md, T grav, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, int pb, int a_idx, int seg) {
    constexpr int NX = 14, NU = 7;
    const int N = dm.N, NBk = dm.NB, kStart = seg * NBk;
    const int iters = (seg < dm.M - 1) ? NBk : NBk - 1;
    const size_t slot = (size_t)pb * dm.A + a_idx;
    T* xs = b.xs + slot * N * NX; T* us = b.us + slot * N * NU; T* ds = b.ds + slot * N * NX;
    const T* xc = b.xb + ((size_t)pb * 2 + b.state[pb].cur) * N * NX;
    const T* uc = b.ucur + (size_t)pb * N * NU; const T* du = b.du + (size_t)pb * N * NU; const T* KT = b.KT + (size_t)pb * N * NX * NU;
    const T alpha = b.alpha[a_idx];
    T xg[NX];
    tl_load14(xg, b.xGoal + (size_t)pb * NX);
    T x[NX];
    if (seg == 0) { tl_load14(x, xc); tl_store14(xs, x); }                 // segment starts: the current state, or what the linear sweep left
    else tl_load14(x, xs + (size_t)kStart * NX);
    ArmTlState<T> st;
    T J = T(0);
    for (int k = 0; k < iters; k++) {
        const int kn = kStart + k;
        T dx[NX], u[NU];
        {
            T xr[NX];
            tl_load14(xr, xc + (size_t)kn * NX);
#pragma unroll
            for (int i = 0; i < NX; i++) dx[i] = x[i] - xr[i];
        }
        const T* Kk = KT + (size_t)kn * (NX * NU);                        // K(r, c) = KT[c + r*14]: row r contiguous
#pragma unroll
        for (int r = 0; r < NU; r++) {
            T row[NX];
            tl_load14(row, Kk + r * NX);
            T acc = alpha * du[(size_t)kn * NU + r];
#pragma unroll
            for (int c = 0; c < NX; c++) acc += row[c] * dx[c];
            u[r] = uc[(size_t)kn * NU + r] - acc;
            us[(size_t)kn * NU + r] = u[r];
        }
        J += arm_tl_cost<T>(cw, x, u, xg, false);
        T qdd[7];
        arm_tl_dynamics<T>(md, grav, st, qdd, x, x + 7, u);
        T xn[NX];
#pragma unroll
        for (int i = 0; i < 7; i++) { xn[i] = x[i] + dt * x[7 + i]; xn[7 + i] = x[7 + i] + dt * qdd[i]; }     // Euler (utils/integrators.cuh:24-36)
        if (k < NBk - 1) {
            tl_store14(xs + (size_t)(kn + 1) * NX, xn);
#pragma unroll
            for (int i = 0; i < NX; i++) x[i] = xn[i];
        } else {                                                          // last step of a non-final segment: defect against the next segment's start
            const int ks = (seg + 1) * NBk;
            T xnext[NX], e[NX], sdef = T(0);
            tl_load14(xnext, xs + (size_t)ks * NX);
#pragma unroll
            for (int i = 0; i < NX; i++) { e[i] = xn[i] - xnext[i]; sdef += tabs(e[i]); }
            tl_store14(ds + (size_t)(ks - 1) * NX, e);
            b.dpart[slot * dm.M + seg] = sdef;
        }
    }
    if (seg == dm.M - 1) {                                                // terminal knot: its (unused) control is carried along
        const int kn = N - 1;
        T u[NU];
#pragma unroll
        for (int r = 0; r < NU; r++) { u[r] = uc[(size_t)kn * NU + r]; us[(size_t)kn * NU + r] = u[r]; }
        J += arm_tl_cost<T>(cw, x, u, xg, true);
        b.dpart[slot * dm.M + seg] = T(0);
    }
    b.Jpart[slot * dm.M + seg] = J;
    b.parts_fresh[pb] = 1;                                                // every thread of the problem stores the same value
}

// the line-search kernel's first step when the thread-lane forward pass ran: J[a] = sum of the segment partial sums in order, dmax[a] = max
template <typename T>
PDDP_HD void tl_reduce_parts(const Buffers<T>& b, const Dims& dm, int pb) {
    for (int a = 0; a < dm.A; a++) {
        const size_t slot = (size_t)pb * dm.A + a;
        T J = T(0), mx = T(0);
        for (int s = 0; s < dm.M; s++) { J += b.Jpart[slot * dm.M + s]; mx = tmax(mx, b.dpart[slot * dm.M + s]); }
        b.J[slot] = J; b.dmax[slot] = mx;
    }
}

// Next-iteration setup of knot k of problem pb (nis_body / arm_lg_nis_body): adopt the winner, g_k, (mode 1: H_k), and the Jacobian of the
// dynamics through emit(col, row, dqdd) -- the caller turns it into [A B] rows 7..13 (k_nis_tl stages it through LDS).  Returns false when
// this knot has no Jacobian to write (rejected / failed iteration, final knot, finished problem).
template <typename T, typename Emit>
PDDP_HD bool arm_tl_nis_knot(const ArmTlModel<T>& md, T grav, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, int mode, int k, int pb, Emit emit) {
    constexpr int NX = 14, NU = 7, NM = 21;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    const size_t knot = (size_t)pb * N + k;
    T* xc = b.xb + (((size_t)pb * 2 + st.cur) * N + k) * NX;
    T* uc = b.ucur + knot * NU;
    T x[NX], u[NU];
    if (mode == 0) {
        if (st.accepted != 1) return false;                               // rejected, or the backward pass failed: nothing moved
        const size_t wknot = ((size_t)pb * dm.A + st.alphaIndex) * N + k;
        tl_load14(x, b.xs + wknot * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) u[i] = b.us[wknot * NU + i];
        tl_store14(xc, x);
#pragma unroll
        for (int i = 0; i < NU; i++) uc[i] = u[i];
        if (dm.M > 1 && dm.on_defect_boundary(k)) { T d[NX]; tl_load14(d, b.ds + wknot * NX); tl_store14(b.dcur + knot * NX, d); }
        if (st.done) return false;                                        // final accepted step: solution copied, no derivatives needed
    } else {
        tl_load14(x, xc);
#pragma unroll
        for (int i = 0; i < NU; i++) u[i] = uc[i];
    }
    const bool fin = (k == N - 1);
    const T w1 = fin ? cw.QF1 : cw.Q1, w2 = fin ? cw.QF2 : cw.Q2, w3 = fin ? T(0) : cw.R;       // ArmPlant::weight
    T xg[NX];
    tl_load14(xg, b.xGoal + (size_t)pb * NX);
    T* g = b.g + knot * NM;
#pragma unroll
    for (int i = 0; i < 7; i++) { g[i] = w1 * (x[i] - xg[i]); g[7 + i] = w2 * (x[7 + i] - xg[7 + i]); g[14 + i] = w3 * u[i]; }
    if (mode == 1) {                                                      // H_k = diag(weight): constant over the solve, written once
        T* H = b.H + knot * (NM * NM);
        for (int e = 0; e < NM * NM; e++) { const int i = e / NM, j = e % NM; H[e] = i != j ? T(0) : (i < 7 ? w1 : (i < NX ? w2 : w3)); }
    }
    if (fin) return false;
    ArmTlState<T> ts;
    T qdd[7];
    arm_tl_dynamics<T>(md, grav, ts, qdd, x, x + 7, u);
    arm_tl_gradient<T>(md, grav, ts, x + 7, qdd, emit);
    return true;
}

// [A B] entry (row, col) of the Euler step from the Jacobian of the dynamics: I + dt [0 I 0; dqdd]   (utils/integrators.cuh:38-53)
template <typename T> PDDP_HD T tl_AB_const(int row, int col, T dt) { return T(col == row ? 1 : 0) + (col == row + 7 ? dt : T(0)); }   // rows 0..6

}  // namespace pddp

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
