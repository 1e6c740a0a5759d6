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

#include "ee_cost.hpp"
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
// Default: tl for float handles of >= 512 problems with the joint-space cost and an iiwa-structured model; lg otherwise (float64 handles keep
// the reference's operation order, which the 1e-9 parity tests rely on; the end-effector cost family and the initial rollout exist on lane
// groups only; below ~512 problems a thread per knot leaves most of the GPU idle and the lane groups' shorter critical path wins:
// profiles/r02_path_sweep.txt -- one problem: setup 33 us on lane groups, 145 us on thread lanes; 1024 problems: 0.37 ms vs 0.24 ms).
enum FpPath { kFpCoop = 0, kFpLg = 1, kFpTl = 2 };
constexpr int kFpTlMinBatch = 512;
inline FpPath select_fp_path(const char* env, bool is_float, bool ee_cost, bool tl_model_ok, int batch) {
    const bool tl_possible = tl_model_ok;          // (joint-space and end-effector cost)
    (void)ee_cost;
    if (env) {
        if (env[0] == 'c') return kFpCoop;
        if (env[0] == 'l') return kFpLg;
        if (env[0] == 't' && env[1] == 'l' && (env[2] == '2' || env[2] == '4')) return kFpLg;      // "tl2" / "tl4": the split rollout kernels of a few-problem handle (a lane-group handle otherwise; "tl4" also selects them on a double handle)
        if (env[0] == 't' && tl_possible) return kFpTl;
    }
    return (is_float && tl_possible && batch >= kFpTlMinBatch) ? kFpTl : kFpLg;
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
    T cost;
    if (final_knot) cost = T(0.5) * (cw.QF1 * sq + cw.QF2 * sv);
    else {
#pragma unroll
        for (int i = 0; i < 7; i++) su += u[i] * u[i];
        cost = T(0.5) * (cw.Q1 * sq + cw.Q2 * sv + cw.R * su);
    }
    if (cw.limits) {                                                      // USE_LIMITS_FLAG (cost_arm.cuh:136-139,147-150): added after the halving, in index order
        const int n = final_knot ? 14 : 21;
        for (int i = 0; i < n; i++) cost += arm_limit_term<T>(x, u, i, 0);
    }
    return cost;
}

// The control law of one knot, u = uc - (alpha du + K (x - xr))   (computeControlKT, DDPHelpers/fpHelpers.cuh:202-221): ONE definition with explicit
// fused multiply-adds, because two kernels evaluate it for the accepted candidate -- the rollout that produced its states and the setup kernel that
// adopts them (arm_tl_adopt_knot) -- and both must get the same bits.  Kk[98] = K(r, c) at [c + 14 r].
template <typename T> PDDP_HD T tl_fma(T a, T b, T c);
template <> PDDP_HD float tl_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> PDDP_HD double tl_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T>
PDDP_HD void tl_control_law(T* u, T alpha, const T* du, const T* Kk, const T* x, const T* xr, const T* uc) {
    T dx[14];
#pragma unroll
    for (int i = 0; i < 14; i++) dx[i] = x[i] - xr[i];
#pragma unroll
    for (int rr = 0; rr < 7; rr++) {
        T acc = alpha * du[rr];
#pragma unroll
        for (int c = 0; c < 14; c++) acc = tl_fma(Kk[rr * 14 + c], dx[c], acc);
        u[rr] = uc[rr] - acc;
    }
}

// One rollout = one (problem, candidate, shooting segment).  The per-step operands (gain K_k, reference state, nominal control, feed-forward)
// are handed in as pointers: global memory on the host, the wave's LDS staging area in k_fp_tl.
//   begin(): start state (the current state for segment 0, else what the linear sweep left in the candidate's slot)
//   step():  control law, running cost, dynamics, Euler step; the last step of a non-final segment produces the boundary defect
//   end():   terminal knot (last segment), partial sums out
// Where the trajectory goes is the Sink's business.  The production sweep keeps every candidate's (state | control | pad) RECORD of every knot, 22 elements, in the
// knot-major array xw (below) and its boundary defects in its slot of ds: TlStagedSink (pddp_tl.hip: float sweep, records through the wave's LDS area) and TlRunSink /
// TlStateSink (the same records with per-lane stores); the setup kernel adopts the accepted candidate's records (arm_tl_adopt_knot).  CandidateSink (teacher-forcing
// hook): x, u, d of every candidate as the reference keeps them, and the records.  xu(k, x, u) hands a knot's state and control over together, right after the control law.
template <typename T>
struct TlRollout {
    T x[14]; T J; T sdef;
    int pb, a_idx, seg, kStart, iters;
    T alpha;
};
// xw: the candidates' records KNOT-major, [problem][knot][candidate][xw_rec] (null: not kept; xw_rec = 22: state 14 | control 7 | pad).  The 8 candidates of a
// (problem, segment) sit in adjacent lanes, so one step of a wave's rollouts fills whole 704-byte runs -- the candidate-major slots of xs would get 56-byte pieces 7 KB
// apart, which cost 1.8 x their bytes in HBM writes (profiles/r02b_b16384: 105 KB written per problem for 57 KB of states).  With the control in the record the setup
// kernel READS the accepted control (84 bytes per knot) instead of recomputing it from the gain, the old state and the feed-forward (~0.7 KB per knot): that kernel is
// bound by its HBM bytes (profiles/r03_stored_controls_experiment.md).  xwk = xw + ((problem * N) * A + candidate) * xw_rec, knot stride A * xw_rec.
template <typename T> PDDP_HD void tl_store7(T* p, const T* v) {
#pragma unroll
    for (int i = 0; i < 3; i++) { TlPair<T> t; t.a = v[2 * i]; t.b = v[2 * i + 1]; *reinterpret_cast<TlPair<T>*>(p + 2 * i) = t; }
    TlPair<T> t; t.a = v[6]; t.b = T(0); *reinterpret_cast<TlPair<T>*>(p + 6) = t;      // (with the record's pad: whole records are written, no holes in the lines)
}
template <typename T> struct TlCandidateSink {         // candidate slot of xs / us / ds (the reference's arrays) and of xw
    T* xs; T* us; T* ds; T* xwk; int xw_stride; int uoff;
    PDDP_HD void x(int k, const T* v) const { tl_store14(xs + (size_t)k * 14, v); if (xwk) tl_store14(xwk + (size_t)k * xw_stride, v); }
    PDDP_HD void u(int k, const T* v) const {
#pragma unroll
        for (int i = 0; i < 7; i++) us[(size_t)k * 7 + i] = v[i];
        if (xwk && uoff) tl_store7(xwk + (size_t)k * xw_stride + uoff, v);
    }
    PDDP_HD void xu(int k, const T*, const T* v) const { u(k, v); }          // (state, control) of a knot together: these sinks store the state when it is produced (x())
    PDDP_HD void xu_last(int k, const T*, const T* v) const { u(k, v); }
    PDDP_HD void d(int k, const T* v) const { tl_store14(ds + (size_t)k * 14, v); }
};
template <typename T> struct TlStateSink {             // states to xw (xs without it), boundary defects to the candidate's slot of ds; controls dropped
    T* xs; T* ds; T* xwk; int xw_stride; int uoff;
    PDDP_HD void x(int k, const T* v) const { if (xwk) tl_store14(xwk + (size_t)k * xw_stride, v); else tl_store14(xs + (size_t)k * 14, v); }
    PDDP_HD void u(int k, const T* v) const { if (xwk && uoff) tl_store7(xwk + (size_t)k * xw_stride + uoff, v); }
    PDDP_HD void xu(int k, const T*, const T* v) const { u(k, v); }
    PDDP_HD void xu_last(int k, const T*, const T* v) const { u(k, v); }
    PDDP_HD void d(int k, const T* v) const { tl_store14(ds + (size_t)k * 14, v); }
};

// The same destination as TlStateSink for a rollout whose x() calls come in knot order (k_fp_tl: begin at kStart, then kStart + 1, ...): a RUNNING pointer instead of
// base + k * stride -- the 64-bit multiply-add per store is what the compiler does not strength-reduce across the inlined dynamics.
template <typename T> struct TlRunSink {
    T* ds; mutable T* cur; int stride; int uoff;
    PDDP_HD void x(int, const T* v) const { tl_store14(cur, v); cur += stride; }
    PDDP_HD void u(int, const T* v) const { if (uoff) tl_store7(cur - stride + uoff, v); }      // (the knot's state went out last: cur is one record ahead)
    PDDP_HD void xu(int k, const T*, const T* v) const { u(k, v); }
    PDDP_HD void xu_last(int k, const T*, const T* v) const { u(k, v); }
    PDDP_HD void d(int k, const T* v) const { tl_store14(ds + (size_t)k * 14, v); }
};
template <typename T>
PDDP_HD TlRunSink<T> tl_run_sink(const Buffers<T>& b, const Dims& dm, int pb, int a_idx, int kStart) {
    const size_t slot = (size_t)pb * dm.A + a_idx;
    if (b.xw) return TlRunSink<T>{b.ds + slot * dm.N * 14, b.xw + (((size_t)pb * dm.N + kStart) * dm.A + a_idx) * b.xw_rec, dm.A * b.xw_rec, b.xw_rec > 14 ? 14 : 0};
    return TlRunSink<T>{b.ds + slot * dm.N * 14, b.xs + (slot * dm.N + kStart) * 14, 14, 0};
}

template <typename T, typename Sink>
PDDP_HD void tl_rollout_begin(TlRollout<T>& r, const Buffers<T>& b, const Dims& dm, int pb, int a_idx, int seg, const T* xcur, const Sink& sink) {
    r.pb = pb; r.a_idx = a_idx; r.seg = seg; r.kStart = seg * dm.NB;
    r.iters = (seg < dm.M - 1) ? dm.NB : dm.NB - 1;
    r.alpha = b.alpha[a_idx]; r.J = T(0); r.sdef = T(0);
    if (seg == 0) tl_load14(r.x, xcur);
    else tl_load14(r.x, b.xs + (((size_t)pb * dm.A + a_idx) * dm.N + r.kStart) * 14);
    sink.x(r.kStart, r.x);                                                // (a candidate slot already holds it for seg > 0; the winner's buffer does not)
}
// step k of the segment (knot kn = kStart + k): Kk[98] = K(r, c) at [c + 14 r], xr[14], uc[7], du[7] of knot kn, xg[14] the goal
template <typename T, typename Sink>
PDDP_HD void tl_rollout_step(TlRollout<T>& r, const ArmTlModel<T>& md, T grav, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, int k,
                             const T* Kk, const T* xr, const T* uc, const T* du, const T* xg, const Sink& sink) {
    constexpr int NX = 14, NU = 7;
    const int kn = r.kStart + k;
    T u[NU];
    tl_control_law<T>(u, r.alpha, du, Kk, r.x, xr, uc);
    sink.xu(kn, r.x, u);
    r.J += arm_tl_cost<T>(cw, r.x, u, xg, false);
    ArmTlState<T> st;
    T qdd[7], xn[NX];
    arm_tl_dynamics<T>(md, grav, st, qdd, r.x, r.x + 7, u);
#pragma unroll
    for (int i = 0; i < 7; i++) { xn[i] = r.x[i] + dt * r.x[7 + i]; xn[7 + i] = r.x[7 + i] + dt * qdd[i]; }     // Euler (utils/integrators.cuh:24-36)
    if (k < dm.NB - 1) {
        sink.x(kn + 1, xn);
#pragma unroll
        for (int i = 0; i < NX; i++) r.x[i] = xn[i];
    } else {                                                          // last step of a non-final segment: defect against the next segment's start
        const int ks = (r.seg + 1) * dm.NB;
        T xnext[NX], e[NX], sdef = T(0);
        tl_load14(xnext, b.xs + (((size_t)r.pb * dm.A + r.a_idx) * dm.N + ks) * NX);
#pragma unroll
        for (int i = 0; i < NX; i++) { e[i] = xn[i] - xnext[i]; sdef += tabs(e[i]); }
        sink.d(ks - 1, e);
        r.sdef = sdef;
    }
}
// ucN: the nominal control of the terminal knot (carried along unchanged)
template <typename T, typename Sink>
PDDP_HD void tl_rollout_end(TlRollout<T>& r, const Dims& dm, const CostWeights<T>& cw, const T* ucN, const T* xg, const Sink& sink) {
    if (r.seg == dm.M - 1) {
        T u[7];
#pragma unroll
        for (int i = 0; i < 7; i++) u[i] = ucN[i];
        sink.xu_last(dm.N - 1, r.x, u);
        r.J += arm_tl_cost<T>(cw, r.x, u, xg, true);
        r.sdef = T(0);
    }
}

// Whole segment with the operands read straight from global memory (host emulation; candidate counts that do not tile a wave).  part != 0: publish the partial sums.
template <typename T, typename Sink>
PDDP_HD void arm_tl_rollout_segment(const ArmTlModel<T>& md, T grav, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, int pb, int a_idx, int seg,
                                    const T* xcur, const Sink& sink, bool part) {
    constexpr int NX = 14, NU = 7;
    const int N = dm.N;
    const T* uc = b.ucur + (size_t)pb * N * NU; const T* du = b.du + (size_t)pb * N * NU; const T* KT = b.KT + (size_t)pb * N * NX * NU;
    T xg[NX];
    tl_load14(xg, b.xGoal + (size_t)pb * NX);
    TlRollout<T> r;
    tl_rollout_begin<T>(r, b, dm, pb, a_idx, seg, xcur, sink);
    for (int k = 0; k < r.iters; k++) {
        const int kn = r.kStart + k;
        T Kk[NX * NU], xr[NX], ucv[NU], duv[NU];
#pragma unroll
        for (int rr = 0; rr < NU; rr++) tl_load14(Kk + rr * NX, KT + (size_t)kn * (NX * NU) + rr * NX);
        tl_load14(xr, xcur + (size_t)kn * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) { ucv[i] = uc[(size_t)kn * NU + i]; duv[i] = du[(size_t)kn * NU + i]; }
        tl_rollout_step<T>(r, md, grav, b, dm, cw, dt, k, Kk, xr, ucv, duv, xg, sink);
    }
    T ucN[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) ucN[i] = uc[(size_t)(N - 1) * NU + i];
    tl_rollout_end<T>(r, dm, cw, ucN, xg, sink);
    if (part) {
        const size_t slot = (size_t)pb * dm.A + a_idx;
        b.Jpart[slot * dm.M + seg] = r.J; b.dpart[slot * dm.M + seg] = r.sdef;
        b.parts_fresh[pb] = 1;                                            // every thread of the problem stores the same value
    }
}
template <typename T>
PDDP_HD TlCandidateSink<T> tl_candidate_sink(const Buffers<T>& b, const Dims& dm, int pb, int a_idx) {
    const size_t slot = (size_t)pb * dm.A + a_idx;
    return TlCandidateSink<T>{b.xs + slot * dm.N * 14, b.us + slot * dm.N * 7, b.ds + slot * dm.N * 14,
                              b.xw ? b.xw + ((size_t)pb * dm.N * dm.A + a_idx) * b.xw_rec : nullptr, dm.A * b.xw_rec, (b.xw && b.xw_rec > 14) ? 14 : 0};
}
template <typename T>
PDDP_HD TlStateSink<T> tl_state_sink(const Buffers<T>& b, const Dims& dm, int pb, int a_idx) {
    const size_t slot = (size_t)pb * dm.A + a_idx;
    return TlStateSink<T>{b.xs + slot * dm.N * 14, b.ds + slot * dm.N * 14, b.xw ? b.xw + ((size_t)pb * dm.N * dm.A + a_idx) * b.xw_rec : nullptr, dm.A * b.xw_rec, (b.xw && b.xw_rec > 14) ? 14 : 0};
}
// After the line search accepted candidate st.alphaIndex (st.cur already points at the NEW half of xb): knot k of the winner becomes the current
// trajectory -- its state from the candidate's record (xw; its slot of xs without xw) into the new half of xb, its control from the record over ucur (without
// records: recomputed from that state with the rollout's own control law and operands -- the gain, the OLD current state / control of this knot, the
// feed-forward; the same tl_control_law, so the same bits the rollout used), the boundary defect from ds into dcur.  Every knot is independent: this is the first step of the setup
// kernel's thread.  Replaces memcpyCurrAKern x3 and the winner -> xp / up / dp copies of nextIterationSetupGPU (nisInitHelpers.cuh:24-32, 270-276).
// Leaves the adopted x[14], u[7] with the caller.
template <typename T>
PDDP_HD void arm_tl_adopt_knot(const Buffers<T>& b, const Dims& dm, int k, int pb, T* x, T* u, bool store_xu = true) {
    constexpr int NX = 14, NU = 7;
    const SolverState<T>& st = b.state[pb];
    const size_t N = dm.N, knot = (size_t)pb * N + k;
    const size_t src = ((size_t)pb * dm.A + st.alphaIndex) * N + k;
    const bool stored_u = b.xw && b.xw_rec > NX;                         // the rollouts left the candidate's control next to its state
    if (b.xw) tl_load14(x, b.xw + (((size_t)pb * N + k) * dm.A + st.alphaIndex) * b.xw_rec); else tl_load14(x, b.xs + src * NX);
    T* uc = b.ucur + knot * NU;
    if (k == dm.N - 1) {                                                  // the terminal knot carries its nominal control along unchanged (tl_rollout_end)
#pragma unroll
        for (int i = 0; i < NU; i++) u[i] = uc[i];
    } else if (stored_u) {
        const T* ur = b.xw + (((size_t)pb * N + k) * dm.A + st.alphaIndex) * b.xw_rec + NX;
#pragma unroll
        for (int i = 0; i < 4; i++) {                                     // four pairs (the last one: u[6] | pad) instead of seven single elements
            const TlPair<T> v = *reinterpret_cast<const TlPair<T>*>(ur + 2 * i);
            u[2 * i] = v.a;
            if (i < 3) u[2 * i + 1] = v.b;
        }
        if (store_xu) {
#pragma unroll
            for (int i = 0; i < NU; i++) uc[i] = u[i];
        }
    } else {
        T ucv[NU];
#pragma unroll
        for (int i = 0; i < NU; i++) ucv[i] = uc[i];
        T xr[NX], duv[NU];
        tl_load14(xr, b.xb + (((size_t)pb * 2 + (1 - st.cur)) * N + k) * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) duv[i] = b.du[knot * NU + i];
        const T* Kg = b.KT + knot * (NX * NU);
        const T alpha = b.alpha[st.alphaIndex];
        T Kk[NX * NU];                                                    // the whole gain first: 49 loads in flight together (row by row costs a memory latency per row)
#pragma unroll
        for (int rr = 0; rr < NU; rr++) tl_load14(Kk + rr * NX, Kg + rr * NX);
        tl_control_law<T>(u, alpha, duv, Kk, x, xr, ucv);
        if (store_xu) {
#pragma unroll
            for (int i = 0; i < NU; i++) uc[i] = u[i];
        }
    }
    if (store_xu) tl_store14(b.xb + (((size_t)pb * 2 + st.cur) * N + k) * NX, x);
    if (dm.M > 1 && dm.on_defect_boundary(k)) {
        T d[NX];
        tl_load14(d, b.ds + src * NX);
        tl_store14(b.dcur + knot * NX, d);
    }
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

// Next-iteration setup of knot k of problem pb (nis_body / arm_lg_nis_body), in two halves:
//   arm_tl_nis_cost: (mode 0: adopt the accepted candidate's knot, arm_tl_adopt_knot), g_k, (mode 1: H_k); returns false when this knot has no Jacobian to
//                    write (rejected / failed iteration, final knot, finished problem)
//   arm_tl_nis_jac:  the Jacobian of the dynamics through emit(col, row, dqdd) -- the caller turns it into [A B] rows 7..13 (k_nis_tl stages it through
//                    LDS in three pieces, flushed at mark(stage); see arm_tl_gradient)
// x[14], u[7]: out -- the current state / control of the knot (mode 0: just adopted from the accepted candidate).
// arm_tl_nis_cost_vals: the same with g_k left in gl[21] for the caller to store (k_nis_tl sends a wave's 64 gradients through LDS: one contiguous run instead of 21
// four-byte stores 84 bytes apart per lane; store_xu = false: likewise the adopted state and control, which the caller then finds in x, u).  Returns 0: nothing moved,
// 4: trajectory adopted only (final accepted step), 1: g computed but no Jacobian wanted (terminal knot), 3: g and Jacobian.
template <typename T>
PDDP_HD int arm_tl_nis_cost_vals(const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, int mode, int k, int pb, T* x, T* u, T* gl, bool store_xu = true) {
    constexpr int NX = 14, NU = 7, NM = 21;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    const size_t knot = (size_t)pb * N + k;
    if (mode == 0) {
        if (!st.win_pending) return 0;                                    // rejected / failed: nothing moved
        arm_tl_adopt_knot<T>(b, dm, k, pb, x, u, store_xu);
        if (st.done) return 4;                                            // final accepted step: the trajectory is adopted, no derivatives needed
    } else {
        tl_load14(x, b.xb + (((size_t)pb * 2 + st.cur) * N + k) * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) u[i] = b.ucur[knot * NU + i];
    }
    const bool fin = (k == N - 1);
    const T w1 = fin ? cw.QF1 : cw.Q1, w2 = fin ? cw.QF2 : cw.Q2, w3 = fin ? T(0) : cw.R;       // ArmPlant::weight
    T xg[NX];
    tl_load14(xg, b.xGoal + (size_t)pb * NX);
#pragma unroll
    for (int i = 0; i < 7; i++) { gl[i] = w1 * (x[i] - xg[i]); gl[7 + i] = w2 * (x[7 + i] - xg[7 + i]); gl[14 + i] = w3 * u[i]; }
    if (cw.limits) {                                                      // USE_LIMITS_FLAG: the gradient only (cost_arm.cuh:176-199)
#pragma unroll
        for (int i = 0; i < NM; i++) if (i < (fin ? NX : NM)) gl[i] += arm_limit_term<T>(x, u, i, 1);
    }
    if (mode == 1) {                                                      // H_k = diag(weight): constant over the solve, written once
        T* H = b.H + knot * (NM * NM);
        for (int e = 0; e < NM * NM; e++) { const int i = e / NM, j = e % NM; H[e] = i != j ? T(0) : (i < 7 ? w1 : (i < NX ? w2 : w3)); }
    }
    return fin ? 1 : 3;
}
template <typename T>
PDDP_HD bool arm_tl_nis_cost(const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, int mode, int k, int pb, T* x, T* u) {
    T gl[21];
    const int r = arm_tl_nis_cost_vals<T>(b, dm, cw, mode, k, pb, x, u, gl);
    if (r & 1) {
        T* g = b.g + ((size_t)pb * dm.N + k) * 21;
#pragma unroll
        for (int i = 0; i < 21; i++) g[i] = gl[i];
    }
    return r == 3;
}
// the Jacobian of the dynamics at (x, u) through emit(col, row, dqdd)
template <typename T, typename Emit, typename Mark>
PDDP_HD void arm_tl_nis_jac(const ArmTlModel<T>& md, T grav, const T* x, const T* u, Emit emit, Mark mark) {
    ArmTlState<T> ts;
    T qdd[7];
    arm_tl_dynamics<T>(md, grav, ts, qdd, x, x + 7, u);
    arm_tl_gradient<T>(md, grav, ts, x + 7, qdd, emit, mark);
}
template <typename T, typename Emit>
PDDP_HD bool arm_tl_nis_knot(const ArmTlModel<T>& md, T grav, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, int mode, int k, int pb, Emit emit) {
    T x[14], u[7];
    if (!arm_tl_nis_cost<T>(b, dm, cw, mode, k, pb, x, u)) return false;
    arm_tl_nis_jac<T>(md, grav, x, u, emit, [](int) {});
    return true;
}

// ------------------------------------------------------------------------------------------------ end-effector cost family on thread lanes
// The same quantities, operations per element and summation orders as ee_cost.hpp / ee_cost_lg.hpp (which restate compute_eePos, plants/dynamics_arm.cuh:1879-1925, the
// end-effector costFunc / costGrad, plants/cost_arm.cuh:206-389, the in-sim accumulation of forwardSimInner, fpHelpers.cuh:259-265,298-300, and the EE branch of
// costGradientHessianKern, nisInitHelpers.cuh:52-84) with the tool point from the thread-lane world chain (plant_arm_tl.hpp arm_tl_world_chain).
template <typename T>
PDDP_HD bool tl_ee_rpy_weighted(const CostWeights<T>& cw) { return cw.Q_EE2 != T(0) || cw.QF_EE2 != T(0); }

// one rollout step with the end-effector cost: control law, the tool point of the CURRENT state, the seven per-joint running sums (costFunc with s_cost: joint 0 also takes
// the end-effector term), dynamics, Euler step.  Every segment runs NB steps; the last one's final step (knot N - 1) only evaluates the cost (forwardSimInner :236, :259-265).
template <typename T, typename Sink>
PDDP_HD void tl_rollout_step_ee(TlRollout<T>& r, T* acc, const ArmTlModel<T>& md, T grav, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, int k,
                                const T* Kk, const T* xr, const T* uc, const T* du, const T* goal, const T* xt, int tshift, const Sink& sink) {
    constexpr int NX = 14, NU = 7;
    const int kn = r.kStart + k, N = dm.N;
    T u[NU];
    tl_control_law<T>(u, r.alpha, du, Kk, r.x, xr, uc);
    sink.xu(kn, r.x, u);
    ArmTlState<T> st;
    arm_tl_trig<T>(st, r.x);
    if (k < dm.NB - 1 || r.seg == dm.M - 1) {                         // not on the "final" state of a non-final segment (:259-265)
        ArmTlFrames<T> fr;
        arm_tl_world_chain<false, T>(md, st.c, st.s, fr);
        T pos[6];
        arm_tl_tool_point<T>(fr, cw.ee_z, tl_ee_rpy_weighted<T>(cw), pos);
#pragma unroll
        for (int ind = 0; ind < NU; ind++) {
            T cost = T(0);
            if (ind == 0) cost += ee_term<T>(cw, pos, goal, kn >= N - 1 - tshift);
            acc[ind] += ee_joint_terms<T>(cw, r.x, u, xt, ind, kn, N, cost);
        }
    }
    if (kn == N - 1) return;                                          // the step out of the last knot is not stored anywhere
    T bias[NU], qdd[NU], xn[NX];
    arm_tl_bias<T>(md, grav, st, r.x + 7, bias);
    arm_tl_factor<T>(md, st);
#pragma unroll
    for (int i = 0; i < NU; i++) qdd[i] = u[i] - bias[i];
    tl_ldl_solve(st, qdd);
#pragma unroll
    for (int i = 0; i < 7; i++) { xn[i] = r.x[i] + dt * r.x[7 + i]; xn[7 + i] = r.x[7 + i] + dt * qdd[i]; }
    if (k < dm.NB - 1) {
        sink.x(kn + 1, xn);
#pragma unroll
        for (int i = 0; i < NX; i++) r.x[i] = xn[i];
    } else {
        const int ks = (r.seg + 1) * dm.NB;
        T xnext[NX], e[NX], sdef = T(0);
        tl_load14(xnext, b.xs + (((size_t)r.pb * dm.A + r.a_idx) * dm.N + ks) * NX);
#pragma unroll
        for (int i = 0; i < NX; i++) { e[i] = xn[i] - xnext[i]; sdef += tabs(e[i]); }
        sink.d(ks - 1, e);
        r.sdef = sdef;
    }
}
// the segment's cost: the seven partial sums added 0..6 (forwardSimKern :298-300)
template <typename T>
PDDP_HD void tl_rollout_end_ee(TlRollout<T>& r, const T* acc, const Dims& dm) {
    r.J = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6];
    if (r.seg == dm.M - 1) r.sdef = T(0);
}
template <typename T, typename Sink>
PDDP_HD void arm_tl_rollout_segment_ee(const ArmTlModel<T>& md, T grav, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, int pb, int a_idx, int seg,
                                       const T* xcur, const Sink& sink, bool part) {
    constexpr int NX = 14, NU = 7;
    const int N = dm.N;
    const T* uc = b.ucur + (size_t)pb * N * NU; const T* du = b.du + (size_t)pb * N * NU; const T* KT = b.KT + (size_t)pb * N * NX * NU;
    T goal[6], xt[NX], acc[NU];
#pragma unroll
    for (int i = 0; i < 6; i++) goal[i] = b.xGoal[(size_t)pb * NX + i];
    tl_load14(xt, b.xTarget + (size_t)pb * NX);
#pragma unroll
    for (int i = 0; i < NU; i++) acc[i] = T(0);
    const int tshift = b.tshift[pb];
    TlRollout<T> r;
    tl_rollout_begin<T>(r, b, dm, pb, a_idx, seg, xcur, sink);
    r.iters = dm.NB;
    for (int k = 0; k < r.iters; k++) {
        const int kn = r.kStart + k;
        T Kk[NX * NU], xr[NX], ucv[NU], duv[NU];
#pragma unroll
        for (int rr = 0; rr < NU; rr++) tl_load14(Kk + rr * NX, KT + (size_t)kn * (NX * NU) + rr * NX);
        tl_load14(xr, xcur + (size_t)kn * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) { ucv[i] = uc[(size_t)kn * NU + i]; duv[i] = du[(size_t)kn * NU + i]; }
        tl_rollout_step_ee<T>(r, acc, md, grav, b, dm, cw, dt, k, Kk, xr, ucv, duv, goal, xt, tshift, sink);
    }
    tl_rollout_end_ee<T>(r, acc, dm);
    if (part) {
        const size_t slot = (size_t)pb * dm.A + a_idx;
        b.Jpart[slot * dm.M + seg] = r.J; b.dpart[slot * dm.M + seg] = r.sdef;
        b.parts_fresh[pb] = 1;
    }
}

// Next-iteration setup of knot k with the end-effector cost: (mode 0: adopt the accepted candidate's knot), tool point + Jacobian, g_k, the Gauss-Newton Hessian -- its only
// dense part, the 7 x 7 position block Jee' Jee (+ Qx on its diagonal), into the compact array b.Hc when the handle keeps one (the matrix-core backward pass rebuilds the rest
// from the cost weights); the reference-layout H_k is written for the final knot (the backward pass starts from it) and in init mode (the API view) -- and for every knot when
// there is no compact array; init mode also leaves the knot's cost in costk (costGrad's d_JT, initAlgGPU).  Returns false when the knot has no Jacobian of the dynamics to write.
// arm_tl_nis_cost_ee_knot: the cost part at a given (x, u).  h_block_only (few problems in flight, no compact array): outside init mode only the position block of H_k is
// rewritten -- the rest of the block is the constant diagonal init mode wrote.
template <typename T>
PDDP_HD bool arm_tl_nis_cost_ee_knot(const ArmTlModel<T>& md, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, int mode, int k, int pb, const T* x, const T* u,
                                     bool h_block_only = false);
template <typename T>
PDDP_HD bool arm_tl_nis_cost_ee(const ArmTlModel<T>& md, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, int mode, int k, int pb, T* x, T* u) {
    constexpr int NX = 14, NU = 7;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    const size_t knot = (size_t)pb * N + k;
    if (mode == 0) {
        if (!st.win_pending) return false;
        arm_tl_adopt_knot<T>(b, dm, k, pb, x, u);
        if (st.done) return false;
    } else {
        tl_load14(x, b.xb + (((size_t)pb * 2 + st.cur) * N + k) * NX);
#pragma unroll
        for (int i = 0; i < NU; i++) u[i] = b.ucur[knot * NU + i];
    }
    return arm_tl_nis_cost_ee_knot<T>(md, b, dm, cw, mode, k, pb, x, u);
}
template <typename T>
PDDP_HD bool arm_tl_nis_cost_ee_knot(const ArmTlModel<T>& md, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, int mode, int k, int pb, const T* x, const T* u,
                                     bool h_block_only) {
    constexpr int NX = 14, NM = 21, NP = 7;
    const int N = dm.N;
    const size_t knot = (size_t)pb * N + k;
    const bool fin = (k == N - 1), fin_ee = k >= N - 1 - b.tshift[pb];
    ArmTlState<T> ts;
    arm_tl_trig<T>(ts, x);
    ArmTlFrames<T> fr;
    arm_tl_world_chain<true, T>(md, ts.c, ts.s, fr);
    T pos[6], dpos[42], goal[6], xt[NX];
    arm_tl_tool_point<T>(fr, cw.ee_z, tl_ee_rpy_weighted<T>(cw), pos);
    arm_tl_tool_jacobian<T>(fr, cw.ee_z, dpos);
#pragma unroll
    for (int i = 0; i < 6; i++) goal[i] = b.xGoal[(size_t)pb * NX + i];
    tl_load14(xt, b.xTarget + (size_t)pb * NX);
    const T Qx = fin ? cw.QF_xEE : cw.Q_xEE, Qxd = fin ? cw.QF_xdEE : cw.Q_xdEE, Ru = fin ? T(0) : cw.R_EE;
    T* g = b.g + knot * NM;
#pragma unroll
    for (int r = 0; r < NP; r++) {                                    // costGrad :330-345
        T dv = T(0);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const T dl = pos[i] - goal[i];
            dv += (fin_ee ? (i < 3 ? cw.QF_EE1 : cw.QF_EE2) : (i < 3 ? cw.Q_EE1 : cw.Q_EE2)) * dl * dpos[r * 6 + i];
        }
        if (cw.smooth_abs) dv /= ee_smooth_abs_divisor<T>(cw, pos, goal, fin_ee);                       // USE_SMOOTH_ABS (cost_arm.cuh:242-251)
        g[r] = dv + Qx * (x[r] - xt[r]);
        g[NP + r] = Qxd * (x[NP + r] - xt[NP + r]);
        g[NX + r] = Ru * u[r];
        if (cw.limits) { g[r] += arm_limit_term<T>(x, u, r, 1); g[NP + r] += arm_limit_term<T>(x, u, NP + r, 1); g[NX + r] += arm_limit_term<T>(x, u, NX + r, 1); }   // USE_LIMITS_FLAG (:341-343)
    }
    T Hqq[NP * NP];                                                   // costGrad :347-379 (unweighted Gauss-Newton block)
#pragma unroll
    for (int c = 0; c < NP; c++)
#pragma unroll
        for (int r = 0; r < NP; r++) {
            T val = T(0);
#pragma unroll
            for (int j = 0; j < 6; j++) val += dpos[r * 6 + j] * dpos[c * 6 + j];
            Hqq[c * NP + r] = (r == c) ? val + Qx : val;
            if (cw.limits && r == c) Hqq[c * NP + r] += arm_limit_term<T>(x, u, r, 2);                   // USE_LIMITS_FLAG on the diagonal of H (:374-376)
        }
    if (b.Hc) {
        T* hc = b.Hc + knot * (NP * NP);
#pragma unroll
        for (int e = 0; e < NP * NP; e++) hc[e] = Hqq[e];
    }
    if (!b.Hc || fin || mode == 1) {
        T* H = b.H + knot * (NM * NM);
        if (!(h_block_only && mode != 1) || cw.limits) for (int e = 0; e < NM * NM; e++) {
            const int c = e / NM, r = e % NM;
            H[e] = (r < NP && c < NP) ? T(0) : (r != c ? T(0) : (r < NX ? Qxd : Ru));
            if (cw.limits && r == c && r >= NP) H[e] += arm_limit_term<T>(x, u, r, 2);
        }
#pragma unroll
        for (int c = 0; c < NP; c++)
#pragma unroll
            for (int r = 0; r < NP; r++) H[c * NM + r] = Hqq[c * NP + r];
    }
    if (mode == 1) {                                                  // the knot's cost as one running sum over the joints (costFunc returning a value, :298-315)
        T cost = T(0);
#pragma unroll
        for (int ind = 0; ind < NP; ind++) {
            if (ind == 0) cost += ee_term<T>(cw, pos, goal, fin_ee);
            cost = ee_joint_terms<T>(cw, x, u, xt, ind, k, N, cost);
        }
        b.costk[knot] = cost;
    }
    return !fin;
}

// [A B] entry (row, col) of the Euler step from the Jacobian of the dynamics: I + dt [0 I 0; dqdd]   (utils/integrators.cuh:38-53)
template <typename T> PDDP_HD T tl_AB_const(int row, int col, T dt) { return T(col == row ? 1 : 0) + (col == row + 7 ? dt : T(0)); }   // rows 0..6

}  // namespace pddp

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
