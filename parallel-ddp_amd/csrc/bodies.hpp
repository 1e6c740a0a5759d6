// Per-workgroup bodies of the sweep kernels, free of any HIP runtime dependency so that tests/hostsim can run the very
// same code on the host with a 1-lane wave.  kernels.hpp wraps them into __global__ entry points.
#pragma once

#include "bp.hpp"
#include "fp.hpp"
#include "fp_tl.hpp"
#include "nis.hpp"
#include "solver_state.hpp"

namespace pddp {

template <typename P, typename T>
PDDP_HD void bp_body(const Wave& w, BpScratch<P, T>& s, const Buffers<T>& b, const Dims& dm, int blk, int pb) {
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    if (st.done) return;
    BpArgs<T> a;
    a.AB = b.AB + (size_t)pb * N * NX * NM;
    a.Pm = (st.pw ? b.Pp : b.P) + (size_t)pb * N * NX * NX;   a.pv = (st.pw ? b.pp : b.p) + (size_t)pb * N * NX;     // written
    a.Pp = (st.pw ? b.P : b.Pp) + (size_t)pb * N * NX * NX;   a.pp = (st.pw ? b.p : b.pp) + (size_t)pb * N * NX;     // previous iteration's
    a.H = b.H + (size_t)pb * N * NM * NM;    a.g = b.g + (size_t)pb * N * NM;
    a.KT = b.KT + (size_t)pb * N * NX * NU;  a.du = b.du + (size_t)pb * N * NU;
    a.dcur = b.dcur + (size_t)pb * N * NX;
    a.ApBK = b.ApBK + (size_t)pb * N * NX * NX;  a.Bdu = b.Bdu + (size_t)pb * N * NX;
    a.xcur = b.xb + ((size_t)pb * 2 + st.cur) * N * NX;
    a.xprev2 = b.xb + ((size_t)pb * 2 + st.cur2) * N * NX;
    a.dJexp = b.dJexp + (size_t)pb * 2 * dm.M;
    a.err = b.err + (size_t)pb * dm.M;
    a.rho = st.rho;
    if (bp_block<P, T>(w, s, dm, blk, a)) { if (w.lane == 0) a.err[blk] = 1; }
}

// true when the forward pass of this sweep has to run for problem pb
template <typename T>
PDDP_HD bool fp_active(const Buffers<T>& b, const Dims& dm, int pb) {
    if (b.state[pb].done) return false;
    const int* err = b.err + (size_t)pb * dm.M;
    for (int i = 0; i < dm.M; i++) if (err[i]) return false;   // backward pass failed: this sweep only raises rho
    return true;
}
template <typename P, typename T>
PDDP_HD FpArgs<T> fp_args(const Buffers<T>& b, const Dims& dm, int pb, int a_idx, T dt, T* segx, T* dnorm, T* segJ = nullptr) {
    constexpr int NX = P::NX, NU = P::NU;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    FpArgs<T> a;
    const size_t slot = (size_t)pb * dm.A + a_idx;
    a.x = b.xs + slot * N * NX; a.u = b.us + slot * N * NU; a.d = b.ds + slot * N * NX;
    a.xcur = b.xb + ((size_t)pb * 2 + st.cur) * N * NX;
    a.ucur = b.ucur + (size_t)pb * N * NU; a.dcur = b.dcur + (size_t)pb * N * NX;
    a.KT = b.KT + (size_t)pb * N * NX * NU; a.du = b.du + (size_t)pb * N * NU;
    a.ApBK = b.ApBK + (size_t)pb * N * NX * NX; a.Bdu = b.Bdu + (size_t)pb * N * NX;
    a.alpha = b.alpha[a_idx]; a.dt = dt; a.segx = segx; a.dnorm = dnorm;
    a.xt = b.xTarget + (size_t)pb * NX; a.segJ = segJ; a.tshift = b.tshift[pb];
    return a;
}
// cost tree-sum and defect max of one candidate, by one wave
template <typename T>
PDDP_HD void fp_reduce(const Wave& w, const Buffers<T>& b, const Dims& dm, int pb, int a_idx, T* cost_k, const T* dnorm, const T* segJ = nullptr) {
    T J;
    if (segJ) { J = 0; for (int i = 0; i < dm.M; i++) J += segJ[i]; }     // end-effector cost: costKern<T,0>, fpHelpers.cuh:169-178
    else J = tree_sum<T>(w, cost_k, dm.N);
    if (w.lane == 0) {
        T mx = 0;
        for (int i = 0; i < dm.M; i++) mx = tmax(mx, dnorm[i]);
        const size_t slot = (size_t)pb * dm.A + a_idx;
        b.J[slot] = J; b.dmax[slot] = mx;
    }
}

// Jsrc / dsrc: where the candidates' total cost / defect norm are read from -- the arrays b.J / b.dmax of the problem (null), or a copy the caller holds closer
// (the rollout kernel that ends with the line search, k_fp_tl4: its wave's LDS)
template <typename T>
PDDP_HD void ls_body(const Buffers<T>& b, const Dims& dm, const SolverParams& sp, int pb, int freeze_exit, const T* Jsrc = nullptr, const T* dsrc = nullptr) {
    SolverState<T> st = b.state[pb];
    if (st.done) { if (st.win_pending) b.state[pb].win_pending = 0; return; }      // the final accepted step was adopted by the previous sweep's winner kernel
    const int* err = b.err + (size_t)pb * dm.M;
    int any = 0;
    for (int i = 0; i < dm.M; i++) any |= err[i];
    if (!Jsrc && b.parts_fresh && b.parts_fresh[pb]) { tl_reduce_parts<T>(b, dm, pb); b.parts_fresh[pb] = 0; }   // thread-lane forward pass: add the per-segment partial sums
    const T* Jp = Jsrc ? Jsrc : b.J + (size_t)pb * dm.A; const T* dp = dsrc ? dsrc : b.dmax + (size_t)pb * dm.A;
    const size_t ho = (size_t)pb * sp.out_stride;
    if (freeze_exit) {   // benchmark mode: never exit, keep writing the same Jout slot
        SolverParams sp2 = sp; sp2.tol_cost = -1e300; sp2.ignore_max_rho_exit = 1;
        const int it = st.iter;
        line_search_accept<T>(st, sp2, dm, any, b.alpha, Jp, dp, b.dJexp + (size_t)pb * 2 * dm.M, b.Jout + ho, b.alphaOut + ho);
        st.done = 0; st.iter = it;
    } else {
        line_search_accept<T>(st, sp, dm, any, b.alpha, Jp, dp, b.dJexp + (size_t)pb * 2 * dm.M, b.Jout + ho, b.alphaOut + ho);
    }
    b.state[pb] = st;
}

// knot k of problem pb.
//   accepted: winner candidate -> current trajectory (x into the other half of xb, u, d), then AB_k, H_k, g_k there;
//   rejected: trajectory and derivatives are unchanged (the reference recomputes identical values);
//   (the reference's P -> Pp, p -> pp copies are a flip of state.pw in the line-search kernel)
// mode 1 = initAlgGPU derivatives (no copies).
template <typename P, int INTEG, typename T>
PDDP_HD void nis_body(const Wave& w, NisScratch<P, INTEG, T>& s, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt,
                      int mode, int k, int pb) {
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    T* xc = b.xb + ((size_t)pb * 2 + st.cur) * N * NX + (size_t)k * NX;
    T* uc = b.ucur + ((size_t)pb * N + k) * NU;
    if (mode == 0) {
        if (st.accepted < 0) return;                           // backward pass failed: nothing moved
        if (st.accepted != 1) return;
        const size_t slot = (size_t)pb * dm.A + st.alphaIndex;
        const T* xw = b.xs + (slot * N + k) * NX; const T* uw = b.us + (slot * N + k) * NU;
        if constexpr (P::PLANT != 4) {                              // closed-form plants whose production rollouts keep records state | control knot-major (k_fp_cf, kernels.hpp)
            if (b.xw) { xw = b.xw + (((size_t)pb * N + k) * dm.A + st.alphaIndex) * (NX + NU); uw = xw + NX; }
        }
        PDDP_FOR(i, NX) xc[i] = xw[i];
        PDDP_FOR(i, NU) uc[i] = uw[i];
        if (dm.M > 1 && dm.on_defect_boundary(k)) {
            const T* dw = b.ds + (slot * N + k) * NX; T* dc = b.dcur + ((size_t)pb * N + k) * NX;
            PDDP_FOR(i, NX) dc[i] = dw[i];
        }
        if (st.done) return;                                   // final accepted step: solution copied, no derivatives needed
        wsync();
    }
    P::load_model(w, s.plant, reinterpret_cast<const typename P::Model*>(b.model));
    nis_knot<P, INTEG, T>(w, s, dm, k, xc, uc, b.xGoal + (size_t)pb * NX, cw, dt,
                          b.AB + ((size_t)pb * N + k) * NX * NM, b.H + ((size_t)pb * N + k) * NM * NM, b.g + ((size_t)pb * N + k) * NM,
                          b.xTarget + (size_t)pb * NX, b.tshift[pb], mode == 1 ? b.costk + (size_t)pb * N + k : nullptr, mode == 1);
}

// cost of the loaded trajectory, prevJ = J + 2 TOL_COST, Jout[0], alphaOut[0], fresh solver state
// (initAlgGPU, nisInitHelpers.cuh:363,385-395, and the locals of runiLQR_GPU, DDPWrappers.cuh:24).
// End-effector cost (stage: 0 joint-space cost, everything in one call; 1 fresh state only, the cost follows; 2 the cost: per-knot
// values the setup kernel left in b.costk (costGrad's d_JT[k], nisInitHelpers.cuh:368) tree-summed (costKern<T,1>, fpHelpers.cuh:179-190),
// or, after an initial rollout, the rollout's own sum (costKern<T,0><<<1,1>>>, nisInitHelpers.cuh:387).  prev_alpha: the winner index
// the previous solve on these buffers ended with.  The reference reads the initial cost from d_JT[*alphaIndex] (:392) although
// costKern<T,1> leaves the sum in d_JT[0] and the cost of KNOT a in d_JT[a]: an MPC solve (which keeps *alphaIndex, MPCHelpers.cuh)
// that follows a solve ending with alphaIndex = a > 0 therefore starts from prevJ = cost of knot a.  Kept, sic.
template <typename P, typename T>
PDDP_HD void init_cost_body(const Wave& w, T* cost_k, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw,
                            const SolverParams& sp, int ignore_first_defect, int rollout, int pb, int stage = 0, int keep_alpha = 0) {
    constexpr int NX = P::NX, NU = P::NU;
    const int N = dm.N;
    const T* x = b.xb + ((size_t)pb * 2 + 0) * N * NX; const T* u = b.ucur + (size_t)pb * N * NU;
    const T* xg = b.xGoal + (size_t)pb * NX;
    T J = 0;
    if (stage == 0) {
        PDDP_FOR(k, N) cost_k[k] = P::cost(cw, x + (size_t)k * NX, u + (size_t)k * NU, xg, k, N);
        wsync();
        J = tree_sum<T>(w, cost_k, N);
    } else if (stage == 2) {
        const size_t ho = (size_t)pb * sp.out_stride;
        if (rollout) J = b.J[(size_t)pb * dm.A];
        else {
            const int a0 = sp.ee_initial_cost_fix ? 0 : b.state[pb].alphaIndex;
            PDDP_FOR(k, N) cost_k[k] = b.costk[(size_t)pb * N + k];
            wsync();
            const T Jk = (a0 > 0 && a0 < N) ? cost_k[a0] : T(0);
            wsync();
            J = tree_sum<T>(w, cost_k, N);
            if (a0 > 0 && a0 < N) J = Jk;
        }
        if (w.lane == 0) {
            b.state[pb].prevJ = J + T(2 * sp.tol_cost);
            b.Jout[ho] = b.state[pb].prevJ - T(2 * sp.tol_cost);
        }
        return;
    }
    if (w.lane == 0) {
        SolverState<T> st;
        st.rho = T(sp.rho_init); st.drho = T(1.0); st.dJ = 0; st.z = 0;
        st.prevJ = J + T(2 * sp.tol_cost);
        st.iter = 1; st.alphaIndex = keep_alpha ? b.state[pb].alphaIndex : 0; st.ignore_defect = ignore_first_defect; st.accepted = 1; st.done = 0;
        st.cur = 0; st.cur2 = 0; st.bp_retries = 0; st.took_step = 0; st.win_pending = 0;
        st.pw = b.state[pb].pw;        // a warm start must read the cost-to-go of the iteration before the previous exit: keep the buffer roles
        b.state[pb] = st;
        const size_t ho = (size_t)pb * sp.out_stride;
        b.Jout[ho] = st.prevJ - T(2 * sp.tol_cost);
        b.alphaOut[ho] = rollout ? 0 : -1;
    }
}

}  // namespace pddp
