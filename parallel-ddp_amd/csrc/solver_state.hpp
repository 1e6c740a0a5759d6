// Device-resident solver state and memory map.
//
// The reference keeps rho, drho, prevJ, iter, alphaIndex, ignore_defect on the HOST and crosses PCIe three times
// per iteration to take the line-search / accept-reject decision (fpHelpers.cuh:374-408, nisInitHelpers.cuh:489-518).
// Here they live in HBM next to the data, one record per problem, and a one-wave kernel takes the decision, so a DDP
// iteration is four back-to-back launches with no host synchronisation.
//
// HBM layout (B = independent problems in flight; every array is [B][...] contiguous, knot-major inside, each knot
// block column-major with leading dimension = rows, as in the reference, nisInitHelpers.cuh:776,797-798):
//   xs[B][A][N][n] us[B][A][N][m] ds[B][A][N][n]   candidate trajectories (pure outputs of the forward pass)
//   xb[B][2][N][n]                                  current trajectory and the one the boundary cost-to-go was
//                                                   computed at (reference d_xp / d_xp2), selected by state.cur/cur2
//   ucur[B][N][m] dcur[B][N][n]                     current controls / defects (reference d_up / d_dp)
//   P[B][N][n*n] p[B][N][n] Pp, pp                  cost-to-go (slot j = knot j+1), double-buffered: the backward pass writes the half
//                                                   state.pw and reads the block-boundary slots of the other one (the previous
//                                                   iteration's, the reference's d_Pp / d_pp); "Pp <- P" is a flip of state.pw
//   AB[B][N][n*(n+m)] H[B][N][(n+m)^2] g[B][N][n+m] derivatives;  KT[B][N][n*m] du[B][N][m]  gains
//   ApBK[B][N][n*n] Bdu[B][N][n]                    sweep operands (M > 1)
//   J[B][A] dmax[B][A] dJexp[B][2M] err[B][M]       line-search inputs
//   Jout[B][max_iter+2] alphaOut[B][max_iter+2]     the reference's observables
#pragma once

#include "pddp_common.hpp"

namespace pddp {

template <typename T>
struct SolverState {
    T rho, drho, prevJ, dJ, z;
    int iter;            // reference `iter`: index of the NEXT Jout/alphaOut slot, starts at 1 (DDPWrappers.cuh:24)
    int alphaIndex;      // slot of the last accepted candidate (reference *alphaIndex)
    int ignore_defect;   // ignoreFirstDefectFlag, cleared on the first accepted small defect (fpHelpers.cuh:404)
    int accepted;        // outcome of the last line search: 1 accepted, 0 rejected, -1 backward pass failed
    int done;            // 0 running, 1 cost tolerance, 2 max_iter, 3 max rho
    int cur, cur2;       // which half of xb holds the current trajectory / the trajectory of the stored Pp,pp
    int bp_retries;
    int pw;              // which half of the cost-to-go double buffer (P = 0, Pp = 1) the next backward pass WRITES; it reads the other one
    int took_step;       // MPC: an accepted iteration of this solve used a step-size index > 0 (MPCHelpers.cuh:986-991)
    int win_pending;     // this sweep's line search accepted a candidate: the thread-lane winner kernel / setup kernel act on it (cleared by the next line search)
};

struct SolverParams {    // read-only per launch (reference macros, config.cuh)
    int max_iter;               // MAX_ITER                :83  (an MPC call lowers it per solve)
    int out_stride;             // row stride of Jout / alphaOut = config.max_iter + 2: fixed at allocation, NOT the per-call iteration limit
    int ignore_max_rho_exit;    // IGNORE_MAX_ROX_EXIT     :105-107
    double tol_cost;            // TOL_COST                :85-87
    double exp_red_min, exp_red_max;   //                  :117-122
    double max_defect;          // MAX_DEFECT_SIZE         :124-126
    double rho_init;            // RHO_INIT                :99-101
    int ee_initial_cost_fix;    // pddp_config.ee_initial_cost_fix (not a reference macro)
};
constexpr double kRhoMax = 10000000.0, kRhoMin = 0.01, kRhoFactor = 1.25;   // config.cuh:102-104

template <typename T>
struct Buffers {
    T *xs, *us, *ds, *xb, *ucur, *dcur;
    T *P, *p, *Pp, *pp, *AB, *H, *g, *KT, *du, *ApBK, *Bdu;
    T *J, *dmax, *dJexp, *alpha, *xGoal, *Jout;
    T *xTarget;          // [B][n] nominal-state target of the end-effector cost (zeros unless set)
    T *costk;            // [B][N] per-knot cost of the loaded trajectory (end-effector cost: written by the setup kernel, d_JT[k] of initAlgGPU)
    int *tshift;         // [B] finalCostShift of the end-effector cost (0 unless the MPC call shifts it)
    T *xw;               // [B][N][A][xw_rec] candidate states knot-major (thread-lane rollouts -> setup kernel, fp_tl.hpp); null otherwise
    int xw_rec;          // elements per (knot, candidate) record of xw: 14 = the state; 22 = state | control | pad (the setup kernel then reads the accepted control instead of recomputing it)
    T *segmap;           // [B][M][16*16] per-segment affine maps of the forward sweep composed by the matrix-core backward pass (bp_mfma.hpp kMxFuseSweep); null otherwise
    T *Hc;               // [B][N][49] the dense 7 x 7 position block Jee' Jee of the end-effector cost's Gauss-Newton Hessian (the rest of H_k is the diagonal of the cost weights):
                         // thread-lane setup -> matrix-core backward pass of end-effector handles (fp_tl.hpp arm_tl_nis_cost_ee, bp_mfma.hpp HQQ); null otherwise
    T *ABc;              // compact [A B] of the arm's Euler step (ab_compact.hpp) when the handle runs the thread-lane setup + matrix-core backward pass; null otherwise
    T *Jpart, *dpart;    // [B][A][M] per-segment partial cost / defect norm of the thread-lane forward pass (fp_tl.hpp); null otherwise
    int *parts_fresh;    // [B] set by that forward pass, consumed by the line-search kernel (which then adds the partial sums into J / dmax)
    int *err, *alphaOut;
    SolverState<T>* state;
    const void* model;   // plant constants (P::Model) in device memory
};

// rho schedule: bpHelpers.cuh:500-501, nisInitHelpers.cuh:494 (up) and :508 (down)
template <typename T> PDDP_HD void rho_increase(SolverState<T>& s) {
    s.drho = tmax(s.drho * T(kRhoFactor), T(kRhoFactor));
    s.rho = tmin(s.rho * s.drho, T(kRhoMax));
}
template <typename T> PDDP_HD void rho_decrease(SolverState<T>& s) {
    s.drho = tmin(s.drho / T(kRhoFactor), T(1.0 / kRhoFactor));
    s.rho = tmax(s.rho * s.drho, T(kRhoMin));
}

// Line search over the A candidates + accept/reject + bookkeeping, executed by one lane.
// forwardSimGPU host part (fpHelpers.cuh:373-408) followed by acceptRejectTrajGPU (nisInitHelpers.cuh:489-518).
// `any_bp_err`: the backward pass of this sweep reported a failed inversion (backwardPassGPU's retry, bpHelpers.cuh:497-511);
// then only rho is raised and the pass is repeated by the next sweep.
template <typename T>
PDDP_HD void line_search_accept(SolverState<T>& st, const SolverParams& sp, const Dims& dm, int any_bp_err, const T* alpha,
                                const T* J, const T* dmax, T* dJexp, T* Jout, int* alphaOut) {
    if (st.done) return;
    st.win_pending = 0;
    if (any_bp_err) {
        rho_increase(st);
        st.accepted = -1;
        st.bp_retries++;
        if ((st.rho == T(kRhoMax) && !sp.ignore_max_rho_exit) || st.bp_retries > 200) st.done = 3;
        return;
    }
    for (int i = 1; i < dm.M; i++) { dJexp[0] += dJexp[2 * i]; dJexp[1] += dJexp[2 * i + 1]; }
    T dJ = -1, z = 0;
    int aidx = st.alphaIndex;
    for (int i = 0; i < dm.A; i++) {
        const T cdJ = st.prevJ - J[i];
        const bool JFlag = cdJ >= T(0) && cdJ > dJ;
        const T cz = cdJ / (alpha[i] * dJexp[0] + T(0.5) * alpha[i] * alpha[i] * dJexp[1]);
        const bool zFlag = T(sp.exp_red_min) < cz && cz < T(sp.exp_red_max);
        const bool dFlag = (dm.M == 1 || st.ignore_defect) ? true : dmax[i] < T(sp.max_defect);
        if (JFlag && zFlag && dFlag) {
            if (dmax[i] < T(sp.max_defect)) st.ignore_defect = 0;
            aidx = i; dJ = cdJ; z = cz;
        }
    }
    st.z = z;
    st.cur2 = st.cur;                                   // xp2 <- xp (fpHelpers.cuh:371), by index instead of by copy
    if (dJ < T(0)) {                                    // reject: keep the trajectory, raise rho
        rho_increase(st);
        st.alphaIndex = 0; alphaOut[st.iter] = -1; Jout[st.iter] = st.prevJ;
        st.accepted = 0; st.dJ = dJ;
        if (st.rho == T(kRhoMax) && !sp.ignore_max_rho_exit) { st.done = 3; return; }
    } else {                                            // accept: lower rho, relative decrease test
        rho_decrease(st);
        dJ = dJ / st.prevJ; st.prevJ = J[aidx];
        st.alphaIndex = aidx; alphaOut[st.iter] = aidx; Jout[st.iter] = J[aidx];
        if (aidx > 0) st.took_step = 1;
        st.accepted = 1; st.dJ = dJ; st.win_pending = 1;
        st.cur = 1 - st.cur;                            // the winner is copied into the other half of xb by the NIS launch
        if (dJ < T(sp.tol_cost)) { st.done = 1; return; }
    }
    if (st.iter == sp.max_iter) st.done = 2; else { st.iter += 1; st.pw ^= 1; }   // Pp <- P, pp <- p of nextIterationSetupGPU (:266-267), by index
}

}  // namespace pddp
