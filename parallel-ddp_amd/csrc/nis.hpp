// Next-iteration setup for one knot: dynamics/cost derivatives at the current trajectory.
//
// Replaces integratorGradientKern (DDPHelpers/nisInitHelpers.cuh:205-221) and costGradientHessianKern (:46-93, the
// joint-cost branch, one THREAD per knot writing a strided 21x21 block) with one wavefront per knot that writes
// AB_k, H_k, g_k coalesced.  The copies the reference issues around them (memcpyCurrAKern x3, P->Pp, p->pp,
// winner -> xp/up/dp, :266-276) are folded into the same launch: see kernels.hip.
#pragma once

#include "ee_cost.hpp"
#include "integrators.hpp"

namespace pddp {

template <typename P, int INTEG, typename T>
struct NisScratch {
    typename P::Scratch plant;
    typename P::GradScratch pgrad;
    IntegGradScratch<P, INTEG, T> integ;
    T x[P::NX], u[P::NU];
    EeScratch<T> ee;
    T qdd[P::NPOS];
    FdScratch<P, T> fd;
};

// xk/uk: the knot's state and control (global).  Writes ABk (k < N-1), Hk, gk (global).
template <typename P, int INTEG, typename T>
PDDP_HD void nis_knot(const Wave& w, NisScratch<P, INTEG, T>& s, const Dims& dm, int k, const T* xk, const T* uk, const T* xg,
                      const CostWeights<T>& cw, T dt, T* ABk, T* Hk, T* gk, const T* xt = nullptr, int tshift = 0, T* cost_out = nullptr, bool write_const_H = true) {
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU;
    PDDP_FOR(i, NX) s.x[i] = xk[i];
    PDDP_FOR(i, NU) s.u[i] = uk[i];
    wsync();
    if constexpr (P::PLANT == 4) {
        if (cw.ee) {   // costGradientHessianKern, end-effector branch (nisInitHelpers.cuh:52-84): the kinematics come with the dynamics gradient
            if (k < dm.N - 1) integrator_gradient<P, INTEG>(w, s.plant, s.pgrad, s.integ, ABk, s.x, s.u, dt);
            else P::dynamics(w, s.plant, s.qdd, s.x, s.u);
            wsync(w);
            ee_position<T>(w, s.plant, cw, s.ee);
            ee_jacobian<T>(w, s.plant, cw, s.ee);
            ee_cost_grad<T>(w, s.ee, cw, xg, xt, s.x, s.u, k, dm.N, tshift, Hk, gk);
            if (cost_out && w.lane == 0) *cost_out = ee_cost_knot<T>(s.ee, cw, xg, xt, s.x, s.u, k, dm.N, tshift);
            return;
        }
    }
    if constexpr (P::kPluginCost) {
        // a cost file in the reference's form (ref_plugin.hpp): costGrad is scalar code that the reference runs with one THREAD per knot (costGradientHessianKern,
        // nisInitHelpers.cuh:86-92) -- one lane here; H_k may be any matrix the user's costGrad writes, the backward pass reads all of it
        if (w.lane == 0) P::cost_grad(cw, Hk, gk, s.x, s.u, xg, k, dm.N);
    } else {
    // the diagonal joint-space / closed-form cost Hessian does not depend on the trajectory: the reference rewrites it with every setup (costGradientHessianKern), here it is
    // written when the problem is loaded (init mode) and left alone afterwards -- 1 KB per quadrotor knot and sweep that nobody needs again (the arm's thread-lane setup
    // kernels have always done so)
    if (write_const_H) PDDP_FOR(e, NM * NM) { const int i = e / NM, j = e % NM; Hk[e] = (i == j) ? P::weight(cw, i, k, dm.N) : T(0); }
    PDDP_FOR(i, NM) {
        T gv = P::weight(cw, i, k, dm.N) * (i < NX ? (s.x[i] - xg[i]) : s.u[i - NX]);
        if constexpr (P::PLANT == 4) { if (cw.limits && (k < dm.N - 1 || i < NX)) gv += arm_limit_term<T>(s.x, s.u, i, 1); }      // USE_LIMITS_FLAG: the gradient only, H stays (cost_arm.cuh:176-199)
        gk[i] = gv;
    }
    }
    if (k < dm.N - 1) {
        if (INTEG == 1 && cw.fd_eps > 0.0) integrator_gradient_fd<P, T>(w, s.plant, s.fd, ABk, s.x, s.u, dt, cw.fd_eps);   // USE_FINITE_DIFF
        else integrator_gradient<P, INTEG>(w, s.plant, s.pgrad, s.integ, ABk, s.x, s.u, dt);
    }
}

}  // namespace pddp
