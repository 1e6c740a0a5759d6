// Next-iteration setup of the KUKA arm on lane groups: one 8-lane group per knot (8 knots per wave), the analytic
// gradient of plant_arm_lg.hpp, AB_k / g_k written straight from registers.
//
// Same outputs as nis_body()/nis_knot() (bodies.hpp, nis.hpp), which replace integratorGradientKern + costGradientHessianKern
// + memcpyCurrAKern x3 + the D2D copies of nextIterationSetupGPU (DDPHelpers/nisInitHelpers.cuh:205-221, 46-93, 24-32,
// 247-279).  One difference in data movement: for the diagonal quadratic costs of this path the cost Hessian H_k does not
// depend on the trajectory, so it is written once per solve (init mode) and not again every iteration (the reference
// rewrites the same numbers every time).
#pragma once

#include "nis.hpp"
#include "plant_arm_lg.hpp"
#include "solver_state.hpp"

namespace pddp {

// knot k of problem pb; mode 1 = initAlgGPU derivatives (no copies, writes H), mode 0 = iteration (see nis_body)
template <typename L, typename T>
PDDP_HD void arm_lg_nis_body(const ArmLgConst<L>& c, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, int mode, int k, int pb) {
    using V = typename L::V;
    constexpr int NX = 14, NU = 7, NM = 21, NP = 7;
    const typename L::M act = L::all_true();
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    T* xc = b.xb + ((size_t)pb * 2 + st.cur) * N * NX + (size_t)k * NX;
    T* uc = b.ucur + ((size_t)pb * N + k) * NU;
    V q, qd, u;
    if (mode == 0) {
        if (st.accepted < 0) return;                           // backward pass failed: nothing moved
        const bool bnd = dm.M > 1 && dm.on_defect_boundary(k);
        if (bnd) {                                             // Pp <- P, pp <- p at the slots the next backward pass reads
            const size_t o = ((size_t)pb * N + k);
            for (int t = 0; t < NX * NX / 7; t++)
                L::scatter(b.Pp + o * NX * NX, [t](int l) { return l + 7 * t; }, L::gather(b.P + o * NX * NX, [t](int l) { return l + 7 * t; }), act);
            for (int t = 0; t < 2; t++)
                L::scatter(b.pp + o * NX, [t](int l) { return l + 7 * t; }, L::gather(b.p + o * NX, [t](int l) { return l + 7 * t; }), act);
        }
        if (st.accepted != 1) return;                          // rejected: trajectory and derivatives are unchanged
        const size_t slot = (size_t)pb * dm.A + st.alphaIndex;
        const T* xw = b.xs + (slot * N + k) * NX; const T* uw = b.us + (slot * N + k) * NU;
        q = L::gather(xw, [](int l) { return l; }); qd = L::gather(xw, [](int l) { return l + NP; }); u = L::gather(uw, [](int l) { return l; });
        L::scatter(xc, [](int l) { return l; }, q, act); L::scatter(xc, [](int l) { return l + NP; }, qd, act); L::scatter(uc, [](int l) { return l; }, u, act);
        if (bnd) {
            const T* dw = b.ds + (slot * N + k) * NX; T* dc = b.dcur + ((size_t)pb * N + k) * NX;
            L::scatter(dc, [](int l) { return l; }, L::gather(dw, [](int l) { return l; }), act);
            L::scatter(dc, [](int l) { return l + NP; }, L::gather(dw, [](int l) { return l + NP; }), act);
        }
        if (st.done) return;                                   // final accepted step: solution copied, no derivatives needed
    } else {
        q = L::gather(xc, [](int l) { return l; }); qd = L::gather(xc, [](int l) { return l + NP; }); u = L::gather(uc, [](int l) { return l; });
    }
    const T* xg = b.xGoal + (size_t)pb * NX;
    T* ABk = b.AB + ((size_t)pb * N + k) * NX * NM; T* Hk = b.H + ((size_t)pb * N + k) * NM * NM; T* gk = b.g + ((size_t)pb * N + k) * NM;
    const bool fin = (k == N - 1);
    const T w1 = fin ? cw.QF1 : cw.Q1, w2 = fin ? cw.QF2 : cw.Q2, w3 = fin ? T(0) : cw.R;       // ArmPlant::weight
    // cost gradient g_k = weight .* [x - xg; u]
    L::scatter(gk, [](int l) { return l; }, V(w1) * (q - L::gather(xg, [](int l) { return l; })), act);
    L::scatter(gk, [](int l) { return l + NP; }, V(w2) * (qd - L::gather(xg, [](int l) { return l + NP; })), act);
    L::scatter(gk, [](int l) { return l + NX; }, V(w3) * u, act);
    if (mode == 1) {                                           // H_k = diag(weight): constant over the solve
        for (int t = 0; t < NM * NM / 7; t++) {
            const V hv = L::make([=](int l) { const int e = l + 7 * t, i = e / NM, j = e % NM; return i != j ? T(0) : (i < NP ? w1 : (i < NX ? w2 : w3)); });
            L::scatter(Hk, [t](int l) { return l + 7 * t; }, hv, act);
        }
    }
    if (fin) return;
    // Euler: AB = I + dt [0 I 0; dqdd]   (integrator_gradient, INTEG == 1; utils/integrators.cuh:38-53)
    for (int ky = 0; ky < NM; ky++) {                          // rows 0..6 (positions): constants
        const V v = L::make([=](int l) { return T(ky == l ? 1 : 0) + dt * T(l + NP == ky ? 1 : 0); });
        L::scatter(ABk, [ky](int l) { return ky * NX + l; }, v, act);
    }
    ArmLgState<L> as;
    const V qdd = arm_lg_dynamics<L>(c, as, q, qd, u);
    arm_lg_gradient<L>(c, as, qd, qdd, [&](int jj, const V& val) {      // rows 7..13: lane l owns row 7 + l
        const V dlt = L::make([jj](int l) { return T(jj == l + 7 ? 1 : 0); });
        L::scatter(ABk, [jj](int l) { return jj * 14 + 7 + l; }, dlt + V(dt) * val, act);
    });
}

}  // namespace pddp
