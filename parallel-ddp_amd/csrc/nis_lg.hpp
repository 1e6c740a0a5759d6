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
    const unsigned N = dm.N;
    const SolverState<T>& st = b.state[pb];
    // wave-uniform bases (b.*) + 32-bit element offsets (lanegroup.hpp gather_at / scatter_at)
    const unsigned knot = (unsigned)pb * N + k;                               // block index of this knot in every per-knot array
    const unsigned oxc = (((unsigned)pb * 2 + st.cur) * N + k) * NX, ouc = knot * NU;
    V q, qd, u;
    if (mode == 0) {
        if (st.accepted < 0) return;                           // backward pass failed: nothing moved
        const bool bnd = dm.M > 1 && dm.on_defect_boundary(k);
        if (st.accepted != 1) return;                          // rejected: trajectory and derivatives are unchanged
        const unsigned wknot = ((unsigned)pb * dm.A + st.alphaIndex) * N + k;  // this knot in the winner's candidate slot
        q = L::gather_at(b.xs, wknot * NX, [](int l) { return l; }); qd = L::gather_at(b.xs, wknot * NX, [](int l) { return l + NP; });
        u = L::gather_at(b.us, wknot * NU, [](int l) { return l; });
        L::scatter_at(b.xb, oxc, [](int l) { return l; }, q, act); L::scatter_at(b.xb, oxc, [](int l) { return l + NP; }, qd, act);
        L::scatter_at(b.ucur, ouc, [](int l) { return l; }, u, act);
        if (bnd) {
            L::scatter_at(b.dcur, knot * NX, [](int l) { return l; }, L::gather_at(b.ds, wknot * NX, [](int l) { return l; }), act);
            L::scatter_at(b.dcur, knot * NX, [](int l) { return l + NP; }, L::gather_at(b.ds, wknot * NX, [](int l) { return l + NP; }), act);
        }
        if (st.done) return;                                   // final accepted step: solution copied, no derivatives needed
    } else {
        q = L::gather_at(b.xb, oxc, [](int l) { return l; }); qd = L::gather_at(b.xb, oxc, [](int l) { return l + NP; });
        u = L::gather_at(b.ucur, ouc, [](int l) { return l; });
    }
    const unsigned oxg = (unsigned)pb * NX, oAB = knot * (NX * NM), oH = knot * (NM * NM), og = knot * NM;
    const bool fin = (k == (int)N - 1);
    const T w1 = fin ? cw.QF1 : cw.Q1, w2 = fin ? cw.QF2 : cw.Q2, w3 = fin ? T(0) : cw.R;       // ArmPlant::weight
    // cost gradient g_k = weight .* [x - xg; u]
    L::scatter_at(b.g, og, [](int l) { return l; }, V(w1) * (q - L::gather_at(b.xGoal, oxg, [](int l) { return l; })), act);
    L::scatter_at(b.g, og, [](int l) { return l + NP; }, V(w2) * (qd - L::gather_at(b.xGoal, oxg, [](int l) { return l + NP; })), act);
    L::scatter_at(b.g, og, [](int l) { return l + NX; }, V(w3) * u, act);
    if (mode == 1) {                                           // H_k = diag(weight): constant over the solve
        for (int t = 0; t < NM * NM / 7; t++) {
            const V hv = L::make([=](int l) { const int e = l + 7 * t, i = e / NM, j = e % NM; return i != j ? T(0) : (i < NP ? w1 : (i < NX ? w2 : w3)); });
            L::scatter_at(b.H, oH, [t](int l) { return l + 7 * t; }, hv, act);
        }
    }
    if (fin) return;
    // Euler: AB = I + dt [0 I 0; dqdd]   (integrator_gradient, INTEG == 1; utils/integrators.cuh:38-53)
    for (int ky = 0; ky < NM; ky++) {                          // rows 0..6 (positions): constants
        const V v = L::make([=](int l) { return T(ky == l ? 1 : 0) + dt * T(l + NP == ky ? 1 : 0); });
        L::scatter_at(b.AB, oAB, [ky](int l) { return ky * NX + l; }, v, act);
    }
    ArmLgState<L> as;
    const V qdd = arm_lg_dynamics<L>(c, as, q, qd, u);
    arm_lg_gradient<L>(c, as, qd, qdd, [&](int jj, const V& val) {      // rows 7..13: lane l owns row 7 + l
        const V dlt = L::make([jj](int l) { return T(jj == l + 7 ? 1 : 0); });
        L::scatter_at(b.AB, oAB, [jj](int l) { return jj * 14 + 7 + l; }, dlt + V(dt) * val, act);
    });
}

}  // namespace pddp
