// Next-iteration setup of the KUKA arm on lane groups: one 8-lane group per knot (8 knots per wave), the analytic
// gradient of plant_arm_lg.hpp, AB_k / g_k written straight from registers.
//
// Same outputs as nis_body()/nis_knot() (bodies.hpp, nis.hpp), which replace integratorGradientKern + costGradientHessianKern
// + memcpyCurrAKern x3 + the D2D copies of nextIterationSetupGPU (DDPHelpers/nisInitHelpers.cuh:205-221, 46-93, 24-32,
// 247-279).  One difference in data movement: for the diagonal quadratic costs of this path the cost Hessian H_k does not
// depend on the trajectory, so it is written once per solve (init mode) and not again every iteration (the reference
// rewrites the same numbers every time).
#pragma once

#include "ee_cost_lg.hpp"
#include "nis.hpp"
#include "plant_arm_lg.hpp"
#include "solver_state.hpp"

namespace pddp {

// AB_k rows of the Euler step (shared by the joint-space and the end-effector variants)
template <typename L, typename T>
PDDP_HD void arm_lg_write_AB(const ArmLgConst<L>& c, const Buffers<T>& b, T dt, unsigned oAB, typename L::V q, typename L::V qd, typename L::V u) {
    using V = typename L::V;
    constexpr int NX = 14, NM = 21, NP = 7;
    const typename L::M act = L::all_true();
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

// End-effector branch of costGradientHessianKern (nisInitHelpers.cuh:52-84) for one knot: AB_k as before, then the tool point, its
// Jacobian, g_k, the full H_k (it depends on the trajectory here: rewritten every iteration) and, in init mode, the knot's cost.
template <typename L, typename T>
PDDP_HD void arm_lg_nis_ee(const ArmLgConst<L>& c, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, int mode, int k, int pb,
                           typename L::V q, typename L::V qd, typename L::V u) {
    using V = typename L::V;
    constexpr int NX = 14, NM = 21, NP = 7;
    const typename L::M act = L::all_true(), first_lane = L::lane_is(0), last_lane = L::lane_is(6);
    const unsigned N = dm.N, knot = (unsigned)pb * N + k;
    const unsigned oxg = (unsigned)pb * NX, oAB = knot * (NX * NM), oH = knot * (NM * NM), og = knot * NM;
    const bool fin = (k == (int)N - 1), fin_ee = k >= (int)N - 1 - b.tshift[pb];
    if (!fin) arm_lg_write_AB<L, T>(c, b, dt, oAB, q, qd, u);
    // ---- kinematics again, on free registers: link frames (bit for bit those of the dynamics), joint axis, frame of link 7 for everyone
    V Tw[12], S[6], Te[12];
    arm_lg_world_frames<L>(c, q, Tw);
    {
        V z[3] = {Tw[6], Tw[7], Tw[8]}, p[3] = {Tw[9], Tw[10], Tw[11]};
        S[0] = z[0]; S[1] = z[1]; S[2] = z[2];
        lg_cross3(S + 3, p, z);
    }
#pragma unroll
    for (int e = 0; e < 12; e++) Te[e] = L::template bcast<6>(Tw[e]);
    V pos[6], goal[6];
    lg_tool_point<L, T>(cw, Te, pos);
#pragma unroll
    for (int i = 0; i < 6; i++) goal[i] = L::gather_at(b.xGoal, oxg, [i](int) { return i; });
    // d rpy / d (rotation entries): compute_eePos :1899-1911.  4x4 index -> here: Tee[0,1,2] = Te[0,1,2], Tee[6] = Te[5], Tee[10] = Te[8]
    V fac[7];
    {
        const V f3 = Te[5] * Te[5] + Te[8] * Te[8];
        const V f4 = V(T(1)) / (Te[2] * Te[2] + f3);
        const V f5 = V(T(1)) / (Te[1] * Te[1] + Te[0] * Te[0]);
        const V sq = L::vsqrt(f3);
        fac[0] = -Te[5] / f3; fac[1] = Te[8] / f3;
        fac[2] = Te[2] * Te[5] * f4 / sq; fac[3] = Te[2] * Te[8] * f4 / sq; fac[4] = -sq * f4;
        fac[5] = -Te[1] * f5; fac[6] = Te[0] * f5;
    }
    V d[6];                                                    // column `lane` of the Jacobian
    {
        V dc0[3], dc1[3], dc2[3], dp[3];
        lg_cross3(dc0, S, Te); lg_cross3(dc1, S, Te + 3); lg_cross3(dc2, S, Te + 6); lg_cross3(dp, S, Te + 9);
#pragma unroll
        for (int i = 0; i < 3; i++) d[i] = dc2[i] * V(cw.ee_z) + (dp[i] + S[3 + i]);
        d[3] = fac[0] * dc2[2] + fac[1] * dc1[2];
        d[4] = fac[2] * dc1[2] + fac[3] * dc2[2] + fac[4] * dc0[2];
        d[5] = fac[5] * dc0[0] + fac[6] * dc0[1];
    }
    const V tq = L::gather_at(b.xTarget, oxg, [](int l) { return l; }), tv = L::gather_at(b.xTarget, oxg, [](int l) { return l + NP; });
    const T Qx = fin ? cw.QF_xEE : cw.Q_xEE, Qxd = fin ? cw.QF_xdEE : cw.Q_xdEE, Ru = fin ? T(0) : cw.R_EE;
    // ---- g_k (costGrad :330-345)
    {
        V dv = V(T(0));
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const V dl = pos[i] - goal[i];
            dv = dv + V(fin_ee ? (i < 3 ? cw.QF_EE1 : cw.QF_EE2) : (i < 3 ? cw.Q_EE1 : cw.Q_EE2)) * dl * d[i];
        }
        L::scatter_at(b.g, og, [](int l) { return l; }, dv + V(Qx) * (q - tq), act);
        L::scatter_at(b.g, og, [](int l) { return l + NP; }, V(Qxd) * (qd - tv), act);
        L::scatter_at(b.g, og, [](int l) { return l + NX; }, V(Ru) * u, act);
    }
    // ---- H_k (costGrad :347-379): column cc, rows (l, l + 7, l + 14)
    {
        V dcol[7][6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            dcol[0][j] = L::template bcast<0>(d[j]); dcol[1][j] = L::template bcast<1>(d[j]); dcol[2][j] = L::template bcast<2>(d[j]);
            dcol[3][j] = L::template bcast<3>(d[j]); dcol[4][j] = L::template bcast<4>(d[j]); dcol[5][j] = L::template bcast<5>(d[j]);
            dcol[6][j] = L::template bcast<6>(d[j]);
        }
#pragma unroll
        for (int cc = 0; cc < NP; cc++) {
            V val = d[0] * dcol[cc][0];
#pragma unroll
            for (int j = 1; j < 6; j++) val = val + d[j] * dcol[cc][j];
            val = L::sel(L::lane_is(cc), val + V(Qx), val);
            L::scatter_at(b.H, oH, [cc](int l) { return cc * NM + l; }, val, act);
            L::scatter_at(b.H, oH, [cc](int l) { return cc * NM + l + NP; }, V(T(0)), act);
            L::scatter_at(b.H, oH, [cc](int l) { return cc * NM + l + NX; }, V(T(0)), act);
        }
        for (int cc = NP; cc < NM; cc++) {
            const T wd = cc < NX ? Qxd : Ru;
            const V diag = L::make([=](int l) { return (l + (cc < NX ? NP : NX)) == cc ? wd : T(0); });
            L::scatter_at(b.H, oH, [cc](int l) { return cc * NM + l; }, V(T(0)), act);
            L::scatter_at(b.H, oH, [cc](int l) { return cc * NM + l + NP; }, cc < NX ? diag : V(T(0)), act);
            L::scatter_at(b.H, oH, [cc](int l) { return cc * NM + l + NX; }, cc < NX ? V(T(0)) : diag, act);
        }
    }
    // ---- the knot's cost as ONE running sum over the joints (costFunc returning a value, :298-315), init mode only
    if (mode == 1) {
        const V ee = lg_ee_term<L, T>(cw, pos, goal, fin_ee);   // every lane holds the frame of link 7: the same value everywhere
        V acc = lg_ee_joint_terms<L, T>(cw, q, qd, u, tq, tv, fin, ee);            // correct for lane 0 (0 + ee = ee)
#pragma unroll
        for (int s = 1; s < 7; s++) acc = L::sel(first_lane, acc, lg_ee_joint_terms<L, T>(cw, q, qd, u, tq, tv, fin, L::up(acc)));
        L::scatter(b.costk, [knot](int) { return (int)knot; }, acc, last_lane);
    }
}

// knot k of problem pb; mode 1 = initAlgGPU derivatives (no copies, writes H), mode 0 = iteration (see nis_body)
template <typename L, typename T, bool EE = false>
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
    if constexpr (EE) { arm_lg_nis_ee<L, T>(c, b, dm, cw, dt, mode, k, pb, q, qd, u); return; }
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
    arm_lg_write_AB<L, T>(c, b, dt, oAB, q, qd, u);
}

}  // namespace pddp
