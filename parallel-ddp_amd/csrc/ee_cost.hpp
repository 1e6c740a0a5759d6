// End-effector cost family of the KUKA arm (EE_COST 1 with USE_EE_VEL_COST 0, USE_SMOOTH_ABS 0, USE_LIMITS_FLAG 0 -- the
// configuration of examples/WAFR_MPC_examples.cu:4-37), wave-cooperative.
//
// Replaces compute_eePos (plants/dynamics_arm.cuh:1879-1925), eeCost / deeCost / nominalStateCost / dNominalStateCost and the
// end-effector costFunc / costGrad (plants/cost_arm.cuh:206-389).
//
//   cost_k = 1/2 sum_{i<6} w_i (ee_i - goal_i)^2                              ee = (x, y, z, roll, pitch, yaw) of the tool point
//          + sum_{j<7} [ 1/2 R_EE u_j^2 (k < N-1) + 1/2 (Qx (q_j - xt_j)^2 + Qxd (qd_j - xt_{7+j})^2) ]
//   with the FINAL weights of the end-effector term from knot N-1-timeShift on, of the nominal-state term at knot N-1 only;
//   g_k   = [ Jee' W (ee - goal) + Qx (q - xt) ; Qxd (qd - xt) ; R_EE u ]
//   H_k   = [ Jee' Jee + Qx I, 0, 0 ; 0, Qxd I, 0 ; 0, 0, R_EE I ]           -- Gauss-Newton, and UNWEIGHTED: the reference has the
//           weight factor commented out (cost_arm.cuh:366), so all six rows of Jee enter with weight 1 whatever W is.
//
// Summation orders are the reference's: inside the rollout the cost is accumulated in seven per-joint partial sums (joint 0
// carries the end-effector term) that are added 0..6 at the end of a shooting segment, the segments then in order
// (fpHelpers.cuh:259-265,298-300, costKern<T,0> :169-178); in initAlgGPU one scalar per knot, tree-summed (costKern<T,1> :179-190).
//
// The Jacobian is not the reference's dT chain (compute_dT_dTA_dJ): with the world-frame joint axes S_k = [z_k; o_k x z_k] that the
// dynamics leave in LDS, d p_ee / d q_k = z_k x p_ee + (o_k x z_k) and d R / d q_k = skew(z_k) R -- the same numbers up to rounding.
#pragma once

#include <cmath>

#include "plants.hpp"

namespace pddp {

template <typename T>
struct EeScratch {
    T pos[6];        // tool point position and roll / pitch / yaw           (s_eePos)
    T fac[7];        // d rpy / d (rotation entries)                         (s_temp)
    T dpos[6 * 7];   // d pos / d q_k, [k][6]                                (s_deePos)
    T acc[7];        // per-joint running cost of a rollout                  (s_cost)
};

// tool point of the last link from its world transform (valid after P::dynamics / P::gradient at the same state); one lane.
template <typename T>
PDDP_HD void ee_position(const Wave& w, const ArmScratch<T>& s, const CostWeights<T>& cw, EeScratch<T>& e) {
    if (w.lane == 0) {
        const T* Te = &s.Tw[16 * (kArmNB - 1)];
        for (int i = 0; i < 3; i++) e.pos[i] = Te[8 + i] * cw.ee_z + Te[12 + i];          // EE_ON_LINK_X = EE_ON_LINK_Y = 0 (:48-49)
        e.pos[3] = tatan2<T>(Te[6], Te[10]);
        e.pos[4] = tatan2<T>(-Te[2], tsqrt<T>(Te[6] * Te[6] + Te[10] * Te[10]));
        e.pos[5] = tatan2<T>(Te[1], Te[0]);
    }
    wsync(w);
}

// ... and its Jacobian with respect to the joint angles (joint velocities do not enter)
template <typename T>
PDDP_HD void ee_jacobian(const Wave& w, const ArmScratch<T>& s, const CostWeights<T>& cw, EeScratch<T>& e) {
    const T* Te = &s.Tw[16 * (kArmNB - 1)];
    if (w.lane == 0) {
        const T f3 = Te[6] * Te[6] + Te[10] * Te[10];
        const T f4 = T(1) / (Te[2] * Te[2] + f3);
        const T f5 = T(1) / (Te[1] * Te[1] + Te[0] * Te[0]);
        const T sq = tsqrt<T>(f3);
        e.fac[0] = -Te[6] / f3;
        e.fac[1] = Te[10] / f3;
        e.fac[2] = Te[2] * Te[6] * f4 / sq;
        e.fac[3] = Te[2] * Te[10] * f4 / sq;
        e.fac[4] = -sq * f4;
        e.fac[5] = -Te[1] * f5;
        e.fac[6] = Te[0] * f5;
    }
    wsync(w);
    PDDP_FOR(k, kArmNB) {
        const T* S = &s.S[6 * k];                          // [z_k; o_k x z_k]
        T dc0[3], dc1[3], dc2[3], dp[3];                   // derivatives of the three rotation columns and of the origin of link 7
        cross3(dc0, S, Te); cross3(dc1, S, Te + 4); cross3(dc2, S, Te + 8); cross3(dp, S, Te + 12);
        T* d = &e.dpos[6 * k];
        for (int i = 0; i < 3; i++) d[i] = dc2[i] * cw.ee_z + (dp[i] + S[3 + i]);
        d[3] = e.fac[0] * dc2[2] + e.fac[1] * dc1[2];
        d[4] = e.fac[2] * dc1[2] + e.fac[3] * dc2[2] + e.fac[4] * dc0[2];
        d[5] = e.fac[5] * dc0[0] + e.fac[6] * dc0[1];
    }
    wsync(w);
}

// 1/2 (ee - goal)' W (ee - goal): eeCost, cost_arm.cuh:208-222
template <typename T>
PDDP_HD T ee_term(const CostWeights<T>& cw, const T* pos, const T* goal, bool fin) {
    T cost = 0;
    for (int i = 0; i < 6; i++) {
        const T dl = pos[i] - goal[i];
        cost += T(0.5) * (fin ? (i < 3 ? cw.QF_EE1 : cw.QF_EE2) : (i < 3 ? cw.Q_EE1 : cw.Q_EE2)) * dl * dl;
    }
    if (cw.smooth_abs) cost = tsqrt<T>(T(2) * cost + cw.sa2) - cw.sa;       // USE_SMOOTH_ABS (cost_arm.cuh:218-220)
    return cost;
}
// USE_SMOOTH_ABS: what deeCost divides the tool-point gradient by (cost_arm.cuh:242-251): sqrt(sum_i w_i delta_i^2 + alpha^2)
template <typename T>
PDDP_HD T ee_smooth_abs_divisor(const CostWeights<T>& cw, const T* pos, const T* goal, bool fin) {
    T val2 = 0;
    for (int i = 0; i < 6; i++) {
        const T dl = pos[i] - goal[i];
        val2 += (fin ? (i < 3 ? cw.QF_EE1 : cw.QF_EE2) : (i < 3 ? cw.Q_EE1 : cw.Q_EE2)) * dl * dl;
    }
    val2 += cw.sa2;
    return tsqrt<T>(val2);
}
// control + nominal-state terms of joint `ind`: cost_arm.cuh:287-288, 257-264
template <typename T>
PDDP_HD T ee_joint_terms(const CostWeights<T>& cw, const T* x, const T* u, const T* xt, int ind, int k, int N, T cost) {
    cost += T(0.5) * (k == N - 1 ? T(0) : cw.R_EE) * u[ind] * u[ind];
    const T Qq = (k == N - 1 ? cw.QF_xEE : cw.Q_xEE), Qqd = (k == N - 1 ? cw.QF_xdEE : cw.Q_xdEE);
    const T dq = x[ind] - xt[ind], dqd = x[ind + kArmNB] - xt[ind + kArmNB];
    cost += T(0.5) * (Qq * dq * dq + Qqd * dqd * dqd);
    if (cw.limits) { cost += arm_limit_term<T>(x, u, ind, 0); cost += arm_limit_term<T>(x, u, ind + kArmNB, 0); cost += arm_limit_term<T>(x, u, ind + 2 * kArmNB, 0); }   // USE_LIMITS_FLAG (cost_arm.cuh:289-291,310-312)
    return cost;
}
// the rollout's accumulation (costFunc with s_cost, cost_arm.cuh:277-294): e.acc[ind] += cost of joint ind at knot k
template <typename T>
PDDP_HD void ee_cost_accumulate(const Wave& w, EeScratch<T>& e, const CostWeights<T>& cw, const T* goal, const T* xt, const T* x, const T* u, int k,
                                int N, int tshift) {
    PDDP_FOR(ind, kArmNB) {
        T cost = 0;
        if (ind == 0) cost += ee_term<T>(cw, e.pos, goal, k >= N - 1 - tshift);
        e.acc[ind] += ee_joint_terms<T>(cw, x, u, xt, ind, k, N, cost);
    }
}
// one scalar per knot (costFunc returning a value, cost_arm.cuh:298-315): what costGrad stores in d_JT[k] during initAlgGPU; one lane
template <typename T>
PDDP_HD T ee_cost_knot(const EeScratch<T>& e, const CostWeights<T>& cw, const T* goal, const T* xt, const T* x, const T* u, int k, int N, int tshift) {
    T cost = 0;
    for (int ind = 0; ind < kArmNB; ind++) {
        if (ind == 0) cost += ee_term<T>(cw, e.pos, goal, k >= N - 1 - tshift);
        cost = ee_joint_terms<T>(cw, x, u, xt, ind, k, N, cost);
    }
    return cost;
}

// g_k and H_k (costGrad, cost_arm.cuh:319-380); x, u in LDS, Hk / gk global, coalesced writes of the whole 21 x 21 block
template <typename T>
PDDP_HD void ee_cost_grad(const Wave& w, const EeScratch<T>& e, const CostWeights<T>& cw, const T* goal, const T* xt, const T* x, const T* u, int k,
                          int N, int tshift, T* Hk, T* gk) {
    constexpr int NP = kArmNB, NX = 14, NM = 21;
    const bool fin_ee = k >= N - 1 - tshift, fin = (k == N - 1);
    PDDP_FOR(r, NM) {
        T val = 0;
        if (r < NP) {
            T dv = 0;
            for (int i = 0; i < 6; i++) {
                const T dl = e.pos[i] - goal[i];
                dv += (fin_ee ? (i < 3 ? cw.QF_EE1 : cw.QF_EE2) : (i < 3 ? cw.Q_EE1 : cw.Q_EE2)) * dl * e.dpos[r * 6 + i];
            }
            if (cw.smooth_abs) dv /= ee_smooth_abs_divisor<T>(cw, e.pos, goal, fin_ee);
            val += dv;
        }
        if (r < NX) val += (r < NP ? (fin ? cw.QF_xEE : cw.Q_xEE) : (fin ? cw.QF_xdEE : cw.Q_xdEE)) * (x[r] - xt[r]);
        else val += (fin ? T(0) : cw.R_EE) * u[r - NX];
        if (cw.limits) val += arm_limit_term<T>(x, u, r, 1);                                            // cost_arm.cuh:341-343
        gk[r] = val;
    }
    PDDP_FOR(el, NM * NM) {
        const int c = el / NM, r = el % NM;
        T val = 0;
        if (r < NP && c < NP) for (int j = 0; j < 6; j++) val += e.dpos[r * 6 + j] * e.dpos[c * 6 + j];
        if (r == c) {
            if (r < NX) val += (r < NP ? (fin ? cw.QF_xEE : cw.Q_xEE) : (fin ? cw.QF_xdEE : cw.Q_xdEE));
            else val += fin ? T(0) : cw.R_EE;
            if (cw.limits) val += arm_limit_term<T>(x, u, r, 2);                                        // cost_arm.cuh:374-376
        }
        Hk[el] = val;
    }
}

}  // namespace pddp
