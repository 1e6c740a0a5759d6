// End-effector cost family on lane groups (lanegroup.hpp): the same quantities, operations and summation orders as ee_cost.hpp
// (which restates compute_eePos, plants/dynamics_arm.cuh:1879-1925, and the end-effector costFunc / costGrad, plants/cost_arm.cuh:206-389),
// laid out like the rest of the lane-group family: lane l of a group = joint l.
//
//   * the tool point comes off lane 6's world frame (link 7); roll / pitch / yaw are only evaluated when they carry weight
//     (_Q_EE2 = _QF_EE2 = 0 in the reference's configuration: a zero weight multiplies them away exactly);
//   * rollout cost: lane l keeps the running partial sum of joint l (s_cost[l]); lane 0's also takes the end-effector term;
//   * Jacobian: lane k computes column k, d p_ee / d q_k = z_k x p_ee + o_k x z_k and d R / d q_k = skew(z_k) R, from its own
//     joint axis and the broadcast frame of link 7; the 7 x 7 Gauss-Newton block J'J is formed from 42 intra-group broadcasts.
#pragma once

#include "ee_cost.hpp"
#include "plant_arm_lg.hpp"

namespace pddp {

template <typename T>
PDDP_HD bool ee_rpy_weighted(const CostWeights<T>& cw) { return cw.Q_EE2 != T(0) || cw.QF_EE2 != T(0); }

// tool point (and rpy when weighted, else 0) of THIS lane's link frame Te[3*col + row]
template <typename L, typename T>
PDDP_HD void lg_tool_point(const CostWeights<T>& cw, const typename L::V* Te, typename L::V* pos) {
    using V = typename L::V;
#pragma unroll
    for (int i = 0; i < 3; i++) pos[i] = Te[6 + i] * V(cw.ee_z) + Te[9 + i];
    if (ee_rpy_weighted<T>(cw)) {
        pos[3] = L::vatan2(Te[5], Te[8]);                                   // atan2(R21, R22)
        pos[4] = L::vatan2(-Te[2], L::vsqrt(Te[5] * Te[5] + Te[8] * Te[8]));
        pos[5] = L::vatan2(Te[1], Te[0]);
    } else {
        pos[3] = V(T(0)); pos[4] = V(T(0)); pos[5] = V(T(0));
    }
}
// eeCost (cost_arm.cuh:208-222) of this lane's pos[]; goal[i] per lane (uniform inside a group)
template <typename L, typename T>
PDDP_HD typename L::V lg_ee_term(const CostWeights<T>& cw, const typename L::V* pos, const typename L::V* goal, bool fin) {
    using V = typename L::V;
    V cost = V(T(0));
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const V dl = pos[i] - goal[i];
        cost = cost + V(T(0.5) * (fin ? (i < 3 ? cw.QF_EE1 : cw.QF_EE2) : (i < 3 ? cw.Q_EE1 : cw.Q_EE2))) * dl * dl;
    }
    return cost;
}
// 0.5 R u^2 + 0.5 (Qx dq^2 + Qxd dqd^2) of this lane's joint, added to `cost` in the reference's order (cost_arm.cuh:287-288)
template <typename L, typename T>
PDDP_HD typename L::V lg_ee_joint_terms(const CostWeights<T>& cw, typename L::V q, typename L::V qd, typename L::V u, typename L::V tq, typename L::V tv,
                                        bool fin, typename L::V cost) {
    using V = typename L::V;
    cost = cost + V(T(0.5) * (fin ? T(0) : cw.R_EE)) * u * u;
    const V dq = q - tq, dv = qd - tv;
    cost = cost + V(T(0.5)) * (V(fin ? cw.QF_xEE : cw.Q_xEE) * dq * dq + V(fin ? cw.QF_xdEE : cw.Q_xdEE) * dv * dv);
    return cost;
}

}  // namespace pddp
