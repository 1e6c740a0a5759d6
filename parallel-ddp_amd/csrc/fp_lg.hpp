// Forward pass of the KUKA arm on lane groups (lanegroup.hpp): one 8-lane group per (line-search candidate, shooting
// segment), 8 rollouts per wave, no LDS in the rollout loop.
//
// Same results as fp.hpp (which restates forwardSweepInner / forwardSimInner / computeControlKT / costKern / defectKern,
// DDPHelpers/fpHelpers.cuh:19-53, 225-275, 202-221, 134-152, 96-111): every sum is accumulated in the same order.
// Lane l of a group holds joint l's position and velocity (state entries l and l+7) and control l.
//   * sweep:   group = candidate alpha; serial over knots, stops after the last segment boundary (nothing it computes
//              beyond that is ever read -- the rollouts overwrite it); writes the M-1 segment start states x[b*NB].
//   * rollout: group = (alpha, segment); per step: control law (14 intra-group broadcasts of dx, one gain row per lane),
//              running cost as a chain over the lanes in the reference's summation order, arm_lg_dynamics, Euler step.
#pragma once

#include "fp.hpp"
#include "plant_arm_lg.hpp"

namespace pddp {

// sum over the 7 active lanes in lane order, starting from `init` (a value every lane holds): returns, in every lane l, the
// partial sum init + t_0 + ... + t_l accumulated left to right -- the order of a serial loop.
template <typename L>
PDDP_HD typename L::V lg_chain_sum(typename L::V init, typename L::V term) {
    using V = typename L::V;
    V acc = init + term;                                   // correct for lane 0
    const typename L::M first = L::lane_is(0);
#pragma unroll
    for (int s = 1; s < 7; s++) acc = L::sel(first, acc, L::up(acc) + term);
    return acc;
}

// 14 broadcasts: bc[c] = entry c of the 14-vector whose entries l / l+7 live in lane l as (lo, hi)
template <typename L>
PDDP_HD void lg_bcast14(typename L::V* bc, typename L::V lo, typename L::V hi) {
    bc[0] = L::template bcast<0>(lo); bc[1] = L::template bcast<1>(lo); bc[2] = L::template bcast<2>(lo); bc[3] = L::template bcast<3>(lo);
    bc[4] = L::template bcast<4>(lo); bc[5] = L::template bcast<5>(lo); bc[6] = L::template bcast<6>(lo);
    bc[7] = L::template bcast<0>(hi); bc[8] = L::template bcast<1>(hi); bc[9] = L::template bcast<2>(hi); bc[10] = L::template bcast<3>(hi);
    bc[11] = L::template bcast<4>(hi); bc[12] = L::template bcast<5>(hi); bc[13] = L::template bcast<6>(hi);
}

// Linear sweep for one candidate (M > 1).  a.x receives the segment start states.
template <typename L, typename T>
PDDP_HD void arm_lg_forward_sweep(const Dims& dm, const FpArgs<T>& a) {
    using V = typename L::V;
    constexpr int NX = 14, NP = 7;
    const typename L::M act = L::all_true();
    V xq = L::gather(a.xcur, [](int l) { return l; }), xv = L::gather(a.xcur, [](int l) { return l + NP; });
    const int k_last = (dm.M - 1) * dm.NB - 1;             // last defect boundary
    for (int k = 0; k <= k_last; k++) {
        const T* Ak = a.ApBK + NX * NX * k;
        const V dq = xq - L::gather(a.xcur, [k](int l) { return NX * k + l; });
        const V dv = xv - L::gather(a.xcur, [k](int l) { return NX * k + l + NP; });
        V bc[14];
        lg_bcast14<L>(bc, dq, dv);
        V vq = L::gather(Ak, [](int l) { return l; }) * bc[0];
        V vv = L::gather(Ak, [](int l) { return l + NP; }) * bc[0];
#pragma unroll
        for (int i = 1; i < NX; i++) {
            vq = vq + L::gather(Ak, [i](int l) { return l + NX * i; }) * bc[i];
            vv = vv + L::gather(Ak, [i](int l) { return l + NP + NX * i; }) * bc[i];
        }
        const bool bnd = dm.on_defect_boundary(k);
        V nq = L::gather(a.xcur, [k](int l) { return NX * (k + 1) + l; });
        V nv = L::gather(a.xcur, [k](int l) { return NX * (k + 1) + l + NP; });
        V aq = -V(a.alpha) * L::gather(a.Bdu, [k](int l) { return NX * k + l; }) + vq;
        V av = -V(a.alpha) * L::gather(a.Bdu, [k](int l) { return NX * k + l + NP; }) + vv;
        if (bnd) {
            aq = aq + L::gather(a.dcur, [k](int l) { return NX * k + l; });
            av = av + L::gather(a.dcur, [k](int l) { return NX * k + l + NP; });
        }                                                  // the cooperative code adds an exact 0 off the boundaries
        xq = nq + aq; xv = nv + av;
        if (bnd) {
            L::scatter(a.x, [k](int l) { return NX * (k + 1) + l; }, xq, act);
            L::scatter(a.x, [k](int l) { return NX * (k + 1) + l + NP; }, xv, act);
        }
    }
}

// running / terminal cost of one knot in the order of ArmPlant::cost (plants.hpp; plants/cost_arm.cuh:130-153): returned in lane 6
template <typename L, typename T>
PDDP_HD typename L::V arm_lg_cost(const CostWeights<T>& cw, typename L::V q, typename L::V qd, typename L::V u, typename L::V gq,
                                  typename L::V gv, bool final_knot) {
    using V = typename L::V;
    const V dq = q - gq, dv = qd - gv;
    const V w1 = V(final_knot ? cw.QF1 : cw.Q1), w2 = V(final_knot ? cw.QF2 : cw.Q2);
    V acc = lg_chain_sum<L>(V(T(0)), w1 * dq * dq);
    acc = lg_chain_sum<L>(L::template bcast<6>(acc), w2 * dv * dv);
    if (!final_knot) acc = lg_chain_sum<L>(L::template bcast<6>(acc), V(cw.R) * u * u);
    return V(T(0.5)) * acc;
}

// Rollout of segment bInd of one candidate.  cost_k: [N] per-knot costs of THIS candidate (LDS or host memory).
// init_rollout: segment starts come from the loaded trajectory (xcur) instead of from the sweep.
template <typename L, typename T>
PDDP_HD void arm_lg_rollout_segment(const ArmLgConst<L>& c, const Dims& dm, const FpArgs<T>& a, int bInd, const CostWeights<T>& cw,
                                    const T* xg, T* cost_k, bool init_rollout) {
    using V = typename L::V;
    constexpr int NX = 14, NU = 7, NP = 7;
    const typename L::M act = L::all_true(), last_lane = L::lane_is(6);
    const int NBk = dm.NB, kStart = bInd * NBk;
    const int iters = (bInd < dm.M - 1) ? NBk : NBk - 1;
    const V gq = L::gather(xg, [](int l) { return l; }), gv = L::gather(xg, [](int l) { return l + NP; });
    const T* xstart = (bInd == 0 || init_rollout) ? a.xcur : a.x;
    V q = L::gather(xstart, [kStart](int l) { return NX * kStart + l; }), qd = L::gather(xstart, [kStart](int l) { return NX * kStart + l + NP; });
    if (bInd == 0 || init_rollout) {
        L::scatter(a.x, [kStart](int l) { return NX * kStart + l; }, q, act);
        L::scatter(a.x, [kStart](int l) { return NX * kStart + l + NP; }, qd, act);
    }
    ArmLgState<L> st;
    for (int k = 0; k < iters; k++) {
        const int kn = kStart + k;
        const V dq = q - L::gather(a.xcur, [kn](int l) { return NX * kn + l; });
        const V dv = qd - L::gather(a.xcur, [kn](int l) { return NX * kn + l + NP; });
        V bc[14];
        lg_bcast14<L>(bc, dq, dv);
        const T* KTk = a.KT + NX * NU * kn;                 // lane r: row r of K = KT[c + r*NX]
        V Kdx = L::gather(KTk, [](int l) { return l * NX; }) * bc[0];
#pragma unroll
        for (int cc = 1; cc < NX; cc++) Kdx = Kdx + L::gather(KTk, [cc](int l) { return cc + l * NX; }) * bc[cc];
        V u = L::gather(a.ucur, [kn](int l) { return NU * kn + l; });
        u = u - (V(a.alpha) * L::gather(a.du, [kn](int l) { return NU * kn + l; }) + Kdx);
        L::scatter(a.u, [kn](int l) { return NU * kn + l; }, u, act);
        if (cost_k) {
            const V J = arm_lg_cost<L, T>(cw, q, qd, u, gq, gv, false);
            L::scatter(cost_k, [kn](int) { return kn; }, J, last_lane);
        }
        const V qdd = arm_lg_dynamics<L>(c, st, q, qd, u);
        const V qn = q + V(a.dt) * qd, qdn = qd + V(a.dt) * qdd;       // Euler (utils/integrators.cuh:24-36)
        if (k < NBk - 1) {
            L::scatter(a.x, [kn](int l) { return NX * (kn + 1) + l; }, qn, act);
            L::scatter(a.x, [kn](int l) { return NX * (kn + 1) + l + NP; }, qdn, act);
            q = qn; qd = qdn;
        } else if (bInd < dm.M - 1) {                        // defect against the next segment's start state
            const int ks = (bInd + 1) * NBk;
            const T* xnext = init_rollout ? a.xcur : a.x;
            const V eq = qn - L::gather(xnext, [ks](int l) { return NX * ks + l; });
            const V ev = qdn - L::gather(xnext, [ks](int l) { return NX * ks + l + NP; });
            L::scatter(a.d, [ks](int l) { return NX * (ks - 1) + l; }, eq, act);
            L::scatter(a.d, [ks](int l) { return NX * (ks - 1) + l + NP; }, ev, act);
            V sdef = lg_chain_sum<L>(V(T(0)), L::vabs(eq));
            sdef = lg_chain_sum<L>(L::template bcast<6>(sdef), L::vabs(ev));
            L::scatter(a.dnorm, [bInd](int) { return bInd; }, sdef, last_lane);
        }
    }
    if (bInd == dm.M - 1) {                                 // terminal knot: its (unused) control is carried along
        const int kn = dm.N - 1;
        const V u = L::gather(a.ucur, [kn](int l) { return NU * kn + l; });
        L::scatter(a.u, [kn](int l) { return NU * kn + l; }, u, act);
        if (cost_k) {
            const V J = arm_lg_cost<L, T>(cw, q, qd, u, gq, gv, true);
            L::scatter(cost_k, [kn](int) { return kn; }, J, last_lane);
        }
        L::scatter(a.dnorm, [bInd](int) { return bInd; }, V(T(0)), last_lane);
    }
}

}  // namespace pddp
