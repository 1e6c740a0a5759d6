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

#include "ee_cost_lg.hpp"
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

// Pointers of the whole batch (wave-uniform) + 32-bit element offsets of this (problem, candidate): one VGPR per address
// instead of a 64-bit pointer pair per array (lanegroup.hpp gather_at / scatter_at).
template <typename T>
struct FpLgArgs {
    T* xs; T* us; T* ds;                      // candidate trajectories [B][A][N][.]
    const T* xb; const T* ucur; const T* dcur; const T* KT; const T* du; const T* ApBK; const T* Bdu; const T* xGoal;
    unsigned slotN;                           // (pb * A + a) * N : knot k of this candidate is block slotN + k of xs / us / ds
    unsigned pbN;                             // pb * N
    unsigned oxc;                             // offset of the current trajectory inside xb
    unsigned oxg;                             // pb * NX
    T alpha, dt;
    T* dnorm;                                 // LDS [M] of this candidate
    const T* xTarget; int tshift;             // end-effector cost: nominal-state target [B][14], finalCostShift of this problem
};
template <typename T>
PDDP_HD FpLgArgs<T> fp_lg_args(const Buffers<T>& b, const Dims& dm, int pb, int a_idx, T dt, T* dnorm) {
    FpLgArgs<T> a;
    a.xs = b.xs; a.us = b.us; a.ds = b.ds; a.xb = b.xb; a.ucur = b.ucur; a.dcur = b.dcur; a.KT = b.KT; a.du = b.du; a.ApBK = b.ApBK; a.Bdu = b.Bdu;
    a.xGoal = b.xGoal;
    a.slotN = ((unsigned)pb * dm.A + a_idx) * dm.N; a.pbN = (unsigned)pb * dm.N;
    a.oxc = ((unsigned)pb * 2 + b.state[pb].cur) * dm.N * 14; a.oxg = (unsigned)pb * 14;
    a.alpha = b.alpha[a_idx]; a.dt = dt; a.dnorm = dnorm;
    a.xTarget = b.xTarget; a.tshift = b.tshift[pb];
    return a;
}

// Linear sweep for one candidate (M > 1).  xs receives the segment start states.
template <typename L, typename T>
PDDP_HD void arm_lg_forward_sweep(const Dims& dm, const FpLgArgs<T>& a) {
    using V = typename L::V;
    constexpr int NX = 14, NP = 7;
    const typename L::M act = L::all_true();
    V xq = L::gather_at(a.xb, a.oxc, [](int l) { return l; }), xv = L::gather_at(a.xb, a.oxc, [](int l) { return l + NP; });
    const int k_last = (dm.M - 1) * dm.NB - 1;             // last defect boundary
    // operands of step k (none depends on the running state): fetched one step ahead, their latency hides behind the previous step's arithmetic
    V Aq[NX], Av[NX], cq, cv, nq, nv, bq, bv, dq_, dv_;
    auto fetch = [&](int k) {
        const unsigned oA = (a.pbN + k) * 196, ob = (a.pbN + k) * 14;
#pragma unroll
        for (int i = 0; i < 14; i++) { Aq[i] = L::gather_at(a.ApBK, oA, [i](int l) { return l + 14 * i; }); Av[i] = L::gather_at(a.ApBK, oA, [i](int l) { return l + 7 + 14 * i; }); }
        cq = L::gather_at(a.xb, a.oxc, [k](int l) { return 14 * k + l; }); cv = L::gather_at(a.xb, a.oxc, [k](int l) { return 14 * k + l + 7; });
        nq = L::gather_at(a.xb, a.oxc, [k](int l) { return 14 * (k + 1) + l; }); nv = L::gather_at(a.xb, a.oxc, [k](int l) { return 14 * (k + 1) + l + 7; });
        bq = L::gather_at(a.Bdu, ob, [](int l) { return l; }); bv = L::gather_at(a.Bdu, ob, [](int l) { return l + 7; });
        dq_ = L::gather_at(a.dcur, ob, [](int l) { return l; }); dv_ = L::gather_at(a.dcur, ob, [](int l) { return l + 7; });
    };
    if (k_last >= 0) fetch(0);
    for (int k = 0; k <= k_last; k++) {
        const V dq = xq - cq, dv = xv - cv;
        V bc[14];
        lg_bcast14<L>(bc, dq, dv);
        V vq = Aq[0] * bc[0];
        V vv = Av[0] * bc[0];
#pragma unroll
        for (int i = 1; i < NX; i++) {
            vq = vq + Aq[i] * bc[i];
            vv = vv + Av[i] * bc[i];
        }
        const bool bnd = dm.on_defect_boundary(k);
        V aq = -V(a.alpha) * bq + vq;
        V av = -V(a.alpha) * bv + vv;
        if (bnd) { aq = aq + dq_; av = av + dv_; }         // the cooperative code adds an exact 0 off the boundaries
        xq = nq + aq; xv = nv + av;
        if (k < k_last) fetch(k + 1);
        if (bnd) {
            L::scatter_at(a.xs, (a.slotN + k + 1) * NX, [](int l) { return l; }, xq, act);
            L::scatter_at(a.xs, (a.slotN + k + 1) * NX, [](int l) { return l + NP; }, xv, act);
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
PDDP_HD void arm_lg_rollout_segment(const ArmLgConst<L>& c, const Dims& dm, const FpLgArgs<T>& a, int bInd, const CostWeights<T>& cw,
                                    T* cost_k, bool init_rollout) {
    using V = typename L::V;
    constexpr int NX = 14, NU = 7, NP = 7;
    const typename L::M act = L::all_true(), last_lane = L::lane_is(6);
    const int NBk = dm.NB, kStart = bInd * NBk;
    const int iters = (bInd < dm.M - 1) ? NBk : NBk - 1;
    const V gq = L::gather_at(a.xGoal, a.oxg, [](int l) { return l; }), gv = L::gather_at(a.xGoal, a.oxg, [](int l) { return l + NP; });
    const bool from_cur = (bInd == 0 || init_rollout);
    const T* xstart = from_cur ? a.xb : a.xs;
    const unsigned ostart = (from_cur ? a.oxc : a.slotN * NX) + NX * kStart;
    V q = L::gather_at(xstart, ostart, [](int l) { return l; }), qd = L::gather_at(xstart, ostart, [](int l) { return l + NP; });
    if (from_cur) {
        L::scatter_at(a.xs, (a.slotN + kStart) * NX, [](int l) { return l; }, q, act);
        L::scatter_at(a.xs, (a.slotN + kStart) * NX, [](int l) { return l + NP; }, qd, act);
    }
    ArmLgState<L> st;
    // operands of one step that do not depend on the state (reference point, gain row, feed-forward): fetched one step ahead so that
    // their global-memory latency hides behind the previous step's dynamics (a wave of this kernel has its SIMD to itself)
    V nxq, nxv, nK[NX], nuc, ndu;
    auto fetch = [&](int kn) {
        const unsigned oKT = (a.pbN + kn) * (NX * NU), oU = (a.pbN + kn) * NU;       // lane r: row r of K = KT[c + r*NX]
        nxq = L::gather_at(a.xb, a.oxc, [kn](int l) { return 14 * kn + l; });
        nxv = L::gather_at(a.xb, a.oxc, [kn](int l) { return 14 * kn + l + 7; });
#pragma unroll
        for (int cc = 0; cc < NX; cc++) nK[cc] = L::gather_at(a.KT, oKT, [cc](int l) { return cc + l * 14; });
        nuc = L::gather_at(a.ucur, oU, [](int l) { return l; });
        ndu = L::gather_at(a.du, oU, [](int l) { return l; });
    };
    fetch(kStart);
    for (int k = 0; k < iters; k++) {
        const int kn = kStart + k;
        const V dq = q - nxq, dv = qd - nxv;
        V bc[14];
        lg_bcast14<L>(bc, dq, dv);
        V Kdx = nK[0] * bc[0];
#pragma unroll
        for (int cc = 1; cc < NX; cc++) Kdx = Kdx + nK[cc] * bc[cc];
        V u = nuc;
        u = u - (V(a.alpha) * ndu + Kdx);
        fetch(kn + 1);                                       // kn + 1 <= N - 1: always inside the arrays
        L::scatter_at(a.us, (a.slotN + kn) * NU, [](int l) { return l; }, u, act);
        if (cost_k) {
            const V J = arm_lg_cost<L, T>(cw, q, qd, u, gq, gv, false);
            L::scatter(cost_k, [kn](int) { return kn; }, J, last_lane);
        }
        const V qdd = arm_lg_dynamics<L, true>(c, st, q, qd, u);
        const V qn = q + V(a.dt) * qd, qdn = qd + V(a.dt) * qdd;       // Euler (utils/integrators.cuh:24-36)
        if (k < NBk - 1) {
            L::scatter_at(a.xs, (a.slotN + kn + 1) * NX, [](int l) { return l; }, qn, act);
            L::scatter_at(a.xs, (a.slotN + kn + 1) * NX, [](int l) { return l + NP; }, qdn, act);
            q = qn; qd = qdn;
        } else if (bInd < dm.M - 1) {                        // defect against the next segment's start state
            const int ks = (bInd + 1) * NBk;
            const T* xnext = init_rollout ? a.xb : a.xs;
            const unsigned onext = (init_rollout ? a.oxc : a.slotN * NX) + NX * ks;
            const V eq = qn - L::gather_at(xnext, onext, [](int l) { return l; });
            const V ev = qdn - L::gather_at(xnext, onext, [](int l) { return l + NP; });
            L::scatter_at(a.ds, (a.slotN + ks - 1) * NX, [](int l) { return l; }, eq, act);
            L::scatter_at(a.ds, (a.slotN + ks - 1) * NX, [](int l) { return l + NP; }, ev, act);
            V sdef = lg_chain_sum<L>(V(T(0)), L::vabs(eq));
            sdef = lg_chain_sum<L>(L::template bcast<6>(sdef), L::vabs(ev));
            L::scatter(a.dnorm, [bInd](int) { return bInd; }, sdef, last_lane);
        }
    }
    if (bInd == dm.M - 1) {                                 // terminal knot: its (unused) control is carried along
        const int kn = dm.N - 1;
        const V u = L::gather_at(a.ucur, (a.pbN + kn) * NU, [](int l) { return l; });
        L::scatter_at(a.us, (a.slotN + kn) * NU, [](int l) { return l; }, u, act);
        if (cost_k) {
            const V J = arm_lg_cost<L, T>(cw, q, qd, u, gq, gv, true);
            L::scatter(cost_k, [kn](int) { return kn; }, J, last_lane);
        }
        L::scatter(a.dnorm, [bInd](int) { return bInd; }, V(T(0)), last_lane);
    }
}

// The same rollout with the end-effector cost (forwardSimInner / forwardSimKern with EE_COST, fpHelpers.cuh:225-275, 279-301; fp.hpp's
// `ee` branch): every segment runs NB steps -- the last step of the last segment only for the tool point of the final state --, the
// cost is accumulated on the way (lane l = s_cost[l]) and the segment's sum s_cost[0] + ... + s_cost[6] goes to segJ[bInd].
template <typename L, typename T>
PDDP_HD void arm_lg_rollout_segment_ee(const ArmLgConst<L>& c, const Dims& dm, const FpLgArgs<T>& a, int bInd, const CostWeights<T>& cw,
                                       T* segJ, bool init_rollout) {
    using V = typename L::V;
    constexpr int NX = 14, NU = 7, NP = 7;
    const typename L::M act = L::all_true(), last_lane = L::lane_is(6), first_lane = L::lane_is(0);
    const int NBk = dm.NB, kStart = bInd * NBk, N = dm.N;
    V goal[6];
#pragma unroll
    for (int i = 0; i < 6; i++) goal[i] = L::gather_at(a.xGoal, a.oxg, [i](int) { return i; });
    const V tq = L::gather_at(a.xTarget, a.oxg, [](int l) { return l; }), tv = L::gather_at(a.xTarget, a.oxg, [](int l) { return l + NP; });
    const bool from_cur = (bInd == 0 || init_rollout);
    const T* xstart = from_cur ? a.xb : a.xs;
    const unsigned ostart = (from_cur ? a.oxc : a.slotN * NX) + NX * kStart;
    V q = L::gather_at(xstart, ostart, [](int l) { return l; }), qd = L::gather_at(xstart, ostart, [](int l) { return l + NP; });
    if (from_cur) {
        L::scatter_at(a.xs, (a.slotN + kStart) * NX, [](int l) { return l; }, q, act);
        L::scatter_at(a.xs, (a.slotN + kStart) * NX, [](int l) { return l + NP; }, qd, act);
    }
    ArmLgState<L> st;
    V nxq, nxv, nK[NX], nuc, ndu;
    auto fetch = [&](int kn) {
        const unsigned oKT = (a.pbN + kn) * (NX * NU), oU = (a.pbN + kn) * NU;
        nxq = L::gather_at(a.xb, a.oxc, [kn](int l) { return 14 * kn + l; });
        nxv = L::gather_at(a.xb, a.oxc, [kn](int l) { return 14 * kn + l + 7; });
#pragma unroll
        for (int cc = 0; cc < NX; cc++) nK[cc] = L::gather_at(a.KT, oKT, [cc](int l) { return cc + l * 14; });
        nuc = L::gather_at(a.ucur, oU, [](int l) { return l; });
        ndu = L::gather_at(a.du, oU, [](int l) { return l; });
    };
    fetch(kStart);
    V acc = V(T(0));
    for (int k = 0; k < NBk; k++) {
        const int kn = kStart + k;
        const V dq = q - nxq, dv = qd - nxv;
        V bc[14];
        lg_bcast14<L>(bc, dq, dv);
        V Kdx = nK[0] * bc[0];
#pragma unroll
        for (int cc = 1; cc < NX; cc++) Kdx = Kdx + nK[cc] * bc[cc];
        V u = nuc;
        u = u - (V(a.alpha) * ndu + Kdx);
        if (kn + 1 < N) fetch(kn + 1);
        L::scatter_at(a.us, (a.slotN + kn) * NU, [](int l) { return l; }, u, act);
        const bool costed = (k < NBk - 1 || bInd == dm.M - 1);
        const bool fin_ee = kn >= N - 1 - a.tshift;
        V eeJ = V(T(0));
        const V qdd = arm_lg_dynamics<L, true>(c, st, q, qd, u, [&](const V* Tw) {
            if (costed) { V pos[6]; lg_tool_point<L, T>(cw, Tw, pos); eeJ = lg_ee_term<L, T>(cw, pos, goal, fin_ee); }   // meaningful in lane 6
        });
        if (costed) {
            V cost = L::sel(first_lane, L::template bcast<6>(eeJ), V(T(0)));
            acc = acc + lg_ee_joint_terms<L, T>(cw, q, qd, u, tq, tv, kn == N - 1, cost);
        }
        const V qn = q + V(a.dt) * qd, qdn = qd + V(a.dt) * qdd;       // Euler (utils/integrators.cuh:24-36)
        if (k < NBk - 1) {
            L::scatter_at(a.xs, (a.slotN + kn + 1) * NX, [](int l) { return l; }, qn, act);
            L::scatter_at(a.xs, (a.slotN + kn + 1) * NX, [](int l) { return l + NP; }, qdn, act);
            q = qn; qd = qdn;
        } else if (bInd < dm.M - 1) {                        // defect against the next segment's start state
            const int ks = (bInd + 1) * NBk;
            const T* xnext = init_rollout ? a.xb : a.xs;
            const unsigned onext = (init_rollout ? a.oxc : a.slotN * NX) + NX * ks;
            const V eq = qn - L::gather_at(xnext, onext, [](int l) { return l; });
            const V ev = qdn - L::gather_at(xnext, onext, [](int l) { return l + NP; });
            L::scatter_at(a.ds, (a.slotN + ks - 1) * NX, [](int l) { return l; }, eq, act);
            L::scatter_at(a.ds, (a.slotN + ks - 1) * NX, [](int l) { return l + NP; }, ev, act);
            V sdef = lg_chain_sum<L>(V(T(0)), L::vabs(eq));
            sdef = lg_chain_sum<L>(L::template bcast<6>(sdef), L::vabs(ev));
            L::scatter(a.dnorm, [bInd](int) { return bInd; }, sdef, last_lane);
        }
    }
    if (bInd == dm.M - 1) L::scatter(a.dnorm, [bInd](int) { return bInd; }, V(T(0)), last_lane);
    const V Jseg = lg_chain_sum<L>(V(T(0)), acc);             // s_cost[0] + s_cost[1] + ... + s_cost[6], complete in lane 6
    L::scatter(segJ, [bInd](int) { return bInd; }, Jseg, last_lane);
}

}  // namespace pddp
