// Forward pass for one line-search candidate alpha: linear sweep across shooting-segment boundaries, nonlinear
// rollout of every segment, running/terminal cost and defect norm.
//
// Replaces forwardSweepKern/forwardSweepInner (DDPHelpers/fpHelpers.cuh:57-63 / :19-53), forwardSimKern /
// forwardSimInner / computeControlKT (:279-301 / :225-275 / :202-221), costKern + reduceSum (:134-152,
// utils/cudaUtils.h:187-207) and defectKern + reduceMax (:96-111, cudaUtils.h:160-182).
//
// Differences in data movement (results are unchanged):
//  * every candidate starts from the CURRENT trajectory (xcur,ucur,dcur) instead of from a private copy that
//    memcpyCurrAKern (nisInitHelpers.cuh:24-32) had to refresh each iteration -- candidate slots are pure outputs;
//  * the sweep only has to deliver the M-1 segment start states (everything else it writes is overwritten by the
//    rollout), so it keeps its running state in LDS and stores nothing but those;
//  * the rollout keeps x_k,u_k in LDS, so cost and defect are reduced from LDS in the same launch.
#pragma once

#include "ee_cost.hpp"
#include "integrators.hpp"

namespace pddp {

template <typename P, typename T>
struct SweepScratch {
    T A[P::NX * P::NX];
    T xk[P::NX], dx[P::NX];
};

template <typename P, typename T>
struct SimScratch {
    typename P::Scratch plant;
    IntegScratch<P, T> integ;
    T x[P::NX], xn[P::NX], u[P::NU], dx[P::NX];
    EeScratch<T> ee;
};

template <typename T>
struct FpArgs {
    T* x; T* u; T* d;                         // this candidate's output trajectory [N][n], [N][m], defects [N][n]
    const T* xcur; const T* ucur; const T* dcur;   // current trajectory (reference: d_xp, d_up, winner's d)
    const T* KT; const T* du; const T* ApBK; const T* Bdu;
    T alpha; T dt;
    T* segx;                                  // LDS [M][NX]: segment start states handed from the sweep to the rollouts
    T* dnorm;                                 // LDS [M]: 1-norm of each segment's boundary defect (0 for the last segment)
    // end-effector cost only: goal is the 6-vector, xt the nominal-state target, segJ (LDS [M]) the segment's accumulated cost
    const T* xt; T* segJ; int tshift;
};

// Linear sweep (M > 1): x_{k+1} = xcur_{k+1} + (A-BK)_k (x_k - xcur_k) - alpha (B du)_k + [boundary] d_k, serial in k.
// Stores the segment start states x[b*NB] (b = 1..M-1) into a.x and into the LDS hand-off array a.segx.
template <typename P, typename T>
PDDP_HD void forward_sweep(const Wave& w, SweepScratch<P, T>& s, const Dims& dm, const FpArgs<T>& a) {
    constexpr int NX = P::NX;
    PDDP_FOR(i, NX) s.xk[i] = a.xcur[i];
    wsync();
    for (int k = 0; k < dm.N - 1; k++) {
        const T* Ak = a.ApBK + NX * NX * k;
        PDDP_FOR(e, NX * NX) s.A[e] = Ak[e];
        PDDP_FOR(i, NX) s.dx[i] = s.xk[i] - a.xcur[NX * k + i];
        wsync();
        const bool bnd = dm.on_defect_boundary(k);
        PDDP_FOR(r, NX) {
            T val = 0;
            for (int i = 0; i < NX; i++) val += s.A[r + NX * i] * s.dx[i];
            T xn = a.xcur[NX * (k + 1) + r];
            xn += -a.alpha * a.Bdu[NX * k + r] + val + (bnd ? a.dcur[NX * k + r] : T(0));
            s.xk[r] = xn;
            if (bnd) { a.x[NX * (k + 1) + r] = xn; a.segx[NX * ((k + 1) / dm.NB) + r] = xn; }
        }
        wsync();
    }
}

// Initial rollout (forwardRolloutFlag, nisInitHelpers.cuh:642-648): no sweep, segment bInd starts from the loaded state.
template <typename P, typename T>
PDDP_HD void rollout_seed_segment(const Wave& w, const Dims& dm, const FpArgs<T>& a, int bInd) {
    constexpr int NX = P::NX;
    PDDP_FOR(i, NX) { const T v = a.xcur[NX * bInd * dm.NB + i]; a.segx[NX * bInd + i] = v; a.x[NX * bInd * dm.NB + i] = v; }
}

// PDDP_PHASE_ROLLOUT (teacher forcing of forwardSimKern alone): no sweep -- segment bInd starts from what the candidate's own x holds at its first knot
template <typename P, typename T>
PDDP_HD void rollout_seed_from_candidate(const Wave& w, const Dims& dm, const FpArgs<T>& a, int bInd) {
    constexpr int NX = P::NX;
    PDDP_FOR(i, NX) a.segx[NX * bInd + i] = a.x[NX * bInd * dm.NB + i];
}

// Nonlinear rollout of segment bInd.  Start state: a.segx[bInd] (from the sweep), or xcur[0] for segment 0.
// Writes x[k+1], u[k] for the segment's knots and the boundary defect.  cost_k (optional, LDS [N]) receives the
// per-knot cost of every knot this segment owns.
template <typename P, int INTEG, typename T>
PDDP_HD void forward_sim_segment(const Wave& w, SimScratch<P, T>& s, const Dims& dm, const FpArgs<T>& a, int bInd,
                                 const CostWeights<T>& cw, const T* xg, T* cost_k) {
    constexpr int NX = P::NX, NU = P::NU;
    const int NBk = dm.NB, kStart = bInd * NBk;
    bool ee = false;
    if constexpr (P::PLANT == 4) ee = cw.ee != 0;
    // with the end-effector cost the last segment also runs the control law and the dynamics at knot N-1: the tool point of the
    // final state comes out of the dynamics (forwardSimInner, fpHelpers.cuh:236: iters = N_BLOCKS_F)
    const int iters = (bInd < dm.M - 1 || ee) ? NBk : NBk - 1;
    if constexpr (P::PLANT == 4) { if (ee) { PDDP_FOR(i, 7) s.ee.acc[i] = 0; } }
    if (bInd == 0) { PDDP_FOR(i, NX) { const T v = a.xcur[i]; s.x[i] = v; a.x[i] = v; } }
    else { PDDP_FOR(i, NX) s.x[i] = a.segx[NX * bInd + i]; }
    wsync();
    for (int k = 0; k < iters; k++) {
        const int kn = kStart + k;
        PDDP_FOR(i, NX) s.dx[i] = s.x[i] - a.xcur[NX * kn + i];
        wsync();
        PDDP_FOR(r, NU) {                     // u = ucur - alpha du - K (x - xcur)      (computeControlKT)
            const T* KTk = a.KT + NX * NU * kn;
            T Kdx = 0;
            for (int c = 0; c < NX; c++) Kdx += KTk[c + r * NX] * s.dx[c];
            T uv = a.ucur[NU * kn + r];
            uv -= a.alpha * a.du[NU * kn + r] + Kdx;
            s.u[r] = uv; a.u[NU * kn + r] = uv;
        }
        wsync();
        if (!ee && cost_k && w.lane == 0) cost_k[kn] = P::cost(cw, s.x, s.u, xg, kn, dm.N);
        integrator_step<P, INTEG>(w, s.plant, s.integ, s.xn, s.x, s.u, a.dt);
        if constexpr (P::PLANT == 4) {
            if (ee && (k < NBk - 1 || bInd == dm.M - 1)) {    // not on the "final" state of a non-final segment (:259-265)
                ee_position<T>(w, s.plant, cw, s.ee);
                ee_cost_accumulate<T>(w, s.ee, cw, xg, a.xt, s.x, s.u, kn, dm.N, a.tshift);
                wsync(w);
            }
        }
        if (ee && kn == dm.N - 1) break;                     // the step out of the last knot is not stored anywhere
        if (k < NBk - 1) {
            PDDP_FOR(i, NX) { const T v = s.xn[i]; a.x[NX * (kn + 1) + i] = v; s.x[i] = v; }
        } else if (bInd < dm.M - 1) {         // last step of a non-final segment: defect against the next start state
            PDDP_FOR(i, NX) { const T dv = s.xn[i] - a.segx[NX * (bInd + 1) + i]; a.d[NX * ((bInd + 1) * NBk - 1) + i] = dv; s.dx[i] = dv; }
            wsync();
            if (w.lane == 0) { T sdef = 0; for (int c = 0; c < NX; c++) sdef += tabs(s.dx[c]); a.dnorm[bInd] = sdef; }   // defectKern
        }
        wsync();
    }
    if constexpr (P::PLANT == 4) {
        if (ee) {                                 // forwardSimKern, fpHelpers.cuh:298-300
            if (w.lane == 0) {
                a.segJ[bInd] = s.ee.acc[0] + s.ee.acc[1] + s.ee.acc[2] + s.ee.acc[3] + s.ee.acc[4] + s.ee.acc[5] + s.ee.acc[6];
                if (bInd == dm.M - 1) a.dnorm[bInd] = 0;
            }
            wsync(w);
            return;
        }
    }
    if (bInd == dm.M - 1) {                   // final knot: terminal cost, and its (unused) control is carried along
        const int kn = dm.N - 1;
        PDDP_FOR(r, NU) { const T uv = a.ucur[NU * kn + r]; s.u[r] = uv; a.u[NU * kn + r] = uv; }
        wsync();
        if (cost_k && w.lane == 0) { cost_k[kn] = P::cost(cw, s.x, s.u, xg, kn, dm.N); a.dnorm[bInd] = 0; }
        wsync();
    }
}

// Pairwise tree sum over N per-knot values in the reference's pairing (reduceSum with blockDim.x = N):
// s[t] += s[t+h] for h = N/2, N/4, ..., 2 and finally s[0] += s[1].  `vals` is LDS scratch of N entries.
// Called by a single wave (any number of lanes); N must be a power of two >= 4 (as in the reference).
template <typename T>
PDDP_HD T tree_sum(const Wave& w, T* vals, int N) {
    for (int h = N >> 1; h >= 2; h >>= 1) {
        PDDP_FOR(t, h) vals[t] += vals[t + h];
        wsync();
    }
    return vals[0] + vals[1];
}

}  // namespace pddp
