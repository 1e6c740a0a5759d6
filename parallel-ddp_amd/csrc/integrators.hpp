// Discrete-time step x+ = f_d(x,u) and its Jacobian AB = [A B] for Euler / midpoint / RK3, wave-cooperative.
// Replaces _integrator / _integratorGradient (utils/integrators.cuh:24-53 Euler, :56-120 midpoint, :123-233 RK3),
// including the two behaviours of the reference that change results: the midpoint rule's final update uses the START
// velocity for the position rows (:78), and the RK3 Jacobian builds its stage states from positions where the
// integrator uses velocities (:182,:190-191).
#pragma once

#include "plants.hpp"

namespace pddp {

template <typename P, typename T>
struct IntegScratch {                 // per-wave LDS
    T qdd[P::NPOS], qdd2[P::NPOS], qdd3[P::NPOS];
    T x2[P::NX], x3[P::NX];
};

template <typename P, int INTEG, typename T>
struct IntegGradScratch {             // per-wave LDS; stage arrays only exist for the integrators that use them
    static constexpr int ND = P::NPOS * (P::NX + P::NU), NAB = P::NX * (P::NX + P::NU);
    T qdd1[P::NPOS], qdd2[P::NPOS], qdd3[P::NPOS];
    T d1[ND], d2[INTEG >= 2 ? ND : 1], d3[INTEG == 3 ? ND : 1];
    T xm1[INTEG >= 2 ? P::NX : 1], xm2[INTEG == 3 ? P::NX : 1];
    T T1[INTEG == 3 ? NAB : 1], T2[INTEG == 3 ? NAB : 1];
};

// d(xdot)/d(x,u) entry (r,c) from dqdd:  x = [q; qd]  ->  [0 I 0; dqdd]     (integrators.cuh:17)
template <int NPOS, typename T>
PDDP_HD T dxd(const T* dqdd, int r, int c) {
    return r < NPOS ? T(r + NPOS == c ? 1 : 0) : dqdd[c * NPOS + (r - NPOS)];
}

// USE_FINITE_DIFF (config.cuh:68; finiteDiffInner, DDPHelpers/nisInitHelpers.cuh:138-166): [A B] of the EULER step column by column from two evaluations of
// the plant's dynamics at x, u -+ eps e_col -- position rows are the exact constants of dqddk2dxd (utils/integrators.cuh:21), velocity rows
// (col == row) + dt (qdd+ - qdd-) / (2 eps) with the quotient taken in double like the reference's `delta / (2.0*FINITE_DIFF_EPSILON)`.
template <typename P, typename T>
struct FdScratch { T xp[P::NX], xm[P::NX], up[P::NU], um[P::NU], qp[P::NPOS], qm[P::NPOS]; };
template <typename P, typename T>
PDDP_HD void integrator_gradient_fd(const Wave& w, typename P::Scratch& ps, FdScratch<P, T>& f, T* AB, const T* x, const T* u, T dt, double eps) {
    constexpr int NP = P::NPOS, NX = P::NX, NU = P::NU, NM = NX + NU;
    for (int col = 0; col < NM; col++) {
        PDDP_FOR(i, NX) { const T adj = T(col == i ? eps : 0.0); f.xp[i] = x[i] + adj; f.xm[i] = x[i] - adj; }
        PDDP_FOR(i, NU) { const T adj = T(col == i + NX ? eps : 0.0); f.up[i] = u[i] + adj; f.um[i] = u[i] - adj; }
        wsync(w);
        P::dynamics(w, ps, f.qp, f.xp, f.up);
        wsync(w);
        P::dynamics(w, ps, f.qm, f.xm, f.um);
        wsync(w);
        PDDP_FOR(i, NX) {
            const T delta = i < NP ? (f.xp[i + NP] - f.xm[i + NP]) : (f.qp[i - NP] - f.qm[i - NP]);
            const T dxdd = T(double(delta) / (2.0 * eps));
            const T v = i < NP ? T(i + NP == col ? 1 : 0) : dxdd;
            AB[col * NX + i] = T((i == col ? 1.0 : 0.0) + double(dt * v));
        }
        wsync(w);
    }
}

// xkp1, x, u in LDS (or host memory).  xkp1 must not alias x.
template <typename P, int INTEG, typename T>
PDDP_HD void integrator_step(const Wave& w, typename P::Scratch& ps, IntegScratch<P, T>& s, T* xkp1, const T* x, const T* u, T dt) {
    constexpr int NP = P::NPOS;
    P::dynamics(w, ps, s.qdd, x, u);
    if constexpr (INTEG == 1) {
        PDDP_FOR(i, NP) { xkp1[i] = x[i] + dt * x[i + NP]; xkp1[i + NP] = x[i + NP] + dt * s.qdd[i]; }
        wsync();
    } else if constexpr (INTEG == 2) {
        PDDP_FOR(i, NP) { s.x2[i] = x[i] + T(0.5) * dt * x[i + NP]; s.x2[i + NP] = x[i + NP] + T(0.5) * dt * s.qdd[i]; }
        wsync();
        P::dynamics(w, ps, s.qdd, s.x2, u);
        PDDP_FOR(i, NP) { xkp1[i] = x[i] + dt * x[i + NP]; xkp1[i + NP] = x[i + NP] + dt * s.qdd[i]; }
        wsync();
    } else {
        PDDP_FOR(i, NP) { s.x2[i] = x[i] + T(0.5) * dt * x[i + NP]; s.x2[i + NP] = x[i + NP] + T(0.5) * dt * s.qdd[i]; }
        wsync();
        P::dynamics(w, ps, s.qdd2, s.x2, u);
        PDDP_FOR(i, NP) {
            s.x3[i] = x[i] + dt * (T(2) * s.x2[i + NP] - x[i + NP]);
            s.x3[i + NP] = x[i + NP] + dt * (T(2) * s.qdd2[i] - s.qdd[i]);
        }
        wsync();
        P::dynamics(w, ps, s.qdd3, s.x3, u);
        PDDP_FOR(i, NP) {
            xkp1[i] = x[i] + (dt / T(6)) * (x[i + NP] + T(4) * s.x2[i + NP] + s.x3[i + NP]);
            xkp1[i + NP] = x[i + NP] + (dt / T(6)) * (s.qdd[i] + T(4) * s.qdd2[i] + s.qdd3[i]);
        }
        wsync();
    }
}

// ---- pieces of the RK3 Jacobian for scalar plug-ins (closed-form plants), split so that a caller can batch the scalar parts over MANY knots (k_nis_gl2, kernels.hpp):
//   rk3_stage_chain     one thread: qdd1 = f(x), xm1, qdd2 = f(xm1), xm2   (the reference builds its stage states from positions where the integrator uses velocities: :182,:190-191)
//   rk3_stage_gradient  one thread: the plug-in's gradient at stage 0 / 1 / 2  -> d1 / d2 / d3 (and the stage's qdd again: the same numbers)
//   rk3_assemble        the cooperating set: T1, T2, [A B] from d1, d2, d3
template <typename P, typename T>
PDDP_HD void rk3_stage_chain(IntegGradScratch<P, 3, T>& s, const T* x, const T* u, T dt) {
    constexpr int NP = P::NPOS;
    P::dynamics_eval(s.qdd1, x, u);
    for (int i = 0; i < NP; i++) { s.xm1[i] = x[i] + T(0.5) * dt * x[i + NP]; s.xm1[i + NP] = x[i] + T(0.5) * dt * s.qdd1[i]; }
    P::dynamics_eval(s.qdd2, s.xm1, u);
    for (int i = 0; i < NP; i++) {
        s.xm2[i] = x[i] + dt * x[i + NP] + T(2) * dt * s.xm1[i + NP];
        s.xm2[i + NP] = x[i] + dt * s.qdd1[i] + T(2) * dt * s.qdd2[i];
    }
}
template <typename P, typename T>
PDDP_HD void rk3_stage_gradient(IntegGradScratch<P, 3, T>& s, const T* x, const T* u, int stage) {
    T* dd = stage == 0 ? s.d1 : (stage == 1 ? s.d2 : s.d3);
    T* qq = stage == 0 ? s.qdd1 : (stage == 1 ? s.qdd2 : s.qdd3);
    const T* xx = stage == 0 ? x : (stage == 1 ? s.xm1 : s.xm2);
    P::gradient_eval(dd, qq, xx, u);
}
template <typename P, typename T>
PDDP_HD void rk3_assemble(const Wave& w, IntegGradScratch<P, 3, T>& s, T* ABk, T dt) {
    constexpr int NP = P::NPOS, NX = P::NX, NM = P::NX + P::NU;
        // T1 = X2 (0.5 dt X1 + I) (+ X2's control columns), T2 = X3 (2 dt T1 - dt X1 + I) (+ X3's control columns) with X = d(xdot)/d(x,u) = [0 I 0; dqdd]
        // (:196-224).  The reference sums 12 products per entry; the position rows of every X hold a single 1 and the position columns of dqdd's left factor
        // pick single entries, so most of those products are exact zeros.  Only the non-zero terms are evaluated here, in the reference's index order and with its
        // operations (a zero term adds +-0: the sums are the same numbers) -- ~8 instead of 12 x 2 table look-ups per entry.
        PDDP_FOR(e, NX * NM) {
            const int ky = e / NX, kx = e % NX;
            T val = 0;
            if (kx < NP) {                                          // row kx of X2 = e_{kx+NP}': the single term i = kx + NP
                val += T(1) * (T(0.5) * dt * s.d1[ky * NP + kx] + T(ky == kx + NP ? 1 : 0));
                s.T1[e] = val + T(0);
            } else {
                const int r = kx - NP;
                if (ky < NP) val += s.d2[ky * NP + r] * (T(0.5) * dt * T(0) + T(1));                       // i = ky < NP: X1(i, ky) = 0, identity 1
                else if (ky < NX) val += s.d2[(ky - NP) * NP + r] * (T(0.5) * dt * T(1) + T(0));          // i = ky - NP: X1(i, ky) = 1
                for (int i = NP; i < NX; i++) val += s.d2[i * NP + r] * (T(0.5) * dt * s.d1[ky * NP + (i - NP)] + T(ky == i ? 1 : 0));
                s.T1[e] = val + (ky < NX ? T(0) : s.d2[ky * NP + r]);
            }
        }
        wsync();
        PDDP_FOR(e, NX * NM) {
            const int ky = e / NX, kx = e % NX;
            T val = 0;
            if (kx < NP) {
                const int i = kx + NP;
                val += T(1) * (T(2) * dt * s.T1[ky * NX + i] - dt * s.d1[ky * NP + kx] + T(ky == i ? 1 : 0));
                s.T2[e] = val + T(0);
            } else {
                const int r = kx - NP;
                for (int i = 0; i < NP; i++) val += s.d3[i * NP + r] * (T(2) * dt * s.T1[ky * NX + i] - dt * T(i + NP == ky ? 1 : 0) + T(ky == i ? 1 : 0));
                for (int i = NP; i < NX; i++) val += s.d3[i * NP + r] * (T(2) * dt * s.T1[ky * NX + i] - dt * s.d1[ky * NP + (i - NP)] + T(ky == i ? 1 : 0));
                s.T2[e] = val + (ky < NX ? T(0) : s.d3[ky * NP + r]);
            }
        }
        wsync();
        PDDP_FOR(e, NX * NM) {
            const int ky = e / NX, kx = e % NX;
            ABk[e] = (dt / T(6)) * dxd<NP>(s.d1, kx, ky) + (T(2) * dt / T(3)) * s.T1[e] + (dt / T(6)) * s.T2[e] + T(kx == ky ? 1 : 0);
        }
        wsync();
}

// ABk: NX x (NX+NU) column-major with leading dimension NX, written to `ABk` (global or LDS).
template <typename P, int INTEG, typename T>
PDDP_HD void integrator_gradient(const Wave& w, typename P::Scratch& ps, typename P::GradScratch& pg, IntegGradScratch<P, INTEG, T>& s,
                                 T* ABk, const T* x, const T* u, T dt) {
    constexpr int NP = P::NPOS, NX = P::NX, NM = P::NX + P::NU;
    bool staged = false;                                            // RK3 with a scalar plug-in and >= 3 lanes: see below
    if constexpr (INTEG == 3 && P::kScalarPlugin) staged = w.nlanes >= 3;
    if (!staged) P::gradient(w, ps, pg, s.d1, s.qdd1, x, u);
    if constexpr (INTEG == 1) {
        PDDP_FOR(e, NX * NM) {
            const int ky = e / NX, kx = e % NX;
            ABk[e] = T(ky == kx ? 1 : 0) + dt * dxd<NP>(s.d1, kx, ky);
        }
        wsync();
    } else if constexpr (INTEG == 2) {
        PDDP_FOR(i, NP) { s.xm1[i] = x[i] + T(0.5) * dt * x[i + NP]; s.xm1[i + NP] = x[i + NP] + T(0.5) * dt * s.qdd1[i]; }
        wsync();
        P::gradient(w, ps, pg, s.d2, s.qdd2, s.xm1, u);
        PDDP_FOR(e, NX * NM) {
            const int ky = e / NX, kx = e % NX;
            T val = 0;
            for (int i = 0; i < NX; i++) {
                const T A2 = T(kx == i ? 1 : 0) + T(0.5) * dt * dxd<NP>(s.d2, kx, i);
                const T AB1 = T(ky == i ? 1 : 0) + T(0.5) * dt * dxd<NP>(s.d1, i, ky);
                val += A2 * AB1;
            }
            ABk[e] = val + (ky < NX ? T(0) : T(0.5) * dt * dxd<NP>(s.d2, kx, ky));
        }
        wsync();
    } else {
        if constexpr (P::kScalarPlugin) {
            // a scalar plug-in with at least three lanes in the set: the stage STATES need only the dynamics of the earlier stages (one lane, in turn), then the
            // three stage gradients are independent scalar evaluations -- one lane each, side by side (the same functions on the same operands as below)
            if (staged) {
                if (w.lane == 0) rk3_stage_chain<P, T>(s, x, u, dt);
                wsync();
                if (w.lane < 3) rk3_stage_gradient<P, T>(s, x, u, w.lane);      // ONE pass of the plug-in's code with three lanes active (three guarded calls would run in turn)
                wsync();
            }
        }
        if (!staged) {
        PDDP_FOR(i, NP) { s.xm1[i] = x[i] + T(0.5) * dt * x[i + NP]; s.xm1[i + NP] = x[i] + T(0.5) * dt * s.qdd1[i]; }
        wsync();
        P::gradient(w, ps, pg, s.d2, s.qdd2, s.xm1, u);
        PDDP_FOR(i, NP) {
            s.xm2[i] = x[i] + dt * x[i + NP] + T(2) * dt * s.xm1[i + NP];
            s.xm2[i + NP] = x[i] + dt * s.qdd1[i] + T(2) * dt * s.qdd2[i];
        }
        wsync();
        P::gradient(w, ps, pg, s.d3, s.qdd3, s.xm2, u);
        }
        rk3_assemble<P, T>(w, s, ABk, dt);
    }
}

}  // namespace pddp
