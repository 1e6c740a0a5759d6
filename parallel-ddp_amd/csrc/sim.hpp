// Lock-step plant simulator of the MPC experiment: simulateForward<T, SUBSTEPS> (examples/WAFR_MPC_examples.cu:111-139) driven by the
// trajectory runner's control law getHardwareControls (DDPHelpers/MPCHelpers.cuh:819-858), one wavefront.
//
//   for each of SUBSTEPS sub-steps of elapsed/SUBSTEPS microseconds:
//       tracking error  |tool point(x) - goal|  is accumulated                                   (evNorm, utils/exampleUtils.cuh:84-93)
//       u = u_k - K_k (x - ((1-f) x_k + f x_{k+1})),  k = floor((t - t0)/TIME_STEP), f the fraction: zero-order hold on u and K,
//           first-order hold on the nominal state, in the plan's precision T                      (USE_FEEDBACK_IN_TRAJ_RUNNER 1)
//       x <- integrator<double>(x, u, dt)                                                        (the plant itself runs in double)
//   the error of the final state is added as well and the sum divided by SUBSTEPS (sic: SUBSTEPS + 1 terms, :137-138).
// A time outside the plan (k >= N - 2 or k < 0) aborts: the state is left untouched and the error is 0 (:129-130).
// Difference: the reference evaluates the tool point in T from the state rounded to T; here it comes from the double-precision
// kinematics the dynamics compute anyway, rounded once.
#pragma once

#include "ee_cost.hpp"
#include "integrators.hpp"

namespace pddp {

template <typename PD, typename T>
struct PlantSimScratch {
    typename PD::Scratch plant;
    IntegScratch<PD, double> integ;
    double x[PD::NX], xn[PD::NX], u[PD::NU], qdd[PD::NPOS];
    T dx[PD::NX];
    EeScratch<double> ee;
    double err;
    int fail;
};

template <typename T>
struct PlantSimArgs {
    const T *x, *u, *KT;          // the plan: [N][n], [N][m], [N][n*m] (KT[k][c + r*n] = K(r, c))
    int N;
    double step_us;               // TIME_STEP_LENGTH_IN_us
    double t0_us, elapsed_us;
    int substeps;
    const T* goal;                // tool-point goal xyz (arm), or nullptr: no error metric
    double ee_z;
    T* xActual;                   // in: measured state at t0 + 0; out: state after elapsed_us
    double* out;                  // [2]: average tracking error, failed flag
};

template <typename PD, int INTEG, typename T>
PDDP_HD void plant_sim_body(const Wave& w, PlantSimScratch<PD, T>& s, const void* model, const PlantSimArgs<T>& a) {
    constexpr int NX = PD::NX, NU = PD::NU, NP = PD::NPOS;
    PD::load_model(w, s.plant, reinterpret_cast<const typename PD::Model*>(model));
    PDDP_FOR(i, NX) s.x[i] = (double)a.xActual[i];
    if (w.lane == 0) { s.err = 0; s.fail = 0; }
    wsync(w);
    const double dt_us = a.elapsed_us / (double)a.substeps, dt = dt_us / 1000000.0;
    double tk = a.t0_us;
    CostWeights<double> cwz{}; cwz.ee_z = a.ee_z;
    for (int i = 0; i <= a.substeps; i++) {
        const bool last = (i == a.substeps);
        // control of this sub-step (not needed after the last one)
        const double steps = (tk - a.t0_us) / a.step_us;
        const int k = (int)steps;
        const double frac = steps - (double)k;
        if (!last) {
            if (k >= a.N - 2 || k < 0) { if (w.lane == 0) s.fail = 1; wsync(w); break; }
            PDDP_FOR(ind, NX) {
                const T val = (T)(1.0 - frac) * a.x[(size_t)k * NX + ind] + (T)frac * a.x[(size_t)(k + 1) * NX + ind];
                s.dx[ind] = (T)s.x[ind] - val;
            }
            wsync(w);
            PDDP_FOR(r, NU) {
                T val = a.u[(size_t)k * NU + r];
                for (int c = 0; c < NX; c++) val -= a.KT[(size_t)k * NX * NU + c + r * NX] * s.dx[c];
                s.u[r] = (double)val;
            }
        } else {
            PDDP_FOR(r, NU) s.u[r] = 0.0;                // the final state only needs its kinematics
        }
        wsync(w);
        if (!last) integrator_step<PD, INTEG, double>(w, s.plant, s.integ, s.xn, s.x, s.u, dt);
        else PD::dynamics(w, s.plant, s.qdd, s.x, s.u);
        wsync(w);
        if constexpr (PD::PLANT == 4) {
            if (a.goal) {                                  // evNorm of the state the step STARTED from (its frames are in the scratch)
                ee_position<double>(w, s.plant, cwz, s.ee);
                if (w.lane == 0) {
                    T e2 = 0;
                    for (int c = 0; c < 3; c++) { const T dl = (T)s.ee.pos[c] - a.goal[c]; e2 += dl * dl; }
                    s.err += (double)(T)tsqrt<T>(e2);
                }
                wsync(w);
            }
        }
        if (!last) { PDDP_FOR(e, NX) s.x[e] = s.xn[e]; tk += dt_us; }
        wsync(w);
    }
    if (w.lane == 0) { a.out[0] = s.fail ? 0.0 : s.err / (double)a.substeps; a.out[1] = (double)s.fail; }
    if (!s.fail) { PDDP_FOR(e, NX) a.xActual[e] = (T)s.x[e]; }
    (void)NP;
}

// compute_eePos_scratch (plants/dynamics_arm.cuh:1953-1960): tool point (x, y, z, roll, pitch, yaw) of one state, arm only
template <typename P, typename T>
PDDP_HD void ee_pos_body(const Wave& w, typename P::Scratch& plant, EeScratch<T>& ee, T* xs, T* us, T* qdd, const void* model, T ee_z, const T* x, T* out) {
    P::load_model(w, plant, reinterpret_cast<const typename P::Model*>(model));
    PDDP_FOR(i, P::NX) xs[i] = x[i];
    PDDP_FOR(i, P::NU) us[i] = 0;
    wsync(w);
    P::dynamics(w, plant, qdd, xs, us);
    wsync(w);
    if constexpr (P::PLANT == 4) {
        CostWeights<T> cwz{}; cwz.ee_z = ee_z;
        ee_position<T>(w, plant, cwz, ee);
        PDDP_FOR(i, 6) out[i] = ee.pos[i];
    }
}

}  // namespace pddp
