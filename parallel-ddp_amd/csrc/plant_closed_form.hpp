// Closed-form plants: pendulum, cart-pole, quadrotor.  One lane evaluates them (they are a few dozen flops);
// the expressions, constants and literal types follow plants/dynamics_pend.cuh:30-51, dynamics_cart.cuh:30-76 and
// dynamics_quad.cuh:42-169 (double literals promote the arithmetic exactly as in the reference).
// These closed-form polynomials (with their numeric coefficients) cannot be written differently and stay exact, so the oracle carries the
// same expressions; what pins BOTH is tests/golden/closed_form_plants.json -- the reference's own statements executed in float64 on stored
// inputs by tests/golden/make_closed_form_plants.py (tests/test_closed_form_pins.py: dynamics and every gradient entry, 25 states per plant).
#pragma once
#include "pddp_common.hpp"
namespace pddp {

#define CART_M_CART 10
#define CART_M_POLE 1
#define CART_ML (CART_M_POLE * 0.5)
#define CART_MLL (CART_ML * 0.5)
#define CART_G (-9.81)
#define QUAD_G (-9.81)
#define QUAD_MASS 0.5
#define QUAD_INVMASS 2
template <typename T> PDDP_HD void pend_dynamics_eval(T* qdd, const T* x, const T* u) {
    T s0, c0; cf_sincos<T>(x[0], s0, c0);
    qdd[0] = u[0] + -9.81 * s0;
}
template <typename T> PDDP_HD void pend_gradient_eval(T* dqdd, T* qdd, const T* x, const T* u) {
    T s0, c0; cf_sincos<T>(x[0], s0, c0);
    qdd[0] = u[0] + -9.81 * s0;                                    // = pend_dynamics_eval (the plug-in's dynamics(), dynamics_pend.cuh:46)
    dqdd[0] = -9.81 * c0;
    dqdd[1] = 0.0;
    dqdd[2] = 1;
}
template <typename T> PDDP_HD void cart_dynamics_trig(T* qdd, const T* x, const T* u, T ct, T st) {
    T td2 = x[3] * x[3];
    T H0 = CART_M_CART + CART_M_POLE, H1 = CART_MLL, Hod = CART_ML * ct;
    T TauM = CART_ML * st, Tau0 = TauM * td2 + u[0], Tau1 = TauM * CART_G;
    T det = 1 / (H0 * H1 - Hod * Hod);
    qdd[0] = det * (H1 * Tau0 - Hod * Tau1);
    qdd[1] = det * (H0 * Tau1 - Hod * Tau0);
}
template <typename T> PDDP_HD void cart_dynamics_eval(T* qdd, const T* x, const T* u) {
    T ct, st; cf_sincos<T>(x[1], st, ct);
    cart_dynamics_trig<T>(qdd, x, u, ct, st);
}
template <typename T> PDDP_HD void cart_gradient_eval(T* dqdd, T* qdd, const T* x, const T* u) {
    T ct, st; cf_sincos<T>(x[1], st, ct);
    cart_dynamics_trig<T>(qdd, x, u, ct, st);                      // the plug-in's dynamics() inside dynamicsGradient (dynamics_cart.cuh:52): the same numbers as cart_dynamics_eval
    T td = x[3], td2 = td * td;
    T H0 = CART_M_CART + CART_M_POLE, H1 = CART_MLL;
    T Hod = CART_ML * ct, TauM = CART_ML * st;
    T Tau0 = TauM * td2 + u[0], Tau1 = TauM * CART_G;
    T det = H0 * H1 - Hod * Hod, idet = 1 / det;
    T xddM = H1 * Tau0 - Hod * Tau1, thetaddM = H0 * Tau1 - Hod * Tau0;
    T thetadd_du = idet * (-Hod), thetadd_dthetad = idet * (-2 * Hod * TauM * td);
    T xdd_du = idet * (H1), xdd_dthetad = idet * (2 * H1 * TauM * td);
    T Hod_dtheta = -TauM, Tau0_dtheta = Hod * td2, Tau1_dTheta = Hod * CART_G;
    T thetaddM_dtheta = H0 * Tau1_dTheta - (Hod_dtheta * Tau0 + Hod * Tau0_dtheta);
    T xddM_dtheta = H1 * Tau0_dtheta - (Hod_dtheta * Tau1 + Hod * Tau1_dTheta);
    T idet_dtheta = -2 * Hod * TauM * idet * idet;
    T thetadd_dtheta = idet * thetaddM_dtheta + idet_dtheta * thetaddM;
    T xdd_dtheta = idet * xddM_dtheta + idet_dtheta * xddM;
    dqdd[0] = 0; dqdd[1] = 0;
    dqdd[2] = xdd_dtheta; dqdd[3] = thetadd_dtheta;
    dqdd[4] = 0; dqdd[5] = 0;
    dqdd[6] = xdd_dthetad; dqdd[7] = thetadd_dthetad;
    dqdd[8] = xdd_du; dqdd[9] = thetadd_du;
}
template <typename T> PDDP_HD void quad_dynamics_trig(T* qdd, const T* x, const T* u, T sX3, T cX3, T sX4, T cX4, T sX5, T cX5) {
    T X910 = x[9] * x[10], X911 = x[9] * x[11], X1011 = x[10] * x[11], X11_2 = x[11] * x[11];
    T sumU = u[0] + u[1] + u[2] + u[3];
    qdd[0] = QUAD_INVMASS * sumU * (sX3 * sX5 + cX3 * cX5 * sX4);
    qdd[1] = -QUAD_INVMASS * sumU * (cX5 * sX3 - cX3 * sX4 * sX5);
    qdd[2] = QUAD_G + QUAD_INVMASS * sumU * cX3 * cX4;
    T diffU4 = u[0] - u[1] + u[2] - u[3], diffU2 = u[2] - u[0];
    T invcX4 = 1 / cX4, cX3_2 = cX3 * cX3, cX34 = cX3 * cX4, s2X3 = 2.0 * sX3 * cX3, c2X3 = cf_cos2<T>(x[3]);
    qdd[3] = invcX4 * (0.0005434782609 * (32000.0 * X1011 + 140000.0 * (u[1] - u[3]) * cX4 - 28320.0 * X910 * sX4 - 30160.0 * X1011 * cX3_2 + 1127.0 * diffU4 * cX3 * sX4 - 140000.0 * diffU2 * sX3 * sX4 + 30160.0 * X910 * cX3_2 * sX4 - 30160.0 * X1011 * cX34 * cX34 + 30160.0 * x[10] * x[10] * cX3 * cX4 * sX3 - 30160.0 * X11_2 * cX3 * cX4 * sX3 + 30160.0 * X911 * cX3 * cX4 * sX3 * sX4));
    qdd[4] = 76.08695652 * diffU2 * cX3 - 0.6125 * diffU4 * sX3 - 1.0 * X911 * cX4 - 8.195652174 * X910 * s2X3 - 16.39130435 * X11_2 * cX3_2 * cX4 * sX4 + 16.39130435 * X911 * cX3_2 * cX4 + 16.39130435 * X1011 * cX3 * sX3 * sX4;
    qdd[5] = -invcX4 * (0.0005434782609 * (13240.0 * X910 - 1127.0 * diffU4 * cX3 - 140000.0 * diffU2 * sX3 - 16920.0 * X1011 * sX4 + 7540.0 * X11_2 * s2X3 * 2.0 * sX4 * cX4 - 15080.0 * X910 * c2X3 - 15080.0 * X911 * s2X3 * cX4 + 15080.0 * X1011 * c2X3 * sX4));
}
template <typename T> PDDP_HD void quad_dynamics_eval(T* qdd, const T* x, const T* u) {
    T sX3, cX3, sX4, cX4, sX5, cX5;
    cf_sincos<T>(x[3], sX3, cX3); cf_sincos<T>(x[4], sX4, cX4); cf_sincos<T>(x[5], sX5, cX5);
    quad_dynamics_trig<T>(qdd, x, u, sX3, cX3, sX4, cX4, sX5, cX5);
}
template <typename T> PDDP_HD void quad_gradient_eval(T* dqdd, T* qdd, const T* x, const T* u) {
    enum { NP = 6 };
    T sX3, cX3, sX4, cX4, sX5, cX5;
    cf_sincos<T>(x[3], sX3, cX3); cf_sincos<T>(x[4], sX4, cX4); cf_sincos<T>(x[5], sX5, cX5);
    quad_dynamics_trig<T>(qdd, x, u, sX3, cX3, sX4, cX4, sX5, cX5);      // the plug-in's dynamics() inside dynamicsGradient (dynamics_quad.cuh:95): the same numbers as quad_dynamics_eval
    for (int i = 0; i < NP * 16; i++) dqdd[i] = 0;
    T UMTerm = (u[0] + u[1] + u[2] + u[3]) * QUAD_INVMASS;
    T row6Term = (sX3 * sX5 + cX3 * cX5 * sX4) * QUAD_INVMASS;
    dqdd[0 + 3 * NP] = (cX3 * sX5 - cX5 * sX3 * sX4) * UMTerm;
    dqdd[0 + 4 * NP] = (cX3 * cX4 * cX5) * UMTerm;
    dqdd[0 + 5 * NP] = (cX5 * sX3 - cX3 * sX4 * sX5) * UMTerm;
    dqdd[0 + 12 * NP] = row6Term; dqdd[0 + 13 * NP] = row6Term; dqdd[0 + 14 * NP] = row6Term; dqdd[0 + 15 * NP] = row6Term;
    T row7Term = -(cX5 * sX3 - cX3 * sX4 * sX5) * QUAD_INVMASS;
    dqdd[1 + 3 * NP] = -(cX3 * cX5 + sX3 * sX4 * sX5) * UMTerm;
    dqdd[1 + 4 * NP] = (cX3 * cX4 * sX5) * UMTerm;
    dqdd[1 + 5 * NP] = (sX3 * sX5 + cX3 * cX5 * sX4) * UMTerm;
    dqdd[1 + 12 * NP] = row7Term; dqdd[1 + 13 * NP] = row7Term; dqdd[1 + 14 * NP] = row7Term; dqdd[1 + 15 * NP] = row7Term;
    T row8Term = (cX3 * cX4) / QUAD_MASS;
    dqdd[2 + 3 * NP] = -(cX4 * sX3) * UMTerm;
    dqdd[2 + 4 * NP] = -(cX3 * sX4) * UMTerm;
    dqdd[2 + 12 * NP] = row8Term; dqdd[2 + 13 * NP] = row8Term; dqdd[2 + 14 * NP] = row8Term; dqdd[2 + 15 * NP] = row8Term;
    T X11_2 = x[11] * x[11], X910 = x[9] * x[10], X911 = x[9] * x[11], X1011 = x[10] * x[11];
    T cX3_2 = cX3 * cX3, cX4_2 = cX4 * cX4, cX3_2X4 = cX3_2 * cX4, cX3_2X4_2 = cX3_2 * cX4_2;
    T cX34sX3 = cX3 * cX4 * sX3, cX3sX34 = cX3 * sX3 * sX4, cX34sX34 = cX34sX3 * sX4;
    T invcX4 = 1.0 / cX4, invcX4_2 = invcX4 / cX4;
    T sumDifU02 = -u[0] + u[2], sumDifU0123 = -u[0] + u[1] - u[2] + u[3];
    T row3Term1 = 6.125 * cX3 * sX4 * invcX4, row3Term2 = 76.0869565217 * sX3;
    dqdd[3 + 3 * NP] = invcX4 * (76.0869565217 * cX3 * sX4 * sumDifU02 + 6.125 * sX3 * sX4 * sumDifU0123 + 0.73913043584 * (x[10] * x[10] * (2 * cX3_2 * cX4 - cX4) + X11_2 * cX4 - 2 * X11_2 * cX3_2 * cX4 - X911 * sX4 * cX4 + 2.0 * X1011 * sX3 * cX3 + 2 * X911 * cX3_2 * cX4 * sX4 - 2 * x[11] * cX3 * cX4_2 * sX3 - 2 * x[10] - 2 * X910 * cX3 * sX3 * sX4));
    dqdd[3 + 4 * NP] = invcX4_2 * (X910 + X1011 * sX4 + 76.0869565217 * sX3 * sumDifU02 - 6.125 * cX3 * sumDifU0123 + 0.73913043584 * (X910 * (1 + cX3_2) + X1011 * (1 - cX3_2 * sX4 + cX3_2X4_2 * sX4) + X911 * cX34sX3 * cX4_2));
    dqdd[3 + 9 * NP] = invcX4 * (x[10] * sX4 + 0.73913043584 * (x[10] * (cX3_2 - 1) * sX4 + x[11] * cX34sX34));
    dqdd[3 + 10 * NP] = invcX4 * (x[11] + x[9] * sX4 - 0.73913043584 * (x[11] * (1 + cX3_2 + cX3_2X4_2) - x[9] * (sX4 + cX3_2 * sX4) - 2 * x[10] * cX34sX3));
    dqdd[3 + 11 * NP] = invcX4 * (x[10] - 0.73913043584 * (x[10] + x[10] * cX3_2 + x[9] * cX34sX34 - 2 * x[11] * cX34sX3 - x[10] * cX3_2X4_2));
    dqdd[3 + 12 * NP] = row3Term1 - row3Term2;
    dqdd[3 + 13 * NP] = 76.0869565217 - row3Term1;
    dqdd[3 + 14 * NP] = row3Term1 + row3Term2;
    dqdd[3 + 15 * NP] = -76.0869565217 - row3Term1;
    T row4Term1 = 6.125 * sX3, row4Term2 = 76.0869565217 * cX3;
    dqdd[4 + 3 * NP] = 6.125 * cX3 * sumDifU0123 - 76.0869565217 * sX3 * sumDifU02 + 0.73913043584 * (X910 * (1 - 2 * cX3_2) + X1011 * sX4 * (2 * cX3_2 - 1) + 2 * X11_2 * cX34sX34 - 2 * X911 * cX34sX3);
    dqdd[4 + 4 * NP] = 0.73913043584 * (X11_2 * cX3_2 - 2.0 * X11_2 * cX3_2X4_2 - cX3_2 * sX4 + X1011 * cX34sX3) + X911 * sX4;
    dqdd[4 + 9 * NP] = 0.73913043584 * (x[11] * cX3_2X4 - x[10] * sX3 * cX3) - x[11] * cX4;
    dqdd[4 + 10 * NP] = 0.73913043584 * (x[11] * cX3sX34 - x[9] * sX3 * cX3);
    dqdd[4 + 11 * NP] = 0.73913043584 * (x[9] * cX3_2X4 + x[10] * cX3sX34 - 2.0 * x[11] * cX3_2X4 * sX4) - x[9] * cX4;
    dqdd[4 + 12 * NP] = -row4Term2 - row4Term1;
    dqdd[4 + 13 * NP] = row4Term1;
    dqdd[4 + 14 * NP] = row4Term2 - row4Term1;
    dqdd[4 + 15 * NP] = row4Term1;
    T row5Term1 = -6.125 * cX3 * invcX4, row5Term2 = 76.0869565217 * sX3 * invcX4;
    dqdd[5 + 3 * NP] = invcX4 * (76.0869565217 * cX3 * sumDifU02 + 6.125 * sX3 * sumDifU0123 + 0.73913043584 * (X11_2 * (sX4 * cX4 - 2.0 * cX3_2X4 * sX4) + X911 * (2.0 * cX3_2X4 - cX4) - 2.0 * X910 * sX3 * cX3 + 2.0 * X1011 * cX3sX34));
    dqdd[5 + 4 * NP] = invcX4_2 * (X1011 + sX4 * (X910 + 76.0869565217 * sX3 * sumDifU02 - 6.125 * cX3 * sumDifU0123) - 0.73913043584 * (X1011 * (1 + cX3_2) + X910 * (sX4 - cX3_2 * sX4) + X11_2 * cX34sX3 * cX4_2));
    dqdd[5 + 9 * NP] = invcX4 * (x[10] + 0.73913043584 * (x[10] * (cX3_2 - 1) + x[11] * cX34sX3));
    dqdd[5 + 10 * NP] = invcX4 * (x[9] + x[11] * sX4 - 0.73913043584 * (x[11] * (cX3_2 * sX4 + 1) - x[9] * (1 + cX3_2)));
    dqdd[5 + 11 * NP] = invcX4 * (x[10] * sX4 + 0.73913043584 * (x[9] * cX34sX3 - x[10] * sX4 * (1.0 + cX3_2) - 0.25 * x[11] * cX34sX34));
    dqdd[5 + 12 * NP] = -row5Term1 - row5Term2;
    dqdd[5 + 13 * NP] = row5Term1;
    dqdd[5 + 14 * NP] = -row5Term1 + row5Term2;
    dqdd[5 + 15 * NP] = row5Term1;
}

#undef CART_M_CART
#undef CART_M_POLE
#undef CART_ML
#undef CART_MLL
#undef CART_G
#undef QUAD_G
#undef QUAD_MASS
#undef QUAD_INVMASS
}  // namespace pddp
