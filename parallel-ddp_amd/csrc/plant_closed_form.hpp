// Closed-form plants: pendulum, cart-pole, quadrotor -- the FUNCTIONS of plants/dynamics_pend.cuh:30-51, dynamics_cart.cuh:30-76 and dynamics_quad.cuh:42-169, written
// from their mathematics rather than statement by statement (round 5):
//   * pendulum: qdd = u - g sin(theta);
//   * cart-pole: H(theta) qdd = tau(theta, thetad, u) with a 2 x 2 mass matrix; the gradient by IMPLICIT differentiation, d qdd = H^-1 (d tau - dH qdd), which needs no
//     quotient rule (the reference differentiates the adjugate form term by term -- the same numbers up to rounding);
//   * quadrotor: translational rows = thrust / mass x the third column e(phi, theta, psi) of the attitude matrix, their gradient = thrust x de/d(angle); the rotational
//     rows are the reference's polynomials in the body rates (p, q, r) and the trigonometric functions of roll and pitch -- a CONSTANT TABLE of the plant (numeric
//     coefficients 0.0005434782609, 76.0869565217, 0.73913043584, 6.125 ... and, in the gradient, a few terms that are not the exact derivative: they define the
//     plug-in and are reproduced as they are) -- grouped here by common factors: (2 c3^2 c4 - c4), c3 s3, the rate products.
// The literals are doubles on purpose: in a float handle the polynomials are evaluated in double and rounded at the assignment, as the reference's are.
// Pinned by tests/golden/closed_form_plants.json -- the reference's own statements executed in float64 on stored inputs (tests/test_closed_form_pins.py: dynamics and every
// gradient entry of 25 states per plant, 1e-12) -- for the kernels, the host emulation and the oracle alike.
#pragma once
#include "pddp_common.hpp"
namespace pddp {

constexpr double kCfGravity = -9.81;                                   // (signed: the plug-ins add it)

// ------------------------------------------------------------------------------------------------ pendulum (n = 2, m = 1)
template <typename T> PDDP_HD void pend_dynamics_eval(T* qdd, const T* x, const T* u) {
    T s, c; cf_sincos<T>(x[0], s, c);
    qdd[0] = u[0] + kCfGravity * s;
}
// dqdd[col]: d/d theta, d/d thetad, d/du
template <typename T> PDDP_HD void pend_gradient_eval(T* dqdd, T* qdd, const T* x, const T* u) {
    T s, c; cf_sincos<T>(x[0], s, c);
    qdd[0] = u[0] + kCfGravity * s;
    dqdd[0] = kCfGravity * c; dqdd[1] = 0.0; dqdd[2] = 1;
}

// ------------------------------------------------------------------------------------------------ cart-pole (n = 4: cart position, pole angle, their rates; m = 1)
// H = [mc + mp, ml cos; ml cos, ml l / 2] (ml = mp l / 2 with l = 1),  tau = [ml sin thetad^2 + u; ml sin g]
template <typename T> struct CartPole {
    static constexpr double kCart = 10, kPole = 1, kMl = kPole * 0.5, kMll = kMl * 0.5;
    T h00, h11, h01, b;        // mass matrix entries; b = ml sin(theta)
    T t0, t1, inv;             // generalised forces, 1 / det H
    PDDP_HD CartPole(const T* x, const T* u, T ct, T st) {
        h00 = kCart + kPole; h11 = kMll; h01 = kMl * ct; b = kMl * st;
        t0 = b * (x[3] * x[3]) + u[0]; t1 = b * kCfGravity;
        inv = 1 / (h00 * h11 - h01 * h01);
    }
    // H^-1 (r0, r1) by the adjugate
    PDDP_HD void solve(T r0, T r1, T& a0, T& a1) const { a0 = inv * (h11 * r0 - h01 * r1); a1 = inv * (h00 * r1 - h01 * r0); }
};
template <typename T> PDDP_HD void cart_dynamics_trig(T* qdd, const T* x, const T* u, T ct, T st) {
    const CartPole<T> cp(x, u, ct, st);
    cp.solve(cp.t0, cp.t1, qdd[0], qdd[1]);
}
template <typename T> PDDP_HD void cart_dynamics_eval(T* qdd, const T* x, const T* u) {
    T ct, st; cf_sincos<T>(x[1], st, ct);
    cart_dynamics_trig<T>(qdd, x, u, ct, st);
}
// dqdd[2 col + row], columns x, theta, xd, thetad, u
template <typename T> PDDP_HD void cart_gradient_eval(T* dqdd, T* qdd, const T* x, const T* u) {
    T ct, st; cf_sincos<T>(x[1], st, ct);
    const CartPole<T> cp(x, u, ct, st);
    cp.solve(cp.t0, cp.t1, qdd[0], qdd[1]);                          // (what the plug-in's dynamics() returns: the kernels build stage states from either)
    const T td = x[3];
    // d/d theta:  dH = [0, -b; -b, 0],  d tau = [h01 thetad^2; h01 g]   ->   H^-1 (d tau - dH qdd)
    cp.solve(cp.h01 * (td * td) + cp.b * qdd[1], cp.h01 * kCfGravity + cp.b * qdd[0], dqdd[2], dqdd[3]);
    // d/d thetad: d tau = [2 b thetad; 0]          d/du: d tau = [1; 0]
    cp.solve(2 * cp.b * td, T(0), dqdd[6], dqdd[7]);
    dqdd[8] = cp.inv * cp.h11; dqdd[9] = -(cp.inv * cp.h01);
    dqdd[0] = 0; dqdd[1] = 0; dqdd[4] = 0; dqdd[5] = 0;              // nothing depends on the cart's position or speed
}

// ------------------------------------------------------------------------------------------------ quadrotor (n = 12: position, roll / pitch / yaw, velocity, body rates; m = 4)
template <typename T> struct QuadTrig { T s3, c3, s4, c4, s5, c5; };
// third column of the attitude matrix and the thrust per unit mass (1 / 0.5 kg)
template <typename T> PDDP_HD void quad_thrust_axis(const QuadTrig<T>& g, T* e) {
    e[0] = g.s3 * g.s5 + g.c3 * g.c5 * g.s4;
    e[1] = -(g.c5 * g.s3 - g.c3 * g.s4 * g.s5);
    e[2] = g.c3 * g.c4;
}
template <typename T> PDDP_HD void quad_dynamics_trig(T* qdd, const T* x, const T* u, T sX3, T cX3, T sX4, T cX4, T sX5, T cX5) {
    const QuadTrig<T> g = {sX3, cX3, sX4, cX4, sX5, cX5};
    const T p = x[9], q = x[10], r = x[11], pq = p * q, pr = p * r, qr = q * r, qq = q * q, rr = r * r;
    const T lift = 2 * (u[0] + u[1] + u[2] + u[3]);                  // thrust / mass
    T e[3]; quad_thrust_axis<T>(g, e);
    qdd[0] = lift * e[0]; qdd[1] = lift * e[1]; qdd[2] = kCfGravity + lift * e[2];
    const T roll = u[1] - u[3], pitch = u[2] - u[0], yaw = u[0] - u[1] + u[2] - u[3];          // rotor differences
    const T ic4 = 1 / g.c4, cc = g.c3 * g.c3, c34 = g.c3 * g.c4, s2 = 2.0 * g.s3 * g.c3, c2 = cf_cos2<T>(x[3]);
    const double k = 0.0005434782609;
    qdd[3] = ic4 * (k * (140000.0 * (roll * g.c4 - pitch * g.s3 * g.s4) + 1127.0 * yaw * g.c3 * g.s4 + 32000.0 * qr - 28320.0 * pq * g.s4
                         + 30160.0 * ((pq * g.s4 - qr) * cc - qr * (c34 * c34) + ((qq - rr) + pr * g.s4) * c34 * g.s3)));
    qdd[4] = 76.08695652 * pitch * g.c3 - 0.6125 * yaw * g.s3 - 1.0 * pr * g.c4 - 8.195652174 * pq * s2
             + 16.39130435 * (cc * g.c4 * (pr - rr * g.s4) + qr * g.c3 * g.s3 * g.s4);
    qdd[5] = -ic4 * (k * (13240.0 * pq - 16920.0 * qr * g.s4 - (1127.0 * yaw * g.c3 + 140000.0 * pitch * g.s3)
                          + 15080.0 * ((rr * g.s4 - pr) * s2 * g.c4 + (qr * g.s4 - pq) * c2)));
}
template <typename T> PDDP_HD void quad_dynamics_eval(T* qdd, const T* x, const T* u) {
    QuadTrig<T> g;
    cf_sincos<T>(x[3], g.s3, g.c3); cf_sincos<T>(x[4], g.s4, g.c4); cf_sincos<T>(x[5], g.s5, g.c5);
    quad_dynamics_trig<T>(qdd, x, u, g.s3, g.c3, g.s4, g.c4, g.s5, g.c5);
}
// dqdd[6 col + row], columns 0..11 the state, 12..15 the rotors
template <typename T> PDDP_HD void quad_gradient_eval(T* dqdd, T* qdd, const T* x, const T* u) {
    enum { NP = 6 };
    QuadTrig<T> g;
    cf_sincos<T>(x[3], g.s3, g.c3); cf_sincos<T>(x[4], g.s4, g.c4); cf_sincos<T>(x[5], g.s5, g.c5);
    quad_dynamics_trig<T>(qdd, x, u, g.s3, g.c3, g.s4, g.c4, g.s5, g.c5);          // (the plug-in's dynamics(): the kernels build stage states from either)
    for (int i = 0; i < NP * 16; i++) dqdd[i] = 0;
    auto D = [&](int row, int col) -> T& { return dqdd[row + col * NP]; };
    const T s3 = g.s3, c3 = g.c3, s4 = g.s4, c4 = g.c4, s5 = g.s5, c5 = g.c5;
    // ---- translation: lift x d e / d(roll, pitch, yaw), and 2 e per rotor
    const T lift = (u[0] + u[1] + u[2] + u[3]) * 2;
    T e[3]; quad_thrust_axis<T>(g, e);
    D(0, 3) = (c3 * s5 - c5 * s3 * s4) * lift; D(0, 4) = (c3 * c4 * c5) * lift; D(0, 5) = (c5 * s3 - c3 * s4 * s5) * lift;
    D(1, 3) = -(c3 * c5 + s3 * s4 * s5) * lift; D(1, 4) = (c3 * c4 * s5) * lift; D(1, 5) = e[0] * lift;
    D(2, 3) = -(c4 * s3) * lift; D(2, 4) = -(c3 * s4) * lift;
    const T per_rotor[3] = {T(e[0] * 2), T(e[1] * 2), T(e[2] / 0.5)};
    for (int row = 0; row < 3; row++) for (int m = 0; m < 4; m++) D(row, 12 + m) = per_rotor[row];
    // ---- rotation: the plug-in's table (coefficients and terms as they are; see the header), grouped by common factors
    const T p = x[9], q = x[10], r = x[11], pq = p * q, pr = p * r, qr = q * r, qq = q * q, rr = r * r;
    const T cc = c3 * c3, dd = c4 * c4, cc4 = cc * c4, ccdd = cc * dd;      // c3^2, c4^2, c3^2 c4, c3^2 c4^2
    const T h3 = c3 * c4 * s3, h4 = c3 * s3 * s4, h34 = h3 * s4;            // c3 c4 s3, c3 s3 s4, c3 c4 s3 s4
    const T ic4 = 1.0 / c4, ic44 = ic4 / c4;
    const T pitch = -u[0] + u[2], yawn = -u[0] + u[1] - u[2] + u[3];        // (yawn = minus the yaw difference of the dynamics)
    const double kA = 76.0869565217, kB = 6.125, kK = 0.73913043584;
    const T w = 2 * cc4 - c4;                                                // the factor the roll derivatives share
    // row 3
    D(3, 3) = ic4 * (kA * c3 * s4 * pitch + kB * s3 * s4 * yawn
                     + kK * (w * (qq - rr + pr * s4) + 2.0 * c3 * s3 * (qr - pq * s4 - r * dd) - 2 * q));
    D(3, 4) = ic44 * (pq + qr * s4 + kA * s3 * pitch - kB * c3 * yawn + kK * (pq * (1 + cc) + qr * (1 - cc * s4 + ccdd * s4) + pr * h3 * dd));
    D(3, 9) = ic4 * (q * s4 + kK * (q * (cc - 1) * s4 + r * h34));
    D(3, 10) = ic4 * (r + p * s4 - kK * (r * (1 + cc + ccdd) - p * (s4 + cc * s4) - 2 * q * h3));
    D(3, 11) = ic4 * (q - kK * (q + q * cc + p * h34 - 2 * r * h3 - q * ccdd));
    { const T a = kB * c3 * s4 * ic4, b = kA * s3; D(3, 12) = a - b; D(3, 13) = kA - a; D(3, 14) = a + b; D(3, 15) = -kA - a; }
    // row 4
    D(4, 3) = kB * c3 * yawn - kA * s3 * pitch + kK * ((1 - 2 * cc) * (pq - qr * s4) + 2 * r * (r * h34 - p * h3));
    D(4, 4) = kK * (rr * cc - 2.0 * rr * ccdd - cc * s4 + qr * h3) + pr * s4;
    D(4, 9) = kK * (r * cc4 - q * s3 * c3) - r * c4;
    D(4, 10) = kK * (r * h4 - p * s3 * c3);
    D(4, 11) = kK * (p * cc4 + q * h4 - 2.0 * r * cc4 * s4) - p * c4;
    { const T a = kB * s3, b = kA * c3; D(4, 12) = -b - a; D(4, 13) = a; D(4, 14) = b - a; D(4, 15) = a; }
    // row 5
    D(5, 3) = ic4 * (kA * c3 * pitch + kB * s3 * yawn + kK * (w * (pr - rr * s4) + 2.0 * c3 * s3 * (qr * s4 - pq)));
    D(5, 4) = ic44 * (qr + s4 * (pq + kA * s3 * pitch - kB * c3 * yawn) - kK * (qr * (1 + cc) + pq * (s4 - cc * s4) + rr * h3 * dd));
    D(5, 9) = ic4 * (q + kK * (q * (cc - 1) + r * h3));
    D(5, 10) = ic4 * (p + r * s4 - kK * (r * (cc * s4 + 1) - p * (1 + cc)));
    D(5, 11) = ic4 * (q * s4 + kK * (p * h3 - q * s4 * (1.0 + cc) - 0.25 * r * h34));
    { const T a = -kB * c3 * ic4, b = kA * s3 * ic4; D(5, 12) = -a - b; D(5, 13) = a; D(5, 14) = -a + b; D(5, 15) = a; }
}

}  // namespace pddp
