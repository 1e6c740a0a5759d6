// KUKA iiwa14 forward dynamics and its analytic gradient, ONE THREAD PER EVALUATION ("thread lane", tl): a wave carries 64 independent
// evaluations, no cross-lane traffic, no LDS in the arithmetic.
//
// Why a third formulation.  The lane-group kernels (plant_arm_lg.hpp) spread one evaluation over 8 lanes: every serial recursion over
// the 7 links is executed as 6 sweeps by all lanes, every broadcast is 2-3 DPP moves, one lane of 8 idles -- ~3.0 k wave instructions
// per forward-dynamics evaluation of 8 instances and 13.3 k per gradient.  With one evaluation per LANE the same wave instruction count
// serves 64 instances, recursions cost what they cost serially, and the algorithm can be the cheapest one instead of the reference's:
//
//   * body coordinates (Featherstone): joint i's frame is F_i Rz(q_i) in its parent's; for the iiwa every F_i is a SIGNED PERMUTATION
//     of the axes plus a translation along ONE parent axis (plants/iiwa14.urdf joint origins; plants/dynamics_arm.cuh:353-427), so a
//     Pluecker transform of a motion / force vector is 8 multiply-adds for the rotation about z and 2 for the translation (a dense
//     6x6 transform is 36), and a rigid-body inertia moves between frames in ~30;
//   * rigid-body inertias stay in their 10-parameter form (m, h = m c, I about the frame origin) -- composite inertias are sums of
//     rigid bodies, so the composite-inertia recursion never needs a 6x6;
//   * bias torque by recursive Newton-Euler, mass matrix by the composite-rigid-body algorithm (M_ij = z-moment of Ic_i e_z carried
//     to frame j), M = L D L' (no square roots), qdd = M^-1 (u - C - 0.5 qd)            -- same function as dynamics<T>
//     (plants/dynamics_arm.cuh:2097-2163: world-frame composite inertias + unpivoted Gauss-Jordan on [M | I]);
//   * gradient: d qdd / d(q, qd) = -M^-1 d ID/d(q, qd) at the computed qdd, d qdd/du = M^-1, with the derivatives of the inverse
//     dynamics by forward-mode recursion over the chain: d/dq_j enters only through X_j(q_j) (d(X_j v)/dq_j = -e_z x (X_j v),
//     d(X_j' f)/dq_j = X_j' (e_z x* f)), links above joint j are untouched, links below see a transported tangent
//                                                                                         -- same function as dynamicsGradient<T> (:2167-2289).
// Floating point: this is a different (shorter, better conditioned) operation sequence than the reference's, fused multiply-adds
// allowed; parity is asserted against the URDF-derived float64 fixture and the oracle within the stated bar (tests/test_urdf_pins.py),
// not bit for bit.
#pragma once

#include <type_traits>

#include "iiwa14_model_data.h"
#include "plant_arm.hpp"

// fused multiply-adds for everything in this header (the rest of the library is built with -ffp-contract=off; see the header comment)
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

// A fence for the instruction scheduler (device code only): nothing is moved across it.  The gradient below is ONE basic block of ~4 k instructions; left alone the
// scheduler may interleave neighbouring levels of its sweep.  Measured (tools/tl_probe.py, round 5): fencing the levels does NOT lower the pressure -- k_nis_tl spills 39
// registers with the fences, 30 without: the live state of ONE level is what does not fit -- so the knob is off; it stays for the next compiler.
#ifndef PDDP_TL_FENCE
#define PDDP_TL_FENCE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && PDDP_TL_FENCE
#define PDDP_TL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PDDP_TL_SCHED_FENCE() ((void)0)
#endif

#ifndef PDDP_TL_OPAQUE
#define PDDP_TL_OPAQUE 1
#endif

namespace pddp {

// hides a value's provenance from the optimiser (device code; the host build has registers to spare)
template <typename T> PDDP_HD void tl_opaque(T& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}

// Joint-frame classes of the iiwa14 chain (validated against the loaded model on the host: arm_tl_model_from_tables).
//   rotation of F_i:  ID = identity;  A = [-x, z, y] (columns -e_x, e_z, e_y);  B = [x, z, -y] (columns e_x, e_z, -e_y)
//   translation of F_i along the parent's z (Z) or y (Y) axis
enum ArmTlKind { kTlIdZ = 0, kTlAZ = 1, kTlAY = 2, kTlBZ = 3 };
PDDP_HD constexpr int arm_tl_kind(int link) { return link == 0 ? kTlIdZ : link == 1 ? kTlAZ : (link == 3 || link == 5) ? kTlBZ : kTlAY; }

template <typename T>
struct ArmTlModel {          // compact robot data of the thread-lane kernels (derived from ArmModel on the host)
    T m[kArmNB];             // link mass
    T h[kArmNB][3];          // first moment m c (joint frame)
    T I[kArmNB][6];          // rotational inertia about the joint-frame origin: xx, yy, zz, xy, xz, yz
    T r[kArmNB];             // translation of F_i along its axis (z or y of the parent, by kind)
};

// The two robot models the reference ships (USE_WAFR_URDF 0 / 1; tools/gen_iiwa14_tables.py), as a constant expression: a kernel that
// declares `constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V)` gets every robot constant as an instruction literal -- no registers, no
// loads, and the structural zeros of the inertias disappear from the arithmetic.  A handle whose tables were edited (pddp_set_array) does
// not match either and keeps the lane-group kernels.
template <typename T>
constexpr ArmTlModel<T> arm_tl_builtin(int variant) {
    ArmTlModel<T> o{};
    for (int b = 0; b < kArmNB; b++) {
        const double* S = IIWA14_SPATIAL_INERTIA[variant][b];
        const double* F = IIWA14_JOINT_FRAME[variant][b];
        o.r[b] = (T)F[arm_tl_kind(b) != kTlAY ? 14 : 13];
        o.m[b] = (T)S[21];
        o.h[b][0] = (T)-S[31]; o.h[b][1] = (T)S[30]; o.h[b][2] = (T)-S[24];
        o.I[b][0] = (T)S[0]; o.I[b][1] = (T)S[7]; o.I[b][2] = (T)S[14]; o.I[b][3] = (T)S[6]; o.I[b][4] = (T)S[12]; o.I[b][5] = (T)S[13];
    }
    return o;
}
template <typename T>
inline bool arm_tl_models_equal(const ArmTlModel<T>& a, const ArmTlModel<T>& b) {
    bool eq = true;
    for (int i = 0; i < kArmNB; i++) {
        eq &= a.m[i] == b.m[i] && a.r[i] == b.r[i];
        for (int e = 0; e < 3; e++) eq &= a.h[i][e] == b.h[i][e];
        for (int e = 0; e < 6; e++) eq &= a.I[i][e] == b.I[i][e];
    }
    return eq;
}

// ArmModel tables -> ArmTlModel.  Returns false when the tables are not of the structure the thread-lane kernels hard-wire (then the
// caller keeps the lane-group kernels, which take any joint frames / spatial inertias).
template <typename T>
inline bool arm_tl_model_from_tables(ArmTlModel<T>& o, const ArmModel<T>& t) {
    const double tol = 1e-9;
    bool ok = true;
    auto near = [&](double a, double b) { return std::fabs(a - b) <= tol * (1.0 + std::fabs(b)); };
    for (int b = 0; b < kArmNB; b++) {
        const T* F = t.F + 16 * b; const T* S = t.I + 36 * b;
        const int kind = arm_tl_kind(b);
        // rotation columns (column-major 4x4)
        const double RA[9] = {-1, 0, 0, 0, 0, 1, 0, 1, 0}, RB[9] = {1, 0, 0, 0, 0, 1, 0, -1, 0}, RI[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double* R = kind == kTlIdZ ? RI : kind == kTlBZ ? RB : RA;
        for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) ok &= near(F[4 * c + r], R[3 * c + r]);
        const bool along_z = (kind != kTlAY);
        ok &= near(F[12], 0) && near(F[along_z ? 13 : 14], 0);
        o.r[b] = F[along_z ? 14 : 13];
        // spatial inertia [I  skew(h); skew(h)'  m 1], column-major, [angular; linear]
        o.m[b] = S[21];
        o.h[b][0] = -S[31]; o.h[b][1] = S[30]; o.h[b][2] = -S[24];
        o.I[b][0] = S[0]; o.I[b][1] = S[7]; o.I[b][2] = S[14]; o.I[b][3] = S[6]; o.I[b][4] = S[12]; o.I[b][5] = S[13];
        const double hx = o.h[b][0], hy = o.h[b][1], hz = o.h[b][2], m = o.m[b];
        const double full[36] = {o.I[b][0], o.I[b][3], o.I[b][4], 0, -hz, hy,   o.I[b][3], o.I[b][1], o.I[b][5], hz, 0, -hx,   o.I[b][4], o.I[b][5], o.I[b][2], -hy, hx, 0,
                                 0, hz, -hy, m, 0, 0,   -hz, 0, hx, 0, m, 0,   hy, -hx, 0, 0, 0, m};
        for (int e = 0; e < 36; e++) ok &= near(S[e], full[e]);
    }
    return ok;
}

// ------------------------------------------------------------------------------------------------ exact-zero pruning (round 5)
// A third of the robot constants are exact zeros (links 0, 2, 3, 5 of both models: first moment h_x, products of inertia I_xy, I_xz; the base does not move, so link 0's
// inboard velocity and all of its inboard acceleration but gravity are zeros too), and with the model a constant expression they reach the arithmetic as literals -- but
// IEEE semantics keep `0 * x + y` alive (x might be an infinity or a NaN, y a negative zero): the compiler deletes those terms only under -ffinite-math-only
// -fno-signed-zeros, which was measured at -12 % / -15 % static vector instructions on the rollout / setup kernels and lost again to the spills the flags' other
// rewrites caused (profiles/r04*: "finite-math build: not adopted").  TlZ does the deletion in SOURCE instead, for the zeros only: a product with an operand that the
// optimiser has proven to be the constant zero IS the constant zero, a sum with it is the other operand.  Finite inputs give the same value as the full expression
// (up to the sign of a zero result); everything else about the arithmetic -- operation order, the fused multiply-adds of this header -- is untouched.
// __builtin_constant_p is resolved after inlining and scalar replacement (llvm.is.constant), so a zero is seen wherever the literal ends up, also through the
// callers' arrays; a value that is not a compile-time zero takes the plain operation.
#ifndef PDDP_TLZ_MASK
#define PDDP_TLZ_MASK 31       // which groups of routines prune (measurement knob, tools/tlz_variants.sh): 1 frame changes / inertia products, 2 inertia and rotation transports,
#endif                         // 4 the gradient's composite sweep, 8 the triangular solves, 16 link accelerations
template <typename T, bool ON = true> struct TlZ {
    T v;
    PDDP_HD TlZ() : v(T(0)) {}
    PDDP_HD TlZ(T x) : v(x) {}
    PDDP_HD bool zero() const { return ON && __builtin_constant_p(v) && v == T(0); }
};
template <typename T, bool ON> PDDP_HD TlZ<T, ON> operator*(TlZ<T, ON> a, TlZ<T, ON> b) { return (a.zero() || b.zero()) ? TlZ<T, ON>(T(0)) : TlZ<T, ON>(a.v * b.v); }
template <typename T, bool ON> PDDP_HD TlZ<T, ON> operator+(TlZ<T, ON> a, TlZ<T, ON> b) { return a.zero() ? b : b.zero() ? a : TlZ<T, ON>(a.v + b.v); }
template <typename T, bool ON> PDDP_HD TlZ<T, ON> operator-(TlZ<T, ON> a, TlZ<T, ON> b) { return b.zero() ? a : a.zero() ? TlZ<T, ON>(-b.v) : TlZ<T, ON>(a.v - b.v); }
template <typename T, bool ON> PDDP_HD TlZ<T, ON> operator-(TlZ<T, ON> a) { return a.zero() ? TlZ<T, ON>(T(0)) : TlZ<T, ON>(-a.v); }

// ------------------------------------------------------------------------------------------------ frame changes (compile-time kind)
// parent coordinates -> stage-1 child coordinates (before the joint rotation): w1 = R_F' w
template <int KIND, typename T> PDDP_HD void tl_to_child_axes(T* o, const T* w) {
    if (KIND == kTlIdZ) { o[0] = w[0]; o[1] = w[1]; o[2] = w[2]; }
    else if (KIND == kTlBZ) { o[0] = w[0]; o[1] = w[2]; o[2] = -w[1]; }
    else { o[0] = -w[0]; o[1] = w[2]; o[2] = w[1]; }
}
// stage-1 child coordinates -> parent coordinates: w = R_F w1
template <int KIND, typename T> PDDP_HD void tl_to_parent_axes(T* o, const T* w1) {
    if (KIND == kTlIdZ) { o[0] = w1[0]; o[1] = w1[1]; o[2] = w1[2]; }
    else if (KIND == kTlBZ) { o[0] = w1[0]; o[1] = -w1[2]; o[2] = w1[1]; }
    else { o[0] = -w1[0]; o[1] = w1[2]; o[2] = w1[1]; }
}
// motion vector [w; v] of the parent frame -> child frame:  w_c = Rz' R_F' w,  v_c = Rz' R_F' (v + w x r)
template <int KIND, typename T> PDDP_HD void tl_motion_to_child(T* o, const T* mv, T r, T c, T s) {
    typedef TlZ<T, (PDDP_TLZ_MASK & 1) != 0> Z;
    const Z m0(mv[0]), m1(mv[1]), m2(mv[2]), rz(r), cz(c), sz(s);
    Z vv[3] = {Z(mv[3]), Z(mv[4]), Z(mv[5])};
    if (KIND == kTlAY) { vv[0] = vv[0] - m2 * rz; vv[2] = vv[2] + m0 * rz; }             // w x (0, r, 0) = (-wz r, 0, wx r)
    else { vv[0] = vv[0] + m1 * rz; vv[1] = vv[1] - m0 * rz; }                           // w x (0, 0, r) = (wy r, -wx r, 0)
    Z w1[3] = {m0, m1, m2}, v1[3] = {vv[0], vv[1], vv[2]};
    if (KIND == kTlBZ) { w1[1] = m2; w1[2] = -m1; v1[1] = vv[2]; v1[2] = -vv[1]; }
    else if (KIND != kTlIdZ) { w1[0] = -m0; w1[1] = m2; w1[2] = m1; v1[0] = -vv[0]; v1[1] = vv[2]; v1[2] = vv[1]; }
    o[0] = (cz * w1[0] + sz * w1[1]).v; o[1] = (cz * w1[1] - sz * w1[0]).v; o[2] = w1[2].v;
    o[3] = (cz * v1[0] + sz * v1[1]).v; o[4] = (cz * v1[1] - sz * v1[0]).v; o[5] = v1[2].v;
}
// force vector [n; f] of the child frame -> parent frame:  f_p = R_F Rz f,  n_p = R_F Rz n + r x f_p
template <int KIND, typename T> PDDP_HD void tl_force_to_parent(T* o, const T* fv, T r, T c, T s) {
    typedef TlZ<T, (PDDP_TLZ_MASK & 1) != 0> Z;
    const Z cz(c), sz(s), rz(r), f0(fv[0]), f1(fv[1]), f3(fv[3]), f4(fv[4]);
    const Z n1[3] = {cz * f0 - sz * f1, sz * f0 + cz * f1, Z(fv[2])};
    const Z g1[3] = {cz * f3 - sz * f4, sz * f3 + cz * f4, Z(fv[5])};
    Z p[6];
    if (KIND == kTlIdZ) { p[0] = n1[0]; p[1] = n1[1]; p[2] = n1[2]; p[3] = g1[0]; p[4] = g1[1]; p[5] = g1[2]; }
    else if (KIND == kTlBZ) { p[0] = n1[0]; p[1] = -n1[2]; p[2] = n1[1]; p[3] = g1[0]; p[4] = -g1[2]; p[5] = g1[1]; }
    else { p[0] = -n1[0]; p[1] = n1[2]; p[2] = n1[1]; p[3] = -g1[0]; p[4] = g1[2]; p[5] = g1[1]; }
    if (KIND == kTlAY) { p[0] = p[0] + rz * p[5]; p[2] = p[2] - rz * p[3]; }              // (0, r, 0) x f = (r fz, 0, -r fx)
    else { p[0] = p[0] - rz * p[4]; p[1] = p[1] + rz * p[3]; }                           // (0, 0, r) x f = (-r fy, r fx, 0)
#pragma unroll
    for (int e = 0; e < 6; e++) o[e] = p[e].v;
}
// rigid-body inertia (m, h, I: xx yy zz xy xz yz) of the child frame, expressed in the parent frame, ADDED to (mp, hp, Ip)
template <int KIND, typename T> PDDP_HD void tl_inertia_add_to_parent(T& mp, T* hp, T* Ip, T m, const T* h, const T* I, T r, T c, T s) {
    typedef TlZ<T, (PDDP_TLZ_MASK & 2) != 0> Z;
    const Z cz(c), sz(s), rz(r), mz(m), h0(h[0]), h1v(h[1]), I0(I[0]), I1(I[1]), I3(I[3]), I4(I[4]), I5(I[5]);
    // rotate about z: h1 = Rz h, I1 = Rz I Rz'
    const Z h1[3] = {cz * h0 - sz * h1v, sz * h0 + cz * h1v, Z(h[2])};
    const Z cc = cz * cz, ss = sz * sz, cs = cz * sz;
    const Z d = I0 - I1;
    const Z xx = cc * I0 + ss * I1 - (cs + cs) * I3;
    const Z yy = ss * I0 + cc * I1 + (cs + cs) * I3;
    const Z xy = cs * d + (cc - ss) * I3;
    const Z xz = cz * I4 - sz * I5, yz = sz * I4 + cz * I5, zz(I[2]);
    // axis permutation R_F
    Z h2[3], J[6];
    if (KIND == kTlIdZ) { h2[0] = h1[0]; h2[1] = h1[1]; h2[2] = h1[2]; J[0] = xx; J[1] = yy; J[2] = zz; J[3] = xy; J[4] = xz; J[5] = yz; }
    else if (KIND == kTlBZ) { h2[0] = h1[0]; h2[1] = -h1[2]; h2[2] = h1[1]; J[0] = xx; J[1] = zz; J[2] = yy; J[3] = -xz; J[4] = xy; J[5] = -yz; }      // x' = x, y' = -z, z' = y
    else { h2[0] = -h1[0]; h2[1] = h1[2]; h2[2] = h1[1]; J[0] = xx; J[1] = zz; J[2] = yy; J[3] = -xz; J[4] = -xy; J[5] = yz; }                         // x' = -x, y' = z, z' = y
    // shift the reference point by r along y (AY) or z: I_p = I' - (r h' + h r') + 2 (h.r) 1 - m (r r' - r.r 1)
    const Z mr = mz * rz;
    Z Q[6] = {Z(Ip[0]), Z(Ip[1]), Z(Ip[2]), Z(Ip[3]), Z(Ip[4]), Z(Ip[5])}, g[3] = {Z(hp[0]), Z(hp[1]), Z(hp[2])};
    if (KIND == kTlAY) {
        const Z t = (h2[1] + h2[1]) * rz + mr * rz;
        Q[0] = Q[0] + (J[0] + t); Q[1] = Q[1] + J[1]; Q[2] = Q[2] + (J[2] + t); Q[3] = Q[3] + (J[3] - rz * h2[0]); Q[4] = Q[4] + J[4]; Q[5] = Q[5] + (J[5] - rz * h2[2]);
        g[0] = g[0] + h2[0]; g[1] = g[1] + (h2[1] + mr); g[2] = g[2] + h2[2];
    } else {
        const Z t = (h2[2] + h2[2]) * rz + mr * rz;
        Q[0] = Q[0] + (J[0] + t); Q[1] = Q[1] + (J[1] + t); Q[2] = Q[2] + J[2]; Q[3] = Q[3] + J[3]; Q[4] = Q[4] + (J[4] - rz * h2[0]); Q[5] = Q[5] + (J[5] - rz * h2[1]);
        g[0] = g[0] + h2[0]; g[1] = g[1] + h2[1]; g[2] = g[2] + (h2[2] + mr);
    }
#pragma unroll
    for (int e = 0; e < 6; e++) Ip[e] = Q[e].v;
#pragma unroll
    for (int e = 0; e < 3; e++) hp[e] = g[e].v;
    mp = (Z(mp) + mz).v;
}
// f = I_spatial [w; v] = [I w + h x v ; m v - h x w]
template <typename T> PDDP_HD void tl_inertia_mul(T* o, T m, const T* h, const T* I, const T* mv) {
    typedef TlZ<T, (PDDP_TLZ_MASK & 1) != 0> Z;
    const Z w[3] = {Z(mv[0]), Z(mv[1]), Z(mv[2])}, v[3] = {Z(mv[3]), Z(mv[4]), Z(mv[5])};
    const Z hz[3] = {Z(h[0]), Z(h[1]), Z(h[2])}, J[6] = {Z(I[0]), Z(I[1]), Z(I[2]), Z(I[3]), Z(I[4]), Z(I[5])}, mz(m);
    o[0] = (J[0] * w[0] + J[3] * w[1] + J[4] * w[2] + (hz[1] * v[2] - hz[2] * v[1])).v;
    o[1] = (J[3] * w[0] + J[1] * w[1] + J[5] * w[2] + (hz[2] * v[0] - hz[0] * v[2])).v;
    o[2] = (J[4] * w[0] + J[5] * w[1] + J[2] * w[2] + (hz[0] * v[1] - hz[1] * v[0])).v;
    o[3] = (mz * v[0] - (hz[1] * w[2] - hz[2] * w[1])).v;
    o[4] = (mz * v[1] - (hz[2] * w[0] - hz[0] * w[2])).v;
    o[5] = (mz * v[2] - (hz[0] * w[1] - hz[1] * w[0])).v;
}
// o += v x* f   (spatial force cross product: [w x n + v x f ; w x f])
template <typename T> PDDP_HD void tl_crf_add(T* o, const T* mv, const T* fv) {
    typedef TlZ<T, (PDDP_TLZ_MASK & 1) != 0> Z;
    const Z w[3] = {Z(mv[0]), Z(mv[1]), Z(mv[2])}, v[3] = {Z(mv[3]), Z(mv[4]), Z(mv[5])};
    const Z n[3] = {Z(fv[0]), Z(fv[1]), Z(fv[2])}, f[3] = {Z(fv[3]), Z(fv[4]), Z(fv[5])};
    o[0] = (Z(o[0]) + ((w[1] * n[2] - w[2] * n[1]) + (v[1] * f[2] - v[2] * f[1]))).v;
    o[1] = (Z(o[1]) + ((w[2] * n[0] - w[0] * n[2]) + (v[2] * f[0] - v[0] * f[2]))).v;
    o[2] = (Z(o[2]) + ((w[0] * n[1] - w[1] * n[0]) + (v[0] * f[1] - v[1] * f[0]))).v;
    o[3] = (Z(o[3]) + (w[1] * f[2] - w[2] * f[1])).v;
    o[4] = (Z(o[4]) + (w[2] * f[0] - w[0] * f[2])).v;
    o[5] = (Z(o[5]) + (w[0] * f[1] - w[1] * f[0])).v;
}

template <typename T> PDDP_HD void tl_sincos(T q, T& s, T& c);
// float: two-term Cody-Waite reduction by pi/2 (exact products through the fused multiply-add; joint angles are a few radians) and the
// degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4]: ~1 ulp, ~25 instructions, no branches (the library sincosf carries a
// Payne-Hanek path that costs ~200 instructions per call when inlined seven times per evaluation).
template <> PDDP_HD void tl_sincos<float>(float q, float& s, float& c) {
    const float k = rintf(q * 0.636619772367581343f);
    float r = fmaf(k, -1.57079637050628662109375f, q);
    r = fmaf(k, 4.37113900018624283e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
    const int n = static_cast<int>(k) & 3;
    const float ss = (n & 1) ? cp : sp, cc = (n & 1) ? sp : cp;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
template <> PDDP_HD void tl_sincos<double>(double q, double& s, double& c) { sincos(q, &s, &c); }

// What one evaluation keeps for the gradient (all per thread).
template <typename T>
struct ArmTlState {
    T c[kArmNB], s[kArmNB];
    T v[kArmNB][6];          // link velocities (own frame)
    T L[21];                 // unit lower factor of M = L D L' (row-major strict lower triangle: L[i(i-1)/2 + j], j < i)
    T Dinv[kArmNB];          // 1 / D_i
};

// solve M x = b in place with the stored factors
template <typename T> PDDP_HD void tl_ldl_solve(const T* L, const T* Dinv, T* x) {
    typedef TlZ<T, (PDDP_TLZ_MASK & 8) != 0> Z;                    // (a unit right-hand side -- the control columns, M^-1 e_j -- skips the multiplications by its leading zeros)
#pragma unroll
    for (int i = 1; i < kArmNB; i++)
#pragma unroll
        for (int j = 0; j < i; j++) x[i] = (Z(x[i]) - Z(L[i * (i - 1) / 2 + j]) * Z(x[j])).v;
#pragma unroll
    for (int i = 0; i < kArmNB; i++) x[i] = (Z(x[i]) * Z(Dinv[i])).v;
#pragma unroll
    for (int i = kArmNB - 2; i >= 0; i--)
#pragma unroll
        for (int j = i + 1; j < kArmNB; j++) x[i] = (Z(x[i]) - Z(L[j * (j - 1) / 2 + i]) * Z(x[j])).v;
}
template <typename T> PDDP_HD void tl_ldl_solve(const ArmTlState<T>& st, T* x) { tl_ldl_solve<T>(st.L, st.Dinv, x); }
// compile-time loop over the links with their frame kind as a template argument
template <int I, int END, int STEP> struct TlFor {
    template <typename F> static PDDP_HD void run(F&& f) { f(std::integral_constant<int, I>()); TlFor<I + STEP, END, STEP>::run(f); }
};
template <int END, int STEP> struct TlFor<END, END, STEP> { template <typename F> static PDDP_HD void run(F&&) {} };

// Forward dynamics in three parts (arm_tl_dynamics below runs them in order; the split rollout kernel k_fp_tl2 runs the bias and the factor parts
// on two different wavefronts):
//   arm_tl_trig:    sines / cosines of the joint angles
//   arm_tl_bias:    recursive Newton-Euler with qdd = 0 -> bias[i] = C_i + 0.5 qd_i (gravity as an upward acceleration of the base; joint damping 0.5 qd,
//                   plants/dynamics_arm.cuh:1433-1434; iiwa14.urdf); fills st.v
//   arm_tl_factor:  composite rigid bodies, mass matrix, M = L D L' -> st.L, st.Dinv
template <typename T>
PDDP_HD void arm_tl_trig(ArmTlState<T>& st, const T* q) {
#pragma unroll
    for (int i = 0; i < kArmNB; i++) tl_sincos<T>(q[i], st.s[i], st.c[i]);
}
template <typename T>
PDDP_HD void arm_tl_bias(const ArmTlModel<T>& md, T grav, ArmTlState<T>& st, const T* qd, T* bias) {
    constexpr int NB = kArmNB;
    T f[NB][6];
    {
        T a[6] = {T(0), T(0), T(0), T(0), T(0), grav}, vp[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        TlFor<0, NB, 1>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int K = arm_tl_kind(i);
            T* v = st.v[i];
            typedef TlZ<T, (PDDP_TLZ_MASK & 16) != 0> Z;
            tl_motion_to_child<K>(v, vp, md.r[i], st.c[i], st.s[i]);
            v[2] = (Z(v[2]) + Z(qd[i])).v;
            T an[6];
            tl_motion_to_child<K>(an, a, md.r[i], st.c[i], st.s[i]);
            an[0] = (Z(an[0]) + Z(qd[i]) * Z(v[1])).v; an[1] = (Z(an[1]) - Z(qd[i]) * Z(v[0])).v;              // v x (e_z qd): [w x e_z; vl x e_z] qd
            an[3] = (Z(an[3]) + Z(qd[i]) * Z(v[4])).v; an[4] = (Z(an[4]) - Z(qd[i]) * Z(v[3])).v;
            T Iv[6];
            tl_inertia_mul(f[i], md.m[i], md.h[i], md.I[i], an);
            tl_inertia_mul(Iv, md.m[i], md.h[i], md.I[i], v);
            tl_crf_add(f[i], v, Iv);
#pragma unroll
            for (int e = 0; e < 6; e++) { a[e] = an[e]; vp[e] = v[e]; }
        });
    }
    TlFor<NB - 1, -1, -1>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int K = arm_tl_kind(i);
        bias[i] = f[i][2] + T(0.5) * qd[i];
        if (i > 0) {
            T fp[6];
            tl_force_to_parent<K>(fp, f[i], md.r[i], st.c[i], st.s[i]);
#pragma unroll
            for (int e = 0; e < 6; e++) f[i - 1][e] += fp[e];
        }
    });
}
template <typename T>
PDDP_HD void arm_tl_factor(const ArmTlModel<T>& md, ArmTlState<T>& st) {
    constexpr int NB = kArmNB;
    // ---- composite rigid bodies and the mass matrix (lower triangle, row-major M[i(i+1)/2 + j]) in ONE inward sweep: when level i is reached the composite of the links
    //      i..6 is complete in frame i; its column of M is read off (F = Ic_i e_z carried to the frames j < i) and the composite moves into link i - 1's frame on top of that
    //      link's own inertia.  Only two composites are ever live (until round 5 all seven were kept for a second loop over the columns: 70 numbers at the peak of the
    //      rollout kernel's register pressure); the operations and their order per element are the same.
    T M[28];
    {
        T cm = md.m[NB - 1], ch[3] = {md.h[NB - 1][0], md.h[NB - 1][1], md.h[NB - 1][2]};
        T cI[6] = {md.I[NB - 1][0], md.I[NB - 1][1], md.I[NB - 1][2], md.I[NB - 1][3], md.I[NB - 1][4], md.I[NB - 1][5]};
        TlFor<NB - 1, -1, -1>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            // F = Ic_i [e_z; 0] = [I e_z ; -h x e_z]
            T F[6] = {cI[4], cI[5], cI[2], -ch[1], ch[0], T(0)};
            M[i * (i + 1) / 2 + i] = F[2];
            TlFor<i, 0, -1>::run([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                T Fp[6];
                tl_force_to_parent<arm_tl_kind(j)>(Fp, F, md.r[j], st.c[j], st.s[j]);
#pragma unroll
                for (int e = 0; e < 6; e++) F[e] = Fp[e];
                M[i * (i + 1) / 2 + (j - 1)] = F[2];
            });
            if (i > 0) {
                T pm = md.m[i - 1], ph[3] = {md.h[i - 1][0], md.h[i - 1][1], md.h[i - 1][2]};
                T pI[6] = {md.I[i - 1][0], md.I[i - 1][1], md.I[i - 1][2], md.I[i - 1][3], md.I[i - 1][4], md.I[i - 1][5]};
                tl_inertia_add_to_parent<arm_tl_kind(i)>(pm, ph, pI, cm, ch, cI, md.r[i], st.c[i], st.s[i]);
                cm = pm;
#pragma unroll
                for (int e = 0; e < 3; e++) ch[e] = ph[e];
#pragma unroll
                for (int e = 0; e < 6; e++) cI[e] = pI[e];
            }
        });
    }
    // ---- M = L D L'
#pragma unroll
    for (int i = 0; i < NB; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            T sum = M[i * (i + 1) / 2 + j];
#pragma unroll
            for (int k = 0; k < j; k++) sum -= st.L[i * (i - 1) / 2 + k] * (st.L[j * (j - 1) / 2 + k] * M[k * (k + 1) / 2 + k]);   // M[k][k] holds D_k from here on
            if (j < i) st.L[i * (i - 1) / 2 + j] = sum * st.Dinv[j];
            else { M[i * (i + 1) / 2 + i] = sum; st.Dinv[i] = T(1) / sum; }
        }
    }
}
// Forward dynamics: qdd[7] from q[7], qd[7], u[7].  Fills st (sines, velocities, factors of M) for arm_tl_gradient.
template <typename T>
PDDP_HD void arm_tl_dynamics(const ArmTlModel<T>& md, T grav, ArmTlState<T>& st, T* qdd, const T* q, const T* qd, const T* u) {
    constexpr int NB = kArmNB;
    arm_tl_trig<T>(st, q);
    T bias[NB];
    arm_tl_bias<T>(md, grav, st, qd, bias);
    arm_tl_factor<T>(md, st);
#pragma unroll
    for (int i = 0; i < NB; i++) qdd[i] = u[i] - bias[i];
    tl_ldl_solve(st, qdd);
}

// Gradient of the forward dynamics at (q, qd, u) with qdd from arm_tl_dynamics (st as it left it).
// emit(col, row, value): dqdd(row, col), col 0..6 d/dq, 7..13 d/dqd, 14..20 d/du   (the plug-in layout s_dqdd[col*7 + row]).
// mark(stage) (an std::integral_constant): called when the columns of joints 4..6 (stage 1: columns 4..6, 11..13), then of joints 0..3 (stage 0: columns 0..3 and
// 7..10), then of the controls (stage 2: columns 14..20) have been emitted -- a caller that stages the columns somewhere small can flush in three pieces.
// The gradient in three reusable parts (arm_tl_gradient runs them for every joint; the setup kernel for few problems in flight, k_nis_tl7, gives every
// joint's columns to a different thread):
//   arm_tl_grad_nominal:  the nominal inverse dynamics at the actual qdd -- per link the acceleration a, I v, and the total force Ft through its joint
//   arm_tl_grad_joint<J>: columns J (d/dq_J) and 7 + J (d/dqd_J)         arm_tl_grad_control<J>: column 14 + J (d/du_J = column J of M^-1)
template <typename T>
struct ArmTlNominal { T a[kArmNB][6], Iv[kArmNB][6], Ft[kArmNB][6]; };
template <typename T>
PDDP_HD void arm_tl_grad_nominal(const ArmTlModel<T>& md, T grav, const ArmTlState<T>& st, const T* qd, const T* qdd, ArmTlNominal<T>& nm) {
    constexpr int NB = kArmNB;
    auto& a = nm.a; auto& Iv = nm.Iv; auto& Ft = nm.Ft;
    // ---- nominal inverse dynamics at the actual qdd: per link the acceleration, I v, and the total force through its joint
    {
        T ap[6] = {T(0), T(0), T(0), T(0), T(0), grav};
        TlFor<0, NB, 1>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const T* v = st.v[i];
            tl_motion_to_child<arm_tl_kind(i)>(a[i], ap, md.r[i], st.c[i], st.s[i]);
            a[i][0] += qd[i] * v[1]; a[i][1] -= qd[i] * v[0]; a[i][2] += qdd[i];
            a[i][3] += qd[i] * v[4]; a[i][4] -= qd[i] * v[3];
            tl_inertia_mul(Ft[i], md.m[i], md.h[i], md.I[i], a[i]);
            tl_inertia_mul(Iv[i], md.m[i], md.h[i], md.I[i], v);
            tl_crf_add(Ft[i], v, Iv[i]);
#pragma unroll
            for (int e = 0; e < 6; e++) ap[e] = a[i][e];
        });
        TlFor<NB - 1, 0, -1>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            T fp[6];
            tl_force_to_parent<arm_tl_kind(i)>(fp, Ft[i], md.r[i], st.c[i], st.s[i]);
#pragma unroll
            for (int e = 0; e < 6; e++) Ft[i - 1][e] += fp[e];
        });
    }
}
template <int j, typename T, typename Emit>
PDDP_HD void arm_tl_grad_joint(const ArmTlModel<T>& md, const ArmTlState<T>& st, const T* qd, const T* qdd, const ArmTlNominal<T>& nm, Emit emit) {
    constexpr int NB = kArmNB;
    const auto& a = nm.a; const auto& Iv = nm.Iv; const auto& Ft = nm.Ft;
    T dFq[6], dFv[6];                 // tangent of the total force through the current joint (accumulated from the tip inwards)
    T dtq[NB], dtv[NB];
    // forward over the links at and below joint j.  Only the CURRENT link's tangent motions (velocity, acceleration; d/dq_j and d/dqd_j) are kept: every link's
    // tangent FORCE  df_i = I da_i + dv_i x* (I v_i) + v_i x* (I dv_i)  is formed as soon as its motions exist and stored (12 numbers per link instead of 24) --
    // the same operations on the same operands as forming it on the way back, half the live state at the peak (joint 0: 7 links).
    T dfq[NB][6], dfv[NB][6];
    T dvq[6], daq[6], dvv[6], dav[6];
    auto link_forces = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        T t6[6];
        tl_inertia_mul(dfq[i], md.m[i], md.h[i], md.I[i], daq);
        tl_crf_add(dfq[i], dvq, Iv[i]);
        tl_inertia_mul(t6, md.m[i], md.h[i], md.I[i], dvq);
        tl_crf_add(dfq[i], st.v[i], t6);
        tl_inertia_mul(dfv[i], md.m[i], md.h[i], md.I[i], dav);
        tl_crf_add(dfv[i], dvv, Iv[i]);
        tl_inertia_mul(t6, md.m[i], md.h[i], md.I[i], dvv);
        tl_crf_add(dfv[i], st.v[i], t6);
    };
    {
        const T* v = st.v[j];
        // d(X_j v_p)/dq_j = -e_z x (X_j v_p) = -e_z x v_j (e_z x e_z = 0):  -(e_z x w) = (w_y, -w_x, 0)
        dvq[0] = v[1]; dvq[1] = -v[0]; dvq[2] = T(0); dvq[3] = v[4]; dvq[4] = -v[3]; dvq[5] = T(0);
        // X_j a_p = a_j - e_z qdd_j - v_j x (e_z qd_j)
        const T xa[6] = {a[j][0] - qd[j] * v[1], a[j][1] + qd[j] * v[0], a[j][2] - qdd[j], a[j][3] - qd[j] * v[4], a[j][4] + qd[j] * v[3], a[j][5]};
        daq[0] = xa[1] + qd[j] * dvq[1]; daq[1] = -xa[0] - qd[j] * dvq[0]; daq[2] = T(0);
        daq[3] = xa[4] + qd[j] * dvq[4]; daq[4] = -xa[3] - qd[j] * dvq[3]; daq[5] = T(0);
        // d/dqd_j: dv_j = e_z, da_j = v_j x e_z  (the e_z x e_z qd term vanishes)
        dvv[0] = T(0); dvv[1] = T(0); dvv[2] = T(1); dvv[3] = T(0); dvv[4] = T(0); dvv[5] = T(0);
        dav[0] = v[1]; dav[1] = -v[0]; dav[2] = T(0); dav[3] = v[4]; dav[4] = -v[3]; dav[5] = T(0);
        link_forces(std::integral_constant<int, j>());
    }
    TlFor<j + 1, NB, 1>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int K = arm_tl_kind(i);
        T nvq[6], naq[6], nvv[6], nav[6];
        tl_motion_to_child<K>(nvq, dvq, md.r[i], st.c[i], st.s[i]);
        tl_motion_to_child<K>(naq, daq, md.r[i], st.c[i], st.s[i]);
        naq[0] += qd[i] * nvq[1]; naq[1] -= qd[i] * nvq[0]; naq[3] += qd[i] * nvq[4]; naq[4] -= qd[i] * nvq[3];
        tl_motion_to_child<K>(nvv, dvv, md.r[i], st.c[i], st.s[i]);
        tl_motion_to_child<K>(nav, dav, md.r[i], st.c[i], st.s[i]);
        nav[0] += qd[i] * nvv[1]; nav[1] -= qd[i] * nvv[0]; nav[3] += qd[i] * nvv[4]; nav[4] -= qd[i] * nvv[3];
#pragma unroll
        for (int e = 0; e < 6; e++) { dvq[e] = nvq[e]; daq[e] = naq[e]; dvv[e] = nvv[e]; dav[e] = nav[e]; }
        link_forces(ic);
    });
    // backward: the tangent forces accumulated towards the base; joint j adds e_z x* F_j to the q tangent
    TlFor<NB - 1, -1, -1>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int K = arm_tl_kind(i);
        if (i >= j) {
            if (i == NB - 1) {
#pragma unroll
                for (int e = 0; e < 6; e++) { dFq[e] = dfq[i][e]; dFv[e] = dfv[i][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 6; e++) { dFq[e] += dfq[i][e]; dFv[e] += dfv[i][e]; }
            }
        }
        dtq[i] = dFq[2]; dtv[i] = dFv[2];
        if (i > 0) {
            if (i == j) {                          // d(X_j' F_j)/dq_j = X_j' (e_z x* F_j + dF_j):  e_z x (n; f) = (-n_y, n_x, 0; -f_y, f_x, 0)
                dFq[0] -= Ft[j][1]; dFq[1] += Ft[j][0]; dFq[3] -= Ft[j][4]; dFq[4] += Ft[j][3];
            }
            T fp[6];
            tl_force_to_parent<K>(fp, dFq, md.r[i], st.c[i], st.s[i]);
#pragma unroll
            for (int e = 0; e < 6; e++) dFq[e] = fp[e];
            tl_force_to_parent<K>(fp, dFv, md.r[i], st.c[i], st.s[i]);
#pragma unroll
            for (int e = 0; e < 6; e++) dFv[e] = fp[e];
        }
    });
    dtv[j] += T(0.5);                              // d(0.5 qd)/dqd_j
    // dqdd/dq_j = -M^-1 dtau/dq_j
#pragma unroll
    for (int i = 0; i < NB; i++) { dtq[i] = -dtq[i]; dtv[i] = -dtv[i]; }
    tl_ldl_solve(st, dtq);
    tl_ldl_solve(st, dtv);
#pragma unroll
    for (int i = 0; i < NB; i++) { emit(j, i, dtq[i]); emit(NB + j, i, dtv[i]); }
}
template <int j, typename T, typename Emit>
PDDP_HD void arm_tl_grad_control(const ArmTlState<T>& st, Emit emit) {
    constexpr int NB = kArmNB;
        T e[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) e[i] = (i == j) ? T(1) : T(0);
    tl_ldl_solve(st, e);
#pragma unroll
    for (int i = 0; i < NB; i++) emit(2 * NB + j, i, e[i]);
}
// ---- the whole gradient on one thread: composite form, O(links) composites + one short transport per (row, column) pair.
// With every vector read as a geometric object (coordinates of whichever link frame is current), J_j the axis of joint j, v_l / a_l the link motions (a_l with the base's
// upward acceleration), f_l = I_l a_l + v_l x* I_l v_l, F_k = sum_{l>=k} f_l, Ic_k = sum_{l>=k} I_l:
//   a change of q_j turns everything outboard of joint j rigidly about J_j, PLUS the same motion offsets for every outboard link l:
//       dv_l - J_j x v_l = v_j x J_j =: dv_j             da_l - J_j x a_l = c_j - v_l x dv_j,   c_j := -J_j x a_j - dv_j x v_j
//   so  d f_l / d q_j  = J_j x* f_l + I_l c_j + G_l dv_j,    G_l := d(v x* I v)/dv - I_l crm(v_l)  (a 6x6 that only reads the ANGULAR part of its argument:
//       G x = [G11 x_w ; -2 p x x_w],  G11 = w~ I - I w~ - (h u' + u h') + 2 (u.h) 1 - n~,  (n; p) = I_l v_l, v_l = (w; u) -- 12 numbers, summed over the outboard links
//       like the inertias: Gc_k),  and  d f_l / d qd_j = 2 I_l dv_j + G_l J_j.   With tau_i = J_i . F_i (the rigid turn cancels against dJ_i/dq_j):
//       i <= j:  dtau_i/dq_j = J_i . (J_j x* F_j + Ic_j c_j + Gc_j dv_j)          dtau_i/dqd_j = J_i . (2 Ic_j dv_j + Gc_j J_j)
//       i >  j:  dtau_i/dq_j = (Ic_i J_i) . c_j + (Gc_i' J_i) . dv_j               dtau_i/dqd_j = 2 (Ic_i J_i) . dv_j + (Gc_i' J_i) . J_j
// One sweep k = 6..0 keeps Ic_k, Gc_k, F_k in frame k (each moved one frame inwards per level), forms the two force vectors of column k and the two covectors of
// row k there, and carries those four (21 numbers) through the frames l = k-1..0: (6 + 6 + 6 + 3) x 21 pair transports instead of the 2 x 4 tangent recursions over
// every outboard link and the 2 x 7 force recursions of the chain form (arm_tl_grad_joint, kept for the one-joint-per-thread kernel).  Column k of dtau is complete
// when level k ends: it is solved (d qdd = -M^-1 dtau) and emitted there.  Same function as dynamicsGradient<T> (plants/dynamics_arm.cuh:2167-2289).
template <int KIND, typename T> PDDP_HD void tl_rot_to_parent(T* o, const T* x, T c, T s) {
    typedef TlZ<T, (PDDP_TLZ_MASK & 2) != 0> Z;
    const Z cz(c), sz(s), x0(x[0]), x1(x[1]);
    const Z n1[3] = {cz * x0 - sz * x1, sz * x0 + cz * x1, Z(x[2])};
    if (KIND == kTlIdZ) { o[0] = n1[0].v; o[1] = n1[1].v; o[2] = n1[2].v; }
    else if (KIND == kTlBZ) { o[0] = n1[0].v; o[1] = (-n1[2]).v; o[2] = n1[1].v; }
    else { o[0] = (-n1[0]).v; o[1] = n1[2].v; o[2] = n1[1].v; }
}
template <typename T, typename Emit, typename Mark>
PDDP_HD void arm_tl_gradient(const ArmTlModel<T>& md, T grav, const ArmTlState<T>& st, const T* qd, const T* qdd, Emit emit, Mark mark) {
    constexpr int NB = kArmNB;
    // ---- link accelerations at the actual qdd
    T a[NB][6];
    {
        T ap[6] = {T(0), T(0), T(0), T(0), T(0), grav};
        TlFor<0, NB, 1>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const T* v = st.v[i];
            typedef TlZ<T, (PDDP_TLZ_MASK & 16) != 0> Z;
            tl_motion_to_child<arm_tl_kind(i)>(a[i], ap, md.r[i], st.c[i], st.s[i]);
            a[i][0] = (Z(a[i][0]) + Z(qd[i]) * Z(v[1])).v; a[i][1] = (Z(a[i][1]) - Z(qd[i]) * Z(v[0])).v; a[i][2] = (Z(a[i][2]) + Z(qdd[i])).v;
            a[i][3] = (Z(a[i][3]) + Z(qd[i]) * Z(v[4])).v; a[i][4] = (Z(a[i][4]) - Z(qd[i]) * Z(v[3])).v;
#pragma unroll
            for (int e = 0; e < 6; e++) ap[e] = a[i][e];
        });
    }
    T dtq[NB][NB], dtv[NB][NB];          // [column][row]
    typedef TlZ<T, (PDDP_TLZ_MASK & 4) != 0> Z;                    // (exact-zero pruning: the composites start as zeros, links 0, 2, 3, 5 have h_x = I_xy = I_xz = 0, link 0 only turns about its axis)
    // composites of the links outboard of the current level, in the current frame
    T cm = T(0), ch[3] = {T(0), T(0), T(0)}, cI[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    T G[9] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)}, P[3] = {T(0), T(0), T(0)};       // Gc: G11 row-major, summed linear momentum
    T F[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    TlFor<NB - 1, -1, -1>::run([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // The link's velocity as an OPAQUE copy: I_k v_k and v_k x* I_k v_k below are the same expressions the bias recursion of the forward dynamics evaluated
        // (arm_tl_bias), and common-subexpression elimination would keep those 12 numbers per link alive from there to this level -- 60-odd registers across the whole
        // sweep, which is what the setup kernel spilled (round 5: 30 spilled registers -> PDDP_TL_OPAQUE).  Recomputing them here costs ~25 instructions per link.
        // (link 0 keeps its constants: v_0 = (0, 0, qd_0, 0, 0, 0) is what the exact-zero pruning feeds on.)
        T vk[6];
#pragma unroll
        for (int e = 0; e < 6; e++) { vk[e] = st.v[k][e]; if (PDDP_TL_OPAQUE && k > 0) tl_opaque(vk[e]); }
        const T* v = vk;
        const T* I = md.I[k]; const T* h = md.h[k];
        PDDP_TL_SCHED_FENCE();
        // ---- this link joins the composites
        {
            T Iv[6], f[6];
            tl_inertia_mul(Iv, md.m[k], h, I, v);
            tl_inertia_mul(f, md.m[k], h, I, a[k]);
            tl_crf_add(f, v, Iv);
#pragma unroll
            for (int e = 0; e < 6; e++) F[e] = (Z(F[e]) + Z(f[e])).v;
            cm = (Z(cm) + Z(md.m[k])).v;
#pragma unroll
            for (int e = 0; e < 3; e++) ch[e] = (Z(ch[e]) + Z(h[e])).v;
#pragma unroll
            for (int e = 0; e < 6; e++) cI[e] = (Z(cI[e]) + Z(I[e])).v;
            const Z Im[9] = {Z(I[0]), Z(I[3]), Z(I[4]), Z(I[3]), Z(I[1]), Z(I[5]), Z(I[4]), Z(I[5]), Z(I[2])};
            const Z w[3] = {Z(v[0]), Z(v[1]), Z(v[2])}, u[3] = {Z(v[3]), Z(v[4]), Z(v[5])}, hz[3] = {Z(h[0]), Z(h[1]), Z(h[2])};
            Z A[9];                               // w~ I
#pragma unroll
            for (int c = 0; c < 3; c++) {
                A[0 + c] = w[1] * Im[6 + c] - w[2] * Im[3 + c];
                A[3 + c] = w[2] * Im[0 + c] - w[0] * Im[6 + c];
                A[6 + c] = w[0] * Im[3 + c] - w[1] * Im[0 + c];
            }
            const Z uh2 = Z(T(2)) * (u[0] * hz[0] + u[1] * hz[1] + u[2] * hz[2]);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    Z t = A[3 * r + c] + A[3 * c + r] - hz[r] * u[c] - u[r] * hz[c];
                    if (r == c) t = t + uh2;
                    G[3 * r + c] = (Z(G[3 * r + c]) + t).v;
                }
            G[1] = (Z(G[1]) + Z(Iv[2])).v; G[2] = (Z(G[2]) - Z(Iv[1])).v; G[3] = (Z(G[3]) - Z(Iv[2])).v;                 // - n~
            G[5] = (Z(G[5]) + Z(Iv[0])).v; G[6] = (Z(G[6]) + Z(Iv[1])).v; G[7] = (Z(G[7]) - Z(Iv[0])).v;
            P[0] = (Z(P[0]) + Z(Iv[3])).v; P[1] = (Z(P[1]) + Z(Iv[4])).v; P[2] = (Z(P[2]) + Z(Iv[5])).v;
        }
        // ---- column k's force vectors, row k's covectors
        const T dv[6] = {v[1], (-Z(v[0])).v, T(0), v[4], (-Z(v[3])).v, T(0)};
        auto offsets = [&](auto lc, T* cl) {       // c_l of link l in its own frame
            constexpr int l = decltype(lc)::value;
            const T* al = a[l];
            const Z v0(st.v[l][0]), v1(st.v[l][1]), v2(st.v[l][2]), v3(st.v[l][3]), v4(st.v[l][4]), v5(st.v[l][5]);
            const Z s2 = v0 * v0 + v1 * v1, su = v0 * v3 + v1 * v4;
            cl[0] = (Z(al[1]) + v0 * v2).v; cl[1] = (-Z(al[0]) + v1 * v2).v; cl[2] = (-s2).v;
            cl[3] = (Z(al[4]) + v0 * v5 + v3 * v2).v; cl[4] = (-Z(al[3]) + v1 * v5 + v4 * v2).v; cl[5] = (-(su + su)).v;
        };
        T ck[6];
        offsets(kc, ck);
        T wq[6], wv[6];
        tl_inertia_mul(wq, cm, ch, cI, ck);
        {
            const Z d0(dv[0]), d1(dv[1]), two(T(2)), v0(v[0]), v1(v[1]);
            wq[0] = (Z(wq[0]) + (-Z(F[1]) + Z(G[0]) * d0 + Z(G[1]) * d1)).v; wq[1] = (Z(wq[1]) + (Z(F[0]) + Z(G[3]) * d0 + Z(G[4]) * d1)).v; wq[2] = (Z(wq[2]) + (Z(G[6]) * d0 + Z(G[7]) * d1)).v;
            wq[3] = (Z(wq[3]) + (-Z(F[4]) - two * (Z(P[2]) * v0))).v; wq[4] = (Z(wq[4]) + (Z(F[3]) - two * (Z(P[2]) * v1))).v; wq[5] = (Z(wq[5]) + two * (Z(P[0]) * v0 + Z(P[1]) * v1)).v;
            tl_inertia_mul(wv, cm, ch, cI, dv);
#pragma unroll
            for (int e = 0; e < 6; e++) wv[e] = (Z(wv[e]) + Z(wv[e])).v;
            wv[0] = (Z(wv[0]) + Z(G[2])).v; wv[1] = (Z(wv[1]) + Z(G[5])).v; wv[2] = (Z(wv[2]) + Z(G[8])).v;
            wv[3] = (Z(wv[3]) - two * Z(P[1])).v; wv[4] = (Z(wv[4]) + two * Z(P[0])).v;
        }
        T y[6] = {cI[4], cI[5], cI[2], (-Z(ch[1])).v, ch[0], T(0)}, gr[3] = {G[6], G[7], G[8]};
        dtq[k][k] = wq[2]; dtv[k][k] = wv[2] + T(0.5);
        TlFor<k - 1, -1, -1>::run([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            constexpr int K = arm_tl_kind(l + 1);
            T t6[6], t3[3];
            tl_force_to_parent<K>(t6, wq, md.r[l + 1], st.c[l + 1], st.s[l + 1]);
#pragma unroll
            for (int e = 0; e < 6; e++) wq[e] = t6[e];
            tl_force_to_parent<K>(t6, wv, md.r[l + 1], st.c[l + 1], st.s[l + 1]);
#pragma unroll
            for (int e = 0; e < 6; e++) wv[e] = t6[e];
            tl_force_to_parent<K>(t6, y, md.r[l + 1], st.c[l + 1], st.s[l + 1]);
#pragma unroll
            for (int e = 0; e < 6; e++) y[e] = t6[e];
            tl_rot_to_parent<K>(t3, gr, st.c[l + 1], st.s[l + 1]);
#pragma unroll
            for (int e = 0; e < 3; e++) gr[e] = t3[e];
            dtq[k][l] = wq[2]; dtv[k][l] = wv[2];
            T cl[6];
            offsets(lc, cl);
            const Z v0(st.v[l][0]), v1(st.v[l][1]), v3(st.v[l][3]), v4(st.v[l][4]);
            const Z y0(y[0]), y1(y[1]), y3(y[3]), y4(y[4]);
            dtq[l][k] = (y0 * Z(cl[0]) + y1 * Z(cl[1]) + Z(y[2]) * Z(cl[2]) + y3 * Z(cl[3]) + y4 * Z(cl[4]) + Z(y[5]) * Z(cl[5]) + (Z(gr[0]) * v1 - Z(gr[1]) * v0)).v;
            const Z yd = y0 * v1 - y1 * v0 + y3 * v4 - y4 * v3;
            dtv[l][k] = (yd + yd + Z(gr[2])).v;
        });
        // ---- column k is complete: d qdd / d(q_k, qd_k) = -M^-1 dtau
        PDDP_TL_SCHED_FENCE();
        {
            T cq[NB], cv[NB];
#pragma unroll
            for (int i = 0; i < NB; i++) { cq[i] = -dtq[k][i]; cv[i] = -dtv[k][i]; }
            tl_ldl_solve(st, cq);
            tl_ldl_solve(st, cv);
#pragma unroll
            for (int i = 0; i < NB; i++) { emit(k, i, cq[i]); emit(NB + k, i, cv[i]); }
        }
        if (k == 4) mark(std::integral_constant<int, 1>());
        if (k == 0) mark(std::integral_constant<int, 0>());
        // ---- the composites move one frame inwards
        PDDP_TL_SCHED_FENCE();
        if (k > 0) {
            constexpr int K = arm_tl_kind(k);
            const T r = md.r[k], c = st.c[k], s = st.s[k];
            T t6[6];
            tl_force_to_parent<K>(t6, F, r, c, s);
#pragma unroll
            for (int e = 0; e < 6; e++) F[e] = t6[e];
            T pm = T(0), ph[3] = {T(0), T(0), T(0)}, pI[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
            tl_inertia_add_to_parent<K>(pm, ph, pI, cm, ch, cI, r, c, s);
            cm = pm;
#pragma unroll
            for (int e = 0; e < 3; e++) ch[e] = ph[e];
#pragma unroll
            for (int e = 0; e < 6; e++) cI[e] = pI[e];
            T M1[9], M2[9], p2[3];
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {      // columns, then rows: (R G R')
                const T col[3] = {G[cc], G[3 + cc], G[6 + cc]};
                T o3[3];
                tl_rot_to_parent<K>(o3, col, c, s);
                M1[cc] = o3[0]; M1[3 + cc] = o3[1]; M1[6 + cc] = o3[2];
            }
#pragma unroll
            for (int rr = 0; rr < 3; rr++) tl_rot_to_parent<K>(M2 + 3 * rr, M1 + 3 * rr, c, s);
            tl_rot_to_parent<K>(p2, P, c, s);
            constexpr int ax = (K == kTlAY) ? 1 : 2;       // the frame's offset r lies along this parent axis:  G11 -= 2 r~ p~ = 2 (p r' - (r.p) 1)
#pragma unroll
            for (int i = 0; i < 3; i++)
                if (i != ax) { M2[3 * i + ax] -= T(2) * (r * p2[i]); M2[3 * i + i] += T(2) * (r * p2[ax]); }
#pragma unroll
            for (int e = 0; e < 9; e++) G[e] = M2[e];
#pragma unroll
            for (int e = 0; e < 3; e++) P[e] = p2[e];
        }
    });
    TlFor<0, NB, 1>::run([&](auto jc) { arm_tl_grad_control<decltype(jc)::value, T>(st, emit); });
    mark(std::integral_constant<int, 2>());
}
template <typename T, typename Emit>
PDDP_HD void arm_tl_gradient(const ArmTlModel<T>& md, T grav, const ArmTlState<T>& st, const T* qd, const T* qdd, Emit emit) {
    arm_tl_gradient<T>(md, grav, st, qd, qdd, emit, [](int) {});
}

// ------------------------------------------------------------------------------------------------ tool point of the last link (end-effector cost family)
// World frames by the same chain the dynamics walk in body coordinates: T_i = T_{i-1} F_i Rz(q_i), F_i a signed axis permutation plus a translation along ONE parent axis,
// so a link costs 12 multiply-adds for the rotation (two columns mix, the third is copied) and 3 for the origin.  Same function as compute_eePos
// (plants/dynamics_arm.cuh:1879-1925 through load_Tb / compute_T_TA_J / compute_dT_dTA_dJ); the Jacobian is taken from the world joint axes: d p_ee / d q_k = z_k x (p_ee - o_k),
// d R / d q_k = skew(z_k) R (ee_cost.hpp) -- the same numbers up to rounding, one thread per evaluation.
template <typename T>
struct ArmTlFrames {
    T z[kArmNB][3];          // world joint axes (third column of every link frame; not changed by the joint's own rotation)
    T o[kArmNB][3];          // world origins of the joint frames
    T R[9];                  // world rotation of the LAST link, column-major (R[3 c + r])
};
template <bool JAC, typename T>
PDDP_HD void arm_tl_world_chain(const ArmTlModel<T>& md, const T* c, const T* s, ArmTlFrames<T>& f) {
    T R[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}, o[3] = {T(0), T(0), T(0)};
    TlFor<0, kArmNB, 1>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int K = arm_tl_kind(i);
        // origin: the parent's origin + r along the parent's y (AY) or z axis
        constexpr int ax = (K == kTlAY) ? 1 : 2;
#pragma unroll
        for (int e = 0; e < 3; e++) o[e] = o[e] + md.r[i] * R[3 * ax + e];
        // A = R_parent R_F: columns are signed columns of the parent
        T A0[3], A1[3], A2[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            if (K == kTlIdZ) { A0[e] = R[e]; A1[e] = R[3 + e]; A2[e] = R[6 + e]; }
            else if (K == kTlBZ) { A0[e] = R[e]; A1[e] = R[6 + e]; A2[e] = -R[3 + e]; }
            else { A0[e] = -R[e]; A1[e] = R[6 + e]; A2[e] = R[3 + e]; }
        }
        // R = A Rz(q)
#pragma unroll
        for (int e = 0; e < 3; e++) { R[e] = c[i] * A0[e] + s[i] * A1[e]; R[3 + e] = c[i] * A1[e] - s[i] * A0[e]; R[6 + e] = A2[e]; }
        if (JAC) {
#pragma unroll
            for (int e = 0; e < 3; e++) { f.z[i][e] = A2[e]; f.o[i][e] = o[e]; }
        }
        if (i == kArmNB - 1) {
#pragma unroll
            for (int e = 0; e < 3; e++) f.o[i][e] = o[e];
        }
    });
#pragma unroll
    for (int e = 0; e < 9; e++) f.R[e] = R[e];
}
template <typename T> PDDP_HD T tl_atan2(T y, T x);
template <> PDDP_HD float tl_atan2<float>(float y, float x) { return atan2f(y, x); }
template <> PDDP_HD double tl_atan2<double>(double y, double x) { return atan2(y, x); }
// pos[6] = tool point (EE_ON_LINK_X = EE_ON_LINK_Y = 0, dynamics_arm.cuh:48-49) and roll / pitch / yaw (only when they carry weight: a zero weight multiplies them away exactly)
template <typename T>
PDDP_HD void arm_tl_tool_point(const ArmTlFrames<T>& f, T ee_z, bool rpy, T* pos) {
    const T* R = f.R;
#pragma unroll
    for (int e = 0; e < 3; e++) pos[e] = R[6 + e] * ee_z + f.o[kArmNB - 1][e];
    if (rpy) {
        pos[3] = tl_atan2<T>(R[5], R[8]);                             // Tee[6], Tee[10]
        pos[4] = tl_atan2<T>(-R[2], tsqrt<T>(R[5] * R[5] + R[8] * R[8]));
        pos[5] = tl_atan2<T>(R[1], R[0]);
    } else { pos[3] = T(0); pos[4] = T(0); pos[5] = T(0); }
}
// d pos / d q_k, [k][6]  (s_deePos)
template <typename T>
PDDP_HD void arm_tl_tool_jacobian(const ArmTlFrames<T>& f, T ee_z, T* dpos) {
    const T* R = f.R;
    const T f3 = R[5] * R[5] + R[8] * R[8];
    const T f4 = T(1) / (R[2] * R[2] + f3);
    const T f5 = T(1) / (R[1] * R[1] + R[0] * R[0]);
    const T sq = tsqrt<T>(f3);
    const T fac[7] = {-R[5] / f3, R[8] / f3, R[2] * R[5] * f4 / sq, R[2] * R[8] * f4 / sq, -sq * f4, -R[1] * f5, R[0] * f5};
    T pe[3];                                                           // the tool point
#pragma unroll
    for (int e = 0; e < 3; e++) pe[e] = R[6 + e] * ee_z + f.o[kArmNB - 1][e];
#pragma unroll
    for (int k = 0; k < kArmNB; k++) {
        const T* z = f.z[k];
        T arm[3], dp[3], dc0[3], dc1[3], dc2[3];
#pragma unroll
        for (int e = 0; e < 3; e++) arm[e] = pe[e] - f.o[k][e];
        cross3(dp, z, arm); cross3(dc0, z, R); cross3(dc1, z, R + 3); cross3(dc2, z, R + 6);
        T* d = dpos + 6 * k;
        d[0] = dp[0]; d[1] = dp[1]; d[2] = dp[2];
        d[3] = fac[0] * dc2[2] + fac[1] * dc1[2];
        d[4] = fac[2] * dc1[2] + fac[3] * dc2[2] + fac[4] * dc0[2];
        d[5] = fac[5] * dc0[0] + fac[6] * dc0[1];
    }
}

}  // namespace pddp

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
