// Plant plug-in surface of the kernels.  A plant is a policy struct (compile-time, like the reference's
// `#define PLANT` + config.cuh:240-252 include switch) providing
//     dims NPOS/NX/NU, Model (device constants), Scratch/GradScratch (per-wave LDS),
//     dynamics(), gradient()   -- wave-cooperative, the whole wave is inside the call (the reference's plug-ins
//                                 run with the whole thread block inside them),
//     cost(), cost_grad()      -- single-lane, per knot.
// Plants: 1 pendulum, 2 cart-pole, 3 quadrotor, 4 KUKA iiwa14 (reference PLANT codes, config.cuh:21-61).
#pragma once

#include "plant_arm.hpp"
#include "plant_closed_form.hpp"

namespace pddp {

template <typename T>
struct CostWeights {   // arm joint-space weights (plants/cost_arm.cuh:97-103); other plants use fixed macros
    T Q1, Q2, R, QF1, QF2;
    // end-effector cost family (EE_COST 1, plants/cost_arm.cuh:206-389; ee_cost.hpp): xyz / rpy running and final weights, control weight,
    // nominal-state weights on q and qd, and the tool point's offset along the last link's z axis (EE_ON_LINK_Z, dynamics_arm.cuh:53-65)
    int ee;
    T Q_EE1, Q_EE2, QF_EE1, QF_EE2, R_EE, Q_xEE, QF_xEE, Q_xdEE, QF_xdEE, ee_z;
    int limits;             // USE_LIMITS_FLAG: arm_limit_term below (joint-space cost: cost and gradient; end-effector cost: also the diagonal of H)
    int smooth_abs; T sa2;  // USE_SMOOTH_ABS (end-effector cost): flag and SMOOTH_ABS_ALPHA^2; sa = alpha itself
    T sa;
    double fd_eps;          // > 0: USE_FINITE_DIFF with this FINITE_DIFF_EPSILON (config.cuh:68-71): [A B] by central differences (integrators.hpp); rides here because every setup kernel takes this record
};

// USE_LIMITS_FLAG (plants/cost_arm.cuh:13-94): limitCosts<T, dLevel>(s_x, s_u, ind, k) = qr * quadPen<T, dLevel>(val, limit) -- qr = Q_PL = Q_VL = R_TL = 100, the
// limits of the iiwa scaled by the safety factors 0.8 (getPosLimit / getVelLimit / getTorqueLimit), quadPen = 0 inside the limit, else 0.5 d^2 | sign(val) d | 1
// with d = |val| - limit.  ind: 0..6 positions, 7..13 velocities, 14..20 torques.
template <typename T>
PDDP_HD T arm_limit_term(const T* xk, const T* uk, int ind, int dlevel) {
    T val, limit;
    if (ind < 7) { val = xk[ind]; limit = (T)(ind == 6 ? (3.05432619099 * 0.8) : ((ind % 2) ? (2.09439510239 * 0.8) : (2.96705972839 * 0.8))); }
    else if (ind < 14) {
        const int j = ind - 7;
        val = xk[ind];
        limit = (T)(j > 4 ? (2.356194 * 0.8) : (j == 4 ? (2.268928 * 0.8) : (j == 3 ? (1.308996 * 0.8) : (j == 2 ? (1.745329 * 0.8) : (1.483529 * 0.8)))));
    } else { val = uk[ind - 14]; limit = (T)(300.0 * 0.8); }
    const T delta = (val < T(0) ? -val : val) - limit;
    if (delta < T(0)) return T(100.0) * T(0);
    if (dlevel == 0) return T(100.0) * (T)(0.5 * delta * delta);
    if (dlevel == 1) return T(100.0) * (val < T(0) ? -delta : delta);
    return T(100.0) * T(1);
}

struct EmptyModel { int unused; };
template <typename T> struct EmptyScratch { T unused; };

// ---- older 4-argument diagonal costs of pend / cart / quad (plants/cost_{pend,cart,quad}.cuh) ------------
// weight(i): running weight of state/control index i; QF: final weight.  pow(T,int) promotes to double on the
// host path the oracle restates; we keep the square in double for both precisions.
template <typename P, typename T>
PDDP_HD T diag_cost(const T* xk, const T* uk, const T* xg, int k, int N) {
    T cost = 0.0;
    for (int i = 0; i < P::NX; i++) { const double dl = xk[i] - xg[i]; cost += (T)(k == N - 1 ? P::QF(N) : P::QR(i, N)) * (dl * dl); }
    if (k != N - 1) for (int i = 0; i < P::NU; i++) { const double uu = uk[i]; cost += (T)P::Rw(N) * (uu * uu); }
    return 0.5 * cost;
}
template <typename P, typename T>
PDDP_HD void diag_cost_grad(T* Hk, T* gk, const T* xk, const T* uk, const T* xg, int k, int N) {
    constexpr int NM = P::NX + P::NU;
    for (int i = 0; i < NM; i++) for (int j = 0; j < NM; j++)
        Hk[i * NM + j] = (T)(i != j ? 0.0 : (k == N - 1 ? (i < P::NX ? P::QF(N) : 0.0) : P::QR(i, N)));
    for (int i = 0; i < P::NX; i++) gk[i] = (T)(k == N - 1 ? P::QF(N) : P::QR(i, N)) * (xk[i] - xg[i]);
    for (int i = 0; i < P::NU; i++) gk[i + P::NX] = (T)(k == N - 1 ? 0.0 : P::Rw(N)) * uk[i];
}

#define PDDP_CLOSED_FORM_PLANT(NAME, CODE, NP, NUv, EVAL, GRAD)                                                          \
    template <typename T>                                                                                                \
    struct NAME {                                                                                                        \
        static constexpr int PLANT = CODE, NPOS = NP, NX = 2 * NP, NU = NUv;                                             \
        template <typename U> using Rebind = NAME<U>;                                                                    \
        using Model = EmptyModel;                                                                                        \
        using Scratch = EmptyScratch<T>;                                                                                 \
        using GradScratch = EmptyScratch<T>;                                                                             \
        static PDDP_HD void load_model(const Wave&, Scratch&, const Model*) {}                                           \
        static PDDP_HD void dynamics(const Wave& w, Scratch&, T* qdd, const T* x, const T* u) {                          \
            if (w.lane == 0) EVAL<T>(qdd, x, u);                                                                         \
            wsync();                                                                                                     \
        }                                                                                                                \
        static PDDP_HD void gradient(const Wave& w, Scratch&, GradScratch&, T* dqdd, T* qdd, const T* x, const T* u) {   \
            if (w.lane == 0) GRAD<T>(dqdd, qdd, x, u);                                                                   \
            wsync();                                                                                                     \
        }                                                                                                                \
        /* the plug-in's functions are scalar code: ONE lane evaluates them.  dynamics_on / gradient_eval let a caller put independent \
           evaluations on different lanes of the cooperating set in ONE pass of the code (integrator_gradient: the three stage gradients of RK3) */ \
        static constexpr bool kScalarPlugin = true;                                                                      \
        static constexpr bool kPluginCost = false;   /* H_k is the diagonal of weight(): the setup kernels build it from that */ \
        static PDDP_HD void dynamics_on(const Wave& w, int lane, T* qdd, const T* x, const T* u) { if (w.lane == lane) EVAL<T>(qdd, x, u); } \
        static PDDP_HD void gradient_eval(T* dqdd, T* qdd, const T* x, const T* u) { GRAD<T>(dqdd, qdd, x, u); }       \
        static PDDP_HD void dynamics_eval(T* qdd, const T* x, const T* u) { EVAL<T>(qdd, x, u); }                       \
        static PDDP_HD T cost(const CostWeights<T>&, const T* xk, const T* uk, const T* xg, int k, int N) {              \
            return diag_cost<NAME<T>, T>(xk, uk, xg, k, N);                                                              \
        }                                                                                                                \
        static PDDP_HD void cost_grad(const CostWeights<T>&, T* Hk, T* gk, const T* xk, const T* uk, const T* xg, int k, \
                                      int N) {                                                                           \
            diag_cost_grad<NAME<T>, T>(Hk, gk, xk, uk, xg, k, N);                                                        \
        }                                                                                                                \
        /* diagonal weight of state/control index i at knot k (H_k = diag(weight), g_k = weight .* [x-xg; u]) */          \
        static PDDP_HD T weight(const CostWeights<T>&, int i, int k, int N) {                                            \
            return (T)(k == N - 1 ? (i < NX ? QF(N) : 0.0) : QR(i, N));                                                  \
        }                                                                                                                \
        static PDDP_HD double QR(int i, int N);                                                                          \
        static PDDP_HD double Rw(int N);                                                                                 \
        static PDDP_HD double QF(int N);                                                                                 \
    };

PDDP_CLOSED_FORM_PLANT(PendPlant, 1, 1, 1, pend_dynamics_eval, pend_gradient_eval)
PDDP_CLOSED_FORM_PLANT(CartPlant, 2, 2, 1, cart_dynamics_eval, cart_gradient_eval)
PDDP_CLOSED_FORM_PLANT(QuadPlant, 3, 6, 4, quad_dynamics_eval, quad_gradient_eval)
}  // namespace pddp
// A user plant compiled in as plant 5 (make user PLANT_POLICY=<header>; examples/plants/damped_pendulum.hpp documents what the header provides).
#if defined(PDDP_USER_PLANT_HEADER) && !defined(PDDP_REF_PLANT_FILE)
#include PDDP_USER_PLANT_HEADER
namespace pddp {
PDDP_CLOSED_FORM_PLANT(UserPlant, 5, kUserPlantNPOS, kUserPlantNU, user_plant_dynamics, user_plant_gradient)
template <typename T> PDDP_HD double UserPlant<T>::QR(int i, int N) { return user_plant_QR(i, N); }
template <typename T> PDDP_HD double UserPlant<T>::Rw(int N) { return user_plant_R(N); }
template <typename T> PDDP_HD double UserPlant<T>::QF(int N) { return user_plant_QF(N); }
}  // namespace pddp
#endif
// A plant file + cost file in the REFERENCE'S OWN FORM compiled in as plant 5 (make user PLANT_FILE=... COST_FILE=... NUM_POS=... CONTROL_SIZE=...; csrc/ref_plugin.hpp holds
// the adapter and documents the contract): `dynamics`, `dynamicsGradient`, `costFunc`, `costGrad`, `initI`, `initT` with the reference's signatures
// (plants/dynamics_arm.cuh:2097,2167, plants/cost_arm.cuh:130,158 and the older forms of plants/{dynamics,cost}_{pend,cart,quad}.cuh).  The user's files are included at
// the END of the translation unit (their `#define R`, `Q1`, `GRAVITY` ... must not reach the library's own code: the reference's stale closed-form plug-ins fail to
// compile against its current solver for exactly that reason), so only declarations appear here.
#ifdef PDDP_REF_PLANT_FILE
#ifdef PDDP_USER_PLANT_HEADER
#error "PLANT_POLICY (a policy header) and PLANT_FILE / COST_FILE (a reference-form plug-in) are two ways to build plant 5: give one"
#endif
#define PDDP_USER_PLANT_HEADER "ref_plugin_decl.hpp"    /* switches on every `plant 5` site of the library */
namespace pddp {
constexpr int kUserPlantNPOS = PDDP_REF_NUM_POS, kUserPlantNU = PDDP_REF_CONTROL_SIZE, kUserPlantN = PDDP_REF_NUM_TIME_STEPS;
template <typename T> struct RefPluginTables { T pddp_tab_I[36 * kUserPlantNPOS], pddp_tab_T[36 * kUserPlantNPOS]; };     // what initI / initT fill (d_I, d_Tbody of the reference's entry points)
template <typename T> PDDP_HD void ref_plugin_dynamics(T* qdd, const T* x, const T* u);
template <typename T> PDDP_HD void ref_plugin_gradient(T* dqdd, T* qdd, const T* x, const T* u);
template <typename T> PDDP_HD T ref_plugin_cost(const CostWeights<T>& cw, const T* xk, const T* uk, const T* xg, int k);
template <typename T> PDDP_HD void ref_plugin_cost_grad(const CostWeights<T>& cw, T* Hk, T* gk, const T* xk, const T* uk, const T* xg, int k, int ld_H);
}  // namespace pddp
#include <string>
namespace pddp {
template <typename T> std::string ref_plugin_setup(int N);     // handle creation: fills the tables of initI / initT, checks N and the plug-in's qdd (ref_plugin.hpp); "" or the complaint
template <typename T>
struct UserPlant {
    static constexpr int PLANT = 5, NPOS = kUserPlantNPOS, NX = 2 * kUserPlantNPOS, NU = kUserPlantNU;
    template <typename U> using Rebind = UserPlant<U>;
    using Model = EmptyModel;                             // the tables of initI / initT are per-library constants (ref_plugin.hpp), not per-handle state
    using Scratch = EmptyScratch<T>;
    using GradScratch = EmptyScratch<T>;
    // ONE lane is inside the plug-in: singleLoopVals / doubleLoopVals hand it (start 0, stride 1) and hd__syncthreads is empty -- the host branches of
    // utils/cudaUtils.h:65-88.  Many problems in flight put 64 such lanes into a wave (the thread-serial kernels), a few run a wave per unit with its lane 0 inside.
    static constexpr bool kScalarPlugin = true;
    static constexpr bool kPluginCost = true;             // H_k, g_k come from the user's costGrad (any symmetric H_k, not only a diagonal): the full-H backward pass reads them
    static PDDP_HD void load_model(const Wave&, Scratch&, const Model*) {}
    static PDDP_HD void dynamics(const Wave& w, Scratch&, T* qdd, const T* x, const T* u) { if (w.lane == 0) ref_plugin_dynamics<T>(qdd, x, u); wsync(); }
    static PDDP_HD void gradient(const Wave& w, Scratch&, GradScratch&, T* dqdd, T* qdd, const T* x, const T* u) { if (w.lane == 0) ref_plugin_gradient<T>(dqdd, qdd, x, u); wsync(); }
    static PDDP_HD void dynamics_on(const Wave& w, int lane, T* qdd, const T* x, const T* u) { if (w.lane == lane) ref_plugin_dynamics<T>(qdd, x, u); }
    static PDDP_HD void gradient_eval(T* dqdd, T* qdd, const T* x, const T* u) { ref_plugin_gradient<T>(dqdd, qdd, x, u); }
    static PDDP_HD void dynamics_eval(T* qdd, const T* x, const T* u) { ref_plugin_dynamics<T>(qdd, x, u); }
    static PDDP_HD T cost(const CostWeights<T>& cw, const T* xk, const T* uk, const T* xg, int k, int) { return ref_plugin_cost<T>(cw, xk, uk, xg, k); }
    // H_k is (NX+NU)^2 column-major with leading dimension NX+NU.  The reference's final-knot costGrad writes the state block only (plants/cost_arm.cuh:159-174): the
    // block is cleared first so that the array is deterministic.
    static PDDP_HD void cost_grad(const CostWeights<T>& cw, T* Hk, T* gk, const T* xk, const T* uk, const T* xg, int k, int) {
        constexpr int NM = NX + NU;
        for (int e = 0; e < NM * NM; e++) Hk[e] = T(0);
        for (int e = 0; e < NM; e++) gk[e] = T(0);
        ref_plugin_cost_grad<T>(cw, Hk, gk, xk, uk, xg, k, NM);
    }
    static PDDP_HD T weight(const CostWeights<T>&, int, int, int) { return T(0); }       // (not used: kPluginCost)
};
}  // namespace pddp
#endif
namespace pddp {
#undef PDDP_CLOSED_FORM_PLANT

// cost_pend.cuh:19-24 (QR(1) falls through to R, QR(2) is the control's Q2)
template <typename T> PDDP_HD double PendPlant<T>::QR(int i, int) { return i == 0 ? 1.0 : (i == 2 ? 0.1 : 0.1); }
template <typename T> PDDP_HD double PendPlant<T>::Rw(int) { return 0.1; }
template <typename T> PDDP_HD double PendPlant<T>::QF(int) { return 1000.0; }
// cost_cart.cuh:19-38 (weights depend on the horizon length)
template <typename T> PDDP_HD double CartPlant<T>::QR(int i, int N) {
    if (N == 512) return i == 0 ? 0.01 : (i == 1 ? 0.01 : (i < 4 ? 0.01 : 0.001));
    return i == 0 ? 0.01 : (i == 1 ? 0.01 : (i < 4 ? 0.001 : 0.0001));
}
template <typename T> PDDP_HD double CartPlant<T>::Rw(int N) { return N == 512 ? 0.001 : 0.0001; }
template <typename T> PDDP_HD double CartPlant<T>::QF(int N) { return N == 512 ? 100000.0 : 1000.0; }
// cost_quad.cuh:19-25
template <typename T> PDDP_HD double QuadPlant<T>::QR(int i, int) { return i < 3 ? 0.01 : (i < 6 ? 0.001 : (i < 9 ? 2.0 : (i < 12 ? 2.0 : 5.0))); }
template <typename T> PDDP_HD double QuadPlant<T>::Rw(int) { return 5.0; }
template <typename T> PDDP_HD double QuadPlant<T>::QF(int) { return 1000.0; }

// A scalar plug-in's gradient routine also returns qdd; the kernel families build the midpoint / RK3 stage states from that qdd (serial paths) or from the dynamics routine's
// (the staged three-lane path of integrators.hpp), so the two have to be the SAME numbers -- the reference's plug-ins call dynamics() inside dynamicsGradient.  Checked on the
// host instantiation when a handle of a user plant is created.
template <typename P, typename T>
inline bool scalar_plugin_qdd_is_dynamics() {
    for (int trial = 0; trial < 8; trial++) {
        T x[P::NX], u[P::NU > 0 ? P::NU : 1], q1[P::NPOS], q2[P::NPOS], d[P::NPOS * (P::NX + P::NU)];
        unsigned s = 12345u + 977u * (unsigned)trial;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (T)(((double)(s >> 8) / 8388608.0 - 1.0) * (trial < 4 ? 1.0 : 3.0)); };
        for (int i = 0; i < P::NX; i++) x[i] = rnd();
        for (int i = 0; i < P::NU; i++) u[i] = rnd();
        for (int i = 0; i < P::NPOS; i++) { q1[i] = T(0); q2[i] = T(0); }
        P::dynamics_eval(q1, x, u);
        P::gradient_eval(d, q2, x, u);
        for (int i = 0; i < P::NPOS; i++) if (!(q1[i] == q2[i])) return false;
    }
    return true;
}

// ---- KUKA arm, joint-space cost (plants/cost_arm.cuh:130-153, 158-202) --------------------------------------
template <typename T>
struct ArmPlant {
    static constexpr int PLANT = 4, NPOS = 7, NX = 14, NU = 7;
    template <typename U> using Rebind = ArmPlant<U>;     // the same plant in another precision (the lock-step simulator runs it in double)
    using Model = ArmModel<T>;
    using Scratch = ArmScratch<T>;
    using GradScratch = ArmGradScratch<T>;
    static constexpr bool kScalarPlugin = false;          // dynamics / gradient are cooperative over the whole set
    static constexpr bool kPluginCost = false;
    static PDDP_HD void load_model(const Wave& w, Scratch& s, const Model* m) { arm_load_model(w, s, m); }
    static PDDP_HD void dynamics(const Wave& w, Scratch& s, T* qdd, const T* x, const T* u) { arm_dynamics(w, s, qdd, x, u); }
    static PDDP_HD void gradient(const Wave& w, Scratch& s, GradScratch& g, T* dqdd, T* qdd, const T* x, const T* u) {
        arm_dynamics_gradient(w, s, g, dqdd, qdd, x, u);
    }
    static PDDP_HD T weight(const CostWeights<T>& cw, int i, int k, int N) {
        if (k == N - 1) return i < NPOS ? cw.QF1 : (i < NX ? cw.QF2 : T(0));
        return i < NPOS ? cw.Q1 : (i < NX ? cw.Q2 : cw.R);
    }
    static PDDP_HD T cost(const CostWeights<T>& cw, const T* xk, const T* uk, const T* xg, int k, int N) {
        T cost = 0;
        if (k == N - 1) {
            for (int i = 0; i < NX; i++) { const T dl = xk[i] - xg[i]; cost += (i < NPOS ? cw.QF1 : cw.QF2) * dl * dl; }
        } else {
            for (int i = 0; i < NX; i++) { const T dl = xk[i] - xg[i]; cost += (i < NPOS ? cw.Q1 : cw.Q2) * dl * dl; }
            for (int i = 0; i < NU; i++) cost += cw.R * uk[i] * uk[i];
        }
        cost = T(0.5) * cost;
        if (cw.limits) { const int n = (k == N - 1) ? NX : NX + NU; for (int i = 0; i < n; i++) cost += arm_limit_term<T>(xk, uk, i, 0); }       // cost_arm.cuh:136-139,147-150
        return cost;
    }
    // H_k is (NX+NU)^2 column-major; at the final knot only the NX x NX block and g are defined
    // (cost_arm.cuh:159-174) -- we still write zeros to the rest so the buffer is deterministic.
    static PDDP_HD void cost_grad(const CostWeights<T>& cw, T* Hk, T* gk, const T* xk, const T* uk, const T* xg, int k, int N) {
        constexpr int NM = NX + NU;
        const bool fin = (k == N - 1);
        for (int i = 0; i < NM; i++) for (int j = 0; j < NM; j++) {
            T v = 0;
            if (i == j) v = fin ? (i < NPOS ? cw.QF1 : (i < NX ? cw.QF2 : T(0))) : (i < NPOS ? cw.Q1 : (i < NX ? cw.Q2 : cw.R));
            Hk[i * NM + j] = v;
        }
        for (int i = 0; i < NX; i++) gk[i] = (fin ? (i < NPOS ? cw.QF1 : cw.QF2) : (i < NPOS ? cw.Q1 : cw.Q2)) * (xk[i] - xg[i]);
        for (int i = 0; i < NU; i++) gk[i + NX] = fin ? T(0) : cw.R * uk[i];
    }
};

}  // namespace pddp
