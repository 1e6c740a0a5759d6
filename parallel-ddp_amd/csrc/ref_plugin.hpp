// Adapter for a plant file + cost file in the REFERENCE'S OWN plug-in form (SURVEY.md section 8b "plug-in functions a plant must provide"; config.cuh:240-252 includes
// "plants/cost_<p>.cuh" then "plants/dynamics_<p>.cuh"): the files are compiled in unchanged as plant 5 of lib/libpddp_<tag>.so, lib/libpddp_cpu_<tag>.so and the test tool
//
//     make -C parallel-ddp_amd user PLANT_FILE=<dynamics file> COST_FILE=<cost file> NUM_POS=<n> CONTROL_SIZE=<m> [NUM_TIME_STEPS=<N>] [USER_TAG=<tag>] [PLUGIN_DEFS="-D..."]
//
// What the files have to define -- the names, argument lists and layouts of the reference, as function templates over the element type:
//   dynamics<T>(T *s_qdd, T *s_x, T *s_u, T *d_I, T *d_Tbody, T *s_eePos = nullptr, int reps = 1 [, T *s_eeVel = nullptr])          plants/dynamics_arm.cuh:2097, dynamics_cart.cuh:29
//   dynamicsGradient<T>(T *s_dqdd, T *s_qdd, T *s_x, T *s_u, T *d_I, T *d_Tbody)     s_dqdd[col * NUM_POS + row], cols = q, qd, u      plants/dynamics_arm.cuh:2167, dynamics_cart.cuh:46
//   costFunc<T>(T *xk, T *uk, T *xgk, int k [, T Q1, T Q2, T R, T QF1, T QF2])                                                         plants/cost_arm.cuh:130, cost_cart.cuh:42
//   costGrad<T>(T *Hk, T *gk, T *xk, T *uk, T *xgk, int k, int ld_H [, T Q1, T Q2, T R, T QF1, T QF2])     Hk[col * ld_H + row]          plants/cost_arm.cuh:158, cost_cart.cuh:55
//   initI<T>(T *s_I), initT<T>(T *s_T)      the constant tables handed back as d_I / d_Tbody (36 * NUM_POS elements each; no-ops in the closed-form plants)   dynamics_arm.cuh:73,353
// Both generations of the cost signature are accepted: with the five trailing weights (the arm's current form) the handle's Q1, Q2, R, QF1, QF2 (pddp_config /
// pddp_set_cost) are passed at run time; the older four- / seven-argument form gets none.  costGrad may write ANY H_k (off-diagonal state, control and cross blocks):
// the setup kernels store what it writes and the backward pass reads all of it.
//
// What the adapter supplies (the part of config.cuh / utils/cudaUtils.h such files are written against):
//   NUM_POS, STATE_SIZE, CONTROL_SIZE, NUM_TIME_STEPS, EE_COST 0, MPC_MODE, USE_LIMITS_FLAG, USE_SMOOTH_ABS, USE_EE_VEL_COST           config.cuh:24-52,165-187
//   singleLoopVals, doubleLoopVals, hd__syncthreads, hd__printOnce                                                                         utils/cudaUtils.h:65-97
//   __host__ __device__ __forceinline__ (empty for the host compilers of the CPU entry points)
// with ONE-LANE semantics: the loop helpers return (start 0, stride 1) and hd__syncthreads() is empty -- the host branches of cudaUtils.h (`__CUDA_ARCH__` is never
// defined here, so `#ifdef __CUDA_ARCH__` sections of a plug-in are skipped like in the reference's CPU build).  The reference runs a plug-in with a whole thread block
// inside it; here ONE lane evaluates it, and the library parallelises over problems x knots x candidates instead: the thread-serial kernels put 64 evaluations into a
// wave once a few hundred (problem, segment) units are in flight (kernels.hpp k_fp_ts / k_nis_ts / k_bp_ts), the wave-cooperative kernels run it on lane 0 below that.
// Other helpers of utils/cudaUtils.h (matMult, loadIdentity, ...) are NOT supplied: a plug-in that uses them brings them along.
//
// NUM_TIME_STEPS is a compile-time constant of the reference's cost files (`k == NUM_TIME_STEPS - 1`, `#if NUM_TIME_STEPS == 512`): a handle of plant 5 must be created
// with pddp_config.N equal to it (PDDP_EINVAL otherwise).  A second requirement is checked at creation: the qdd that dynamicsGradient returns has to equal dynamics() bit
// for bit (the reference's plug-ins call dynamics() for it) -- the kernel families build the RK3 / midpoint stage states from one or the other.
//
// This header is included at the END of every translation unit that instantiates plant 5 (pddp_api.hip, cpu_twin.cpp, the test tool), after all library code has been
// parsed: the user's files #define what they like (R, Q1, Q2, GRAVITY, PI ...) and none of it reaches the library -- the collision that keeps the reference's own
// closed-form plug-ins from compiling against its current solver.
#pragma once
#ifdef PDDP_REF_PLANT_FILE

#include <math.h>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <type_traits>
#include <utility>

#include "plants.hpp"

#if !defined(__HIPCC__)
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
#endif
#endif

// ---- the environment of config.cuh
#define NUM_POS PDDP_REF_NUM_POS
#define STATE_SIZE (2 * NUM_POS)
#define CONTROL_SIZE PDDP_REF_CONTROL_SIZE
#ifndef NUM_TIME_STEPS
#define NUM_TIME_STEPS PDDP_REF_NUM_TIME_STEPS
#endif
#ifndef EE_COST
#define EE_COST 0
#endif
#if EE_COST
#error "the reference-form plug-in carries the joint-space cost family (EE_COST 0): the end-effector family needs the arm's kinematics (plant 4, pddp_config.ee_cost)"
#endif
#ifndef MPC_MODE
#define MPC_MODE 0
#endif
#ifndef USE_LIMITS_FLAG
#define USE_LIMITS_FLAG 0
#endif
#ifndef USE_SMOOTH_ABS
#define USE_SMOOTH_ABS 0
#endif
#ifndef USE_EE_VEL_COST
#define USE_EE_VEL_COST 0
#endif

// ---- everything of the adapter that mentions the library's own field names comes BEFORE the user's files (cw.Q2, cw.R would be rewritten by a cost file's `#define Q2`, `#define R`)
namespace pddp {
template <typename T> PDDP_HD void ref_plugin_weights(const CostWeights<T>& cw, T* five) { five[0] = cw.Q1; five[1] = cw.Q2; five[2] = cw.R; five[3] = cw.QF1; five[4] = cw.QF2; }
// host copy of the tables (filled once per library and element type) and the set of DEVICES whose copy of the __device__ symbol has been written: a __device__ variable
// exists once per device, so a handle created on a second device needs its own upload (with one process-wide flag such a handle read all-zero tables)
template <typename T> struct RefPluginHost { static RefPluginTables<T> tab; static bool filled; static std::set<int> uploaded; static std::mutex lock; };
template <typename T> RefPluginTables<T> RefPluginHost<T>::tab;
template <typename T> bool RefPluginHost<T>::filled = false;
template <typename T> std::set<int> RefPluginHost<T>::uploaded;
template <typename T> std::mutex RefPluginHost<T>::lock;
#if defined(__HIPCC__)
template <typename T> __device__ RefPluginTables<T> g_ref_plugin_tables;
#endif
// the tables of initI / initT: filled once per library and element type on the host, mirrored into device memory
template <typename T> PDDP_HD const RefPluginTables<T>& ref_plugin_tables() {
#if defined(__HIP_DEVICE_COMPILE__)
    return g_ref_plugin_tables<T>;
#else
    return RefPluginHost<T>::tab;
#endif
}
// uploads to the CURRENT device (pddp_create has already made the handle's device current) unless that device has its copy; caller holds RefPluginHost<T>::lock
template <typename T> inline bool ref_plugin_upload_tables() {
#if defined(__HIPCC__)
    int pddp_dev_ = -1;
    if (hipGetDevice(&pddp_dev_) != hipSuccess) return false;
    if (RefPluginHost<T>::uploaded.count(pddp_dev_)) return true;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ref_plugin_tables<T>), &RefPluginHost<T>::tab, sizeof(RefPluginTables<T>)) != hipSuccess) return false;
    RefPluginHost<T>::uploaded.insert(pddp_dev_);
#endif
    return true;
}
inline std::string ref_plugin_horizon_complaint(int pddp_n_) {
    return "plant 5 was compiled from a reference-form cost file with NUM_TIME_STEPS = " + std::to_string(kUserPlantN) + " (a compile-time constant of such files): create the handle with N = " +
           std::to_string(kUserPlantN) + " or rebuild with `make user ... NUM_TIME_STEPS=" + std::to_string(pddp_n_) + "`";
}
// pseudo-random probe values in [-scale, scale] for the creation-time check
inline double ref_plugin_probe_value(unsigned& pddp_s_, double pddp_scale_) { pddp_s_ = pddp_s_ * 1664525u + 1013904223u; return ((double)(pddp_s_ >> 8) / 8388608.0 - 1.0) * pddp_scale_; }
}  // namespace pddp

namespace pddp_ref_plugin {
// utils/cudaUtils.h:65-97, host branches: one lane is inside the plug-in
__host__ __device__ __forceinline__ void doubleLoopVals(int* starty, int* dy, int* startx, int* dx) { *starty = 0; *dy = 1; *startx = 0; *dx = 1; }
__host__ __device__ __forceinline__ void singleLoopVals(int* start, int* delta) { *start = 0; *delta = 1; }
__host__ __device__ __forceinline__ void hd__syncthreads() {}
template <int tx, int ty, int bx, int by> __host__ __device__ __forceinline__ int hd__printOnce() { return 1; }

#include PDDP_REF_COST_FILE
#include PDDP_REF_PLANT_FILE
}  // namespace pddp_ref_plugin

// ---- from here on the user's macros are live: only identifiers with the pddp_ prefix (and the names this header declared above) appear below
namespace pddp {

template <typename T> PDDP_HD void ref_plugin_dynamics(T* pddp_qdd_, const T* pddp_x_, const T* pddp_u_) {
    const RefPluginTables<T>& pddp_t_ = ref_plugin_tables<T>();
    pddp_ref_plugin::dynamics<T>(pddp_qdd_, const_cast<T*>(pddp_x_), const_cast<T*>(pddp_u_), const_cast<T*>(pddp_t_.pddp_tab_I), const_cast<T*>(pddp_t_.pddp_tab_T));
}
template <typename T> PDDP_HD void ref_plugin_gradient(T* pddp_dqdd_, T* pddp_qdd_, const T* pddp_x_, const T* pddp_u_) {
    const RefPluginTables<T>& pddp_t_ = ref_plugin_tables<T>();
    pddp_ref_plugin::dynamicsGradient<T>(pddp_dqdd_, pddp_qdd_, const_cast<T*>(pddp_x_), const_cast<T*>(pddp_u_), const_cast<T*>(pddp_t_.pddp_tab_I), const_cast<T*>(pddp_t_.pddp_tab_T));
}

// which generation of the cost signature the file has (the five weights as trailing arguments, or none)
namespace ref_plugin_detail {
template <typename T, typename = void> struct cost_takes_weights : std::false_type {};
template <typename T>
struct cost_takes_weights<T, std::void_t<decltype(pddp_ref_plugin::costFunc<T>(std::declval<T*>(), std::declval<T*>(), std::declval<T*>(), 0, std::declval<T>(), std::declval<T>(),
                                                                                std::declval<T>(), std::declval<T>(), std::declval<T>()))>> : std::true_type {};
template <typename T, typename = void> struct grad_takes_weights : std::false_type {};
template <typename T>
struct grad_takes_weights<T, std::void_t<decltype(pddp_ref_plugin::costGrad<T>(std::declval<T*>(), std::declval<T*>(), std::declval<T*>(), std::declval<T*>(), std::declval<T*>(), 0, 0,
                                                                                std::declval<T>(), std::declval<T>(), std::declval<T>(), std::declval<T>(), std::declval<T>()))>> : std::true_type {};
}  // namespace ref_plugin_detail

template <typename T> PDDP_HD T ref_plugin_cost(const CostWeights<T>& pddp_cw_, const T* pddp_xk_, const T* pddp_uk_, const T* pddp_xg_, int pddp_k_) {
    T* pddp_x_ = const_cast<T*>(pddp_xk_); T* pddp_u_ = const_cast<T*>(pddp_uk_); T* pddp_g_ = const_cast<T*>(pddp_xg_);
    if constexpr (ref_plugin_detail::cost_takes_weights<T>::value) {
        T pddp_w_[5]; ref_plugin_weights<T>(pddp_cw_, pddp_w_);
        return pddp_ref_plugin::costFunc<T>(pddp_x_, pddp_u_, pddp_g_, pddp_k_, pddp_w_[0], pddp_w_[1], pddp_w_[2], pddp_w_[3], pddp_w_[4]);
    } else return pddp_ref_plugin::costFunc<T>(pddp_x_, pddp_u_, pddp_g_, pddp_k_);
}
template <typename T> PDDP_HD void ref_plugin_cost_grad(const CostWeights<T>& pddp_cw_, T* pddp_Hk_, T* pddp_gk_, const T* pddp_xk_, const T* pddp_uk_, const T* pddp_xg_, int pddp_k_, int pddp_ld_) {
    T* pddp_x_ = const_cast<T*>(pddp_xk_); T* pddp_u_ = const_cast<T*>(pddp_uk_); T* pddp_g_ = const_cast<T*>(pddp_xg_);
    if constexpr (ref_plugin_detail::grad_takes_weights<T>::value) {
        T pddp_w_[5]; ref_plugin_weights<T>(pddp_cw_, pddp_w_);
        pddp_ref_plugin::costGrad<T>(pddp_Hk_, pddp_gk_, pddp_x_, pddp_u_, pddp_g_, pddp_k_, pddp_ld_, pddp_w_[0], pddp_w_[1], pddp_w_[2], pddp_w_[3], pddp_w_[4]);
    } else pddp_ref_plugin::costGrad<T>(pddp_Hk_, pddp_gk_, pddp_x_, pddp_u_, pddp_g_, pddp_k_, pddp_ld_);
}

// Called when a handle of plant 5 is created (declared in plants.hpp): fills the tables, checks the two requirements of the header comment.  Returns "" or what is wrong.
template <typename T> std::string ref_plugin_setup(int pddp_n_) {
    if (pddp_n_ != kUserPlantN) return ref_plugin_horizon_complaint(pddp_n_);
    {
        std::lock_guard<std::mutex> pddp_guard_(RefPluginHost<T>::lock);
        if (!RefPluginHost<T>::filled) {
            RefPluginTables<T>& pddp_t_ = RefPluginHost<T>::tab;
            std::memset(&pddp_t_, 0, sizeof(pddp_t_));
            pddp_ref_plugin::initI<T>(pddp_t_.pddp_tab_I);
            pddp_ref_plugin::initT<T>(pddp_t_.pddp_tab_T);
            RefPluginHost<T>::filled = true;
        }
        if (!ref_plugin_upload_tables<T>()) return "hipMemcpyToSymbol of the plug-in's initI / initT tables failed";
    }
    // the qdd of dynamicsGradient against dynamics(), on the host instantiation, at a few states spread over the unit box and beyond
    for (int pddp_trial_ = 0; pddp_trial_ < 8; pddp_trial_++) {
        T pddp_x_[2 * kUserPlantNPOS], pddp_u_[kUserPlantNU > 0 ? kUserPlantNU : 1], pddp_q1_[kUserPlantNPOS], pddp_q2_[kUserPlantNPOS], pddp_d_[kUserPlantNPOS * (2 * kUserPlantNPOS + kUserPlantNU)];
        unsigned pddp_s_ = 12345u + 977u * (unsigned)pddp_trial_;
        for (int pddp_i_ = 0; pddp_i_ < 2 * kUserPlantNPOS; pddp_i_++) pddp_x_[pddp_i_] = (T)ref_plugin_probe_value(pddp_s_, pddp_trial_ < 4 ? 1.0 : 3.0);
        for (int pddp_i_ = 0; pddp_i_ < kUserPlantNU; pddp_i_++) pddp_u_[pddp_i_] = (T)ref_plugin_probe_value(pddp_s_, pddp_trial_ < 4 ? 1.0 : 3.0);
        for (int pddp_i_ = 0; pddp_i_ < kUserPlantNPOS; pddp_i_++) { pddp_q1_[pddp_i_] = T(0); pddp_q2_[pddp_i_] = T(0); }
        ref_plugin_dynamics<T>(pddp_q1_, pddp_x_, pddp_u_);
        ref_plugin_gradient<T>(pddp_d_, pddp_q2_, pddp_x_, pddp_u_);
        if (std::memcmp(pddp_q1_, pddp_q2_, sizeof(pddp_q1_)) != 0)
            return "the plug-in's dynamicsGradient returns a qdd that differs from dynamics() at the same state (the reference's plug-ins call dynamics() for it; the kernel "
                   "families build integrator stage states from either one, so the two have to be the same numbers)";
    }
    return std::string();
}

}  // namespace pddp
#endif  // PDDP_REF_PLANT_FILE
