// TEST TOOL -- NOT PRODUCT CODE.  A user's plant file + cost file in the reference's plug-in form, compiled for the HOST as the callbacks of the oracle's plant 5
// (oracle/oracle.h ora_set_plugin_*): the oracle restates the reference's SOLVER, the plug-in is an input to both sides of a parity test.  The environment is the one
// the reference's CPU build gives such files (utils/cudaUtils.h:65-88 host branches: loop helpers (0, 1), empty hd__syncthreads; config.cuh:24-52 dimensions) -- written
// here independently of the product's adapter (parallel-ddp_amd/csrc/ref_plugin.hpp), with which it shares nothing.
//   g++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off -DPLUGIN_PLANT_FILE='"..."' -DPLUGIN_COST_FILE='"..."' -DPLUGIN_NUM_POS=2 -DPLUGIN_CONTROL_SIZE=2 -DPLUGIN_NUM_TIME_STEPS=64 plugin_shim.cpp
#include <math.h>
#include <type_traits>
#include <utility>

#include "../../oracle/oracle.h"

#define __host__
#define __device__
#define __forceinline__ inline
#define NUM_POS PLUGIN_NUM_POS
#define STATE_SIZE (2 * NUM_POS)
#define CONTROL_SIZE PLUGIN_CONTROL_SIZE
#define NUM_TIME_STEPS PLUGIN_NUM_TIME_STEPS
#define EE_COST 0
#define MPC_MODE 0
#define USE_LIMITS_FLAG 0
#define USE_SMOOTH_ABS 0
#define USE_EE_VEL_COST 0

namespace plug {
inline void doubleLoopVals(int* starty, int* dy, int* startx, int* dx) { *starty = 0; *dy = 1; *startx = 0; *dx = 1; }
inline void singleLoopVals(int* start, int* delta) { *start = 0; *delta = 1; }
inline void hd__syncthreads() {}
#include PLUGIN_COST_FILE
#include PLUGIN_PLANT_FILE
}  // namespace plug

namespace {
template <typename T> struct Tables {
    T I[36 * PLUGIN_NUM_POS], Tb[36 * PLUGIN_NUM_POS];
    Tables() { for (auto& v : I) v = 0; for (auto& v : Tb) v = 0; plug::initI<T>(I); plug::initT<T>(Tb); }
};
template <typename T> Tables<T>& tables() { static Tables<T> t; return t; }

template <typename T> void dyn(T* qdd, const T* x, const T* u) { plug::dynamics<T>(qdd, (T*)x, (T*)u, tables<T>().I, tables<T>().Tb); }
template <typename T> void grad(T* dqdd, T* qdd, const T* x, const T* u) { plug::dynamicsGradient<T>(dqdd, qdd, (T*)x, (T*)u, tables<T>().I, tables<T>().Tb); }

template <typename T, typename = void> struct weights_in_signature : std::false_type {};
template <typename T>
struct weights_in_signature<T, std::void_t<decltype(plug::costFunc<T>((T*)nullptr, (T*)nullptr, (T*)nullptr, 0, T(), T(), T(), T(), T()))>> : std::true_type {};

template <typename T> T cost(const T* xk, const T* uk, const T* xg, int k, T Q1, T Q2, T R_, T QF1, T QF2) {
    if constexpr (weights_in_signature<T>::value) return plug::costFunc<T>((T*)xk, (T*)uk, (T*)xg, k, Q1, Q2, R_, QF1, QF2);
    else return plug::costFunc<T>((T*)xk, (T*)uk, (T*)xg, k);
}
template <typename T> void cgrad(T* Hk, T* gk, const T* xk, const T* uk, const T* xg, int k, int ld, T Q1, T Q2, T R_, T QF1, T QF2) {
    if constexpr (weights_in_signature<T>::value) plug::costGrad<T>(Hk, gk, (T*)xk, (T*)uk, (T*)xg, k, ld, Q1, Q2, R_, QF1, QF2);
    else plug::costGrad<T>(Hk, gk, (T*)xk, (T*)uk, (T*)xg, k, ld);
}
}  // namespace

extern "C" const ora_plugin_f32* plugin_f32(void) { static const ora_plugin_f32 p = {PLUGIN_NUM_POS, PLUGIN_CONTROL_SIZE, dyn<float>, grad<float>, cost<float>, cgrad<float>}; return &p; }
extern "C" const ora_plugin_f64* plugin_f64(void) { static const ora_plugin_f64 p = {PLUGIN_NUM_POS, PLUGIN_CONTROL_SIZE, dyn<double>, grad<double>, cost<double>, cgrad<double>}; return &p; }
