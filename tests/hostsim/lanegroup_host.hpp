// TEST TOOL -- NOT PRODUCT CODE.
// Host implementation of the lane-group policy (parallel-ddp_amd/csrc/lanegroup.hpp): the 8 lanes of one group executed in lock step,
// so that the lane-group kernels' arithmetic (plant_arm_lg.hpp, fp_lg.hpp, nis_lg.hpp, bp_lg.hpp) can be checked against the oracle and
// against the wave-cooperative code on a machine without a GPU.  Only tests/hostsim includes it.
#pragma once

#include "../../parallel-ddp_amd/csrc/lanegroup.hpp"

namespace pddp {

template <typename T>
struct Vec8 {
    T l[kLg];
    Vec8() {}
    Vec8(T s) { for (int i = 0; i < kLg; i++) l[i] = s; }
};
struct Mask8 { bool l[kLg]; };
#define PDDP_V8_BIN(op)                                                                                                         \
    template <typename T> inline Vec8<T> operator op(const Vec8<T>& a, const Vec8<T>& b) { Vec8<T> r; for (int i = 0; i < kLg; i++) r.l[i] = a.l[i] op b.l[i]; return r; } \
    template <typename T> inline Vec8<T> operator op(const Vec8<T>& a, T b) { Vec8<T> r; for (int i = 0; i < kLg; i++) r.l[i] = a.l[i] op b; return r; }               \
    template <typename T> inline Vec8<T> operator op(T a, const Vec8<T>& b) { Vec8<T> r; for (int i = 0; i < kLg; i++) r.l[i] = a op b.l[i]; return r; }
PDDP_V8_BIN(+) PDDP_V8_BIN(-) PDDP_V8_BIN(*) PDDP_V8_BIN(/)
#undef PDDP_V8_BIN
template <typename T> inline Vec8<T> operator-(const Vec8<T>& a) { Vec8<T> r; for (int i = 0; i < kLg; i++) r.l[i] = -a.l[i]; return r; }
template <typename T> inline Vec8<T>& operator+=(Vec8<T>& a, const Vec8<T>& b) { for (int i = 0; i < kLg; i++) a.l[i] = a.l[i] + b.l[i]; return a; }
template <typename T> inline Vec8<T>& operator-=(Vec8<T>& a, const Vec8<T>& b) { for (int i = 0; i < kLg; i++) a.l[i] = a.l[i] - b.l[i]; return a; }
template <typename T> inline Vec8<T>& operator*=(Vec8<T>& a, const Vec8<T>& b) { for (int i = 0; i < kLg; i++) a.l[i] = a.l[i] * b.l[i]; return a; }

// a pair of per-lane values that the device executes with ONE packed instruction (v_pk_mul_f32 / v_pk_add_f32)
template <typename V>
struct LgPair { V x, y; };
template <typename V> inline LgPair<V> operator*(const LgPair<V>& a, const LgPair<V>& b) { return LgPair<V>{a.x * b.x, a.y * b.y}; }
template <typename V> inline LgPair<V> operator+(const LgPair<V>& a, const LgPair<V>& b) { return LgPair<V>{a.x + b.x, a.y + b.y}; }
template <typename V> inline LgPair<V> operator-(const LgPair<V>& a, const LgPair<V>& b) { return LgPair<V>{a.x - b.x, a.y - b.y}; }

template <typename T>
struct LgHost {
    using V = Vec8<T>;
    using V2 = LgPair<Vec8<T>>;
    static V2 pair(const V& a, const V& b) { return V2{a, b}; }
    static V2 splat(const V& a) { return V2{a, a}; }
    template <typename F> static V2 gather2(const T* p, F f) { V2 r; for (int i = 0; i < kLg; i++) { const int o = f(i < 7 ? i : 6); r.x.l[i] = p[o]; r.y.l[i] = p[o + 1]; } return r; }
    using M = Mask8;
    using Scalar = T;
    static constexpr bool kDevice = false;
    static M lane_is(int j) { M m; for (int i = 0; i < kLg; i++) m.l[i] = (i == j); return m; }
    static M lane_lt(int j) { M m; for (int i = 0; i < kLg; i++) m.l[i] = (i < j); return m; }
    static M lane_ge(int j) { M m; for (int i = 0; i < kLg; i++) m.l[i] = (i >= j); return m; }
    static V sel(const M& m, const V& a, const V& b) { V r; for (int i = 0; i < kLg; i++) r.l[i] = m.l[i] ? a.l[i] : b.l[i]; return r; }
    static V up(const V& v) { V r; r.l[0] = T(0); for (int i = 1; i < kLg; i++) r.l[i] = v.l[i - 1]; return r; }
    static V down(const V& v) { V r; for (int i = 0; i < 6; i++) r.l[i] = v.l[i + 1]; r.l[6] = T(0); r.l[7] = T(0); return r; }
    template <int J> static V bcast(const V& v) { return V(v.l[J]); }
    static V bcast_dyn(const V& v, int j) { return V(v.l[j]); }
    template <typename F> static V gather(const T* p, F f) { V r; for (int i = 0; i < kLg; i++) r.l[i] = p[f(i < 7 ? i : 6)]; return r; }
    template <typename F> static void scatter(T* p, F f, const V& v, const M& m) { for (int i = 0; i < 7; i++) if (m.l[i]) p[f(i)] = v.l[i]; }
    template <typename F> static V make(F f) { V r; for (int i = 0; i < kLg; i++) r.l[i] = f(i < 7 ? i : 6); return r; }
    // element `off + f(lane)` of an array addressed by a wave-uniform base and a 32-bit per-group offset
    template <typename F> static V gather_at(const T* p, unsigned off, F f) { V r; for (int i = 0; i < kLg; i++) r.l[i] = p[off + (unsigned)f(i < 7 ? i : 6)]; return r; }
    template <typename F> static void scatter_at(T* p, unsigned off, F f, const V& v, const M& m) { for (int i = 0; i < 7; i++) if (m.l[i]) p[off + (unsigned)f(i)] = v.l[i]; }
    static V vsin(const V& v) { V r; for (int i = 0; i < kLg; i++) r.l[i] = tsin<T>(v.l[i]); return r; }
    static V vcos(const V& v) { V r; for (int i = 0; i < kLg; i++) r.l[i] = tcos<T>(v.l[i]); return r; }
    static void vsincos(const V& v, V& sn, V& cs) { sn = vsin(v); cs = vcos(v); }
    static V vabs(const V& v) { V r; for (int i = 0; i < kLg; i++) r.l[i] = tabs(v.l[i]); return r; }
    static V vatan2(const V& y, const V& x) { V r; for (int i = 0; i < kLg; i++) r.l[i] = tatan2<T>(y.l[i], x.l[i]); return r; }
    static V vsqrt(const V& v) { V r; for (int i = 0; i < kLg; i++) r.l[i] = tsqrt<T>(v.l[i]); return r; }
    static T lane_value(const V& v, int j) { return v.l[j]; }     // host-side extraction (tests)
    static M all_true() { M m; for (int i = 0; i < kLg; i++) m.l[i] = true; return m; }
    static void sched_fence() {}
    static void pin(V&) {}
};


}  // namespace pddp
