// Lane groups: 8 consecutive lanes of a 64-lane wavefront cooperate on ONE instance (one rollout, one knot, ...), so a
// wave carries 8 independent instances and every value lives in registers.
//
// Why.  The wave-cooperative kernels (plant_arm.hpp, fp.hpp) give a whole wave to one instance: most stages keep 7..49
// of the 64 lanes busy and every stage boundary is an LDS round trip, so a forward-dynamics evaluation costs ~19 k
// cycles for ~25 kflop.  The KUKA arm has 7 links: with lane l of a group owning link l (lane 7 shadows lane 6), the
// per-link quantities (transforms, 6x6 inertias, twists, wrenches, one row of [M | I]) are per-lane REGISTER arrays, the
// recursions over links become nearest-neighbour moves (DPP row_shr/row_shl: an operand modifier, no LDS), and the few
// all-to-one accesses (joint axes for the mass matrix, pivot rows of the Gauss-Jordan) are intra-group broadcasts
// (two or three DPP moves: quad_perm, then row_shr/shl:4 under a bank mask).  No LDS, no barriers, 8 instances per wave.
//
// The code using this header is written ONCE against a policy L:
//     L::V            one value per lane of the group (device: T itself; host: Vec8<T>)
//     L::M            one predicate per lane
//     L::lane_is(j), lane_lt(j), lane_ge(j)      predicates on the lane-in-group index
//     L::sel(m,a,b)   per-lane select
//     L::up(v)        value of lane-1 (0 in lane 0);   L::down(v)  value of lane+1 (0 in lanes 6 and 7)
//     L::bcast<j>(v)  value of lane j of the group
//     L::gather(p, f) p[f(lane)] per lane,  L::scatter(p, f, v, m)
// LgDevice<T> (below) is the gfx950 implementation and the only one in the product.  tests/hostsim/lanegroup_host.hpp supplies
// LgHost<T>, which runs the 8 lanes of one group in lock step on the CPU (TEST TOOL): arithmetic is identical operation by
// operation in both, which is what makes the float32 results of the two comparable bit for bit.
#pragma once

#include "pddp_common.hpp"

namespace pddp {

constexpr int kLg = 8;          // lanes per group
constexpr int kLgPerWave = 8;   // groups per 64-lane wave

// ------------------------------------------------------------------------------------------------ device: gfx950
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
template <typename T>
struct LgDevice {
    using V = T;
    typedef T V2 __attribute__((ext_vector_type(2)));      // float: one v_pk_* instruction per operation on both halves
    static __device__ __forceinline__ V2 pair(V a, V b) { V2 r; r.x = a; r.y = b; return r; }
    static __device__ __forceinline__ V2 splat(V a) { V2 r; r.x = a; r.y = a; return r; }
    // p[f(lane)], p[f(lane) + 1] as one 8-byte access (f(lane) must be even and p 8-byte aligned)
    template <typename F> static __device__ __forceinline__ V2 gather2(const T* p, F f) { return *reinterpret_cast<const V2*>(p + f(link())); }
    using M = bool;
    using Scalar = T;
    static constexpr bool kDevice = true;
    static __device__ __forceinline__ int lane() { return static_cast<int>(threadIdx.x) & (kLg - 1); }
    static __device__ __forceinline__ int link() { const int l = lane(); return l < 7 ? l : 6; }   // lane 7 shadows lane 6
    static __device__ __forceinline__ M lane_is(int j) { return lane() == j; }
    static __device__ __forceinline__ M lane_lt(int j) { return lane() < j; }
    static __device__ __forceinline__ M lane_ge(int j) { return lane() >= j; }
    static __device__ __forceinline__ V sel(M m, V a, V b) { return m ? a : b; }

    // PRECONDITION of every cross-lane function below: lane 7 of each group is INACTIVE (EXEC = 0) -- the kernels run all
    // group code inside `if (LgDevice<T>::lane() < 7)`.  A DPP read whose source lane is out of the 16-lane row or
    // disabled returns 0 when bound_ctrl is set (verified on gfx950: tools/probes/dpp_probe.hip), so
    //   up(v):   row_shr:1 -- lane 0 of the first group of a row reads out of the row, lane 0 of the second group reads
    //            the disabled lane 7 of the first: both get the 0 the recursions start from;
    //   down(v): row_shl:1 -- lane 6 reads the disabled lane 7: 0.
    // One DPP operand modifier each: no select, no LDS.
    static __device__ __forceinline__ int dpp32(int v, int ctrl) {
        return ctrl == 0x111 ? __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true);
    }
    template <int CTRL> static __device__ __forceinline__ float dppv(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
    }
    template <int CTRL> static __device__ __forceinline__ double dppv(double v) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
    static __device__ __forceinline__ V up(V v) { return dppv<0x111>(v); }      // row_shr:1
    static __device__ __forceinline__ V down(V v) { return dppv<0x101>(v); }    // row_shl:1

    // intra-group broadcast of lane J (J <= 6) in DPP moves (no LDS crossbar):
    //   1. quad_perm: every lane takes lane (J % 4) of its own quad -- correct in the quad that contains J;
    //   2. row_shr:4 / row_shl:4 restricted by bank_mask to the OTHER quad of each group copies it across;
    //   3. J >= 4 only: lane 3 would have to read the disabled lane 7 in step 2 (it keeps its old value): it takes lane 2's.
    template <int J> static __device__ __forceinline__ int bc32(int v) {
        constexpr int q = J & 3;
        const int t = __builtin_amdgcn_mov_dpp(v, q * 0x55, 0xF, 0xF, true);                          // quad_perm:[q,q,q,q]; no "old" to initialise
        if constexpr (J < 4) return __builtin_amdgcn_update_dpp(t, t, 0x114, 0xF, 0xA, false);        // banks 1,3 <- lane-4
        else {
            const int r = __builtin_amdgcn_update_dpp(t, t, 0x104, 0xF, 0x5, false);                  // banks 0,2 <- lane+4
            return __builtin_amdgcn_update_dpp(r, r, 0xA4, 0xF, 0x5, false);                          // quad_perm:[0,1,2,2] in banks 0,2
        }
    }
    template <int J> static __device__ __forceinline__ float bcv(float v) { return __builtin_bit_cast(float, bc32<J>(__builtin_bit_cast(int, v))); }
    template <int J> static __device__ __forceinline__ double bcv(double v) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = bc32<J>((int)(b & 0xffffffffLL)), hi = bc32<J>((int)(b >> 32));
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
    template <int J> static __device__ __forceinline__ V bcast(V v) { return bcv<J>(v); }
    static __device__ __forceinline__ V bcast_dyn(V v, int j) { return __shfl(v, j, kLg); }
    template <typename F> static __device__ __forceinline__ V gather(const T* p, F f) { return p[f(link())]; }
    template <typename F> static __device__ __forceinline__ void scatter(T* p, F f, V v, M m) { if (m && lane() < 7) p[f(lane())] = v; }
    template <typename F> static __device__ __forceinline__ V make(F f) { return f(link()); }
    // wave-uniform base (SGPR pair) + 32-bit per-lane offset: one VGPR of address instead of a 64-bit pointer pair
    template <typename F> static __device__ __forceinline__ V gather_at(const T* p, unsigned off, F f) { return p[off + (unsigned)f(link())]; }
    template <typename F> static __device__ __forceinline__ void scatter_at(T* p, unsigned off, F f, V v, M m) { if (m && lane() < 7) p[off + (unsigned)f(lane())] = v; }
    static __device__ __forceinline__ V vsin(V v) { return tsin<T>(v); }
    static __device__ __forceinline__ V vcos(V v) { return tcos<T>(v); }
    static __device__ __forceinline__ void vsincos(V v, V& sn, V& cs) { double s_, c_; sincos(static_cast<double>(v), &s_, &c_); sn = static_cast<T>(s_); cs = static_cast<T>(c_); }
    static __device__ __forceinline__ V vabs(V v) { return tabs(v); }
    static __device__ __forceinline__ V vatan2(V y, V x) { return tatan2<T>(y, x); }
    static __device__ __forceinline__ V vsqrt(V v) { return tsqrt<T>(v); }
    static __device__ __forceinline__ M all_true() { return true; }
    // keeps the instruction scheduler from hoisting the next block's loads over this point (register pressure)
    static __device__ __forceinline__ void sched_fence() { __asm__ volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    // forces a value to be materialised HERE: stops the vectoriser from collecting the arithmetic of many unrolled iterations
    // into one late block (which would keep every iteration's loaded operands alive and spill them)
    static __device__ __forceinline__ void pin(V& v) { __asm__ volatile("" : "+v"(v)); }
};
#endif

}  // namespace pddp
