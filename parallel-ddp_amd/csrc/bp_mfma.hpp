// Backward (Riccati-like) pass of the KUKA-sized problem (n = 14, m = 7) on the MATRIX CORES: one wavefront walks one of the M
// blocks of knots of one problem backwards and every dense product of a knot is a chain of v_mfma_f32_16x16x4_f32 (float handles: the production
// path) or v_mfma_f64_16x16x4_f64 (double handles, PDDP_BP=mx: the SAME tile algebra at a precision at which it can be held against the oracle
// decision for decision -- tests/test_f64_benched_family.py; the two instructions differ only in which accumulator row a register holds, Mx<T> below).
//
// Same function as bp_block() (bp.hpp) / arm_lg_bp_block() (bp_lg.hpp), which restate backPassKern and its inner routines
// (DDPHelpers/bpHelpers.cuh:18-420: linearXfrmOrLoad, backprop, invHuu + invertMatrix, computeKTdu, computeCTG, computeFSVars,
// computeExpRed) including the asymmetric placement of the regulariser (rho reaches Hxu and Huu, not the Hux block the gains are computed
// from).  What differs is the arithmetic decomposition, so the float32 results agree with the oracle within the float32 bar
// (tests/test_fp32_bar.py), not bit for bit: sums over the state index run in the matrix core's order (tile rows r, 4+r, 8+r, 12+r per
// instruction, r = 0..3, in the state order mx_state() below), rho B is added to AB'P instead of rho to P.
//
// Layout.  "RB tile" of a matrix X with <= 16 rows and <= 16 columns: four registers t[0..3] per lane; lane (g = lane >> 4, c = lane & 15)
// holds X[4 g + r][c] in t[r] -- tile indices; WHICH state a tile row / column index stands for is mx_state() (positions in registers 0, 1, velocities
// in registers 2, 3, the vector column last), the same on rows and columns.  That is the accumulator layout of the 16x16x4 instruction, and -- with the k index of step r taken as
// 4 g + r -- it is at the same time the A operand of X' (.) and the B operand of (.) X.  So for two RB tiles X, Y with a common row index
//          mfma4(X, Y, C) = C + X' Y        (four instructions, result again an RB tile, rows = columns of X)
// and the whole knot is written as products of that one form; no operand is ever transposed through LDS or shuffled between lanes:
//   W_x, W_u = P' [A | B]                                    (= AB2', rows i)                      8 instructions   (4 with the Euler step's compact [A B])
//   Hxx' = A' W_x + Hxx_cost    Hux = B' W_x + Hux_cost      -Hxu' = (-W_u)' A - Hxu_cost'    Huu = B' W_u + Huu_cost      16       (8)
//   K    = (Huu^-1')' Hux                                    (rows a)                              2
//   T1'  = Huu' K - Hxu'                                     (rows b)                              2
//   P+   = Hxx + T1'' K - K' Hux                             (rows kx: the next knot's P)          4
//   A-BK = A + (-B')' K                                                                           2
//   Psi' <- G' Psi'   (sweep map of the segment, blocks 0..M-2)                                   4                 (2)
// Euler step (compact [A B], CAB): the position rows of B are exact zeros and those of A are [I  dt I]; with the positions in registers 0, 1 of every lane a sum
// over the state rows of B is instructions 2, 3 only, and the position rows' share of A'W and of G'Psi' is W's / Psi''s own registers 0, 1 and dt x them
// (state 7 + s sits two registers above state s in the same lane); the position rows' share of P'A and of W_u'A is the position ROWS of P / W_u read as columns, a
// transposition of two registers per lane through the wave's LDS area (round 4): 24-26 matrix instructions per knot instead of 38.  The float32 matrix instruction executes on
// the SIMD's float32 lanes -- it excludes the vector instructions of the other resident waves for its 32 cycles (tools/probes/mfma_valu_overlap.hip) -- so every
// instruction removed, matrix or vector, is launch time removed.
// The vectors ride along as state index 14 ("column 14" below; tile column 15, mx_state) of the tiles: p is column 14 of P, so g_x = A'p + g_cost, g_u = B'p + g_cost come out as column 14
// of Hxx and Hux, du = Huu^-1 g_u as column 14 of K, Huu'du as column 14 of T1', the new p as column 14 of P+, and -B du as column 14 of
// A - BK.  Only the 7x7 Gauss-Jordan inversion (unpivoted, never failing -- utils/cudaUtils.h:236-292) runs on the vector ALU: lane group g keeps
// rows 2g, 2g + 1 of [Huu | I] as the matrix core delivered them, the pivot row and the pivot-column entries travel through ds_bpermute.
//
// Global memory: every per-knot block is read / written straight in RB order -- two 8-byte pieces per lane (states 2g, 2g + 1 and 7 + 2g, 8 + 2g of a column
// of the column-major blocks) or 56-byte runs across lanes; the bytes are the same as the lane-group kernel's (algorithmic bytes).
#pragma once

#include <hip/hip_runtime.h>

#include "ab_compact.hpp"
#include "solver_state.hpp"

#ifndef PDDP_MX_EXP
#define PDDP_MX_EXP 0        // measurement variants (tools/bp_exp_times.py, profiles/r04_bp_mfma.md): 1 no pivots, 2 no loads, 3 no stores, 4 no matrix instructions, 5 no pivot exchanges
#endif

#ifndef PDDP_MX_STAGE_K
#define PDDP_MX_STAGE_K 0    // prefetching variants: the gains K | du of a knot leave through LDS as 16-byte pieces (four store instructions of 16 / 8 / 16 / 12 bytes per lane) instead of
#endif                       // four 4-byte stores per lane 56 bytes apart.  Measured slower (round 5, profiles/r05_bp_exchange.md: +40 ... +65 us -- the extra LDS round trip sits on the knot's chain): off.
                             // 2 (round 6): the same pieces ONE KNOT LATE -- parked at the end of the knot, read with the next knot's operands, stored behind its prefetch (off the chain):
                             // same bits, +25 ... +60 us, and the counters see the same bytes either way -- L2 merges the 4-byte stores (profiles/r06_bp_mfma.md section 4)
#ifndef PDDP_MX_DMA_MASK
#define PDDP_MX_DMA_MASK 0   // 1: the operand prefetch requests exactly each run's dwords (exec narrowed per run; measurement knob of round 5)
#endif
#ifndef PDDP_MX_ORDER
#define PDDP_MX_ORDER 0      // 1 (compact [A B] instantiations, crossbar pivots): W_u and Huu FIRST, then the pivots with the products that do not need the inverse (W_x, Hxx, Hux, Hxu')
                             // issued between a pivot's exchange and its update -- same operations, same bits, another order; measured: no gain (round 6, profiles/r06_bp_mfma.md section 3)
#endif
#ifndef PDDP_MX_GJ
#define PDDP_MX_GJ 0         // exchanges of the distributed Gauss-Jordan inversion: 0 = through the LDS crossbar (ds_bpermute, rounds 3-5: the product); measured alternatives of
#endif                       // round 5 (profiles/r05_bp_exchange.md): 1 = gfx950's v_permlane16_swap / v_permlane32_swap + DPP row_newbcast moves, 2 = the same with the
                             // pivot-column broadcast folded into v_fmac_f32_dpp, 3 = pivot row through ds_bpermute + v_fmac_f32_dpp.  Same bits in all four (GPU suite green on 1).

namespace pddp {

typedef unsigned mx_w2 __attribute__((ext_vector_type(2)));
// Cross-lane moves of one 32-bit register that stay on the vector ALU (no LDS round trip, no lgkmcnt wait on the knot's serial chain):
//   mx_row_of_group<GO>(w): every lane (g, c) receives lane (GO, c)'s value -- the 16-lane row of lane group GO copied over all four rows.  gfx950's
//     v_permlane16_swap_b32 (vdst's odd rows <-> vsrc's even rows) applied to two copies of w leaves [r0 r0 r2 r2] and [r1 r1 r3 r3]; v_permlane32_swap_b32 (vdst's upper
//     half <-> vsrc's lower half) on two copies of the one that holds row GO leaves that row everywhere (tools/probes/permlane_swap.hip checks both on the device).
//   mx_row_bcast<L>(w): every lane receives lane L OF ITS OWN ROW (DPP row_newbcast, gfx90a+).
template <int GO> __device__ __forceinline__ unsigned mx_row_of_group(unsigned w) {
    const mx_w2 a = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    const unsigned x = (GO & 1) ? a[1] : a[0];
    const mx_w2 b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (GO >> 1) ? b[1] : b[0];
}
template <int L> __device__ __forceinline__ unsigned mx_row_bcast(unsigned w) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x150 + L, 0xf, 0xf, false); }

// Element-type traits of the tile algebra.  Both 16x16x4 instructions take one A / B element per lane -- lane (g = lane >> 4, c = lane & 15) supplies
// A[i = c][k = g], B[k = g][j = c] -- and return four accumulator elements per lane in column c; what differs is the ROW of register r:
//     float  (v_mfma_f32_16x16x4_f32):  row 4 g + r          double (v_mfma_f64_16x16x4_f64):  row g + 4 r
// Everything below is written in terms of (lane group g, register r): tile index q = Mx<T>::q_of(g, r), and a column index c is decomposed the same
// way (g_of(c), r_of(c)), so that rows and columns keep ONE state order and the product algebra is the same for both element types.
template <typename T> struct Mx;
template <> struct Mx<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    typedef float v2u __attribute__((ext_vector_type(2), aligned(4)));
    static __host__ __device__ constexpr int q_of(int g, int r) { return 4 * g + r; }
    static __host__ __device__ constexpr int g_of(int q) { return q >> 2; }
    static __host__ __device__ constexpr int r_of(int q) { return q & 3; }
    static __device__ __forceinline__ v4 mfma(float a, float b, v4 c) {
        if (PDDP_MX_EXP == 4) { c[0] = __builtin_fmaf(a, b, c[0]); return c; }
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // 1 / d to within one unit in the last place: hardware reciprocal + one Newton step (the reference divides, cudaUtils.h:262)
    static __device__ __forceinline__ float recip(float d) { const float x = __builtin_amdgcn_rcpf(d); return __builtin_fmaf(__builtin_fmaf(-d, x, 1.f), x, x); }
    static __device__ __forceinline__ float readlane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
    static __device__ __forceinline__ float from_lane(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    // ds_bpermute with the source lane split into a lane-dependent register (bytes) and a compile-time byte OFFSET in the instruction's offset field -- the compiler
    // does not fold constants into that field and would keep one address register per (pivot, role): nine registers in a kernel that has none to spare.  The result
    // is only valid after lanes_arrived() (the compiler's own wait-count bookkeeping does not see an asm's LDS operation).
    template <int OFF> static __device__ __forceinline__ float from_lane_off(float v, int base_bytes) {
        float r;
        asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(r) : "v"(base_bytes), "v"(v), "n"(OFF));
        return r;
    }
    static __device__ __forceinline__ void lanes_arrived(float& a, float& b, float& c, float& before) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(before)); }
    static __device__ __forceinline__ void lane_arrived(float& a, float& before) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(before)); }
    template <int GO> static __device__ __forceinline__ float row_of_group(float v) { return __uint_as_float(mx_row_of_group<GO>(__float_as_uint(v))); }
    template <int L> static __device__ __forceinline__ float row_bcast(float v) { return __uint_as_float(mx_row_bcast<L>(__float_as_uint(v))); }
    // r - (lane L of r's own 16-lane row) x q as ONE instruction: the broadcast is the DPP control of the fused multiply-add's first operand (VOP2 v_fmac; the same
    // rounding as fma(-col, q, r)).  s_nop 1: a DPP operand written by the preceding vector instruction needs two wait states (inline asm is not seen by the hazard pass).
    template <int L> static __device__ __forceinline__ float fnma_row_bcast(float r, float q) {
        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(q), "n"(L));
        return r;
    }
};
template <> struct Mx<double> {
    typedef double v4 __attribute__((ext_vector_type(4)));
    typedef double v2u __attribute__((ext_vector_type(2), aligned(8)));
    static __host__ __device__ constexpr int q_of(int g, int r) { return g + 4 * r; }
    static __host__ __device__ constexpr int g_of(int q) { return q & 3; }
    static __host__ __device__ constexpr int r_of(int q) { return q >> 2; }
    static __device__ __forceinline__ v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ double recip(double d) { return 1.0 / d; }
    static __device__ __forceinline__ double readlane(double v, int lane) {
        const long long w = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)w, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w >> 32), lane);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    static __device__ __forceinline__ double from_lane(double v, int src_lane) {
        const long long w = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)w), hi = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)(w >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    template <int OFF> static __device__ __forceinline__ double from_lane_off(double v, int base_bytes) {
        const long long w = __double_as_longlong(v);
        unsigned lo, hi;
        asm volatile("ds_bpermute_b32 %0, %2, %3 offset:%5\n\tds_bpermute_b32 %1, %2, %4 offset:%5" : "=&v"(lo), "=&v"(hi) : "v"(base_bytes), "v"((unsigned)w), "v"((unsigned)(w >> 32)), "n"(OFF));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    static __device__ __forceinline__ void lanes_arrived(double& a, double& b, double& c, double& before) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(before)); }
    static __device__ __forceinline__ void lane_arrived(double& a, double& before) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(before)); }
    template <int GO> static __device__ __forceinline__ double row_of_group(double v) {
        const unsigned long long w = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = mx_row_of_group<GO>((unsigned)w), hi = mx_row_of_group<GO>((unsigned)(w >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    template <int L> static __device__ __forceinline__ double row_bcast(double v) {
        const unsigned long long w = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = mx_row_bcast<L>((unsigned)w), hi = mx_row_bcast<L>((unsigned)(w >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    template <int L> static __device__ __forceinline__ double fnma_row_bcast(double r, double q) { return __builtin_fma(-row_bcast<L>(r), q, r); }
};
template <typename T> using mx4t = typename Mx<T>::v4;

template <typename T>
__device__ __forceinline__ mx4t<T> mx_mfma4(const mx4t<T>& X, const mx4t<T>& Y, mx4t<T> acc) {
    acc = Mx<T>::mfma(X[0], Y[0], acc);
    acc = Mx<T>::mfma(X[1], Y[1], acc);
    acc = Mx<T>::mfma(X[2], Y[2], acc);
    acc = Mx<T>::mfma(X[3], Y[3], acc);
    return acc;
}

// State order inside the tiles.  Register r of lane group g (tile index q_of(g, r), as a row; the same decomposition of lane & 15 as a column) keeps state
// mx_state_gr(g, r): registers 0, 1 the POSITION states 2g, 2g + 1 (g = 3, r = 1: padding, 15), registers 2, 3 the VELOCITY states 7 + 2g, 8 + 2g (g = 3, r = 3:
// index 14, the vector column / homogeneous row).  The same map on rows and columns, so the product algebra above is unchanged; what it buys: the rows in which
// the Euler step's B is exactly zero (and A is {1, dt, 0}) fill instructions 0 and 1 completely, so a sum over the state rows of B needs instructions 2 and 3
// only (mx_mfma_hi) -- like the controls (mx_mfma2 below).
__host__ __device__ constexpr int mx_state_gr(int g, int r) { return (r & 2) ? 7 + 2 * g + (r & 1) : (2 * g + r < 7 ? 2 * g + r : 15); }
template <typename T> __host__ __device__ constexpr int mx_state(int q) { return mx_state_gr(Mx<T>::g_of(q), Mx<T>::r_of(q)); }

// tile column from a column of a column-major block (14 states consecutive at col): not for the hot loop (predicated)
template <typename T>
__device__ __forceinline__ mx4t<T> mx_load_col(const T* col, int g, bool lane_ok) {
    mx4t<T> t = {T(0), T(0), T(0), T(0)};
    if (lane_ok) {
        if (g < 3) {
            const typename Mx<T>::v2u lo = *reinterpret_cast<const typename Mx<T>::v2u*>(col + 2 * g), hi = *reinterpret_cast<const typename Mx<T>::v2u*>(col + 7 + 2 * g);
            t[0] = lo[0]; t[1] = lo[1]; t[2] = hi[0]; t[3] = hi[1];
        } else { t[0] = col[6]; t[2] = col[13]; }
    }
    return t;
}
template <typename T>
__device__ __forceinline__ void mx_store_col(T* col, int g, bool lane_ok, const mx4t<T>& t) {
    if (lane_ok) {
        if (g < 3) {
            typename Mx<T>::v2u lo, hi; lo[0] = t[0]; lo[1] = t[1]; hi[0] = t[2]; hi[1] = t[3];
            *reinterpret_cast<typename Mx<T>::v2u*>(col + 2 * g) = lo; *reinterpret_cast<typename Mx<T>::v2u*>(col + 7 + 2 * g) = hi;
        } else { col[6] = t[0]; col[13] = t[2]; }
    }
}
// the same, unconditional (hot loop): both halves are read from clamped addresses, what the tile must not keep is zeroed by the caller's selects
// (unsigned 32-bit BYTE offsets from a wave-uniform pointer: the loads take the scalar-base + vector-offset addressing mode instead of 64-bit vector adds)
template <class V, typename T>
__device__ __forceinline__ V mx_ld(const T* base, unsigned elem_off) {
    return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + (unsigned)sizeof(T) * elem_off);
}
// Buffer form of the hot loop's loads: base = a wave-uniform resource (four scalar registers, set up once per wave), the knot's position a SCALAR byte offset and the
// lane's share a loop-invariant 32-bit vector offset -- no per-access vector address arithmetic (the flat form costs a 64-bit vector add per access and a register
// PAIR per lane offset, which is what pushed the kernel over its 96 registers).  Raw buffer, stride 0, no range limit.
typedef unsigned mx_u2 __attribute__((ext_vector_type(2)));
typedef unsigned mx_u3 __attribute__((ext_vector_type(3)));
typedef unsigned mx_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mx_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000); }
template <typename T> __device__ __forceinline__ T mx_bld(__amdgpu_buffer_rsrc_t r, unsigned vbyte, unsigned sbyte);
template <> __device__ __forceinline__ float mx_bld<float>(__amdgpu_buffer_rsrc_t r, unsigned vbyte, unsigned sbyte) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vbyte, sbyte, 0));
}
template <> __device__ __forceinline__ double mx_bld<double>(__amdgpu_buffer_rsrc_t r, unsigned vbyte, unsigned sbyte) {
    const mx_u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, vbyte, sbyte, 0);
    return __longlong_as_double((long long)(((unsigned long long)w[1] << 32) | w[0]));
}
template <typename T> __device__ __forceinline__ void mx_bst(__amdgpu_buffer_rsrc_t r, T v, unsigned vbyte, unsigned sbyte);
template <> __device__ __forceinline__ void mx_bst<float>(__amdgpu_buffer_rsrc_t r, float v, unsigned vbyte, unsigned sbyte) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, vbyte, sbyte, 0); }
template <> __device__ __forceinline__ void mx_bst<double>(__amdgpu_buffer_rsrc_t r, double v, unsigned vbyte, unsigned sbyte) {
    const unsigned long long w = (unsigned long long)__double_as_longlong(v);
    mx_u2 p; p[0] = (unsigned)w; p[1] = (unsigned)(w >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(p, r, vbyte, sbyte, 0);
}
template <typename T> __device__ __forceinline__ typename Mx<T>::v2u mx_bld2(__amdgpu_buffer_rsrc_t r, unsigned vbyte, unsigned sbyte);
template <> __device__ __forceinline__ Mx<float>::v2u mx_bld2<float>(__amdgpu_buffer_rsrc_t r, unsigned vbyte, unsigned sbyte) {
    const mx_u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, vbyte, sbyte, 0);
    Mx<float>::v2u o; o[0] = __uint_as_float(w[0]); o[1] = __uint_as_float(w[1]); return o;
}
template <> __device__ __forceinline__ Mx<double>::v2u mx_bld2<double>(__amdgpu_buffer_rsrc_t r, unsigned vbyte, unsigned sbyte) {
    const mx_u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, vbyte, sbyte, 0);
    Mx<double>::v2u o;
    o[0] = __longlong_as_double((long long)(((unsigned long long)w[1] << 32) | w[0])); o[1] = __longlong_as_double((long long)(((unsigned long long)w[3] << 32) | w[2]));
    return o;
}
// (a loop-invariant lane offset is hoisted with its 64-bit extension and costs one 64-bit vector add per access; forcing it to stay 32-bit inside the loop
// with an empty asm costs a register copy + shift per access instead -- measured in instructions, no gain)
template <typename T>
__device__ __forceinline__ mx4t<T> mx_load_col_raw(const T* col, int g) {
    const unsigned o = 2u * (unsigned)g;
    const typename Mx<T>::v2u lo = mx_ld<typename Mx<T>::v2u, T>(col, o), hi = mx_ld<typename Mx<T>::v2u, T>(col, o + 7u);
    mx4t<T> t; t[0] = lo[0]; t[1] = lo[1]; t[2] = hi[0]; t[3] = hi[1];
    return t;
}
template <typename T>
__device__ __forceinline__ mx4t<T> mx_mfma_hi(const mx4t<T>& X, const mx4t<T>& Y, mx4t<T> acc) {
    acc = Mx<T>::mfma(X[2], Y[2], acc);
    acc = Mx<T>::mfma(X[3], Y[3], acc);
    return acc;
}

// Control-indexed tiles.  A tile whose ROWS are a control index a = 0..6 comes out of the matrix core in accumulator rows = the lanes of the
// X operand's columns.  Control column b is therefore kept in lane mx_pi(b) = q_of(b >> 1, b & 1) of every tile that has controls in its
// columns (B, W_u, Huu, Huu^-1'), which puts control row b into register b & 1 of lane group b >> 1: a product that sums over the controls
// then needs instructions r = 0, 1 only (mx_mfma2) instead of four half-empty ones.
template <typename T>
__device__ __forceinline__ mx4t<T> mx_mfma2(const mx4t<T>& X, const mx4t<T>& Y, mx4t<T> acc) {
    acc = Mx<T>::mfma(X[0], Y[0], acc);
    acc = Mx<T>::mfma(X[1], Y[1], acc);
    return acc;
}
template <typename T> __host__ __device__ constexpr int mx_pi(int b) { return Mx<T>::q_of(b >> 1, b & 1); }

// One pivot of the distributed Gauss-Jordan inversion (arm_mx_bp_block): lane group g owns rows 2g, 2g + 1 (R0, R1) of [Huu | I].
template <typename T, int PV>
__device__ __forceinline__ void mx_gj_pivot(T& R0, T& R1, int g, int c4, int g64) {
    using X = Mx<T>;
    constexpr int go = PV >> 1;                                           // owner group of the pivot row; its register is PV & 1
    if (PDDP_MX_EXP == 1) return;
    if (PDDP_MX_EXP == 5) {                                               // the arithmetic of a pivot without readlane / ds_bpermute
        const T q = R1 * X::recip(R0 + T(PV));
        const T n0 = X::fma(-R0, q, R0), n1 = X::fma(-R1, q, R1);
        R0 = (g == go && !(PV & 1)) ? q : n0; R1 = (g == go && (PV & 1)) ? q : n1;
        return;
    }
    const T src = (PV & 1) ? R1 : R0;
    const T piv = X::readlane(src, 16 * go + mx_pi<T>(PV));
    if (PDDP_MX_GJ == 1 || PDDP_MX_GJ == 2) {                             // the exchanges on the vector ALU: same values in the same lanes as below, hence the same bits
        const T rp = X::recip(piv);
        const T prow = X::template row_of_group<go>(src);                 // the pivot row: lane group go's register, column for column, in every lane group
        const T q = prow * rp;
        T n0, n1;
        if (PDDP_MX_GJ == 2) { n0 = X::template fnma_row_bcast<mx_pi<T>(PV)>(R0, q); n1 = X::template fnma_row_bcast<mx_pi<T>(PV)>(R1, q); }
        else {
            const T col0 = X::template row_bcast<mx_pi<T>(PV)>(R0), col1 = X::template row_bcast<mx_pi<T>(PV)>(R1);   // a group's own two pivot-column entries
            n0 = X::fma(-col0, q, R0); n1 = X::fma(-col1, q, R1);
        }
        R0 = (g == go && !(PV & 1)) ? q : n0;
        R1 = (g == go && (PV & 1)) ? q : n1;
        return;
    }
    if (PDDP_MX_GJ == 3) {                                                // the pivot row through the LDS crossbar, the pivot-column entries as DPP operands of the update
        T prow = X::template from_lane_off<64 * go>(src, c4);
        T rp = X::recip(piv);
        X::lane_arrived(prow, rp);
        const T q = prow * rp;
        const T n0 = X::template fnma_row_bcast<mx_pi<T>(PV)>(R0, q), n1 = X::template fnma_row_bcast<mx_pi<T>(PV)>(R1, q);
        R0 = (g == go && !(PV & 1)) ? q : n0;
        R1 = (g == go && (PV & 1)) ? q : n1;
        return;
    }
    T prow = X::template from_lane_off<64 * go>(src, c4);
    T col0 = X::template from_lane_off<4 * mx_pi<T>(PV)>(R0, g64), col1 = X::template from_lane_off<4 * mx_pi<T>(PV)>(R1, g64);
    T rp = X::recip(piv);                                                  // (issued before the wait: the reciprocal runs while the lanes travel)
    X::lanes_arrived(prow, col0, col1, rp);
    const T q = prow * rp;                                                // the scaled pivot row; row a loses (its pivot-column entry) x q
    const T n0 = X::fma(-col0, q, R0), n1 = X::fma(-col1, q, R1);
    R0 = (g == go && !(PV & 1)) ? q : n0;
    R1 = (g == go && (PV & 1)) ? q : n1;
}

// Gauss-Jordan on [Huu | I] with one COLUMN per lane (PDDP_MX_GJ == 4): C[a] = entry (a, this lane's column); the pivot column sits in lane mx_pi(pv) of lane group 0.
template <typename T, int NUc, int PV>
__device__ __forceinline__ void mx_gj_column_pivot(T (&C)[NUc]) {
    using X = Mx<T>;
    T s[NUc];
#pragma unroll
    for (int a = 0; a < NUc; a++) s[a] = X::readlane(C[a], mx_pi<T>(PV));
    const T rp = X::recip(s[PV]);
    const T q = C[PV] * rp;
#pragma unroll
    for (int a = 0; a < NUc; a++) C[a] = (a == PV) ? q : X::fma(-s[a], q, C[a]);
}
// the same pivot with the pivot column taken through DPP row broadcasts (PDDP_MX_GJ == 5): every 16-lane row holds a complete copy of the columns, so lane mx_pi(pv) OF THE
// LANE'S OWN ROW has the pivot column -- the pivot by one broadcast move, the six updates as v_fmac_f32_dpp with the broadcast folded in; no v_readlane, no scalar hazards
template <typename T, int NUc, int PV>
__device__ __forceinline__ void mx_gj_column_pivot_dpp(T (&C)[NUc]) {
    using X = Mx<T>;
    const T piv = X::template row_bcast<mx_pi<T>(PV)>(C[PV]);
    const T rp = X::recip(piv);
    const T q = C[PV] * rp;
#pragma unroll
    for (int a = 0; a < NUc; a++) { if (a != PV) C[a] = X::template fnma_row_bcast<mx_pi<T>(PV)>(C[a], q); }
    C[PV] = q;
}
template <typename T, int NUc>
__device__ __forceinline__ void mx_gj_columns(T (&C)[NUc]) {
    static_assert(NUc == 7, "the arm's seven controls");
    if constexpr (PDDP_MX_GJ == 5) {
        mx_gj_column_pivot_dpp<T, NUc, 0>(C); mx_gj_column_pivot_dpp<T, NUc, 1>(C); mx_gj_column_pivot_dpp<T, NUc, 2>(C); mx_gj_column_pivot_dpp<T, NUc, 3>(C);
        mx_gj_column_pivot_dpp<T, NUc, 4>(C); mx_gj_column_pivot_dpp<T, NUc, 5>(C); mx_gj_column_pivot_dpp<T, NUc, 6>(C);
        return;
    }
    mx_gj_column_pivot<T, NUc, 0>(C); mx_gj_column_pivot<T, NUc, 1>(C); mx_gj_column_pivot<T, NUc, 2>(C); mx_gj_column_pivot<T, NUc, 3>(C);
    mx_gj_column_pivot<T, NUc, 4>(C); mx_gj_column_pivot<T, NUc, 5>(C); mx_gj_column_pivot<T, NUc, 6>(C);
}

// The crossbar pivot in two halves (PDDP_MX_ORDER): issue = the pivot through v_readlane, the pivot row and the group's two pivot-column entries through ds_bpermute, the
// reciprocal; finish = wait for the lanes and update the group's two rows.  Matrix instructions placed between the two run while the exchange is in flight.
template <typename T> struct MxGjFlight { T prow, col0, col1, rp; };
template <typename T, int PV>
__device__ __forceinline__ void mx_gj_issue(MxGjFlight<T>& f, const T& R0, const T& R1, int c4, int g64) {
    using X = Mx<T>;
    constexpr int go = PV >> 1;
    const T src = (PV & 1) ? R1 : R0;
    const T piv = X::readlane(src, 16 * go + mx_pi<T>(PV));
    f.prow = X::template from_lane_off<64 * go>(src, c4);
    f.col0 = X::template from_lane_off<4 * mx_pi<T>(PV)>(R0, g64); f.col1 = X::template from_lane_off<4 * mx_pi<T>(PV)>(R1, g64);
    f.rp = X::recip(piv);
}
template <typename T, int PV>
__device__ __forceinline__ void mx_gj_finish(MxGjFlight<T>& f, T& R0, T& R1, int g) {
    using X = Mx<T>;
    constexpr int go = PV >> 1;
    X::lanes_arrived(f.prow, f.col0, f.col1, f.rp);
    const T q = f.prow * f.rp;
    const T n0 = X::fma(-f.col0, q, R0), n1 = X::fma(-f.col1, q, R1);
    R0 = (g == go && !(PV & 1)) ? q : n0;
    R1 = (g == go && (PV & 1)) ? q : n1;
}

// the read-only operands of one knot as they come from memory
template <typename T, bool FS, bool DIAGH>
struct MxKnotIn {
    mx4t<T> A0, B1;        // A(i, kx = c), B(i, control of this lane)
    T BT0, BT1;            // FS: B(kx = c, b = 2g + r)
    mx4t<T> CXX;           // full H: Hcost(kx, ky = c) | g_x in column 14;   diagonal H: g_x only (lane of column 14)
    T CUX0, CUX1;          // full H: Hcost(14 + b, kx = c) | g_u;             diagonal H: g_u only
    T CXU0, CXU1;          // full H: Hcost(kx = c, 14 + b)
    T CUU0, CUU1;          // full H: Hcost(14 + a, 14 + control of this lane)
    T hx, hu;              // unused
};

// Every lane reads UNCONDITIONALLY from an address clamped into the arrays (no divergent control flow around the loads: the compiler would turn
// every predicated load into its own exec-masked branch region) and what a lane must not use is replaced by 0 afterwards.  The clamped
// addresses stay inside the allocation because the loop never touches the last knot's blocks (ks <= N - 2): an over-read of up to one row / two
// elements past a knot's block lands in the next knot's block.
template <typename T, bool FS, bool DIAGH>
__device__ __forceinline__ void mx_load_knot(MxKnotIn<T, FS, DIAGH>& k, const T* ABk, const T* Hk, const T* gk, int g, int c, int ub) {
    constexpr int NX = 14, NU = 7, NM = 21;
    const int u0 = 2 * g, sc = mx_state<T>(c);
    const bool cx = sc < NX, cu = ub < NU, c14 = (sc == NX), g3 = g < 3;
    const int cc = cx ? sc : NX - 1, uc = cu ? ub : NU - 1;           // clamped column indices
    const mx4t<T> a0 = mx_load_col_raw<T>(ABk + cc * NX, g);
    const mx4t<T> b1 = mx_load_col_raw<T>(ABk + (NX + uc) * NX, g);
#pragma unroll
    for (int r = 0; r < 4; r++) { const bool ok = g3 || !(r & 1); k.A0[r] = (cx && ok) ? a0[r] : T(0); k.B1[r] = (cu && ok) ? b1[r] : T(0); }
    if (FS) {
        const T t0 = ABk[NX * NX + cc + NX * u0], t1 = ABk[NX * NX + cc + NX * (u0 + 1)];
        k.BT0 = cx ? t0 : T(0); k.BT1 = (cx && u0 + 1 < NU) ? t1 : T(0);
    }
    if (DIAGH) {
        const mx4t<T> gx = mx_load_col_raw<T>(gk, g);
        const T gu0 = gk[NX + u0], gu1 = gk[NX + (u0 + 1 < NU ? u0 + 1 : NU - 1)];
#pragma unroll
        for (int r = 0; r < 4; r++) k.CXX[r] = (c14 && (g3 || !(r & 1))) ? gx[r] : T(0);
        k.CUX0 = c14 ? gu0 : T(0); k.CUX1 = (c14 && u0 + 1 < NU) ? gu1 : T(0);
        (void)Hk;                                                     // the diagonal comes from the cost weights (arm_mx_bp_block): no Hessian traffic at all
        k.hx = T(0); k.hu = T(0);
    } else {
        const mx4t<T> xx = mx_load_col_raw<T>(c14 ? gk : Hk + cc * NM, g);
        const T* pu = c14 ? gk + NX + u0 : Hk + cc * NM + NX + u0;
        const T ux0 = pu[0], ux1 = pu[1];
        const T xu0 = Hk[(NX + u0) * NM + cc], xu1 = Hk[(NX + u0 + 1) * NM + cc];
        const T uu0 = Hk[(NX + uc) * NM + NX + u0], uu1 = Hk[(NX + uc) * NM + NX + u0 + 1];
        const bool v1 = u0 + 1 < NU;
#pragma unroll
        for (int r = 0; r < 4; r++) k.CXX[r] = ((cx || c14) && (g3 || !(r & 1))) ? xx[r] : T(0);
        k.CUX0 = (cx || c14) ? ux0 : T(0); k.CUX1 = ((cx || c14) && v1) ? ux1 : T(0);
        k.CXU0 = cx ? xu0 : T(0); k.CXU1 = (cx && v1) ? xu1 : T(0);
        k.CUU0 = cu ? uu0 : T(0); k.CUU1 = (cu && v1) ? uu1 : T(0);
    }
}

// The same operands from the COMPACT [A B] (ab_compact.hpp): only the velocity rows 7..13 of a column are in memory, seven consecutive elements -- exactly the
// rows registers 2, 3 keep: one two-element load per tile and lane at element 2g of the column (g = 3: the second element belongs to the next column and is
// discarded); the constant rows {1, dt, 0} of the Euler step (registers 0, 1) come from compares and B's are zero (never multiplied: mx_mfma_hi).
// chA + offA / chB + offB: this lane's column of the knot's share of its piece, + 2g (ch*: wave-uniform, off*: the lane's element offset); offT0 / offT1: B(state of
// column c, controls 2g, 2g + 1).
template <typename T, bool FS, bool DIAGH>
__device__ __forceinline__ void mx_load_knot_compact(MxKnotIn<T, FS, DIAGH>& k, __amdgpu_buffer_rsrc_t rab, unsigned soCh, unsigned voA, unsigned soB, unsigned voB, unsigned voT0,
                                                     unsigned voT1, __amdgpu_buffer_rsrc_t rg, unsigned soG, int g, int c, int ub, T dt) {
    constexpr int NX = 14, NU = 7;
    constexpr unsigned E = (unsigned)sizeof(T);
    using V2 = typename Mx<T>::v2u;
    const int u0 = 2 * g, sc = mx_state<T>(c);
    const bool cx = sc < NX, cu = ub < NU, c14 = (sc == NX), g3 = g < 3;
    if (PDDP_MX_EXP == 2) {
        const T f = T(1e-3) * T(c + 1);
        k.A0[0] = cx ? ((sc == u0) ? T(1) : ((sc == u0 + 7) ? dt : T(0))) : T(0); k.A0[1] = k.A0[0]; k.A0[2] = cx ? f : T(0); k.A0[3] = (cx && g3) ? -f : T(0);
        k.B1[0] = T(0); k.B1[1] = T(0); k.B1[2] = cu ? f : T(0); k.B1[3] = (cu && g3) ? f : T(0);
        k.BT0 = f; k.BT1 = f;
        for (int r = 0; r < 4; r++) k.CXX[r] = c14 ? f : T(0);
        k.CUX0 = c14 ? f : T(0); k.CUX1 = k.CUX0; k.hx = T(0); k.hu = T(0);
        return;
    }
    const V2 ta = mx_bld2<T>(rab, voA, soCh);
    const V2 tb = mx_bld2<T>(rab, voB, soB);
    k.A0[0] = cx ? ((sc == u0) ? T(1) : ((sc == u0 + 7) ? dt : T(0))) : T(0);                // position rows 2g, 2g + 1 of A; B has zeros there
    k.A0[1] = (cx && g3) ? ((sc == u0 + 1) ? T(1) : ((sc == u0 + 8) ? dt : T(0))) : T(0);
    k.A0[2] = cx ? ta[0] : T(0); k.A0[3] = (cx && g3) ? ta[1] : T(0);                       // velocity rows 7 + 2g, 8 + 2g
    k.B1[0] = T(0); k.B1[1] = T(0);
    k.B1[2] = cu ? tb[0] : T(0); k.B1[3] = (cu && g3) ? tb[1] : T(0);
    if (FS) {
        const T t0 = mx_bld<T>(rab, voT0, soB), t1 = mx_bld<T>(rab, voT1, soB);
        k.BT0 = (cx && sc >= 7) ? t0 : T(0); k.BT1 = (cx && sc >= 7 && u0 + 1 < NU) ? t1 : T(0);
    }
    static_assert(DIAGH, "the compact [A B] is produced by the thread-lane setup kernel, whose cost Hessian is the joint-space diagonal");
    const unsigned vg = E * 2u * (unsigned)g;
    const V2 gl = mx_bld2<T>(rg, vg, soG), gh = mx_bld2<T>(rg, vg + E * 7u, soG);
    const T gu0 = mx_bld<T>(rg, vg + E * (unsigned)NX, soG), gu1 = mx_bld<T>(rg, E * (unsigned)(NX + (u0 + 1 < NU ? u0 + 1 : NU - 1)), soG);
    const mx4t<T> gx = {gl[0], gl[1], gh[0], gh[1]};
#pragma unroll
    for (int r = 0; r < 4; r++) k.CXX[r] = (c14 && (g3 || !(r & 1))) ? gx[r] : T(0);
    k.CUX0 = c14 ? gu0 : T(0); k.CUX1 = (c14 && u0 + 1 < NU) ? gu1 : T(0);
    k.hx = T(0); k.hu = T(0);
}

// ---- Operand prefetch through LDS (float handles, compact [A B]).  A knot's operands are four contiguous runs -- its share of the three pieces of the compact [A B]
// (56 / 42 / 49 floats) and its cost gradient (21) -- so four LDS-direct buffer loads (buffer_load_dword ... lds: lane i's dword lands at M0 + 4 i; no registers) bring
// the NEXT knot's operands into the other half of a double buffer while the current knot computes; the tile operands are then LDS reads (~100 cycles) instead of
// HBM loads (~2 k cycles) on the wave's serial chain, and -- gfx9's single in-order vmcnt -- the loads are issued BEFORE the knot's stores instead of queueing behind
// their acknowledgements.  Inline assembly because the compiler's own LDS-DMA bookkeeping waits for EVERY outstanding transfer before any read of the target array
// (it would wait for the prefetch it just issued); here the wait is explicit: s_waitcnt vmcnt(n) at the top of a knot, n = the store instructions the previous knot issued
// behind its prefetch = everything but those stores has landed (the transfers are older than the stores).  Runs are fetched as 64 dwords: the over-read stays inside the arrays (abc_floats' slack; the cost
// gradient's run is limited to 32 lanes and the loop never reaches the last knot).
typedef int mx_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mx_i4 mx_desc(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    mx_i4 d = {(int)(unsigned)a, (int)(unsigned)(a >> 32), -1, 0x00020000};
    return d;
}
constexpr int kMxDmaRegion = 64, kMxDmaBuf = 5 * kMxDmaRegion;        // dwords: regions piece 0 | piece 1 | piece 2 | cost gradient | (end-effector cost) position block of H
template <bool HQQ>
__device__ __forceinline__ void mx_dma_knot(const float* buf, mx_i4 dab, mx_i4 dg, mx_i4 dh, unsigned lane4, unsigned s0, unsigned s1, unsigned s2, unsigned sg, unsigned sh) {
    const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)buf;
    unsigned keep, keep_lo;                                           // exec is narrowed to each run's length and put back as it was
#if PDDP_MX_DMA_MASK
    // every run fetches exactly its dwords (56 | 42 | 49 of the compact [A B]'s pieces, 21 of the cost gradient) instead of 64 / 64 / 64 / 32: a third fewer bytes requested
    asm volatile(
        "s_mov_b32 %[keep], exec_hi\n\ts_mov_b32 %[keep_lo], exec_lo\n\t"
        "s_mov_b32 exec_hi, 0x00ffffff\n\ts_mov_b32 m0, %[l0]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dab], %[s0] offen lds\n\t"
        "s_mov_b32 exec_hi, 0x000003ff\n\ts_mov_b32 m0, %[l1]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dab], %[s1] offen lds\n\t"
        "s_mov_b32 exec_hi, 0x0001ffff\n\ts_mov_b32 m0, %[l2]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dab], %[s2] offen lds\n\t"
        "s_mov_b32 exec_hi, 0\n\ts_mov_b32 exec_lo, 0x001fffff\n\ts_mov_b32 m0, %[l3]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dg], %[sg] offen lds\n\t"
        "s_mov_b32 exec_lo, %[keep_lo]\n\ts_mov_b32 exec_hi, %[keep]"
        : [keep] "=&s"(keep), [keep_lo] "=&s"(keep_lo)
        : [l0] "s"(l0), [l1] "s"(l0 + 4u * kMxDmaRegion), [l2] "s"(l0 + 8u * kMxDmaRegion), [l3] "s"(l0 + 12u * kMxDmaRegion), [v] "v"(lane4), [dab] "s"(dab), [dg] "s"(dg),
          [s0] "s"(s0), [s1] "s"(s1), [s2] "s"(s2), [sg] "s"(sg)
        : "memory", "m0");
#else
    (void)keep_lo;
    asm volatile(
        "s_mov_b32 m0, %[l0]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dab], %[s0] offen lds\n\t"
        "s_mov_b32 m0, %[l1]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dab], %[s1] offen lds\n\t"
        "s_mov_b32 m0, %[l2]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dab], %[s2] offen lds\n\t"
        "s_mov_b32 %[keep], exec_hi\n\ts_mov_b32 exec_hi, 0\n\ts_mov_b32 m0, %[l3]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dg], %[sg] offen lds\n\t"
        "s_mov_b32 exec_hi, %[keep]"
        : [keep] "=&s"(keep)
        : [l0] "s"(l0), [l1] "s"(l0 + 4u * kMxDmaRegion), [l2] "s"(l0 + 8u * kMxDmaRegion), [l3] "s"(l0 + 12u * kMxDmaRegion), [v] "v"(lane4), [dab] "s"(dab), [dg] "s"(dg),
          [s0] "s"(s0), [s1] "s"(s1), [s2] "s"(s2), [sg] "s"(sg)
        : "memory", "m0");
#endif
    if constexpr (HQQ)                                                // the knot's 49-float position block (b.Hc; 64 dwords: the over-read stays inside the array's slack)
        asm volatile("s_mov_b32 m0, %[l4]\n\ts_nop 0\n\tbuffer_load_dword %[v], %[dh], %[sh] offen lds" :: [l4] "s"(l0 + 16u * kMxDmaRegion), [v] "v"(lane4), [dh] "s"(dh), [sh] "s"(sh) : "memory", "m0");
}
// the knot's tile operands from its LDS copy (the selects of mx_load_knot_compact)
// (aA .. aG: the lane's LDS BYTE addresses inside half 0 of the double buffer, kept opaque so that a knot pays one add per address; half: byte offset of the half)
template <bool FS>
__device__ __forceinline__ void mx_lds_knot(MxKnotIn<float, FS, true>& k, unsigned half, unsigned aA, unsigned aB, unsigned aT, unsigned aG, int g, int c, int ub, float dt) {
    constexpr int NX = 14, NU = 7;
    typedef const float __attribute__((address_space(3))) * lptr;
    const int u0 = 2 * g, sc = mx_state<float>(c);
    const bool cx = sc < NX, cu = ub < NU, c14 = (sc == NX), g3 = g < 3;
    const lptr pA = (lptr)(size_t)(aA + half), pB = (lptr)(size_t)(aB + half), pT = (lptr)(size_t)(aT + half), pG = (lptr)(size_t)(aG + half);
    const float ta0 = pA[0], ta1 = pA[1], tb0 = pB[0], tb1 = pB[1];
    k.A0[0] = cx ? ((sc == u0) ? 1.f : ((sc == u0 + 7) ? dt : 0.f)) : 0.f;
    k.A0[1] = (cx && g3) ? ((sc == u0 + 1) ? 1.f : ((sc == u0 + 8) ? dt : 0.f)) : 0.f;
    k.A0[2] = cx ? ta0 : 0.f; k.A0[3] = (cx && g3) ? ta1 : 0.f;
    k.B1[0] = 0.f; k.B1[1] = 0.f;
    k.B1[2] = cu ? tb0 : 0.f; k.B1[3] = (cu && g3) ? tb1 : 0.f;
    if (FS) {
        const float t0 = pT[0], t1 = pT[7];
        k.BT0 = (cx && sc >= 7) ? t0 : 0.f; k.BT1 = (cx && sc >= 7 && u0 + 1 < NU) ? t1 : 0.f;
    }
    const float gx[4] = {pG[0], pG[1], pG[7], pG[8]};
    const float gu0 = pG[NX], gu1 = pG[NX + 1];
#pragma unroll
    for (int r = 0; r < 4; r++) k.CXX[r] = (c14 && (g3 || !(r & 1))) ? gx[r] : 0.f;
    k.CUX0 = c14 ? gu0 : 0.f; k.CUX1 = (c14 && u0 + 1 < NU) ? gu1 : 0.f;
    k.hx = 0.f; k.hu = 0.f;
}

// One (problem, block of knots).  lds: kMxLds = 400 elements of this wave.  FS: M > 1 (write the forward-sweep operands A - B K, B du).
// DIAGH: the cost Hessian of every running knot is the joint-space cost's diag(Q1 x 7, Q2 x 7, R x 7) (plants/cost_arm.cuh:158-202, ArmPlant::weight):
// the setup kernel wrote exactly those numbers into H, so they are taken from the launch arguments (hq1, hq2, hr) and H is not read in the loop.
// CAB: [A B] comes from the compact array b.ABc (ab_compact.hpp; dt rebuilds the constant rows).  keepP = 0: only the cost-to-go slots a later pass reads are
// written -- the one in front of the block's first knot, which the neighbouring block's next pass starts from (the reference's d_Pp / d_pp boundary slots); the
// per-knot P, p of the interior are by-products nobody reads (not an output of runiLQR_GPU; pddp_config.boundary_cost_to_go_only).  The default, the phase hook
// and MPC handles (whose warm start shifts the whole array, MPCHelpers.cuh:602-655) pass keepP = 1.
// flags bit 1 (kMxFuseSweep): the block also composes its shooting segment's effect on the forward sweep's two sequences -- with G_k = [A - B K, B du; 0, 1] (15 x 15)
// the segment's map is Psi = G_last ... G_first, and Psi' <- G_k' Psi' is one more mfma4 per knot on tiles this pass holds anyway (Psi' again in accumulator layout) --
// and leaves Psi' in b.segmap[problem][block] (256 elements) INSTEAD of writing A - B K and B du of every knot (107 KB per problem, and the linear sweep kernel that would
// read them back): k_sweep_maps composes the M - 1 maps.  Same mathematics as k_sweep_wg's per-segment tiles (pddp_tl.hip).
// HQQ (with DIAGH): the end-effector cost -- the running knots' Hessian is diag(hq1 x 7, hq2 x 7, hr x 7) (nominal-state and control weights) plus the dense 7 x 7 position
// block Jee' Jee of b.Hc (the thread-lane setup kernel's compact output, fp_tl.hpp arm_tl_nis_cost_ee; its diagonal already carries hq1): one two-element load per lane and
// knot into the position rows (registers 0, 1) of the position columns -- 196 bytes per knot instead of the 1764-byte reference-layout block.
// The gains of one knot, staged in memory order (K(b, kx) at 14 b + kx: 392 contiguous bytes of KT; du(b) at 100 + b), as 16-byte pieces: lanes 0..23 one piece of K each,
// lane 24 its last 8 bytes, lanes 25, 26 the 16 + 12 bytes of du -- FOUR store instructions (kMxGainStores).
__device__ __forceinline__ void mx_store_gain_pieces(const mx_u4& w, __amdgpu_buffer_rsrc_t rKT, __amdgpu_buffer_rsrc_t rdu, int lane, unsigned knot) {
    const unsigned soK = knot * (unsigned)(14 * 7 * 4), sod = knot * (unsigned)(7 * 4);
    if (lane < 24) __builtin_amdgcn_raw_buffer_store_b128(w, rKT, 16u * (unsigned)lane, soK, 0);
    else if (lane == 24) { mx_u2 h; h[0] = w[0]; h[1] = w[1]; __builtin_amdgcn_raw_buffer_store_b64(h, rKT, 384u, soK, 0); }
    else if (lane == 25) __builtin_amdgcn_raw_buffer_store_b128(w, rdu, 0u, sod, 0);
    else if (lane == 26) { mx_u3 t; t[0] = w[0]; t[1] = w[1]; t[2] = w[2]; __builtin_amdgcn_raw_buffer_store_b96(t, rdu, 16u, sod, 0); }
}
constexpr int kMxGainStores = 4, kMxCtgStores = 3;                    // store instructions per knot of the prefetching (compact [A B], float) variants: K rows 2g, 2g + 1 and du(2g), du(2g + 1) | [P | p] as 16-byte pieces
constexpr int kMxKeepP = 1, kMxFuseSweep = 2, kMxLds = 400, kMxDmaFloats = 2 * 5 * 64 + 256 + 112;     // (float handles: two operand buffers of five 64-dword regions + the staging areas of [P | p] and of [K | du])
template <typename T, bool FS, bool DIAGH, bool CAB, bool FUSE, bool HQQ = false>
__device__ void arm_mx_bp_block(T* lds, const Buffers<T>& b, const Dims& dm, int pb, int blk, T hq1, T hq2, T hr, T dt, int flags) {
    static_assert(!HQQ || DIAGH, "the compact position block rides on the diagonal-Hessian path");
    using X = Mx<T>;
    using mx4 = mx4t<T>;
    const int keepP = flags & kMxKeepP;
    const bool fuse = FUSE && blk < dm.M - 1;                          // (the last block's segment has no boundary after it)
    constexpr int NX = 14, NU = 7, NM = 21, SZP = NX * NX, SZAB = NX * NM, SZH = NM * NM;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, u0 = 2 * g, sc = mx_state<T>(c);      // sc: the state (14: the vector) of this lane's column
    const int cg = X::g_of(c), cr = X::r_of(c);                        // this lane's column index as (lane group, register) of the row order
    const int ub = (cr < 2) ? 2 * cg + cr : 8;                         // the control whose column this lane holds in control-column tiles (8: none)
    const SolverState<T>& st = b.state[pb];
    if (st.done) return;
    const int N = dm.N, NBk = dm.NB;
    const T rho = st.rho;
    const size_t halfP = (size_t)(b.Pp - b.P), halfp = (size_t)(b.pp - b.p), knot0 = (size_t)pb * N;
    T* Pw = b.P + (st.pw ? halfP : 0) + knot0 * SZP;                  // the half of the cost-to-go double buffer this pass writes ...
    const T* Pr = b.P + (st.pw ? 0 : halfP) + knot0 * SZP;            // ... and the one whose block-boundary slots it reads (reference d_Pp)
    T* pw = b.p + (st.pw ? halfp : 0) + knot0 * NX;
    const T* pr = b.p + (st.pw ? 0 : halfp) + knot0 * NX;
    const T* AB = b.AB + knot0 * SZAB; const T* H = b.H + knot0 * SZH; const T* gg = b.g + knot0 * NM;
    T* KT = b.KT + knot0 * (NX * NU); T* du = b.du + knot0 * NU; T* ApBK = b.ApBK + knot0 * SZP; T* Bdu = b.Bdu + knot0 * NX;
    const T* dcur = b.dcur + knot0 * NX;
    const T* xc = b.xb + ((size_t)pb * 2 + st.cur) * N * NX; const T* xp2 = b.xb + ((size_t)pb * 2 + st.cur2) * N * NX;
    const bool cx = sc < NX, cu = ub < NU, c14 = (sc == NX);          // lane holds a state column / a control column / the vector column (lane & 15 == 15)
    bool diag[4];                                                     // register r of this lane sits on the tile's diagonal (its row index equals the lane's column index)
#pragma unroll
    for (int r = 0; r < 4; r++) diag[r] = (X::q_of(g, r) == c);

    int ks = NBk * (blk + 1) - 1, iterCount;
    mx4 Pa;                                                           // [P | p] in tile order: P(mx_state_gr(g, r), mx_state(c)), p in the vector column
    if (ks == N - 1) {                                                // last block: the final cost (bpHelpers.cuh:362-367)
        const T* Hf = H + (size_t)ks * SZH; const T* gf = gg + (size_t)ks * NM;
        Pa = mx_load_col<T>(cx ? Hf + sc * NM : gf, g, cx || c14);
        if (keepP) mx_store_col<T>(cx ? Pw + (size_t)(ks - 1) * SZP + sc * NX : pw + (size_t)(ks - 1) * NX, g, cx || c14, Pa);
        ks--; iterCount = NBk - 2;
    } else {                                                          // boundary cost-to-go of the previous iteration + linear transform (:18-34)
        iterCount = NBk - 1;
        const T* bP = Pr + (size_t)ks * SZP;
        Pa = mx_load_col<T>(bP + sc * NX, g, cx);
        if (lane < NX) {                                              // p = (Pp dx + pp) + Pp d: lane = row (the block's first knot is a defect boundary, :73)
            T dot = T(0), val = T(0);
            for (int j = 0; j < NX; j++) {
                const T pj = bP[lane + NX * j];
                dot += pj * (xc[NX * (ks + 1) + j] - xp2[NX * (ks + 1) + j]);
                val += dcur[(size_t)ks * NX + j] * pj;
            }
            lds[lane] = (dot + pr[(size_t)ks * NX + lane]) + val;
        }
        wsync();
        if (c14) {
#pragma unroll
            for (int r = 0; r < 4; r++) { const int sr = mx_state_gr(g, r); Pa[r] = (sr < NX) ? lds[sr] : T(0); }
        }
        wsync();
    }
    T dJ00 = T(0), dJ01 = T(0), dJ10 = T(0), dJ11 = T(0);            // per-control partial sums of the expected reduction (lanes of the vector column: controls 2g, 2g+1)
    const mx4 zero = {T(0), T(0), T(0), T(0)};
    // Three transposition areas of 8 rows x 16 tile columns, ONE write address and ONE read address per lane (the areas differ by constant offsets):
    //   a lane group writes its registers 0, 1 (tile rows 2g, 2g + 1; lane group 3's register 1 is the padding row, parked in row 7) at [row][lane & 15];
    //   a lane reads row pj = the position state / control its column belongs to (pj == ub on the control lanes), tile columns q_of(g, r).
    T* ldsI = lds + 16;                                               // Huu^-1: entry (a, b) at [a * 16 + mx_pi(b)]
    T* ldsP = lds + 144;                                              // compact [A B]: the position rows of [P | p] ...
    T* ldsW = lds + 272;                                              // ... and of W_u
    lds[16 + lane] = T(0); lds[80 + lane] = T(0);                     // the inverse's tile column of "control 7" (and row 7) stay 0
    const int pj = cx ? (sc < 7 ? sc : sc - 7) : 0;                   // the position state whose row of A has its nonzero in this lane's column, and that entry
    const T fpos = cx ? (sc < 7 ? T(1) : dt) : T(0);
    const int wa = (2 * g) * 16 + c, ra = pj * 16 + X::q_of(g, 0);
    wsync();
    mx4 PsiT = zero;                                                  // Psi'(m, i) = Psi(i, m): starts as the identity of the 15 x 15 augmented map
    if (FUSE) {
#pragma unroll
        for (int r = 0; r < 4; r++) PsiT[r] = (diag[r] && sc <= NX) ? T(1) : T(0);
    }
    MxKnotIn<T, FS, DIAGH> in;
    T hq0 = T(0), hq1v = T(0);                                        // HQQ: this lane's two entries of the knot's position block
    const T* ABk = AB + (size_t)ks * SZAB; const T* Hk = H + (size_t)ks * SZH; const T* gk = gg + (size_t)ks * NM;   // running block pointers (wave-uniform)
    T* KTk = KT + (size_t)ks * (NX * NU); T* duk = du + (size_t)ks * NU; T* Fk = ApBK + (size_t)ks * SZP; T* Bduk = Bdu + (size_t)ks * NX;
    T* Pk = Pw + (size_t)(ks - 1) * SZP; T* pk = pw + (size_t)(ks - 1) * NX;
    // compact [A B]: per-lane element offsets of this lane's columns inside a knot's share of their pieces, and the per-knot strides of those pieces
    unsigned oA = 0, sA = 0, oB = 0, oT0 = 0, oT1 = 0;
    const unsigned krel0 = (unsigned)(knot0 & 63);
    const __amdgpu_buffer_rsrc_t rab = mx_rsrc(CAB ? b.ABc + (knot0 >> 6) * kAbcChunk : nullptr), rgg = mx_rsrc(gg), rKT = mx_rsrc(KT), rdu = mx_rsrc(du);
    const __amdgpu_buffer_rsrc_t rPw = mx_rsrc(Pw), rpw = mx_rsrc(pw);
    const unsigned voKT = (unsigned)sizeof(T) * (unsigned)(u0 * NX + (cx ? sc : 0)), vodu = (unsigned)sizeof(T) * (unsigned)u0;
    if (CAB) {
        const int cc = cx ? sc : NX - 1, uc = cu ? ub : NU - 1, roff = 2 * g;
        const int pa = abc_piece(cc);
        sA = abc_piece_cols(pa) * 7;
        oA = abc_piece_off(pa) + abc_col_in_piece(cc) * 7 + roff;
        oB = abc_piece_off(2) + uc * 7 + roff;
        const int tr = cx && sc >= 7 ? sc - 7 : 0, u1 = u0 + 1 < NU ? u0 + 1 : NU - 1;
        oT0 = abc_piece_off(2) + u0 * 7 + tr; oT1 = abc_piece_off(2) + u1 * 7 + tr;
    }
    constexpr bool DMA = CAB && sizeof(T) == 4 && PDDP_MX_EXP != 2;   // operand prefetch through LDS (above)
    const float* dmaLds = reinterpret_cast<const float*>(lds) + kMxLds;
    unsigned aA = 0, aB = 0, aT = 0, aG = 0, aH = 0;
    int par = 0;
    mx_i4 dab = {0, 0, 0, 0}, dgg = {0, 0, 0, 0}, dhc = {0, 0, 0, 0};
    if constexpr (DMA) {
        const int cc = cx ? sc : NX - 1, uc = cu ? ub : NU - 1;
        const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)dmaLds;
        aA = l0 + 4u * (unsigned)(abc_piece(cc) * kMxDmaRegion + abc_col_in_piece(cc) * 7 + 2 * g);
        aB = l0 + 4u * (unsigned)(2 * kMxDmaRegion + uc * 7 + 2 * g);
        aT = l0 + 4u * (unsigned)(2 * kMxDmaRegion + u0 * 7 + (cx && sc >= 7 ? sc - 7 : 0));
        aG = l0 + 4u * (unsigned)(3 * kMxDmaRegion + 2 * g);
        asm volatile("" : "+v"(aA), "+v"(aB), "+v"(aT), "+v"(aG));
        dab = mx_desc(b.ABc + (knot0 >> 6) * kAbcChunk); dgg = mx_desc(gg);
        if constexpr (HQQ) {
            aH = l0 + 4u * (unsigned)(4 * kMxDmaRegion + (sc < 7 ? sc : 6) * 7 + (g < 3 ? 2 * g : 5));
            asm volatile("" : "+v"(aH));
            dhc = mx_desc(b.Hc + knot0 * 49);
        }
    }
    auto dma_issue = [&](int knot, int half) {                        // (wave-uniform byte offsets of the knot's four runs)
        const unsigned Gr = krel0 + (unsigned)knot, kk = Gr & 63u, soCh = (Gr >> 6) * (unsigned)(kAbcChunk * 4);
        mx_dma_knot<HQQ>(dmaLds + half * kMxDmaBuf, dab, dgg, dhc, 4u * (unsigned)lane, soCh + kk * 224u, soCh + (unsigned)(abc_piece_off(1) * 4) + kk * 168u,
                         soCh + (unsigned)(abc_piece_off(2) * 4) + kk * 196u, (unsigned)knot * (unsigned)(NM * 4), (unsigned)knot * 196u);
    };
    if constexpr (DMA) { if (iterCount >= 0) dma_issue(ks, 0); }
    for (int iter = iterCount; iter >= 0; iter--, ks--) {
        // No register prefetch of the next knot: measured on MI355X (profiles/r02_bp_mfma_experiments.md) the 18 registers it costs are worth more
        // as a fifth resident wave per SIMD (95 registers -> 5 waves: 0.53 ms for 4096 problems, against 0.58 ms with prefetch and 4 waves).
        if constexpr (DMA) {
            if constexpr (sizeof(T) == 4) {
                // This knot's operands are in LDS once everything OLDER than the previous knot's stores has landed: the wait count is the number of store instructions a knot
                // issues behind its prefetch -- kMxGainStores, plus kMxCtgStores when every knot's cost-to-go is written.  (Until round 5 the count was 4 in both modes: with
                // every slot written the chain then also sat out the acknowledgements of three gain stores per knot.)  tests/test_isa_invariants.py holds the emitted
                // loop to these counts: a store merged, split or added by a compiler or an edit fails the CPU suite instead of racing the LDS reads.
                if (PDDP_MX_EXP == 3 || iter == iterCount) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if PDDP_MX_STAGE_K == 2                                              // (the block's first knot issued no gain stores behind its prefetch: they leave one knot late)
                else if (iter == iterCount - 1) { if (keepP) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kMxCtgStores) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
#ifdef PDDP_MX_CTG_WAIT                                               // (measurement variant: -DPDDP_MX_CTG_WAIT=4 is round 4's count)
                else if (keepP) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PDDP_MX_CTG_WAIT) : "memory");
#else
                else if (keepP) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kMxGainStores + kMxCtgStores) : "memory");
#endif
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kMxGainStores) : "memory");
                mx_lds_knot<FS>(in, (unsigned)par * (unsigned)(4 * kMxDmaBuf), aA, aB, aT, aG, g, c, ub, dt);
                if constexpr (HQQ) {
                    typedef const float __attribute__((address_space(3))) * lptr;
                    const lptr pH = (lptr)(size_t)(aH + (unsigned)par * (unsigned)(4 * kMxDmaBuf));
                    hq0 = pH[0]; hq1v = pH[1];
                }
#if PDDP_MX_STAGE_K == 2
                mx_u4 gpiece = {0u, 0u, 0u, 0u};                       // the previous knot's gains, staged in memory order at the end of its body: read with this knot's operands
                if (iter != iterCount) gpiece = *reinterpret_cast<const mx_u4*>(const_cast<float*>(dmaLds) + 2 * kMxDmaBuf + 256 + 4 * (lane < 27 ? lane : 26));
#endif
                if (iter > 0) dma_issue(ks - 1, par ^ 1);
#if PDDP_MX_STAGE_K == 2
                if (iter != iterCount) mx_store_gain_pieces(gpiece, rKT, rdu, lane, (unsigned)(ks + 1));
#endif
                par ^= 1;
            }
        } else if constexpr (CAB) {
            const unsigned Gr = krel0 + (unsigned)ks, kk = Gr & 63u;                              // knot index from the start of the problem's first chunk
            const unsigned soCh = (Gr >> 6) * (unsigned)(kAbcChunk * sizeof(T));                  // (wave-uniform byte offsets; the lane parts are loop-invariant)
            mx_load_knot_compact<T, FS, DIAGH>(in, rab, soCh, (unsigned)sizeof(T) * (oA + kk * sA), soCh + kk * (unsigned)(49 * sizeof(T)), (unsigned)sizeof(T) * oB,
                                               (unsigned)sizeof(T) * oT0, (unsigned)sizeof(T) * oT1, rgg, (unsigned)ks * (unsigned)(NM * sizeof(T)), g, c, ub, dt);
        } else mx_load_knot<T, FS, DIAGH>(in, ABk, Hk, gk, g, c, ub);
        const MxKnotIn<T, FS, DIAGH>& k = in;
        ABk -= SZAB; Hk -= SZH; gk -= NM;
        // ---- cost blocks in tile form
        mx4 CXX = k.CXX, CUX = {k.CUX0, k.CUX1, T(0), T(0)}, CXU = zero, CUU = zero;
        if (DIAGH) {
#pragma unroll
            for (int r = 0; r < 4; r++) if (cx) CXX[r] = diag[r] ? (sc < 7 ? hq1 : hq2) : T(0);
            CUU[0] = (cu && u0 == ub) ? hr : T(0); CUU[1] = (cu && u0 + 1 == ub) ? hr : T(0);
            if (HQQ) {                                                // rows 2g, 2g + 1 of column sc of the position block (symmetric: read as column sc, rows 2g..)
                typename X::v2u hv;
                if constexpr (DMA) { hv[0] = hq0; hv[1] = hq1v; }     // (prefetched with the knot's other operands)
                else {
                    const T* hk = b.Hc + (knot0 + (size_t)ks) * 49;
                    hv = mx_ld<typename X::v2u, T>(hk, (unsigned)((sc < 7 ? sc : 6) * 7 + (g < 3 ? 2 * g : 5)));
                }
                if (sc < 7) { CXX[0] = (g < 3) ? hv[0] : hv[1]; CXX[1] = (g < 3) ? hv[1] : T(0); }
            }
        } else {
            CXU[0] = k.CXU0; CXU[1] = k.CXU1; CUU[0] = k.CUU0; CUU[1] = k.CUU1;
        }
        // ---- W = P' [A | B]  (AB2', rows i = column of P); the B columns take rho B (backprop, :39-64)
        // Compact [A B]: the position rows of A are [I  dt I], so their share of P'A is the position ROWS of P read as columns -- W(i, kx) gets P(kx, i) for a position
        // column kx, dt P(kx - 7, i) for a velocity column: one nonzero term per element, the bits instructions 0, 1 would have produced.  The transposition goes through
        // the wave's LDS area (one 8-byte-pair write, one 16-byte read per lane: the LDS pipe, not the float32 lanes the matrix instructions occupy).
        constexpr bool REORDER = PDDP_MX_ORDER && CAB && PDDP_MX_GJ == 0 && PDDP_MX_EXP == 0;
        mx4 W0, W1, Hxx, Hux, HxuT, Huu; mx4 W0a;
        T R0, R1;
        const int e = 2 * cg + cr - 2;                                // identity column of this lane (cr >= 2)
        const int c4 = 4 * c, g64 = 64 * g;                           // ds_bpermute addresses: source lane x 4 bytes = these + a constant per pivot
        if constexpr (REORDER) {
            // both transpositions first (they need P and W_u only), so that nothing between the pivots waits on the LDS queue behind an exchange
            ldsP[wa] = Pa[0]; ldsP[wa + 16] = Pa[1];
            wsync();
            mx4 lowP;
#pragma unroll
            for (int r = 0; r < 4; r++) lowP[r] = fpos * ldsP[ra + X::q_of(0, r)];
            wsync();
            W1 = mx_mfma_hi<T>(Pa, k.B1, zero);
            W1[2] += rho * k.B1[2]; W1[3] += rho * k.B1[3];
            ldsW[wa] = W1[0]; ldsW[wa + 16] = W1[1];
            wsync();
            const mx4 lowW = {fpos * ldsW[ra], fpos * ldsW[ra + X::q_of(0, 1)], T(0), T(0)};
            wsync();
            Huu = mx_mfma_hi<T>(k.B1, W1, zero) + CUU;
            {
                const bool left = cr < 2;
                R0 = left ? Huu[0] : (e == u0 ? T(1) : T(0));
                R1 = left ? Huu[1] : ((e == u0 + 1 && u0 + 1 < NU) ? T(1) : T(0));
            }
            MxGjFlight<T> fl;
            mx_gj_issue<T, 0>(fl, R0, R1, c4, g64); __builtin_amdgcn_sched_barrier(0);
            W0 = mx_mfma_hi<T>(Pa, k.A0, lowP);
            W0a = c14 ? Pa : W0;
            __builtin_amdgcn_sched_barrier(0);
            mx_gj_finish<T, 0>(fl, R0, R1, g); mx_gj_issue<T, 1>(fl, R0, R1, c4, g64); __builtin_amdgcn_sched_barrier(0);
            { const mx4 HxxLow = {W0a[0], W0a[1], dt * W0a[0], dt * W0a[1]}; Hxx = mx_mfma_hi<T>(k.A0, W0a, HxxLow) + CXX; }
            __builtin_amdgcn_sched_barrier(0);
            mx_gj_finish<T, 1>(fl, R0, R1, g); mx_gj_issue<T, 2>(fl, R0, R1, c4, g64); __builtin_amdgcn_sched_barrier(0);
            Hux = mx_mfma_hi<T>(k.B1, W0a, zero) + CUX;
            __builtin_amdgcn_sched_barrier(0);
            mx_gj_finish<T, 2>(fl, R0, R1, g); mx_gj_issue<T, 3>(fl, R0, R1, c4, g64); __builtin_amdgcn_sched_barrier(0);
            HxuT = mx_mfma_hi<T>(W1, k.A0, lowW) + CXU;
            __builtin_amdgcn_sched_barrier(0);
            mx_gj_finish<T, 3>(fl, R0, R1, g);
            mx_gj_pivot<T, 4>(R0, R1, g, c4, g64); mx_gj_pivot<T, 5>(R0, R1, g, c4, g64); mx_gj_pivot<T, 6>(R0, R1, g, c4, g64);
            if (cr >= 2 && e < NU) { ldsI[wa - X::q_of(0, 2)] = R0; ldsI[wa + 16 - X::q_of(0, 2)] = R1; }
            wsync();
        } else {
        if constexpr (CAB) {
            ldsP[wa] = Pa[0]; ldsP[wa + 16] = Pa[1];
            wsync();
            mx4 low;
#pragma unroll
            for (int r = 0; r < 4; r++) low[r] = fpos * ldsP[ra + X::q_of(0, r)];
            wsync();
            W0 = mx_mfma_hi<T>(Pa, k.A0, low);
        } else W0 = mx_mfma4<T>(Pa, k.A0, zero);
        W1 = CAB ? mx_mfma_hi<T>(Pa, k.B1, zero) : mx_mfma4<T>(Pa, k.B1, zero);               // (compact [A B] = Euler step: the position rows of B are exact zeros)
        if (CAB) { W1[2] += rho * k.B1[2]; W1[3] += rho * k.B1[3]; } else W1 = W1 + rho * k.B1;
        W0a = c14 ? Pa : W0;                                                            // vector column := p
        // ---- H blocks (:66-93): products first, cost added after, like the reference
        // Hxx(kx, ky) | g_x.  Compact [A B]: the position rows of A are [I  dt I], so their share of A'W is W's position rows themselves (output position rows)
        // and dt x the same registers (output velocity rows: state 7 + s sits two registers above state s in the same lane) -- what instructions 0, 1 would have
        // produced, bit for bit (one nonzero term per element); instructions 2, 3 add the velocity rows
        const mx4 HxxLow = {W0a[0], W0a[1], dt * W0a[0], dt * W0a[1]};
        Hxx = (CAB ? mx_mfma_hi<T>(k.A0, W0a, HxxLow) : mx_mfma4<T>(k.A0, W0a, zero)) + CXX;
        Hux = (CAB ? mx_mfma_hi<T>(k.B1, W0a, zero) : mx_mfma4<T>(k.B1, W0a, zero)) + CUX;                                    // Hux(b, kx)  | g_u      (no rho: the block K is computed from)
                                                                                      // Hxu(kx, b) as [b][kx]   (with rho)
        if constexpr (CAB) {                                          // the position rows' share of W_u'A: W_u(kx, b) / dt W_u(kx - 7, b), transposed through LDS as above
            ldsW[wa] = W1[0]; ldsW[wa + 16] = W1[1];                  // (lanes without a control hold zeros)
            wsync();
            const mx4 low = {fpos * ldsW[ra], fpos * ldsW[ra + X::q_of(0, 1)], T(0), T(0)};
            wsync();
            HxuT = mx_mfma_hi<T>(W1, k.A0, low) + CXU;
        } else HxuT = mx_mfma4<T>(W1, k.A0, zero) + CXU;
        Huu = (CAB ? mx_mfma_hi<T>(k.B1, W1, zero) : mx_mfma4<T>(k.B1, W1, zero)) + CUU;                                     // Huu(a, b)               (with rho)
        // ---- Huu^-1: unpivoted Gauss-Jordan on [Huu | I] (invHuu :192-204, invertMatrix cudaUtils.h:236-292).  The rows stay where the matrix core left them:
        //      lane group g owns rows 2g, 2g + 1 (R0, R1); lane mx_pi(j) of a group keeps column j of Huu, the lane two registers further along the column order
        //      (q_of(j >> 1, (j & 1) + 2)) column j of the identity part.  Per pivot
        //      the pivot row and a group's own two pivot-column entries travel through ds_bpermute (LDS crossbar: not the float32 lanes the matrix instructions
        //      need), the pivot itself through v_readlane; every group then updates its two rows -- 8 vector instructions per pivot instead of 18 with all seven
        //      rows replicated in every lane.  Same operations per element as before: R[a] -= R[a][pv] * (R[pv] / R[pv][pv]).
        {
            const bool left = cr < 2;
            R0 = left ? Huu[0] : (e == u0 ? T(1) : T(0));
            R1 = left ? Huu[1] : ((e == u0 + 1 && u0 + 1 < NU) ? T(1) : T(0));
        }
        if constexpr (PDDP_MX_GJ == 4 || PDDP_MX_GJ == 5) {
            // Round 6: the same elimination with every lane holding a whole COLUMN of [Huu | I] (seven registers; the four lane groups carry identical copies).  Huu goes through
            // the inverse's LDS area once (the tile's rows 2g, 2g + 1 -> rows of that area), a lane reads its column back, and a pivot is seven v_readlane of the pivot column
            // (wave-uniform: they travel as scalar operands), one reciprocal, one multiply, six multiply-subtracts -- no LDS round trip per pivot.  Measured with the pivots'
            // exchanges removed (profiles/r06_bp_mfma.md): the seven dependent ds_bpermute rounds are a quarter of the launch and of one problem's backward pass.  Per element
            // the operations are the distributed form's: R[a] -= R[a][pv] * (R[pv] * (1 / R[pv][pv])), the pivot row becomes the scaled row -- the same bits.
            if (cr < 2) { ldsI[wa] = Huu[0]; if (u0 + 1 < NU) ldsI[wa + 16] = Huu[1]; }
            wsync();
            T Cc[NU];
#pragma unroll
            for (int a = 0; a < NU; a++) { const T v = ldsI[a * 16 + c]; Cc[a] = (cr < 2) ? v : (e == a ? T(1) : T(0)); }
            wsync();
            mx_gj_columns<T, NU>(Cc);
            if (g == 0 && cr >= 2 && e < NU) {
#pragma unroll
                for (int a = 0; a < NU; a++) ldsI[a * 16 + c - X::q_of(0, 2)] = Cc[a];      // column e of the inverse: entry (a, e) at [a * 16 + mx_pi(e)]
            }
            wsync();
        } else {
        mx_gj_pivot<T, 0>(R0, R1, g, c4, g64); mx_gj_pivot<T, 1>(R0, R1, g, c4, g64); mx_gj_pivot<T, 2>(R0, R1, g, c4, g64); mx_gj_pivot<T, 3>(R0, R1, g, c4, g64);
        mx_gj_pivot<T, 4>(R0, R1, g, c4, g64); mx_gj_pivot<T, 5>(R0, R1, g, c4, g64); mx_gj_pivot<T, 6>(R0, R1, g, c4, g64);
        if (cr >= 2 && e < NU) { ldsI[wa - X::q_of(0, 2)] = R0; ldsI[wa + 16 - X::q_of(0, 2)] = R1; }   // identity column e sits two registers along: tile column mx_pi(e)
        wsync();
        }
        }
        mx4 InvT = zero;                                                                          // [b = 2g + r][a = control of this lane] = Huu^-1(a, b)
        if (cu) { InvT[0] = ldsI[ra]; InvT[1] = ldsI[ra + X::q_of(0, 1)]; }
        wsync();
        // ---- gains (computeKTdu :208-220): K(a, kx) | du(a), rows a = 2g + r
        const mx4 Kp = mx_mfma2<T>(InvT, Hux, zero);
        if (PDDP_MX_EXP != 3) {                                       // K row a = 2g + r: 14 elements of KT (lane -> state of its column); du(a) from the lane of the vector column
            // (buffer stores: the lane's offset is loop-invariant, the knot's a scalar -- kMxGainStores = FOUR store instructions per knot, which the prefetch's s_waitcnt counts on)
            constexpr unsigned E = (unsigned)sizeof(T);
            const unsigned soK = (unsigned)ks * (unsigned)(NX * NU) * E, sod = (unsigned)ks * (unsigned)NU * E;
            if constexpr (DMA && PDDP_MX_STAGE_K == 2) {
                // one knot late: the gains are parked in memory order now (two LDS writes per lane, nothing waits for them) and leave at the top of the NEXT knot, read with
                // its operands and stored behind its prefetch -- the same FOUR store instructions per knot in the same place of the order the wait counts on, the LDS trip
                // off the chain (VERDICT r5 task 5)
                float* stgK = const_cast<float*>(dmaLds) + 2 * kMxDmaBuf + 256;
                if (cx) { stgK[u0 * NX + sc] = Kp[0]; if (u0 + 1 < NU) stgK[(u0 + 1) * NX + sc] = Kp[1]; }
                else if (c14) { stgK[100 + u0] = Kp[0]; if (u0 + 1 < NU) stgK[100 + u0 + 1] = Kp[1]; }
                wsync();
            } else if constexpr (DMA && PDDP_MX_STAGE_K) {
                // K (7 rows of 14: 392 contiguous bytes of KT) and du (28 bytes) staged in memory order -- K(b, kx) at 14 b + kx, du(b) at 100 + b -- and written as 16-byte
                // pieces: lanes 0..23 one piece of K each, lane 24 its last 8 bytes, lanes 25, 26 the 16 + 12 bytes of du.  FOUR store instructions, like the per-lane form
                // (kMxGainStores: the prefetch's wait counts them), a quarter of the memory transactions.
                float* stgK = const_cast<float*>(dmaLds) + 2 * kMxDmaBuf + 256;
                if (cx) { stgK[u0 * NX + sc] = Kp[0]; if (u0 + 1 < NU) stgK[(u0 + 1) * NX + sc] = Kp[1]; }
                else if (c14) { stgK[100 + u0] = Kp[0]; if (u0 + 1 < NU) stgK[100 + u0 + 1] = Kp[1]; }
                wsync();
                if (lane < 24) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const mx_u4*>(stgK + 4 * lane), rKT, 16u * (unsigned)lane, soK, 0);
                else if (lane == 24) __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const mx_u2*>(stgK + 96), rKT, 384u, soK, 0);
                else if (lane == 25) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const mx_u4*>(stgK + 100), rdu, 0u, sod, 0);
                else if (lane == 26) { const mx_u4 w = *reinterpret_cast<const mx_u4*>(stgK + 104); mx_u3 t; t[0] = w[0]; t[1] = w[1]; t[2] = w[2]; __builtin_amdgcn_raw_buffer_store_b96(t, rdu, 16u, sod, 0); }
#ifdef PDDP_MX_TEST_EXTRA_STORE                                       // (tests/test_isa_invariants.py: a fifth gain store on purpose must turn the check red)
                if (c14) mx_bst<T>(rdu, Kp[1], vodu, sod);
#endif
            } else if constexpr (CAB) {
                if (cx) { mx_bst<T>(rKT, Kp[0], voKT, soK); if (u0 + 1 < NU) mx_bst<T>(rKT, Kp[1], voKT + (unsigned)NX * E, soK); }
#ifdef PDDP_MX_TEST_EXTRA_STORE                                       // (tests/test_isa_invariants.py: a fifth gain store on purpose must turn the check red)
                if (c14) mx_bst<T>(rdu, Kp[1], vodu, sod);
#endif
                if (c14) {
                    mx_bst<T>(rdu, Kp[0], vodu, sod);
                    asm volatile("" ::: "memory");                   // (the two stores of du are adjacent in memory: they must stay TWO instructions, the wait count above counts them)
                    if (u0 + 1 < NU) mx_bst<T>(rdu, Kp[1], vodu + E, sod);
                }
            } else if (sc <= NX) {                                    // few problems in flight (reference-layout [A B]): two stores from one address select -- the shorter instruction stream
                T* q0 = cx ? KTk + u0 * NX + sc : duk + u0;
                q0[0] = Kp[0];
                if (u0 + 1 < NU) (cx ? q0 + NX : q0 + 1)[0] = Kp[1];
            }
        }
        const bool do_ctg = (iter != 0 || blk != 0);                  // the cost-to-go in front of knot 0 is never used (:396)
        // T1(kx, b) = sum_a K(a,kx) Huu(a,b) - Hxu(kx,b) as [b][kx]; its column 14 is Huu' du
        const mx4 T1t = mx_mfma2<T>(Huu, Kp, zero) - HxuT;
        // ---- expected reduction (computeExpRed :317-334): du . g_u and du . Huu du, per control
        if (FUSE) {                                                   // (two accumulators instead of four: the fused variant is at its register limit)
            dJ00 += Kp[0] * Hux[0] + Kp[1] * Hux[1]; dJ10 += Kp[0] * T1t[0] + Kp[1] * T1t[1];
        } else {
            dJ00 += Kp[0] * Hux[0]; dJ01 += Kp[1] * Hux[1];
            dJ10 += Kp[0] * T1t[0]; dJ11 += Kp[1] * T1t[1];
        }
        if (FS) {                                                     // A - B K | B du  (computeFSVars :281-312)
            const mx4 BT = {k.BT0, k.BT1, T(0), T(0)};                                            // [b][kx = c] = B(kx, b)
            const mx4 BK = mx_mfma2<T>(BT, Kp, zero);
            mx4 Gt = c14 ? BK : k.A0 - BK;                                                        // [A - B K | B du]; the padding / vector rows are zero
            if (!FUSE) mx_store_col<T>(cx ? Fk + sc * NX : Bduk, g, cx || c14, Gt);
            if (FUSE && fuse) {
                if (c14 && g == 3) Gt[3] = T(1);                                                  // G(14, 14) = 1: the homogeneous coordinate (lane group 3, register 3)
                if (CAB) {                                                                        // position rows of G are [I  dt I | 0] (B is zero there): as for Hxx
                    const mx4 low = {PsiT[0], PsiT[1], dt * PsiT[0], dt * PsiT[1]};
                    PsiT = mx_mfma_hi<T>(Gt, PsiT, low);
                } else PsiT = mx_mfma4<T>(Gt, PsiT, zero);
            }
        }
        if (do_ctg) {                                                 // new cost-to-go (computeCTG :225-276): P(kx, ky) | p(kx)
            mx4 val = mx_mfma2<T>(T1t, Kp, zero);
            val = mx_mfma2<T>(-Kp, Hux, val);
            mx4 Pn = Hxx + val;
            if (g == 3) { Pn[1] = T(0); Pn[3] = T(0); }               // the padding and vector rows carry by-products of the vector column: keep them clean
            if ((keepP && PDDP_MX_EXP != 3) || iter == 0) {
                if constexpr (DMA) {
                    // [P | p] of a knot are two contiguous blocks (784 + 56 bytes): staged in LDS in memory order (column sc of P at 14 sc, p at 196 = 4 x 49) they leave as
                    // 16-byte pieces -- lanes 0..48 one piece of P each, lanes 49..52 the pieces of p -- instead of two 8-byte stores per lane 56 bytes apart
                    float* stgQ = const_cast<float*>(dmaLds) + 2 * kMxDmaBuf;
                    if (cx || c14) {
                        float* q = stgQ + (cx ? sc * NX : SZP);
                        if (g < 3) { q[u0] = Pn[0]; q[u0 + 1] = Pn[1]; q[7 + u0] = Pn[2]; q[8 + u0] = Pn[3]; } else { q[6] = Pn[0]; q[13] = Pn[2]; }
                    }
                    wsync();
                    const unsigned soP = (unsigned)(ks - 1) * (unsigned)(SZP * 4), sop = (unsigned)(ks - 1) * (unsigned)(NX * 4);
                    const mx_u4 piece = *reinterpret_cast<const mx_u4*>(stgQ + 4 * lane);
                    if (lane < 49) __builtin_amdgcn_raw_buffer_store_b128(piece, rPw, 16u * (unsigned)lane, soP, 0);
                    else if (lane < 52) __builtin_amdgcn_raw_buffer_store_b128(piece, rpw, 16u * (unsigned)(lane - 49), sop, 0);
                    else if (lane == 52) { mx_u2 h; h[0] = piece[0]; h[1] = piece[1]; __builtin_amdgcn_raw_buffer_store_b64(h, rpw, 48u, sop, 0); }
                    wsync();
                } else mx_store_col<T>(cx ? Pk + sc * NX : pk, g, cx || c14, Pn);
            }
            Pa = Pn;
        }
        KTk -= NX * NU; duk -= NU; Fk -= SZP; Bduk -= NX; Pk -= SZP; pk -= NX;
    }
#if PDDP_MX_STAGE_K == 2
    if constexpr (DMA) {
        if (iterCount >= 0) {                                         // the last knot's gains
            const mx_u4 gpiece = *reinterpret_cast<const mx_u4*>(const_cast<float*>(dmaLds) + 2 * kMxDmaBuf + 256 + 4 * (lane < 27 ? lane : 26));
            mx_store_gain_pieces(gpiece, rKT, rdu, lane, (unsigned)(ks + 1));
        }
    }
#endif
    if (FUSE && fuse) {                                               // Psi' of this segment, row-major [16][16]
        T* o = b.segmap + ((size_t)pb * dm.M + blk) * 256;
#pragma unroll
        for (int r = 0; r < 4; r++) o[mx_state_gr(g, r) * 16 + sc] = PsiT[r];        // in state order (the padding index lands in row / column 15)
    }
    // dJexp[2 blk], [2 blk + 1]: the 7 per-control partial sums in order (vector-column lanes 15, 31, 47, 63 hold controls 2g, 2g + 1)
    {
        T a0 = dJ00, a1 = dJ10;
        a0 = a0 + X::readlane(dJ01, 15); a1 = a1 + X::readlane(dJ11, 15);
        a0 = (a0 + X::readlane(dJ00, 31)) + X::readlane(dJ01, 31); a1 = (a1 + X::readlane(dJ10, 31)) + X::readlane(dJ11, 31);
        a0 = (a0 + X::readlane(dJ00, 47)) + X::readlane(dJ01, 47); a1 = (a1 + X::readlane(dJ10, 47)) + X::readlane(dJ11, 47);
        a0 = a0 + X::readlane(dJ00, 63); a1 = a1 + X::readlane(dJ10, 63);
        if (lane == 15) {
            T* dJexp = b.dJexp + (size_t)pb * 2 * dm.M;
            dJexp[2 * blk] = a0; dJexp[2 * blk + 1] = a1;
            b.err[(size_t)pb * dm.M + blk] = 0;                       // the generic 7x7 inversion never reports failure (utils/cudaUtils.h:291)
        }
    }
}

}  // namespace pddp
