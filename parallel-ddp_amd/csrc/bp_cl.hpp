// Backward pass of the 12-state / 4-control closed-form plant (the quadrotor, BASELINE configs[4]) with the device full: 16 lanes per block of knots, lane = COLUMN.
//
// Same function as bp_block() (bp.hpp), which restates backPassKern and its inner routines (DDPHelpers/bpHelpers.cuh:18-420), and the same operations per output
// element in the same order (sums over the state / control index ascending, the regulariser added to P's diagonal entry for the B columns before the multiply, the
// 4x4 adjugate inverse with its det > 0 test) -- tests/test_closed_form_serial.py holds the two against each other bit for bit.  What differs is who computes what:
// bp_block strides every matrix over the cooperating lanes entry by entry (index = e / NM, e % NM at run time, both operands of every product from LDS); at n + m = 16
// that is ~1250 vector instructions per knot for two 32-lane units, and the kernel is instruction-issue bound (counters: profiles/r04_quad.md).  Here lane c of a
// 16-lane unit keeps COLUMN c of [A B] (12 registers, straight from memory), of AB2 = [A B]'(P + rho) (12), and of H (16); the other operand of a product is a
// broadcast read (all 16 lanes the same LDS address, 16 bytes at a time), every index but the lane's own column is a compile-time constant, and only the pieces that
// change hands between lanes go through LDS:  AB2 (for H = AB2'[A B]),  Huu, Hux, g_u (lanes 12..15 -> everybody),  K, du,  B (for A - B K).
// Four units per wavefront; no barriers (wsync: LDS operations of a wave retire in order).
#pragma once

#include "bp.hpp"

namespace pddp {

template <typename T>
struct BpClLds {                       // one unit
    T Pm[144], pv[12], dx[12];
    T X[12 * 16];                      // AB2[j][column]
    T Huu[16], adj[16], Hinv[16];
    T HUX[48];                         // H[kx * NM + 12 + j]  at [kx * 4 + j]
    T G[16];
    T K[48];                           // bp_block's s.K[ky + NU * kx]  at [kx * 4 + ky]
    T du[4];
    T Bm[48];                          // AB[(12 + j) * NX + kx]  at [j * 12 + kx]
};

#define PDDP_CL_UNROLL _Pragma("unroll")

// c: this lane's column (0..15).  diag_w: the running knots' cost Hessian is diag(P::weight) (joint-space / closed-form cost, not overridden through pddp_set_array("H")):
// taken from the weights instead of 1 KB of H per knot.  Returns 1 on a failed Huu inversion (uniform over the unit).
template <typename P, typename T>
__device__ __forceinline__ int bp_cl_block(BpClLds<T>& s, int c, const Dims& dm, int blk, const BpArgs<T>& a, bool diag_h, T wdiag) {
    constexpr int NX = 12, NU = 4, NM = 16;
    static_assert(P::NX == NX && P::NU == NU, "column-lane backward pass: 12 states, 4 controls");
    const int N = dm.N, M = dm.M, NBk = dm.NB;
    const T rho = a.rho, rr = c >= NX ? rho : T(0);
    int ks = NBk * (blk + 1) - 1, iterCount;
    bool lin = true;
    T dj0 = 0, dj1 = 0;                                       // lanes 0..3: s.dJ[ind], s.dJ[NU + ind] of bp_block
    if (ks == N - 1) {
        T* Pprev = a.Pm + NX * NX * (ks - 1); T* pprev = a.pv + NX * (ks - 1);
        const T* Hf = a.H + NM * NM * ks; const T* gf = a.g + NM * ks;
        for (int e = c; e < NX * NX; e += 16) { const T v = Hf[(e % NX) + NM * (e / NX)]; s.Pm[e] = v; Pprev[e] = v; }
        if (c < NX) { const T v = gf[c]; s.pv[c] = v; pprev[c] = v; }
        ks--; iterCount = NBk - 2; lin = false;
    } else {
        iterCount = NBk - 1;
        const T* bP = a.Pp + NX * NX * ks;
        for (int e = c; e < NX * NX; e += 16) s.Pm[e] = bP[e];
        if (c < NX) s.dx[c] = a.xcur[NX * (ks + 1) + c] - a.xprev2[NX * (ks + 1) + c];
    }
    wsync();
    if (lin) {
        const T* bp = a.pp + NX * ks;
        T v = 0;
        if (c < NX) { T dot = 0; for (int j = 0; j < NX; j++) dot += s.Pm[c + NX * j] * s.dx[j]; v = dot + bp[c]; }
        wsync();
        if (c < NX) s.pv[c] = v;
        wsync();
    }
    for (int iter = iterCount; iter >= 0; iter--, ks--) {
        const T* bAB = a.AB + NX * NM * ks; const T* bH = a.H + NM * NM * ks; const T* bg = a.g + NM * ks;
        const T* bd = a.dcur + NX * ks;
        T col[NX];                                            // column c of [A B]
        {
            typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
            constexpr int W = 16 / sizeof(T);
            const V* src = reinterpret_cast<const V*>(bAB + c * NX);
            PDDP_CL_UNROLL for (int q = 0; q < NX / W; q++) { const V v = src[q]; PDDP_CL_UNROLL for (int e = 0; e < W; e++) col[q * W + e] = v[e]; }
        }
        const T gcost = bg[c];
        T hcost[NM];
        if (!diag_h) { PDDP_CL_UNROLL for (int ky = 0; ky < NM; ky++) hcost[ky] = bH[ky * NM + c]; }
        // AB2[ky][c] = sum_j AB[c][j] (P[ky][j] + [B column and ky == j] rho)
        T ab2[NX];
        PDDP_CL_UNROLL for (int ky = 0; ky < NX; ky++) {
            T val = 0;
            PDDP_CL_UNROLL for (int j = 0; j < NX; j++) val += col[j] * (s.Pm[ky * NX + j] + (ky == j ? rr : T(0)));
            ab2[ky] = val;
            if (ky & 1) __builtin_amdgcn_sched_barrier(0);    // (keeps the scheduler from hoisting all 144 broadcast reads in front of the arithmetic: 256 registers, one wave per SIMD)
        }
        const bool bnd = M > 1 && dm.on_defect_boundary(iter);        // tests the loop counter like the reference (bpHelpers.cuh:73)
        T padd = 0;
        if (bnd && c < NX) { T val = 0; for (int j = 0; j < NX; j++) val += bd[j] * s.Pm[c + j * NX]; padd = val; }
        PDDP_CL_UNROLL for (int j = 0; j < NX; j++) s.X[j * 16 + c] = ab2[j];
        if (c >= NX) { PDDP_CL_UNROLL for (int kx = 0; kx < NX; kx++) s.Bm[(c - NX) * NX + kx] = col[kx]; }
        wsync();
        if (bnd && c < NX) s.pv[c] += padd;
        wsync();
        // H[ky][c] = sum_j AB2[j][ky] AB[c][j] + H_cost;   g[c] = sum_j p[j] AB[c][j] + g_cost
        T h[NM];
        PDDP_CL_UNROLL for (int ky = 0; ky < NM; ky++) {
            T dot = 0;
            PDDP_CL_UNROLL for (int j = 0; j < NX; j++) dot += s.X[j * 16 + ky] * col[j];
            h[ky] = dot + (diag_h ? (ky == c ? wdiag : T(0)) : hcost[ky]);
            if (ky & 1) __builtin_amdgcn_sched_barrier(0);
        }
        T gc;
        { T dot = 0; PDDP_CL_UNROLL for (int j = 0; j < NX; j++) dot += s.pv[j] * col[j]; gc = dot + gcost; }
        // what lanes 12..15 hold of H and g, for everybody:  Huu (A2 of invHuu_dim4), Hux, g
        s.G[c] = gc;
        if (c >= NX) {
            PDDP_CL_UNROLL for (int q = 0; q < NU; q++) s.Huu[q * 4 + (c - NX)] = h[NX + q];          // A2[e] = H[oHUU + (e % 4) + NM (e / 4)]
            PDDP_CL_UNROLL for (int kx = 0; kx < NX; kx++) s.HUX[kx * 4 + (c - NX)] = h[kx];          // H[oHUX + kx NM + j]
        }
        wsync();
        {                                                     // adjugate inverse with a det > 0 test (invHuu_dim4): lane e computes cofactor e
            const int ky = c / 4, kx = c % 4;
            const int r0 = (kx + 1) % 4, c0 = (ky + 1) % 4, r1 = (r0 + 1) % 4, c1 = (c0 + 1) % 4, r2 = (r1 + 1) % 4, c2 = (c1 + 1) % 4;
            const T* A2 = s.Huu;
            const T f0 = A2[c0 * 4 + r0], f1 = A2[c0 * 4 + r1], f2 = A2[c0 * 4 + r2];
            const T f3 = A2[c1 * 4 + r0], f4 = A2[c1 * 4 + r1], f5 = A2[c1 * 4 + r2];
            const T f6 = A2[c2 * 4 + r0], f7 = A2[c2 * 4 + r1], f8 = A2[c2 * 4 + r2];
            const T cdet = f0 * f4 * f8 + f3 * f7 * f2 + f6 * f1 * f5 - f2 * f4 * f6 - f5 * f7 * f0 - f8 * f1 * f3;
            const T mine = ((kx + ky) % 2 ? T(-1) : T(1)) * cdet;
            s.adj[ky * 4 + kx] = mine;
            wsync();
            const T val = T(1) / (s.adj[0] * A2[0] + s.adj[1] * A2[1] + s.adj[2] * A2[2] + s.adj[3] * A2[3]);
            if (val <= T(0)) return 1;
            s.Hinv[kx * 4 + ky] = val * mine;
            wsync();
        }
        T* bKT = a.KT + NX * NU * ks; T* bdu = a.du + NU * ks;
        T k[NU] = {T(0), T(0), T(0), T(0)};                   // lane kx < 12: K[ky][kx] = bp_block's s.K[ky + NU kx]
        if (c < NX) {
            PDDP_CL_UNROLL for (int ky = 0; ky < NU; ky++) {
                T dot = 0;
                PDDP_CL_UNROLL for (int j = 0; j < NU; j++) dot += s.Hinv[ky + NU * j] * s.HUX[c * 4 + j];
                k[ky] = dot; s.K[c * 4 + ky] = dot; bKT[c + NX * ky] = dot;
            }
        }
        if (c < NU) {
            T dot = 0;
            PDDP_CL_UNROLL for (int j = 0; j < NU; j++) dot += s.Hinv[c + NU * j] * s.G[NX + j];
            s.du[c] = dot; bdu[c] = dot;
        }
        wsync();
        const bool do_ctg = (iter != 0 || blk != 0);
        T w[NU];                                              // lane kx < 12: (K' Huu - Hxu)[kx][ky] = bp_block's AB2[ky NX + kx]
        PDDP_CL_UNROLL for (int ky = 0; ky < NU; ky++) {
            T val = 0;
            PDDP_CL_UNROLL for (int j = 0; j < NU; j++) val += k[j] * s.Huu[ky * 4 + j];
            w[ky] = val - h[NX + ky];
        }
        if (M > 1 && c < NX) {                                // forward-sweep operands: lane ky writes column ky of A - B K, lane kx entry kx of B du
            T* bApBK = a.ApBK + NX * NX * ks; T* bBdu = a.Bdu + NX * ks;
            T outc[NX];
            PDDP_CL_UNROLL for (int kx = 0; kx < NX; kx++) {
                T val = 0;
                PDDP_CL_UNROLL for (int j = 0; j < NU; j++) val += s.Bm[j * NX + kx] * k[j];
                outc[kx] = col[kx] - val;
            }
            {
                typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
                constexpr int W = 16 / sizeof(T);
                V* dst = reinterpret_cast<V*>(bApBK + c * NX);
                PDDP_CL_UNROLL for (int q = 0; q < NX / W; q++) { V v; PDDP_CL_UNROLL for (int e = 0; e < W; e++) v[e] = outc[q * W + e]; dst[q] = v; }
            }
            T val = 0;
            PDDP_CL_UNROLL for (int j = 0; j < NU; j++) val += s.Bm[j * NX + c] * s.du[j];
            bBdu[c] = val;
        }
        if (c < NU) {                                         // expected reduction, per-lane partial sums (computeExpRed)
            T dot = 0;
            PDDP_CL_UNROLL for (int j = 0; j < NU; j++) dot += s.Huu[j * 4 + c] * s.du[j];
            dj0 += s.du[c] * s.G[NX + c];
            dj1 += s.du[c] * dot;
        }
        if (do_ctg) {
            T* Pprev = a.Pm + NX * NX * (ks - 1); T* pprev = a.pv + NX * (ks - 1);
            T pn[NX]; T pvn = 0;
            if (c < NX) {
                PDDP_CL_UNROLL for (int ky = 0; ky < NX; ky++) {
                    T val = 0;
                    PDDP_CL_UNROLL for (int j = 0; j < NU; j++) val += w[j] * s.K[ky * 4 + j] - k[j] * s.HUX[ky * 4 + j];
                    pn[ky] = h[ky] + val;
                    if ((ky & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                T val = 0;
                PDDP_CL_UNROLL for (int j = 0; j < NU; j++) val += s.du[j] * w[j] - k[j] * s.G[NX + j];
                pvn = gc + val;
            }
            wsync();                                          // every lane is done with the old P, p
            if (c < NX) {
                PDDP_CL_UNROLL for (int ky = 0; ky < NX; ky++) { s.Pm[ky * NX + c] = pn[ky]; Pprev[ky * NX + c] = pn[ky]; }
                s.pv[c] = pvn; pprev[c] = pvn;
            }
        }
        wsync();
    }
    s.G[c] = dj0; s.X[c] = dj1;                               // dJexp of the block = the sum of the per-lane partials in lane order (bp_block): lanes 1..3 hand theirs over
    wsync();
    if (c == 0) {
        T a0 = s.G[0], a1 = s.X[0];
        for (int j = 1; j < NU; j++) { a0 += s.G[j]; a1 += s.X[j]; }
        a.dJexp[2 * blk] = a0; a.dJexp[2 * blk + 1] = a1; a.err[blk] = 0;
    }
    return 0;
}

}  // namespace pddp
