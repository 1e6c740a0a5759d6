// Backward (Riccati-like) pass of the KUKA-sized problem (n = 14, m = 7) on lane groups: one 8-lane group walks one of
// the M blocks of knots of one problem backwards, 8 (problem, block) pairs per wave.
//
// Same operations per output element, in the same order, as bp_block() (bp.hpp), which restates backPassKern and its inner
// routines (DDPHelpers/bpHelpers.cuh:18-420) including the asymmetric placement of the regulariser -- the float32 results are
// bit-identical to it (tests/test_lanegroup.py).  What changes is the decomposition:
//   * the stacked variable z = [x; u] has 21 = 3 x 7 entries: lane l owns entries l, l+7, l+14 (q_l, qd_l, u_l).  It keeps
//     the three columns of AB_k that belong to them in registers (42), computes the matching three rows of AB2 = AB'(P+rho E)
//     and the three columns of H = AB2 AB + H_cost (63 registers), i.e. every product has ONE operand in registers and streams
//     the other from the group's LDS region with wave-uniform-per-group addresses;
//   * lane r holds row r of [Huu | I] (it falls out of its own H column), so the unpivoted Gauss-Jordan is the same
//     broadcast-pivot-row scheme as in plant_arm_lg.hpp;
//   * K, du, Huu, Hux, gu are exchanged through ~250 floats of LDS, P and p live in LDS for the whole block.
// The cooperative kernel spends two LDS reads per multiply-add on 64 lanes for one block; here a lane does 12-63
// multiply-adds per LDS read and a wave carries 8 blocks.
#pragma once

#include "bp.hpp"
#include "fp_lg.hpp"
#include "solver_state.hpp"

namespace pddp {

constexpr int kBpLgAB = 0, kBpLgP = 294, kBpLgp = 490, kBpLgW = 504, kBpLgGu = 798, kBpLgDu = 805, kBpLgFloats = 816;
// W region (294): AB2[ky*21 + kx] while H is formed; then K[ky + 7 kx] at +0 (98), Huu[j + 7 ky] at +98 (49), Hux[j + 7 kx] at +147 (98)
constexpr int kBpLgK = 0, kBpLgHuu = 98, kBpLgHux = 147;

// lds: this group's region (kBpLgFloats elements).  Returns nothing: the generic 7x7 inversion never reports failure
// (utils/cudaUtils.h:291), so err[blk] is always cleared.
template <typename L, typename T>
PDDP_HD void arm_lg_bp_block(T* lds, const Dims& dm, int blk, const BpArgs<T>& a) {
    using V = typename L::V;
    constexpr int NX = 14, NU = 7, NM = 21;
    const typename L::M act = L::all_true();
    T* ABl = lds + kBpLgAB; T* Pl = lds + kBpLgP; T* pl = lds + kBpLgp; T* W = lds + kBpLgW; T* gul = lds + kBpLgGu; T* dul = lds + kBpLgDu;
    const int N = dm.N, M = dm.M, NBk = dm.NB;
    const T rho = a.rho;
    int ks = NBk * (blk + 1) - 1, iterCount;
    V dJ0 = V(T(0)), dJ1 = V(T(0));                       // per-lane partial sums of the expected reduction (computeExpRed)
    if (ks == N - 1) {                                    // last block: cost-to-go at N-1 is the final cost (bpHelpers.cuh:362-367)
        T* Pprev = a.Pm + NX * NX * (ks - 1); T* pprev = a.pv + NX * (ks - 1);
        const T* Hf = a.H + NM * NM * ks; const T* gf = a.g + NM * ks;
        for (int t = 0; t < 28; t++) {
            const V v = L::gather(Hf, [t](int l) { const int e = l + 7 * t; return (e % 14) + 21 * (e / 14); });
            L::scatter(Pl, [t](int l) { return l + 7 * t; }, v, act); L::scatter(Pprev, [t](int l) { return l + 7 * t; }, v, act);
        }
        for (int t = 0; t < 2; t++) {
            const V v = L::gather(gf, [t](int l) { return l + 7 * t; });
            L::scatter(pl, [t](int l) { return l + 7 * t; }, v, act); L::scatter(pprev, [t](int l) { return l + 7 * t; }, v, act);
        }
        ks--; iterCount = NBk - 2;
        wsync();
    } else {                                              // boundary cost-to-go of the PREVIOUS iteration + linear transform
        iterCount = NBk - 1;
        const T* bP = a.Pp + NX * NX * ks; const T* bp = a.pp + NX * ks;
        for (int t = 0; t < 28; t++) L::scatter(Pl, [t](int l) { return l + 7 * t; }, L::gather(bP, [t](int l) { return l + 7 * t; }), act);
        wsync();
        // p = pp + Pp (x - xp2)   (linearXfrmOrLoad): lane l rows l, l+7
        V d0 = V(T(0)), d1 = V(T(0));
        for (int j = 0; j < NX; j++) {
            const T dxj = a.xcur[NX * (ks + 1) + j] - a.xprev2[NX * (ks + 1) + j];
            d0 = d0 + L::gather(Pl, [j](int l) { return l + NX * j; }) * V(dxj);
            d1 = d1 + L::gather(Pl, [j](int l) { return l + 7 + NX * j; }) * V(dxj);
        }
        L::scatter(pl, [](int l) { return l; }, d0 + L::gather(bp, [](int l) { return l; }), act);
        L::scatter(pl, [](int l) { return l + 7; }, d1 + L::gather(bp, [](int l) { return l + 7; }), act);
        wsync();
    }
    for (int iter = iterCount; iter >= 0; iter--, ks--) {
        const T* bAB = a.AB + NX * NM * ks; const T* bH = a.H + NM * NM * ks; const T* bg = a.g + NM * ks;
        const T* bd = a.dcur + NX * ks;
        // ---- stage AB_k through LDS (coalesced global read), then each lane takes its three columns
        for (int t = 0; t < 42; t++) L::scatter(ABl, [t](int l) { return l + 7 * t; }, L::gather(bAB, [t](int l) { return l + 7 * t; }), act);
        wsync();
        V ABc[3][14];
#pragma unroll
        for (int cI = 0; cI < 3; cI++)
#pragma unroll
            for (int j = 0; j < NX; j++) ABc[cI][j] = L::gather(ABl, [cI, j](int l) { return (l + 7 * cI) * NX + j; });
        // ---- AB2(kx, ky) = sum_j AB(j,kx) (P(j,ky) + rho [kx >= 14, ky == j]) for the lane's three kx, all ky  -> W[ky*21 + kx]
        for (int ky = 0; ky < NX; ky++) {
            V v0 = V(T(0)), v1 = V(T(0)), v2 = V(T(0));
#pragma unroll
            for (int j = 0; j < NX; j++) {
                const T pj = Pl[ky * NX + j];
                v0 = v0 + ABc[0][j] * V(pj);               // P + 0 is exact
                v1 = v1 + ABc[1][j] * V(pj);
                v2 = v2 + ABc[2][j] * V(ky == j ? pj + rho : pj);
            }
            L::scatter(W, [ky](int l) { return ky * NM + l; }, v0, act);
            L::scatter(W, [ky](int l) { return ky * NM + l + 7; }, v1, act);
            L::scatter(W, [ky](int l) { return ky * NM + l + 14; }, v2, act);
        }
        if (M > 1 && dm.on_defect_boundary(iter)) {       // p += P d  (tests the loop counter like the reference, :73)
            V s0 = V(T(0)), s1 = V(T(0));
            for (int j = 0; j < NX; j++) {
                s0 = s0 + V(bd[j]) * L::gather(Pl, [j](int l) { return l + j * NX; });
                s1 = s1 + V(bd[j]) * L::gather(Pl, [j](int l) { return l + 7 + j * NX; });
            }
            wsync();                                       // every lane has read p's inputs (P) -- p itself is only read below
            L::scatter(pl, [](int l) { return l; }, L::gather(pl, [](int l) { return l; }) + s0, act);
            L::scatter(pl, [](int l) { return l + 7; }, L::gather(pl, [](int l) { return l + 7; }) + s1, act);
        }
        wsync();
        // ---- H(ky, kx) = sum_j AB2(ky, j) AB(j, kx) + H_cost for the lane's three kx, all 21 ky; g(kx) = sum_j p_j AB(j,kx) + g_cost
        V Hc[3][21];
#pragma unroll
        for (int cI = 0; cI < 3; cI++)
#pragma unroll
            for (int ky = 0; ky < NM; ky++) Hc[cI][ky] = V(T(0));
        V gc[3] = {V(T(0)), V(T(0)), V(T(0))};
#pragma unroll
        for (int j = 0; j < NX; j++) {
#pragma unroll
            for (int ky = 0; ky < NM; ky++) {
                const T w = W[j * NM + ky];                // AB2[ky + NM*j]
#pragma unroll
                for (int cI = 0; cI < 3; cI++) Hc[cI][ky] = Hc[cI][ky] + V(w) * ABc[cI][j];
            }
            const T pj = pl[j];
#pragma unroll
            for (int cI = 0; cI < 3; cI++) gc[cI] = gc[cI] + V(pj) * ABc[cI][j];
        }
#pragma unroll
        for (int cI = 0; cI < 3; cI++) {
#pragma unroll
            for (int ky = 0; ky < NM; ky++) Hc[cI][ky] = Hc[cI][ky] + L::gather(bH, [cI, ky](int l) { return ky * NM + l + 7 * cI; });
            gc[cI] = gc[cI] + L::gather(bg, [cI](int l) { return l + 7 * cI; });
        }
        wsync();                                          // all reads of AB2 (W) done: the region is reused below
        // ---- Huu row of this lane, published Huu / Hux / gu, Gauss-Jordan on [Huu | I]
        V A[14], Huur[7];
#pragma unroll
        for (int ky = 0; ky < NU; ky++) {
            Huur[ky] = Hc[2][14 + ky]; A[ky] = Huur[ky];
            A[7 + ky] = L::sel(L::lane_is(ky), V(T(1)), V(T(0)));
            L::scatter(W + kBpLgHuu, [ky](int l) { return l + 7 * ky; }, Huur[ky], act);
        }
#pragma unroll
        for (int kx = 0; kx < NX; kx++) L::scatter(W + kBpLgHux, [kx](int l) { return l + 7 * kx; }, Hc[2][kx], act);
        L::scatter(gul, [](int l) { return l; }, gc[2], act);
#define PDDP_LG_PIV(PV)                                                                                     \
        {                                                                                                   \
            V rowp[8];                                                                                      \
            _Pragma("unroll") for (int kc = 0; kc < 8; kc++) rowp[kc] = L::template bcast<PV>(A[PV + kc]);  \
            const V colp = A[PV];                                                                           \
            const V inv = V(T(1)) / rowp[0];                                                                \
            const typename L::M isp = L::lane_is(PV);                                                       \
            _Pragma("unroll") for (int kc = 0; kc < 8; kc++) A[PV + kc] = L::sel(isp, A[PV + kc] * inv, A[PV + kc] - colp * inv * rowp[kc]); \
        }
        PDDP_LG_PIV(0) PDDP_LG_PIV(1) PDDP_LG_PIV(2) PDDP_LG_PIV(3) PDDP_LG_PIV(4) PDDP_LG_PIV(5) PDDP_LG_PIV(6)
#undef PDDP_LG_PIV
        wsync();
        // ---- K row of this lane: K(l, kx) = sum_j Hinv(l, j) Hux(j, kx);  du_l = sum_j Hinv(l, j) gu_j
        T* bKT = a.KT + NX * NU * ks; T* bdu = a.du + NU * ks;
        V du;
        {
#pragma unroll
            for (int kx = 0; kx < NX; kx++) {
                V dot = A[7] * V(W[kBpLgHux + 7 * kx]);
#pragma unroll
                for (int j = 1; j < NU; j++) dot = dot + A[7 + j] * V(W[kBpLgHux + j + 7 * kx]);
                L::scatter(W + kBpLgK, [kx](int l) { return l + NU * kx; }, dot, act);
                L::scatter(bKT, [kx](int l) { return kx + NX * l; }, dot, act);
            }
            du = A[7] * V(gul[0]);
#pragma unroll
            for (int j = 1; j < NU; j++) du = du + A[7 + j] * V(gul[j]);
            L::scatter(dul, [](int l) { return l; }, du, act);
            L::scatter(bdu, [](int l) { return l; }, du, act);
        }
        wsync();
        const bool do_ctg = (iter != 0 || blk != 0);      // the cost-to-go in front of knot 0 is never used (:396)
        // ---- T1(kx, ky) = sum_j K(j,kx) Huu(j,ky) - Hxu(kx,ky)  for the lane's rows kx = l, l+7 ("K'Huu - Hxu")
        V T1[2][7];
        if (do_ctg) {
#pragma unroll
            for (int rI = 0; rI < 2; rI++)
#pragma unroll
                for (int ky = 0; ky < NU; ky++) {
                    V val = L::gather(W + kBpLgK, [rI](int l) { return (l + 7 * rI) * NU; }) * V(W[kBpLgHuu + 7 * ky]);
#pragma unroll
                    for (int j = 1; j < NU; j++) val = val + L::gather(W + kBpLgK, [rI, j](int l) { return (l + 7 * rI) * NU + j; }) * V(W[kBpLgHuu + j + 7 * ky]);
                    T1[rI][ky] = val - Hc[rI][14 + ky];
                }
        }
        if (M > 1) {                                      // forward-sweep operands A - B K and B du (computeFSVars)
            T* bApBK = a.ApBK + NX * NX * ks; T* bBdu = a.Bdu + NX * ks;
            // column ky = l, l+7 of ApBK (all kx): ApBK(kx,ky) = A(kx,ky) - sum_j B(kx,j) K(j,ky)
#pragma unroll
            for (int cI = 0; cI < 2; cI++) {
                V Kc[7];
#pragma unroll
                for (int j = 0; j < NU; j++) Kc[j] = L::gather(W + kBpLgK, [cI, j](int l) { return (l + 7 * cI) * NU + j; });
#pragma unroll
                for (int kx = 0; kx < NX; kx++) {
                    V val = V(ABl[196 + kx]) * Kc[0];
#pragma unroll
                    for (int j = 1; j < NU; j++) val = val + V(ABl[196 + kx + NX * j]) * Kc[j];
                    L::scatter(bApBK, [cI, kx](int l) { return (l + 7 * cI) * NX + kx; }, ABc[cI][kx] - val, act);
                }
            }
#pragma unroll
            for (int rI = 0; rI < 2; rI++) {               // Bdu rows l, l+7
                V val = L::gather(ABl, [rI](int l) { return 196 + l + 7 * rI; }) * V(dul[0]);
#pragma unroll
                for (int j = 1; j < NU; j++) val = val + L::gather(ABl, [rI, j](int l) { return 196 + l + 7 * rI + NX * j; }) * V(dul[j]);
                L::scatter(bBdu, [rI](int l) { return l + 7 * rI; }, val, act);
            }
        }
        {                                                 // expected reduction, per-lane partial sums (computeExpRed)
            V dot = Huur[0] * V(dul[0]);
#pragma unroll
            for (int j = 1; j < NU; j++) dot = dot + Huur[j] * V(dul[j]);
            dJ0 = dJ0 + du * gc[2];
            dJ1 = dJ1 + du * dot;
        }
        wsync();
        if (do_ctg) {                                     // new cost-to-go: rows l, l+7 of P, entries l, l+7 of p
            T* Pprev = a.Pm + NX * NX * (ks - 1); T* pprev = a.pv + NX * (ks - 1);
#pragma unroll
            for (int rI = 0; rI < 2; rI++) {
                V Kr[7];                                   // K(j, kx) for this row kx
#pragma unroll
                for (int j = 0; j < NU; j++) Kr[j] = L::gather(W + kBpLgK, [rI, j](int l) { return (l + 7 * rI) * NU + j; });
#pragma unroll
                for (int ky = 0; ky < NX; ky++) {
                    V val = T1[rI][0] * V(W[kBpLgK + ky * NU]) - Kr[0] * V(W[kBpLgHux + 7 * ky]);
#pragma unroll
                    for (int j = 1; j < NU; j++) val = val + (T1[rI][j] * V(W[kBpLgK + ky * NU + j]) - Kr[j] * V(W[kBpLgHux + j + 7 * ky]));
                    const V v = Hc[rI][ky] + val;
                    L::scatter(Pl, [rI, ky](int l) { return ky * NX + l + 7 * rI; }, v, act);
                    L::scatter(Pprev, [rI, ky](int l) { return ky * NX + l + 7 * rI; }, v, act);
                }
                V val = V(dul[0]) * T1[rI][0] - Kr[0] * V(gul[0]);
#pragma unroll
                for (int j = 1; j < NU; j++) val = val + (V(dul[j]) * T1[rI][j] - Kr[j] * V(gul[j]));
                const V v = gc[rI] + val;
                L::scatter(pl, [rI](int l) { return l + 7 * rI; }, v, act);
                L::scatter(pprev, [rI](int l) { return l + 7 * rI; }, v, act);
            }
        }
        wsync();
    }
    // dJexp[2 blk], dJexp[2 blk + 1]: the 7 partial sums in lane order
    const V a0 = lg_chain_sum<L>(V(T(0)), dJ0), a1 = lg_chain_sum<L>(V(T(0)), dJ1);
    const typename L::M last = L::lane_is(6);
    L::scatter(a.dJexp, [blk](int) { return 2 * blk; }, a0, last);
    L::scatter(a.dJexp, [blk](int) { return 2 * blk + 1; }, a1, last);
}

// (problem pb, block blk): pointer set-up of bp_body() (bodies.hpp) around arm_lg_bp_block
template <typename L, typename T>
PDDP_HD void arm_lg_bp_body(T* lds, const Buffers<T>& b, const Dims& dm, int blk, int pb, bool write_err) {
    constexpr int NX = 14, NU = 7, NM = 21;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    if (st.done) return;
    BpArgs<T> a;
    a.AB = b.AB + (size_t)pb * N * NX * NM;
    a.Pm = b.P + (size_t)pb * N * NX * NX;   a.pv = b.p + (size_t)pb * N * NX;
    a.Pp = b.Pp + (size_t)pb * N * NX * NX;  a.pp = b.pp + (size_t)pb * N * NX;
    a.H = b.H + (size_t)pb * N * NM * NM;    a.g = b.g + (size_t)pb * N * NM;
    a.KT = b.KT + (size_t)pb * N * NX * NU;  a.du = b.du + (size_t)pb * N * NU;
    a.dcur = b.dcur + (size_t)pb * N * NX;
    a.ApBK = b.ApBK + (size_t)pb * N * NX * NX;  a.Bdu = b.Bdu + (size_t)pb * N * NX;
    a.xcur = b.xb + ((size_t)pb * 2 + st.cur) * N * NX;
    a.xprev2 = b.xb + ((size_t)pb * 2 + st.cur2) * N * NX;
    a.dJexp = b.dJexp + (size_t)pb * 2 * dm.M;
    a.err = b.err + (size_t)pb * dm.M;
    a.rho = st.rho;
    arm_lg_bp_block<L, T>(lds, dm, blk, a);
    if (write_err) a.err[blk] = 0;                        // the generic 7x7 inversion never reports failure (utils/cudaUtils.h:291)
}

}  // namespace pddp
