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

// LDS map of one group (floats): B block of AB (98), P (196; between the H stage and the end of the knot it holds Hxx, which the new
// cost-to-go overwrites element by element), p (14), W (294), gu (7), du (7) = 2480 bytes -> 8 groups = 19.4 KB per wave, 8 waves per CU.
constexpr int kBpLgB = 0, kBpLgP = 98, kBpLgp = 294, kBpLgW = 308, kBpLgGu = 602, kBpLgDu = 609, kBpLgFloats = 620;
constexpr int kBpLgWs = 14;    // row stride of AB2 in W
// W region (294): AB2 row-contiguous, AB2(kx,ky) at [kx*14 + ky] while H is formed;
// then K[ky + 7 kx] at +0 (98), Huu[j + 7 ky] at +98 (49), Hux[j + 7 kx] at +147 (98)
constexpr int kBpLgK = 0, kBpLgHuu = 98, kBpLgHux = 147;

// lds: this group's region (kBpLgFloats elements).  Returns nothing: the generic 7x7 inversion never reports failure
// (utils/cudaUtils.h:291), so err[blk] is always cleared.
// Global arrays are addressed as (wave-uniform base pointer of the whole batch) + (32-bit element offset of this problem): the
// per-group part of every address is one VGPR, not a 64-bit pointer pair (a dozen of those overflowed the register file).
template <typename T>
struct BpLgArgs {
    const T* AB; T* Pm; T* pv; const T* Pp; const T* pp; const T* H; const T* g; T* KT; T* du; const T* dcur; T* ApBK; T* Bdu; const T* xb; T* dJexp;
    unsigned pbN;          // pb * N: knot k of this problem is element block pbN + k of every per-knot array
    unsigned oPw, opw, oPr, opr;   // element offsets of the written / read half of the cost-to-go double buffer (Pm, pv are its base: P, p)
    unsigned oxc, oxp2;    // offsets of the current trajectory / the trajectory of the stored boundary cost-to-go inside xb
    unsigned odJ;          // pb * 2M
    T rho;
};

template <typename L, typename T>
PDDP_HD void arm_lg_bp_block(T* lds, const Dims& dm, int blk, const BpLgArgs<T>& a) {
    using V = typename L::V;
    constexpr int NX = 14, NU = 7, NM = 21;
    const typename L::M act = L::all_true();
    T* Bl = lds + kBpLgB; T* Pl = lds + kBpLgP; T* pl = lds + kBpLgp; T* W = lds + kBpLgW; T* gul = lds + kBpLgGu; T* dul = lds + kBpLgDu;
    const int N = dm.N, M = dm.M, NBk = dm.NB;
    const T rho = a.rho;
    int ks = NBk * (blk + 1) - 1, iterCount;
    V dJ0 = V(T(0)), dJ1 = V(T(0));                       // per-lane partial sums of the expected reduction (computeExpRed)
    if (ks == N - 1) {                                    // last block: cost-to-go at N-1 is the final cost (bpHelpers.cuh:362-367)
        const unsigned oPprev = a.oPw + (a.pbN + ks - 1) * (NX * NX), opprev = a.opw + (a.pbN + ks - 1) * NX, oHf = (a.pbN + ks) * (NM * NM), ogf = (a.pbN + ks) * NM;
        for (int t = 0; t < 28; t++) {
            const V v = L::gather_at(a.H, oHf, [t](int l) { const int e = l + 7 * t; return (e % 14) + 21 * (e / 14); });
            L::scatter(Pl, [t](int l) { return l + 7 * t; }, v, act); L::scatter_at(a.Pm, oPprev, [t](int l) { return l + 7 * t; }, v, act);
        }
        for (int t = 0; t < 2; t++) {
            const V v = L::gather_at(a.g, ogf, [t](int l) { return l + 7 * t; });
            L::scatter(pl, [t](int l) { return l + 7 * t; }, v, act); L::scatter_at(a.pv, opprev, [t](int l) { return l + 7 * t; }, v, act);
        }
        ks--; iterCount = NBk - 2;
        wsync();
    } else {                                              // boundary cost-to-go of the PREVIOUS iteration + linear transform
        iterCount = NBk - 1;
        const unsigned obP = a.oPr + (a.pbN + ks) * (NX * NX), obp = a.opr + (a.pbN + ks) * NX;
        for (int t = 0; t < 28; t++) L::scatter(Pl, [t](int l) { return l + 7 * t; }, L::gather_at(a.Pm, obP, [t](int l) { return l + 7 * t; }), act);
        wsync();
        // p = pp + Pp (x - xp2)   (linearXfrmOrLoad): lane l rows l, l+7
        V d0 = V(T(0)), d1 = V(T(0));
        for (int j = 0; j < NX; j++) {
            const T dxj = a.xb[a.oxc + NX * (ks + 1) + j] - a.xb[a.oxp2 + NX * (ks + 1) + j];
            d0 = d0 + L::gather(Pl, [j](int l) { return l + NX * j; }) * V(dxj);
            d1 = d1 + L::gather(Pl, [j](int l) { return l + 7 + NX * j; }) * V(dxj);
        }
        L::scatter(pl, [](int l) { return l; }, d0 + L::gather_at(a.pv, obp, [](int l) { return l; }), act);
        L::scatter(pl, [](int l) { return l + 7; }, d1 + L::gather_at(a.pv, obp, [](int l) { return l + 7; }), act);
        wsync();
    }
    // this lane's three columns of AB (z entries l, l+7, l+14), fetched one knot ahead: the global-memory latency hides behind the
    // previous knot's arithmetic
    V ABn[3][14];
#pragma unroll
    for (int cI = 0; cI < 3; cI++)
#pragma unroll
        for (int j = 0; j < NX; j++) ABn[cI][j] = L::gather_at(a.AB, (a.pbN + ks) * (NX * NM), [cI, j](int l) { return (l + 7 * cI) * NX + j; });
    for (int iter = iterCount; iter >= 0; iter--, ks--) {
        const unsigned knot = a.pbN + ks;                  // element block of this knot in every per-knot array
        const unsigned obH = knot * (NM * NM), obg = knot * NM, obd = knot * NX;
        V ABc[3][14];
#pragma unroll
        for (int cI = 0; cI < 3; cI++)
#pragma unroll
            for (int j = 0; j < NX; j++) ABc[cI][j] = ABn[cI][j];
        // cost Hessian / gradient of the lane's columns: the loads are issued here and consumed after the AB2 stage
        V Hc[3][21], gc[3];
#pragma unroll
        for (int cI = 0; cI < 3; cI++) {
#pragma unroll
            for (int ky = 0; ky < NM; ky++) Hc[cI][ky] = L::gather_at(a.H, obH, [cI, ky](int l) { return ky * NM + l + 7 * cI; });
            gc[cI] = L::gather_at(a.g, obg, [cI](int l) { return l + 7 * cI; });
        }
        // the B block (columns 14..20) is needed by every lane later: its owners publish it, B(kx, l) at Bl[kx + 14 l]
#pragma unroll
        for (int kx = 0; kx < NX; kx++) L::scatter(Bl, [kx](int l) { return kx + NX * l; }, ABc[2][kx], act);
        // ---- AB2(kx, ky) = sum_j AB(j,kx) (P(j,ky) + rho [kx >= 14, ky == j]) for the lane's three kx, all ky  -> W[ky*21 + kx]
#pragma nounroll
        for (int ky = 0; ky < NX; ky++) {
            V v0 = V(T(0)), v1 = V(T(0)), v2 = V(T(0));
#pragma unroll
            for (int j = 0; j < NX; j++) {
                const T pj = Pl[ky * NX + j];
                v0 = v0 + ABc[0][j] * V(pj);               // P + 0 is exact
                v1 = v1 + ABc[1][j] * V(pj);
                v2 = v2 + ABc[2][j] * V(ky == j ? pj + rho : pj);
            }
            L::scatter(W, [ky](int l) { return l * kBpLgWs + ky; }, v0, act);
            L::scatter(W, [ky](int l) { return (l + 7) * kBpLgWs + ky; }, v1, act);
            L::scatter(W, [ky](int l) { return (l + 14) * kBpLgWs + ky; }, v2, act);
        }
        if (M > 1 && dm.on_defect_boundary(iter)) {       // p += P d  (tests the loop counter like the reference, :73)
            V s0 = V(T(0)), s1 = V(T(0));
            for (int j = 0; j < NX; j++) {
                const T dj = a.dcur[obd + j];
                s0 = s0 + V(dj) * L::gather(Pl, [j](int l) { return l + j * NX; });
                s1 = s1 + V(dj) * L::gather(Pl, [j](int l) { return l + 7 + j * NX; });
            }
            wsync();                                       // every lane has read p's inputs (P) -- p itself is only read below
            L::scatter(pl, [](int l) { return l; }, L::gather(pl, [](int l) { return l; }) + s0, act);
            L::scatter(pl, [](int l) { return l + 7; }, L::gather(pl, [](int l) { return l + 7; }) + s1, act);
        }
        wsync();
        // ---- H(ky, kx) = sum_j AB2(ky, j) AB(j, kx) + H_cost for the lane's three kx, all 21 ky; g(kx) = sum_j p_j AB(j,kx) + g_cost
#pragma unroll
        for (int ky = 0; ky < NM; ky++) {                  // one output row at a time: 14 uniform LDS reads feed 42 multiply-adds
            T w[14];
#pragma unroll
            for (int j = 0; j < NX; j++) w[j] = W[ky * kBpLgWs + j];            // AB2[ky + NM*j]
#pragma unroll
            for (int cI = 0; cI < 3; cI++) {
                V dot = V(w[0]) * ABc[cI][0];
#pragma unroll
                for (int j = 1; j < NX; j++) dot = dot + V(w[j]) * ABc[cI][j];
                Hc[cI][ky] = dot + Hc[cI][ky];
                L::pin(Hc[cI][ky]);
            }
            L::sched_fence();
        }
        {
            T pv[14];
#pragma unroll
            for (int j = 0; j < NX; j++) pv[j] = pl[j];
#pragma unroll
            for (int cI = 0; cI < 3; cI++) {
                V dot = V(pv[0]) * ABc[cI][0];
#pragma unroll
                for (int j = 1; j < NX; j++) dot = dot + V(pv[j]) * ABc[cI][j];
                gc[cI] = dot + gc[cI];
            }
        }
        if (iter > 0) {                                   // prefetch the next knot's columns of AB
#pragma unroll
            for (int cI = 0; cI < 3; cI++)
#pragma unroll
                for (int j = 0; j < NX; j++) ABn[cI][j] = L::gather_at(a.AB, (knot - 1) * (NX * NM), [cI, j](int l) { return (l + 7 * cI) * NX + j; });
        }
        wsync();                                          // all reads of AB2 (W) done: the region is reused below
        // ---- Huu row of this lane, published Huu / Hux / gu, Gauss-Jordan on [Huu | I]
        V A[14], Huur[7], Hxu[2][7];
#pragma unroll
        for (int rI = 0; rI < 2; rI++) {                   // Hxx rows l, l+7 -> over P in LDS (P's last reader was the AB2 stage), Hxu stays in registers
#pragma unroll
            for (int ky = 0; ky < NX; ky++) L::scatter(Pl, [rI, ky](int l) { return ky * NX + l + 7 * rI; }, Hc[rI][ky], act);
#pragma unroll
            for (int ky = 0; ky < NU; ky++) Hxu[rI][ky] = Hc[rI][14 + ky];
        }
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
        const unsigned obKT = knot * (NX * NU), obdu = knot * NU;
        V du;
        {
#pragma nounroll
            for (int kx = 0; kx < NX; kx++) {
                V dot = A[7] * V(W[kBpLgHux + 7 * kx]);
#pragma unroll
                for (int j = 1; j < NU; j++) dot = dot + A[7 + j] * V(W[kBpLgHux + j + 7 * kx]);
                L::scatter(W + kBpLgK, [kx](int l) { return l + NU * kx; }, dot, act);
                L::scatter_at(a.KT, obKT, [kx](int l) { return kx + NX * l; }, dot, act);
            }
            du = A[7] * V(gul[0]);
#pragma unroll
            for (int j = 1; j < NU; j++) du = du + A[7 + j] * V(gul[j]);
            L::scatter(dul, [](int l) { return l; }, du, act);
            L::scatter_at(a.du, obdu, [](int l) { return l; }, du, act);
        }
        wsync();
        const bool do_ctg = (iter != 0 || blk != 0);      // the cost-to-go in front of knot 0 is never used (:396)
        // K(j, kx) for the lane's rows / columns kx = l, l+7: used by T1, A - B K and the new cost-to-go
        V Kr[2][7];
#pragma unroll
        for (int rI = 0; rI < 2; rI++)
#pragma unroll
            for (int j = 0; j < NU; j++) Kr[rI][j] = L::gather(W + kBpLgK, [rI, j](int l) { return (l + 7 * rI) * NU + j; });
        // ---- T1(kx, ky) = sum_j K(j,kx) Huu(j,ky) - Hxu(kx,ky)  for the lane's rows kx = l, l+7 ("K'Huu - Hxu")
        V T1[2][7];
        if (do_ctg) {
#pragma unroll
            for (int ky = 0; ky < NU; ky++) {
                T huu[7];
#pragma unroll
                for (int jj = 0; jj < NU; jj++) huu[jj] = W[kBpLgHuu + jj + 7 * ky];
#pragma unroll
                for (int rI = 0; rI < 2; rI++) {
                    V val = Kr[rI][0] * V(huu[0]);
#pragma unroll
                    for (int jj = 1; jj < NU; jj++) val = val + Kr[rI][jj] * V(huu[jj]);
                    T1[rI][ky] = val - Hxu[rI][ky];
                }
                L::sched_fence();
            }
        }
        if (M > 1) {                                      // forward-sweep operands A - B K and B du (computeFSVars)
            const unsigned obApBK = knot * (NX * NX), obBdu = knot * NX;
            // columns ky = l, l+7 of ApBK (all kx): ApBK(kx,ky) = A(kx,ky) - sum_j B(kx,j) K(j,ky)
#pragma unroll
            for (int kx = 0; kx < NX; kx++) {
                T brow[7];
#pragma unroll
                for (int jj = 0; jj < NU; jj++) brow[jj] = Bl[kx + NX * jj];
#pragma unroll
                for (int cI = 0; cI < 2; cI++) {
                    V val = V(brow[0]) * Kr[cI][0];
#pragma unroll
                    for (int jj = 1; jj < NU; jj++) val = val + V(brow[jj]) * Kr[cI][jj];
                    L::scatter_at(a.ApBK, obApBK, [cI, kx](int l) { return (l + 7 * cI) * NX + kx; }, ABc[cI][kx] - val, act);
                }
                L::sched_fence();
            }
            T dv[7];
#pragma unroll
            for (int jj = 0; jj < NU; jj++) dv[jj] = dul[jj];
#pragma unroll
            for (int rI = 0; rI < 2; rI++) {               // Bdu rows l, l+7
                V val = L::gather(Bl, [rI](int l) { return l + 7 * rI; }) * V(dv[0]);
#pragma unroll
                for (int jj = 1; jj < NU; jj++) val = val + L::gather(Bl, [rI, jj](int l) { return l + 7 * rI + NX * jj; }) * V(dv[jj]);
                L::scatter_at(a.Bdu, obBdu, [rI](int l) { return l + 7 * rI; }, val, act);
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
            const unsigned oPprev = a.oPw + (knot - 1) * (NX * NX), opprev = a.opw + (knot - 1) * NX;
#pragma nounroll
            for (int ky = 0; ky < NX; ky++) {
                T kk[7], hx[7];
#pragma unroll
                for (int jj = 0; jj < NU; jj++) { kk[jj] = W[kBpLgK + ky * NU + jj]; hx[jj] = W[kBpLgHux + jj + 7 * ky]; }
#pragma unroll
                for (int rI = 0; rI < 2; rI++) {
                    V val = T1[rI][0] * V(kk[0]) - Kr[rI][0] * V(hx[0]);
#pragma unroll
                    for (int jj = 1; jj < NU; jj++) val = val + (T1[rI][jj] * V(kk[jj]) - Kr[rI][jj] * V(hx[jj]));
                    const V v = L::gather(Pl, [rI, ky](int l) { return ky * NX + l + 7 * rI; }) + val;       // Hxx(kx, ky), overwritten in place
                    L::scatter(Pl, [rI, ky](int l) { return ky * NX + l + 7 * rI; }, v, act);
                    L::scatter_at(a.Pm, oPprev, [rI, ky](int l) { return ky * NX + l + 7 * rI; }, v, act);
                }
            }
            T dv[7], gv[7];
#pragma unroll
            for (int jj = 0; jj < NU; jj++) { dv[jj] = dul[jj]; gv[jj] = gul[jj]; }
#pragma unroll
            for (int rI = 0; rI < 2; rI++) {
                V val = V(dv[0]) * T1[rI][0] - Kr[rI][0] * V(gv[0]);
#pragma unroll
                for (int jj = 1; jj < NU; jj++) val = val + (V(dv[jj]) * T1[rI][jj] - Kr[rI][jj] * V(gv[jj]));
                const V v = gc[rI] + val;
                L::scatter(pl, [rI](int l) { return l + 7 * rI; }, v, act);
                L::scatter_at(a.pv, opprev, [rI](int l) { return l + 7 * rI; }, v, act);
            }
        }
        wsync();
    }
    // dJexp[2 blk], dJexp[2 blk + 1]: the 7 partial sums in lane order
    const V a0 = lg_chain_sum<L>(V(T(0)), dJ0), a1 = lg_chain_sum<L>(V(T(0)), dJ1);
    const typename L::M last = L::lane_is(6);
    L::scatter_at(a.dJexp, a.odJ, [blk](int) { return 2 * blk; }, a0, last);
    L::scatter_at(a.dJexp, a.odJ, [blk](int) { return 2 * blk + 1; }, a1, last);
}

// (problem pb, block blk): pointer set-up of bp_body() (bodies.hpp) around arm_lg_bp_block
template <typename L, typename T>
PDDP_HD void arm_lg_bp_body(T* lds, const Buffers<T>& b, const Dims& dm, int blk, int pb, bool write_err) {
    constexpr int NX = 14, NU = 7, NM = 21;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    if (st.done) return;
    BpLgArgs<T> a;
    a.AB = b.AB; a.Pm = b.P; a.pv = b.p; a.Pp = b.Pp; a.pp = b.pp; a.H = b.H; a.g = b.g; a.KT = b.KT; a.du = b.du; a.dcur = b.dcur;
    a.ApBK = b.ApBK; a.Bdu = b.Bdu; a.xb = b.xb; a.dJexp = b.dJexp;
    a.pbN = (unsigned)pb * N;
    {   // P and Pp (p and pp) are the two halves of one allocation: Pp = P + half
        const unsigned halfP = (unsigned)(b.Pp - b.P), halfp = (unsigned)(b.pp - b.p);
        a.oPw = st.pw ? halfP : 0u; a.opw = st.pw ? halfp : 0u; a.oPr = st.pw ? 0u : halfP; a.opr = st.pw ? 0u : halfp;
    }
    a.oxc = ((unsigned)pb * 2 + st.cur) * N * NX; a.oxp2 = ((unsigned)pb * 2 + st.cur2) * N * NX;
    a.odJ = (unsigned)pb * 2 * dm.M;
    a.rho = st.rho;
    arm_lg_bp_block<L, T>(lds, dm, blk, a);
    if (write_err) b.err[(size_t)pb * dm.M + blk] = 0;                        // the generic 7x7 inversion never reports failure (utils/cudaUtils.h:291)
}

}  // namespace pddp
