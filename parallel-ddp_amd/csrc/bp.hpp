// Backward (Riccati-like) pass: one wavefront owns one of the M blocks of knots and walks it backwards.
//
// Replaces backPassKern (DDPHelpers/bpHelpers.cuh:339-420) and the inner routines it calls:
// linearXfrmOrLoad :18-34, backprop :39-93, computeKTdu_dim1 :98-128, invHuu_dim4 :132-188, invHuu :192-204 with
// invertMatrix (utils/cudaUtils.h:236-292), computeKTdu :208-220, computeCTG :225-276, computeFSVars :281-312,
// computeExpRed :317-334.  Results follow the reference's index expressions, including the asymmetric placement of
// the Tassa regulariser (rho enters the x-row/u-col block of H and Huu, not the u-row/x-col block that K is
// computed from -- SURVEY.md section 8(a) "quirks").
//
// Data movement differs from the reference: P/p/K/du/H/g of the current knot live in this wave's LDS for the whole
// block, AB_k streams in once (coalesced), the cost Hessian is added straight from global memory, and the wave
// keeps the dJexp partial sums in LDS lanes exactly like the reference's per-thread partials (:326-328, :416).
#pragma once

#include "lanegroup.hpp"

#include "plants.hpp"

namespace pddp {

template <typename P, typename T>
struct BpScratch {
    static constexpr int NX = P::NX, NU = P::NU, NM = P::NX + P::NU;
    T Pm[NX * NX], pv[NX];
    T AB[NX * NM], AB2[NX * NM];
    T H[NM * NM], g[NM];
    T K[NU * NX], du[NU];
    T Huu[2 * NU * NU + 2 * NU + 2];
    T dx[NX];
    T dJ[2 * NU];
};

// Pointers of ONE problem, as the reference passes them (DDPWrappers.cuh:56-58).  xcur is the current trajectory
// (the reference's h_d_x[alphaIndex], equal to d_xp at this point), xprev2 is d_xp2, dcur the current defects.
template <typename T>
struct BpArgs {
    const T* AB; T* Pm; T* pv; const T* Pp; const T* pp; const T* H; const T* g;
    T* KT; T* du; const T* dcur; T* ApBK; T* Bdu; const T* xcur; const T* xprev2;
    T* dJexp;   // [2*M]
    int* err;   // [M]
    T rho;
    T* Hrw = nullptr; T* grw = nullptr;   // host CPU path only (cpu_twin.cpp): the reference's CPU backprop accumulates H, g of every knot IN PLACE
                                          // (bpHelpers.cuh:90-91) where its GPU kernel writes shared memory (:86-87); the kernels leave these null
};

// returns 1 (uniformly over the wave) on a failed Huu inversion
template <typename P, typename T>
PDDP_HD int bp_block(const Wave& w, BpScratch<P, T>& s, const Dims& dm, int blk, const BpArgs<T>& a) {
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU;
    constexpr int oHXU = NX * NM, oHUU = NX * NM + NX, oHUX = NX, oB = NX * NX;
    const int N = dm.N, M = dm.M, NBk = dm.NB;
    const T rho = a.rho;
    int ks = NBk * (blk + 1) - 1, iterCount;
    bool lin = true;                          // LINEAR_TRANSFORM_SWITCH 1 (config.cuh:81)
    PDDP_FOR(i, 2 * NU) s.dJ[i] = 0;
    if (ks == N - 1) {                        // last block: cost-to-go at N-1 is the final cost (bpHelpers.cuh:362-367)
        T* Pprev = a.Pm + NX * NX * (ks - 1); T* pprev = a.pv + NX * (ks - 1);
        const T* Hf = a.H + NM * NM * ks; const T* gf = a.g + NM * ks;
        PDDP_FOR(e, NX * NX) { const T v = Hf[(e % NX) + NM * (e / NX)]; s.Pm[e] = v; Pprev[e] = v; }
        PDDP_FOR(e, NX) { const T v = gf[e]; s.pv[e] = v; pprev[e] = v; }
        ks--; iterCount = NBk - 2; lin = false;
    } else {                                  // boundary cost-to-go of the PREVIOUS iteration (FORCE_PARALLEL, :369)
        iterCount = NBk - 1;
        const T* bP = a.Pp + NX * NX * ks;
        PDDP_FOR(e, NX * NX) s.Pm[e] = bP[e];
        PDDP_FOR(e, NX) s.dx[e] = a.xcur[NX * (ks + 1) + e] - a.xprev2[NX * (ks + 1) + e];
    }
    wsync(w);
    if (lin) {                                // p = pp + Pp (x - xp2)   (linearXfrmOrLoad)
        const T* bp = a.pp + NX * ks;
        PDDP_FOR(r, NX) { T dot = 0; for (int j = 0; j < NX; j++) dot += s.Pm[r + NX * j] * s.dx[j]; s.pv[r] = dot + bp[r]; }
        wsync(w);
    }
    for (int iter = iterCount; iter >= 0; iter--, ks--) {
        const T* bAB = a.AB + NX * NM * ks; const T* bH = a.H + NM * NM * ks; const T* bg = a.g + NM * ks;
        const T* bd = a.dcur + NX * ks;
        PDDP_FOR(e, NX * NM) s.AB[e] = bAB[e];
        wsync(w);
        PDDP_FOR(e, NX * NM) {                // AB2 = AB' (P + rho I on the B rows)
            const int ky = e / NM, kx = e % NM;
            T val = 0;
            for (int j = 0; j < NX; j++) val += s.AB[kx * NX + j] * (s.Pm[ky * NX + j] + ((kx >= NX && ky == j) ? rho : T(0)));
            s.AB2[e] = val;
        }
        if (M > 1 && dm.on_defect_boundary(iter)) {   // p += P d  (tests the loop counter like the reference, :73)
            PDDP_FOR(r, NX) { T val = 0; for (int j = 0; j < NX; j++) val += bd[j] * s.Pm[r + j * NX]; s.pv[r] += val; }
        }
        wsync(w);
        PDDP_FOR(e, NM * NM) {                // H = (AB2 AB)' + H_cost
            const int ky = e / NM, kx = e % NM;
            T dot = 0;
            for (int j = 0; j < NX; j++) dot += s.AB2[ky + NM * j] * s.AB[kx * NX + j];
            s.H[e] = dot + bH[e];
        }
        PDDP_FOR(kx, NM) {                    // g = AB' p + g_cost
            T dot = 0;
            for (int j = 0; j < NX; j++) dot += s.pv[j] * s.AB[kx * NX + j];
            s.g[kx] = dot + bg[kx];
        }
        wsync(w);
        if (a.Hrw) {
            PDDP_FOR(e, NM * NM) a.Hrw[(size_t)NM * NM * ks + e] = s.H[e];
            PDDP_FOR(kx, NM) a.grw[(size_t)NM * ks + kx] = s.g[kx];
            wsync(w);
        }
        T* bKT = a.KT + NX * NU * ks; T* bdu = a.du + NU * ks;
        if (NU == 1) {                        // scalar Huu (computeKTdu_dim1)
            if (s.H[oHUU] <= T(0)) return 1;
            const T val = T(1) / s.H[oHUU];
            PDDP_FOR(ky, NX) { const T k = s.H[oHUX + ky * NM] * val; s.K[ky] = k; bKT[ky] = k; }
            if (w.lane == 0) { const T v = s.g[oHUX] * val; s.du[0] = v; bdu[0] = v; }
            wsync(w);
        } else {
            T* Hinv;
            if (NU == 4) {                    // adjugate inverse with a det > 0 test (invHuu_dim4)
                T* A2 = &s.Huu[16]; T* adj = &s.Huu[0];
                PDDP_FOR(e, 16) A2[e] = s.H[oHUU + (e % 4) + NM * (e / 4)];
                wsync(w);
                PDDP_FOR(e, 16) {
                    const int ky = e / 4, kx = e % 4;
                    const int r0 = (kx + 1) % 4, c0 = (ky + 1) % 4, r1 = (r0 + 1) % 4, c1 = (c0 + 1) % 4, r2 = (r1 + 1) % 4, c2 = (c1 + 1) % 4;
                    const T f0 = A2[c0 * 4 + r0], f1 = A2[c0 * 4 + r1], f2 = A2[c0 * 4 + r2];
                    const T f3 = A2[c1 * 4 + r0], f4 = A2[c1 * 4 + r1], f5 = A2[c1 * 4 + r2];
                    const T f6 = A2[c2 * 4 + r0], f7 = A2[c2 * 4 + r1], f8 = A2[c2 * 4 + r2];
                    const T cdet = f0 * f4 * f8 + f3 * f7 * f2 + f6 * f1 * f5 - f2 * f4 * f6 - f5 * f7 * f0 - f8 * f1 * f3;
                    adj[ky * 4 + kx] = ((kx + ky) % 2 ? T(-1) : T(1)) * cdet;
                }
                wsync(w);
                const T val = T(1) / (adj[0] * A2[0] + adj[1] * A2[1] + adj[2] * A2[2] + adj[3] * A2[3]);
                if (val <= T(0)) return 1;
                wsync(w);                      // every lane has read A2[0..3] before it is overwritten
                PDDP_FOR(e, 16) { const int ky = e / 4, kx = e % 4; A2[kx * 4 + ky] = val * adj[ky * 4 + kx]; }
                wsync(w);
                Hinv = A2;
            }
#if defined(__HIP_DEVICE_COMPILE__)
            else if (w.block && NU == 7) {    // workgroup variant (k_bp_wide, the latency case): the 7 pivots cost 14 barriers below; here lane r of
                // the first lane group keeps row r of [Huu | I] in registers and the pivot rows travel by DPP (lanegroup.hpp) -- the same
                // operations per element (bp_lg.hpp uses the same scheme), no barrier per pivot
                T* A = &s.Huu[0];
                if (threadIdx.x < 7) {        // lane 7 of the group must be inactive for the broadcasts
                    using L = LgDevice<T>;
                    const int r = threadIdx.x;
                    T R[14];
#pragma unroll
                    for (int kc = 0; kc < 7; kc++) { R[kc] = s.H[oHUU + r + NM * kc]; R[7 + kc] = T(r == kc ? 1 : 0); }
#define PDDP_WIDE_PIV(PV)                                                                                   \
                    {                                                                                       \
                        T rowp[8];                                                                          \
                        _Pragma("unroll") for (int kc = 0; kc < 8; kc++) rowp[kc] = L::template bcast<PV>(R[PV + kc]);  \
                        const T colp = R[PV];                                                               \
                        const T inv = T(1) / rowp[0];                                                       \
                        _Pragma("unroll") for (int kc = 0; kc < 8; kc++) R[PV + kc] = (r == PV) ? R[PV + kc] * inv : R[PV + kc] - colp * inv * rowp[kc]; \
                    }
                    PDDP_WIDE_PIV(0) PDDP_WIDE_PIV(1) PDDP_WIDE_PIV(2) PDDP_WIDE_PIV(3) PDDP_WIDE_PIV(4) PDDP_WIDE_PIV(5) PDDP_WIDE_PIV(6)
#undef PDDP_WIDE_PIV
#pragma unroll
                    for (int kc = 0; kc < 7; kc++) A[NU * NU + r + NU * kc] = R[7 + kc];
                }
                wsync(w);
                Hinv = &A[NU * NU];
            }
#endif
            else {                            // [Huu | I] unpivoted Gauss-Jordan, never reports failure (invHuu)
                T* A = &s.Huu[0]; T* gjC = &s.Huu[2 * NU * NU]; T* gjR = gjC + NU;
                PDDP_FOR(e, NU * NU) { const int ky = e / NU, kx = e % NU; A[e] = s.H[oHUU + kx + NM * ky]; A[NU * NU + e] = T(kx == ky ? 1 : 0); }
                wsync(w);
                for (int piv = 0; piv < NU; piv++) {
                    PDDP_FOR(kr, NU) gjC[kr] = A[kr + piv * NU];
                    PDDP_FOR(kc, NU + 1) gjR[kc] = A[piv + (piv + kc) * NU];
                    wsync(w);
                    PDDP_FOR(e, NU * (NU + 1)) {
                        const int kr = e % NU, kc = e / NU;
                        const T inv = T(1) / gjR[0];
                        T& v = A[kr + (kc + piv) * NU];
                        if (kr == piv) v *= inv; else v -= gjC[kr] * inv * gjR[kc];
                    }
                    wsync(w);
                }
                Hinv = &A[NU * NU];
            }
            PDDP_FOR(e, NU * NX) {            // K = Huu^-1 Hux ; KT stored transposed (n x m per knot)
                const int kx = e / NU, ky = e % NU;
                T dot = 0;
                for (int j = 0; j < NU; j++) dot += Hinv[ky + NU * j] * s.H[oHUX + kx * NM + j];
                s.K[ky + NU * kx] = dot;
                bKT[kx + NX * ky] = dot;
            }
            PDDP_FOR(r, NU) {                 // du = Huu^-1 gu
                T dot = 0;
                for (int j = 0; j < NU; j++) dot += Hinv[r + NU * j] * s.g[oHUX + j];
                s.du[r] = dot; bdu[r] = dot;
            }
            wsync(w);
        }
        const bool do_ctg = (iter != 0 || blk != 0);   // the cost-to-go in front of knot 0 is never used (:396)
        if (do_ctg) {
            PDDP_FOR(e, NX * NU) {            // K' Huu - Hxu  (into AB2, free by now)
                const int ky = e / NX, kx = e % NX;
                T val = 0;
                for (int j = 0; j < NU; j++) val += s.K[kx * NU + j] * s.H[oHUU + ky * NM + j];
                s.AB2[e] = val - s.H[oHXU + kx + NM * ky];
            }
        }
        if (M > 1) {                          // forward-sweep operands A - B K and B du (computeFSVars)
            T* bApBK = a.ApBK + NX * NX * ks; T* bBdu = a.Bdu + NX * ks;
            PDDP_FOR(e, NX * NX) {
                const int ky = e / NX, kx = e % NX;
                T val = 0;
                for (int j = 0; j < NU; j++) val += s.AB[oB + kx + NX * j] * s.K[ky * NU + j];
                bApBK[e] = s.AB[e] - val;
            }
            PDDP_FOR(kx, NX) { T val = 0; for (int j = 0; j < NU; j++) val += s.AB[oB + kx + NX * j] * s.du[j]; bBdu[kx] = val; }
        }
        PDDP_FOR(ind, NU) {                   // expected reduction, per-lane partial sums (computeExpRed)
            T dot = 0;
            for (int j = 0; j < NU; j++) dot += s.H[oHUU + ind + NM * j] * s.du[j];
            s.dJ[ind] += s.du[ind] * s.g[oHUX + ind];
            s.dJ[NU + ind] += s.du[ind] * dot;
        }
        wsync(w);
        if (do_ctg) {
            T* Pprev = a.Pm + NX * NX * (ks - 1); T* pprev = a.pv + NX * (ks - 1);
            PDDP_FOR(e, NX * NX) {
                const int ky = e / NX, kx = e % NX;
                T val = 0;
                for (int j = 0; j < NU; j++) val += s.AB2[kx + NX * j] * s.K[ky * NU + j] - s.K[kx * NU + j] * s.H[oHUX + ky * NM + j];
                const T v = s.H[kx + ky * NM] + val;
                s.Pm[e] = v; Pprev[e] = v;
            }
            PDDP_FOR(kx, NX) {
                T val = 0;
                for (int j = 0; j < NU; j++) val += s.du[j] * s.AB2[kx + NX * j] - s.K[kx * NU + j] * s.g[oHUX + j];
                const T v = s.g[kx] + val;
                s.pv[kx] = v; pprev[kx] = v;
            }
        }
        wsync(w);
    }
    if (w.lane == 0) {
        T a0 = s.dJ[0], a1 = s.dJ[NU];
        for (int j = 1; j < NU; j++) { a0 += s.dJ[j]; a1 += s.dJ[NU + j]; }
        a.dJexp[2 * blk] = a0; a.dJexp[2 * blk + 1] = a1; a.err[blk] = 0;
    }
    return 0;
}

}  // namespace pddp
