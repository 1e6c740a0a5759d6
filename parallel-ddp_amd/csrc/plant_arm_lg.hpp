// KUKA iiwa14 forward dynamics on a LANE GROUP: lane l of an 8-lane group owns link l, everything is in registers.
//
// Same algorithm and the same floating-point operations, in the same order per output element, as arm_dynamics()
// (plant_arm.hpp), which restates the reference's dynamics<T> (plants/dynamics_arm.cuh:2097-2163): world-frame link
// transforms, Pluecker transforms, world inertias Iw = TA' (I TA), composite inertias, twists, velocity-product
// accelerations, body / net wrenches, mass matrix M_ij = S_min . (Ic_max S_max), unpivoted Gauss-Jordan on [M | I],
// qdd = Minv tau.  Terms that multiply a structural zero of TA (its upper-right 3x3 block) or the constant last row
// (0 0 0 1) of a homogeneous transform are dropped: adding an exact zero does not change a sum.
//
// What changes is WHERE things live (lanegroup.hpp): per-link data = per-lane registers; the chain recursions
// (T_i = T_{i-1} Tb_i, v_i, JdotV_i forwards; Ic_i, net wrench backwards) are written as 6 sweeps of "own + neighbour"
// (after sweep s the first/last s+1 links hold their final value, computed from final neighbours -- the same additions
// in the same order as the serial loop); row r of [M | I] lives in lane r.
#pragma once

#include "lanegroup.hpp"
#include "plant_arm.hpp"

namespace pddp {

// Robot constants: pointers to the model tables (in LDS inside the kernels -- lane l reads I[36 l + e], 7 distinct banks,
// the 8 groups of a wave read the same addresses = broadcast -- so that they do not pin 48 VGPRs for a whole rollout).
template <typename L>
struct ArmLgConst {
    const typename L::Scalar* Itab;    // [7][36] link spatial inertias
    const typename L::Scalar* Ftab;    // [7][16] fixed joint frames (row 3 of every frame is 0 0 0 1)
    typename L::Scalar grav;
    PDDP_HD typename L::V I(int e) const { return L::gather(Itab, [e](int b) { return 36 * b + e; }); }
    // rows (2 rp, 2 rp + 1) of column i of I as one pair (8-byte aligned: 36 b + 6 i + 2 rp is even)
    PDDP_HD typename L::V2 I2(int rp, int i) const { return L::gather2(Itab, [rp, i](int b) { return 36 * b + 6 * i + 2 * rp; }); }
    PDDP_HD typename L::V F(int col, int r) const { return L::gather(Ftab, [col, r](int b) { return 16 * b + 4 * col + r; }); }
};

template <typename L, typename T>
PDDP_HD void arm_lg_load_const(ArmLgConst<L>& c, const ArmModel<T>* mdl) { c.Itab = mdl->I; c.Ftab = mdl->F; c.grav = mdl->grav; }

// o = a x b on 3-vectors of per-lane values
template <typename V> PDDP_HD void lg_cross3(V* o, const V* a, const V* b) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename V> PDDP_HD V lg_dot6(const V* a, const V* b) {
    V s = a[0] * b[0];            // 0 + a0 b0 is exact
    s = s + a[1] * b[1]; s = s + a[2] * b[2]; s = s + a[3] * b[3]; s = s + a[4] * b[4]; s = s + a[5] * b[5];
    return s;
}
// o = A v, A column-major 6x6 in registers
template <typename V> PDDP_HD void lg_mat6_mul(V* o, const V* A, const V* v) {
#pragma unroll
    for (int r = 0; r < 6; r++) {
        V s = A[r] * v[0];
#pragma unroll
        for (int c = 1; c < 6; c++) s = s + A[r + 6 * c] * v[c];
        o[r] = s;
    }
}

// o = A v with A stored as 18 row pairs: A2[3*c + rp] = (A(2 rp, c), A(2 rp + 1, c)).  Same operations per element as lg_mat6_mul,
// two rows per packed instruction.
template <typename L>
PDDP_HD void lg_mat6_mul2(typename L::V* o, const typename L::V2* A2, const typename L::V* v) {
#pragma unroll
    for (int rp = 0; rp < 3; rp++) {
        typename L::V2 s = A2[rp] * L::splat(v[0]);
#pragma unroll
        for (int c = 1; c < 6; c++) s = s + A2[rp + 3 * c] * L::splat(v[c]);
        o[2 * rp] = s.x; o[2 * rp + 1] = s.y;
    }
}

// Working set that outlives arm_lg_dynamics() (the gradient needs it); all per lane = per link.
template <typename L>
struct ArmLgState {
    typename L::V S[6], v[6], JdV[6], t1[6], t2[6], Fj[6], Wn[6];
    typename L::V Iw[36], Ic[36];     // world / composite inertia, column-major 6x6            (PACK = false)
    typename L::V2 Iw2[18], Ic2[18];  // the same as row pairs: [3*col + rp] = rows (2 rp, 2 rp + 1)  (PACK = true); only one form is live
    typename L::V Minv[7];        // row `lane` of M^-1
};

// World frame of every link: Tw[3*col + row], rows 0..2 of T_i = T_{i-1} Tb_i(q_i), Tb = F Rz(q) (lane = link).
template <typename L>
PDDP_HD void arm_lg_world_frames(const ArmLgConst<L>& c, typename L::V q, typename L::V* Tw) {
    using V = typename L::V;
    using T = typename L::Scalar;
    V sn, cs;
    L::vsincos(q, sn, cs);
    // ---- link transform Tb = F Rz(q), rows 0..2, col-major index 3*col + r
    V Tb[12];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const V f0 = c.F(0, r), f1 = c.F(1, r);
        Tb[r] = cs * f0 + sn * f1;
        Tb[3 + r] = -sn * f0 + cs * f1;
        Tb[6 + r] = c.F(2, r);
        Tb[9 + r] = c.F(3, r);
    }
    // ---- world transforms T_i = T_{i-1} Tb_i.  Every sweep EVERY lane recomputes from its predecessor's current value: lanes
    // below s already hold (and reproduce, bit for bit) their final value, lane s becomes final in sweep s.  Lane 0's
    // predecessor is the identity (up() gives 0, the diagonal gets +1), and I * Tb_0 = Tb_0 exactly.
#pragma unroll
    for (int e = 0; e < 12; e++) Tw[e] = Tb[e];
    const V e0 = L::sel(L::lane_is(0), V(T(1)), V(T(0)));
#pragma unroll
    for (int s = 1; s < 7; s++) {
        V P[12];
#pragma unroll
        for (int e = 0; e < 12; e++) P[e] = L::up(Tw[e]);
        P[0] = P[0] + e0; P[4] = P[4] + e0; P[8] = P[8] + e0;
#pragma unroll
        for (int ky = 0; ky < 4; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                V val = P[kx] * Tb[3 * ky];
                val = val + P[kx + 3] * Tb[3 * ky + 1];
                val = val + P[kx + 6] * Tb[3 * ky + 2];
                if (ky == 3) val = val + P[kx + 9];
                Tw[3 * ky + kx] = val;
            }
    }
}

struct LgNoHook { template <typename V> PDDP_HD void operator()(const V*) const {} };

// q, qd, u: this lane's joint position, velocity, torque.  Returns this lane's qdd.
// PACK: do the 6x6 products two rows at a time with packed instructions (v_pk_mul_f32 / v_pk_add_f32).  Same operations per element;
// it shortens the rollout step by ~14 % but needs even-aligned register pairs, which costs the gradient kernel its second wave per
// SIMD -- so the forward pass uses PACK = true and next-iteration setup PACK = false.
// frame_hook(Tw): called once with the link frames (the end-effector cost reads the tool point off lane 6 there, ee_cost_lg.hpp).
template <typename L, bool PACK = false, typename Hook = LgNoHook>
PDDP_HD typename L::V arm_lg_dynamics(const ArmLgConst<L>& c, ArmLgState<L>& st, typename L::V q, typename L::V qd, typename L::V u,
                                      Hook frame_hook = Hook()) {
    using V = typename L::V;
    using T = typename L::Scalar;
    V Tw[12];
    arm_lg_world_frames<L>(c, q, Tw);
    frame_hook(Tw);
    // R(row, col) = Tw[3*col + row]; p = Tw[9..11]
    // ---- Pluecker transform TA = [R' 0; K R'],  K = skew(-R'p) R'   (Rt[row][col] = R(col,row); Kb[row][col])
    V tt[3];
    // indexing: the cooperative code reads a 4x4 column-major Tw4[4*col + row]; here Tw[3*col + row] (rows 0..2), so
    //   t[k] = -(Tw4[4k] Tw4[12] + Tw4[4k+1] Tw4[13] + Tw4[4k+2] Tw4[14]) = -(R(0,k) p0 + R(1,k) p1 + R(2,k) p2), R(j,k) = Tw[3k + j]
#pragma unroll
    for (int k = 0; k < 3; k++) tt[k] = -((Tw[3 * k] * Tw[9] + Tw[3 * k + 1] * Tw[10]) + Tw[3 * k + 2] * Tw[11]);
    V Rt[9], Kb[9];               // [3*col + row]
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
        for (int row = 0; row < 3; row++) {
            const int i1 = (row + 1) % 3, i2 = (row + 2) % 3;
            Rt[3 * col + row] = Tw[3 * row + col];                                   // Tw4[col + 4 row]
            Kb[3 * col + row] = -tt[i2] * Tw[3 * i1 + col] + tt[i1] * Tw[3 * i2 + col];
        }
    // joint axis S = [z; p x z]
    {
        V z[3] = {Tw[6], Tw[7], Tw[8]}, p[3] = {Tw[9], Tw[10], Tw[11]};
        st.S[0] = z[0]; st.S[1] = z[1]; st.S[2] = z[2];
        lg_cross3(st.S + 3, p, z);
    }
    using V2 = typename L::V2;
    V2 ITA2[18];                   // PACK
    V ITA[36];                     // !PACK
    if constexpr (PACK) {
    // TA column cc (6 entries): cc < 3: [Rt[.][cc]; Kb[.][cc]] ; cc >= 3: [0; Rt[.][cc-3]]
    // ---- ITA = I TA, two rows per packed instruction: ITA2[3*cc + rp] = rows (2 rp, 2 rp + 1) of column cc
#pragma unroll
    for (int cc = 0; cc < 6; cc++)
#pragma unroll
        for (int rp = 0; rp < 3; rp++) {
            V2 val;
            if (cc < 3) {
                val = c.I2(rp, 0) * L::splat(Rt[3 * cc]);
                val = val + c.I2(rp, 1) * L::splat(Rt[3 * cc + 1]);
                val = val + c.I2(rp, 2) * L::splat(Rt[3 * cc + 2]);
                val = val + c.I2(rp, 3) * L::splat(Kb[3 * cc]);
                val = val + c.I2(rp, 4) * L::splat(Kb[3 * cc + 1]);
                val = val + c.I2(rp, 5) * L::splat(Kb[3 * cc + 2]);
            } else {
                val = c.I2(rp, 3) * L::splat(Rt[3 * (cc - 3)]);
                val = val + c.I2(rp, 4) * L::splat(Rt[3 * (cc - 3) + 1]);
                val = val + c.I2(rp, 5) * L::splat(Rt[3 * (cc - 3) + 2]);
            }
            ITA2[3 * cc + rp] = val;
        }
    } else {
    // TA column cc (6 entries): cc < 3: [Rt[.][cc]; Kb[.][cc]] ; cc >= 3: [0; Rt[.][cc-3]]
    // ---- ITA = I TA   (col-major 6x6: ITA[6*cc + r])
#pragma unroll
    for (int cc = 0; cc < 6; cc++)
#pragma unroll
        for (int r = 0; r < 6; r++) {
            V val;
            if (cc < 3) {
                val = c.I(r) * Rt[3 * cc];
                val = val + c.I(r + 6) * Rt[3 * cc + 1];
                val = val + c.I(r + 12) * Rt[3 * cc + 2];
                val = val + c.I(r + 18) * Kb[3 * cc];
                val = val + c.I(r + 24) * Kb[3 * cc + 1];
                val = val + c.I(r + 30) * Kb[3 * cc + 2];
            } else {
                val = c.I(r + 18) * Rt[3 * (cc - 3)];
                val = val + c.I(r + 24) * Rt[3 * (cc - 3) + 1];
                val = val + c.I(r + 30) * Rt[3 * (cc - 3) + 2];
            }
            ITA[6 * cc + r] = val;
        }
    }
    // ---- twists v_i = S_i qd_i + v_{i-1}
    {
        V sq[6];
#pragma unroll
        for (int e = 0; e < 6; e++) { sq[e] = st.S[e] * qd; st.v[e] = sq[e]; }
#pragma unroll
        for (int s = 1; s < 7; s++)
#pragma unroll
            for (int e = 0; e < 6; e++) st.v[e] = sq[e] + L::up(st.v[e]);
    }
    if constexpr (PACK) {
    // ---- world inertia Iw = TA' ITA :  Iw(r, cc) = sum_i TA(i, r) ITA(i, cc), rows (2 rp, 2 rp + 1) per packed instruction.
    // TA(i, r): r < 3: i < 3 ? Rt[3r+i] : Kb[3r+i-3];  r >= 3: i < 3 ? 0 : Rt[3(r-3)+i-3].  The mixed pair (r = 2, 3) carries exact
    // zeros in its second half for i < 3 (0 * x + ... = the value the scalar code starts from).
    {
        const V zero = V(T(0));
        V2 TAp[6][3];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            TAp[i][0] = L::pair(i < 3 ? Rt[i] : Kb[i - 3], i < 3 ? Rt[3 + i] : Kb[3 + i - 3]);
            TAp[i][1] = L::pair(i < 3 ? Rt[6 + i] : Kb[6 + i - 3], i < 3 ? zero : Rt[i - 3]);
            TAp[i][2] = L::pair(i < 3 ? zero : Rt[3 + i - 3], i < 3 ? zero : Rt[6 + i - 3]);
        }
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
            V ita[6];
#pragma unroll
            for (int rp = 0; rp < 3; rp++) { ita[2 * rp] = ITA2[3 * cc + rp].x; ita[2 * rp + 1] = ITA2[3 * cc + rp].y; }
#pragma unroll
            for (int rp = 0; rp < 3; rp++) {
                V2 val;
                if (rp < 2) {
                    val = TAp[0][rp] * L::splat(ita[0]);
#pragma unroll
                    for (int i = 1; i < 6; i++) val = val + TAp[i][rp] * L::splat(ita[i]);
                } else {
                    val = TAp[3][rp] * L::splat(ita[3]);
                    val = val + TAp[4][rp] * L::splat(ita[4]);
                    val = val + TAp[5][rp] * L::splat(ita[5]);
                }
                st.Iw2[3 * cc + rp] = val;
            }
        }
    }
    } else {
    // ---- world inertia Iw = TA' ITA :  Iw[6*cc + r] = sum_i TA[6*r + i] ITA[6*cc + i]
#pragma unroll
    for (int cc = 0; cc < 6; cc++)
#pragma unroll
        for (int r = 0; r < 6; r++) {
            V val;
            if (r < 3) {
                val = Rt[3 * r] * ITA[6 * cc];
                val = val + Rt[3 * r + 1] * ITA[6 * cc + 1];
                val = val + Rt[3 * r + 2] * ITA[6 * cc + 2];
                val = val + Kb[3 * r] * ITA[6 * cc + 3];
                val = val + Kb[3 * r + 1] * ITA[6 * cc + 4];
                val = val + Kb[3 * r + 2] * ITA[6 * cc + 5];
            } else {
                val = Rt[3 * (r - 3)] * ITA[6 * cc + 3];
                val = val + Rt[3 * (r - 3) + 1] * ITA[6 * cc + 4];
                val = val + Rt[3 * (r - 3) + 2] * ITA[6 * cc + 5];
            }
            st.Iw[6 * cc + r] = val;
        }
    }
    // ---- velocity-product acceleration of this link: qd (crm(v) S)
    V cvs[6];
    {
        V o[6], t3[3];
        lg_cross3(o, st.v, st.S);
        lg_cross3(o + 3, st.v, st.S + 3);
        lg_cross3(t3, st.v + 3, st.S);
        o[3] = o[3] + t3[0]; o[4] = o[4] + t3[1]; o[5] = o[5] + t3[2];
#pragma unroll
        for (int e = 0; e < 6; e++) cvs[e] = qd * o[e];
    }
    // ---- composite inertias Ic_i = Ic_{i+1} + Iw_i, JdotV_i = cvs_i + JdotV_{i-1}
    if constexpr (PACK) {
#pragma unroll
        for (int e = 0; e < 18; e++) st.Ic2[e] = st.Iw2[e];
#pragma unroll
        for (int s = 1; s < 7; s++)
#pragma unroll
            for (int e = 0; e < 18; e++) { st.Ic2[e].x = L::down(st.Ic2[e].x) + st.Iw2[e].x; st.Ic2[e].y = L::down(st.Ic2[e].y) + st.Iw2[e].y; }
    } else {
#pragma unroll
        for (int e = 0; e < 36; e++) st.Ic[e] = st.Iw[e];
#pragma unroll
        for (int s = 1; s < 7; s++)
#pragma unroll
            for (int e = 0; e < 36; e++) st.Ic[e] = L::down(st.Ic[e]) + st.Iw[e];
    }
#pragma unroll
    for (int e = 0; e < 6; e++) st.JdV[e] = cvs[e];
#pragma unroll
    for (int s = 1; s < 7; s++)
#pragma unroll
        for (int e = 0; e < 6; e++) st.JdV[e] = cvs[e] + L::up(st.JdV[e]);
    // ---- Iw v, Iw (JdotV + g), Ic S
    {
        V ag[6];
#pragma unroll
        for (int e = 0; e < 6; e++) ag[e] = e == 5 ? st.JdV[e] + V(c.grav) : st.JdV[e];
        if constexpr (PACK) {
            lg_mat6_mul2<L>(st.t1, st.Iw2, st.v);
            lg_mat6_mul2<L>(st.t2, st.Iw2, ag);
            lg_mat6_mul2<L>(st.Fj, st.Ic2, st.S);
        } else {
            lg_mat6_mul(st.t1, st.Iw, st.v);
            lg_mat6_mul(st.t2, st.Iw, ag);
            lg_mat6_mul(st.Fj, st.Ic, st.S);
        }
    }
    // ---- body wrench Wb = crf(v) (Iw v) + Iw (JdotV + g); net wrench Wn_i = Wn_{i+1} + Wb_i
    V Wb[6];
    {
        V o[6], t3[3];
        lg_cross3(o, st.v, st.t1);
        lg_cross3(t3, st.v + 3, st.t1 + 3);
        o[0] = o[0] + t3[0]; o[1] = o[1] + t3[1]; o[2] = o[2] + t3[2];
        lg_cross3(o + 3, st.v, st.t1 + 3);
#pragma unroll
        for (int e = 0; e < 6; e++) Wb[e] = o[e] + st.t2[e];
    }
#pragma unroll
    for (int e = 0; e < 6; e++) st.Wn[e] = Wb[e];
#pragma unroll
    for (int s = 1; s < 7; s++)
#pragma unroll
        for (int e = 0; e < 6; e++) st.Wn[e] = L::down(st.Wn[e]) + Wb[e];
    // ---- mass matrix row of this lane: M(l, k) = S_min(l,k) . Fj_max(l,k)
    V A[14];                       // row `lane` of [M | I]
    {
        V Lw[7];                   // Lw[k] = S_k . Fj_lane  (meaningful for k <= lane)
#define PDDP_LG_ROW(K)                                                                                      \
        { V Sk[6]; _Pragma("unroll") for (int e = 0; e < 6; e++) Sk[e] = L::template bcast<K>(st.S[e]); Lw[K] = lg_dot6(Sk, st.Fj); }
        PDDP_LG_ROW(0) PDDP_LG_ROW(1) PDDP_LG_ROW(2) PDDP_LG_ROW(3) PDDP_LG_ROW(4) PDDP_LG_ROW(5) PDDP_LG_ROW(6)
#undef PDDP_LG_ROW
        // upper part: M(l, k) for k > l is lane k's Lw[l]
#define PDDP_LG_COL(K)                                                                                      \
        {                                                                                                   \
            V up_val = Lw[K];                                                                               \
            _Pragma("unroll") for (int j = 0; j < K; j++) { const V cand = L::template bcast<K>(Lw[j]); up_val = L::sel(L::lane_is(j), cand, up_val); } \
            A[K] = up_val;                                                                                  \
        }
        PDDP_LG_COL(0) PDDP_LG_COL(1) PDDP_LG_COL(2) PDDP_LG_COL(3) PDDP_LG_COL(4) PDDP_LG_COL(5) PDDP_LG_COL(6)
#undef PDDP_LG_COL
#pragma unroll
        for (int k = 0; k < 7; k++) A[7 + k] = L::sel(L::lane_is(k), V(T(1)), V(T(0)));
    }
    // ---- joint torques tau = u - (S . Wn + 0.5 qd)
    const V tau = u - (lg_dot6(st.S, st.Wn) + V(T(0.5)) * qd);
    // ---- unpivoted Gauss-Jordan on [M | I]
#define PDDP_LG_PIV(PV)                                                                                     \
    {                                                                                                       \
        V rowp[8];                                                                                          \
        _Pragma("unroll") for (int kc = 0; kc < 8; kc++) rowp[kc] = L::template bcast<PV>(A[PV + kc]);      \
        const V colp = A[PV];                                                                               \
        const V inv = V(T(1)) / rowp[0];                                                                    \
        const typename L::M isp = L::lane_is(PV);                                                           \
        _Pragma("unroll") for (int kc = 0; kc < 8; kc++) A[PV + kc] = L::sel(isp, A[PV + kc] * inv, A[PV + kc] - colp * inv * rowp[kc]); \
    }
    PDDP_LG_PIV(0) PDDP_LG_PIV(1) PDDP_LG_PIV(2) PDDP_LG_PIV(3) PDDP_LG_PIV(4) PDDP_LG_PIV(5) PDDP_LG_PIV(6)
#undef PDDP_LG_PIV
#pragma unroll
    for (int k = 0; k < 7; k++) st.Minv[k] = A[7 + k];
    // ---- qdd_l = sum_i Minv(l,i) tau_i
    V qdd;
    {
        V tk[7];
        tk[0] = L::template bcast<0>(tau); tk[1] = L::template bcast<1>(tau); tk[2] = L::template bcast<2>(tau); tk[3] = L::template bcast<3>(tau);
        tk[4] = L::template bcast<4>(tau); tk[5] = L::template bcast<5>(tau); tk[6] = L::template bcast<6>(tau);
        qdd = st.Minv[0] * tk[0];
#pragma unroll
        for (int i = 1; i < 7; i++) qdd = qdd + st.Minv[i] * tk[i];
    }
    return qdd;
}


// ------------------------------------------------------------------------------------------------ gradient on a lane group
// o = crm(a) b and o = crf(a) f on per-lane 6-vectors, operation order of crm_mul / crf_mul (pddp_common.hpp)
template <typename V> PDDP_HD void lg_crm_mul(V* o, const V* a, const V* b) {
    V t[3];
    lg_cross3(o, a, b);
    lg_cross3(o + 3, a, b + 3);
    lg_cross3(t, a + 3, b);
    o[3] = o[3] + t[0]; o[4] = o[4] + t[1]; o[5] = o[5] + t[2];
}
template <typename V> PDDP_HD void lg_crf_mul(V* o, const V* a, const V* f) {
    V t[3];
    lg_cross3(o, a, f);
    lg_cross3(t, a + 3, f + 3);
    o[0] = o[0] + t[0]; o[1] = o[1] + t[1]; o[2] = o[2] + t[2];
    lg_cross3(o + 3, a, f + 3);
}

// Analytic gradient of the forward dynamics, same derivation and the same operations per output element as
// arm_dynamics_gradient() (plant_arm.hpp, stages G1..G7).  Lane i owns link i = row i of every Jacobian column; the loop runs
// over the derivative index k (q_k and qd_k together), so only one column's worth of pair quantities is live:
//     from lane k:   S_k, v_k, Ic_k                                  (dynamic intra-group broadcasts)
//     own:           dS_i/dq_k, dv_i/dq_k, tmpM_ik, d(JdotV)/d(q,qd)_k (prefix chain), dWb (suffix chain)
//     dM/dq_k:       lane i builds its lower-triangular row entries (c <= i) from lane c's dS_c/dq_k and S_c; the upper
//                    ones are the transposed entries of the lanes above (dM is symmetric by construction, bit for bit)
// st must hold the state arm_lg_dynamics<L, false>() left for (q, qd, u); qdd is this lane's acceleration.
// emit(jj, val): column jj (0..6 d/dq, 7..13 d/dqd, 14..20 d/du) of dqdd, row = lane.
template <typename L, typename Emit>
PDDP_HD void arm_lg_gradient(const ArmLgConst<L>& c, const ArmLgState<L>& st, typename L::V qd, typename L::V qdd, Emit emit) {
    using V = typename L::V;
    using T = typename L::Scalar;
    const V zero = V(T(0));
    // loop-invariant broadcasts: every joint axis and every acceleration
    V Sall[7][6], qddall[7];
#define PDDP_LG_ALL(K) { _Pragma("unroll") for (int e = 0; e < 6; e++) Sall[K][e] = L::template bcast<K>(st.S[e]); qddall[K] = L::template bcast<K>(qdd); }
    PDDP_LG_ALL(0) PDDP_LG_ALL(1) PDDP_LG_ALL(2) PDDP_LG_ALL(3) PDDP_LG_ALL(4) PDDP_LG_ALL(5) PDDP_LG_ALL(6)
#undef PDDP_LG_ALL
    V ag[6];                                                  // JdotV + g
#pragma unroll
    for (int e = 0; e < 6; e++) ag[e] = e == 5 ? st.JdV[e] + V(c.grav) : st.JdV[e];

    for (int k = 0; k < 7; k++) {
        const typename L::M below = L::lane_lt(k);            // i <  k
        const typename L::M above = L::lane_ge(k + 1);        // i >  k   (k < i)
        V Sk[6], vk[6];
#pragma unroll
        for (int e = 0; e < 6; e++) { Sk[e] = L::bcast_dyn(st.S[e], k); vk[e] = L::bcast_dyn(st.v[e], k); }
        // ---- G1
        V dS[6], dvq[6], tmpM[6];
        {
            V t6[6], dlt[6];
            lg_crm_mul(t6, Sk, st.S);
#pragma unroll
            for (int e = 0; e < 6; e++) { dS[e] = L::sel(above, t6[e], zero); dlt[e] = st.v[e] - vk[e]; }
            lg_crm_mul(t6, Sk, dlt);
#pragma unroll
            for (int e = 0; e < 6; e++) dvq[e] = L::sel(above, t6[e], zero);
            // k <= i: crf(S_k) F_i ;  k > i: crf(S_k)(Ic_k S_i) - Ic_k (crm(S_k) S_i)
            V lo[6], hi[6], Ick[36], a6[6], b6[6], c6[6];
            lg_crf_mul(lo, Sk, st.Fj);
#pragma unroll
            for (int e = 0; e < 36; e++) Ick[e] = L::bcast_dyn(st.Ic[e], k);
            lg_mat6_mul(a6, Ick, st.S);
            lg_crf_mul(hi, Sk, a6);
            lg_crm_mul(b6, Sk, st.S);
            lg_mat6_mul(c6, Ick, b6);
#pragma unroll
            for (int e = 0; e < 6; e++) tmpM[e] = L::sel(below, hi[e] - c6[e], lo[e]);
        }
        // ---- G2: increment of d(JdotV)/dq_k and closed form d(JdotV)/dqd_k
        V term[6], dJq[6];
        {
            V a6[6], b6[6];
            lg_crm_mul(a6, dvq, st.S);
            lg_crm_mul(b6, st.v, dS);
#pragma unroll
            for (int e = 0; e < 6; e++) term[e] = L::sel(above, (a6[e] + b6[e]) * qd, zero);
            lg_crm_mul(a6, vk, Sk);
#pragma unroll
            for (int e = 0; e < 6; e++) dJq[e] = L::sel(below, zero, dvq[e] + a6[e]);
        }
        // ---- G3: prefix over links
        {
            V inc[6];
#pragma unroll
            for (int e = 0; e < 6; e++) inc[e] = term[e];
#pragma unroll
            for (int s = 1; s < 7; s++)
#pragma unroll
                for (int e = 0; e < 6; e++) term[e] = L::up(term[e]) + inc[e];
        }
        // ---- G4: body-wrench derivatives (zero for links below k)
        V dWq[6], dWv[6];
        {
            V t6[6], r1[6], r2[6], r3[6], r4[6], in1[6], in2[6];
            lg_crm_mul(t6, Sk, ag);
#pragma unroll
            for (int e = 0; e < 6; e++) t6[e] = term[e] - t6[e];
            lg_mat6_mul(r1, st.Iw, t6);
            lg_crf_mul(r2, Sk, st.t2);
            lg_crf_mul(r3, dvq, st.t1);
            lg_crm_mul(t6, Sk, st.v);
#pragma unroll
            for (int e = 0; e < 6; e++) t6[e] = dvq[e] - t6[e];
            lg_mat6_mul(in1, st.Iw, t6);
            lg_crf_mul(in2, Sk, st.t1);
#pragma unroll
            for (int e = 0; e < 6; e++) in1[e] = in1[e] + in2[e];
            lg_crf_mul(r4, st.v, in1);
#pragma unroll
            for (int e = 0; e < 6; e++) dWq[e] = L::sel(below, zero, r1[e] + r2[e] + r3[e] + r4[e]);
            lg_mat6_mul(r1, st.Iw, dJq);
            // r2 of the qd column is crf(S_k)(Iw v) = in2
            lg_mat6_mul(t6, st.Iw, Sk);
            lg_crf_mul(r3, st.v, t6);
#pragma unroll
            for (int e = 0; e < 6; e++) dWv[e] = L::sel(below, zero, r1[e] + in2[e] + r3[e]);
        }
        // ---- G5: suffix over links (net-wrench derivatives)
        {
            V iq[6], iv[6];
#pragma unroll
            for (int e = 0; e < 6; e++) { iq[e] = dWq[e]; iv[e] = dWv[e]; }
#pragma unroll
            for (int s = 1; s < 7; s++)
#pragma unroll
                for (int e = 0; e < 6; e++) { dWq[e] = L::down(dWq[e]) + iq[e]; dWv[e] = L::down(dWv[e]) + iv[e]; }
        }
        // ---- dM/dq_k row of this lane and (dM/dq_k) qdd
        V mq;
        {
            V Lw[7];
#define PDDP_LG_DM(C)                                                                                                     \
            { V dSc[6]; _Pragma("unroll") for (int e = 0; e < 6; e++) dSc[e] = L::template bcast<C>(dS[e]);              \
              Lw[C] = lg_dot6(dSc, st.Fj) + lg_dot6(Sall[C], tmpM); }
            PDDP_LG_DM(0) PDDP_LG_DM(1) PDDP_LG_DM(2) PDDP_LG_DM(3) PDDP_LG_DM(4) PDDP_LG_DM(5) PDDP_LG_DM(6)
#undef PDDP_LG_DM
            V row[7];
#define PDDP_LG_DMT(C)                                                                                                    \
            { V uv = Lw[C]; _Pragma("unroll") for (int j = 0; j < C; j++) { const V cand = L::template bcast<C>(Lw[j]); uv = L::sel(L::lane_is(j), cand, uv); } row[C] = uv; }
            PDDP_LG_DMT(0) PDDP_LG_DMT(1) PDDP_LG_DMT(2) PDDP_LG_DMT(3) PDDP_LG_DMT(4) PDDP_LG_DMT(5) PDDP_LG_DMT(6)
#undef PDDP_LG_DMT
            mq = row[0] * qddall[0];
#pragma unroll
            for (int cc = 1; cc < 7; cc++) mq = mq + row[cc] * qddall[cc];
        }
        // ---- G6: dtau columns
        V tq, tv;
        {
            V val = lg_dot6(st.S, dWq);
            val = val + lg_dot6(dS, st.Wn);
            tq = -val - mq;                                    // -(val + 0) - mq
            const V val2 = lg_dot6(st.S, dWv);
            tv = -(val2 + L::sel(L::lane_is(k), V(T(0.5)), zero));
        }
        // ---- G7: dqdd columns = Minv dtau
        V oq, ov;
        {
            V bq[7], bv[7];
            bq[0] = L::template bcast<0>(tq); bq[1] = L::template bcast<1>(tq); bq[2] = L::template bcast<2>(tq); bq[3] = L::template bcast<3>(tq);
            bq[4] = L::template bcast<4>(tq); bq[5] = L::template bcast<5>(tq); bq[6] = L::template bcast<6>(tq);
            bv[0] = L::template bcast<0>(tv); bv[1] = L::template bcast<1>(tv); bv[2] = L::template bcast<2>(tv); bv[3] = L::template bcast<3>(tv);
            bv[4] = L::template bcast<4>(tv); bv[5] = L::template bcast<5>(tv); bv[6] = L::template bcast<6>(tv);
            oq = st.Minv[0] * bq[0]; ov = st.Minv[0] * bv[0];
#pragma unroll
            for (int i = 1; i < 7; i++) { oq = oq + st.Minv[i] * bq[i]; ov = ov + st.Minv[i] * bv[i]; }
        }
        emit(k, oq);
        emit(7 + k, ov);
    }
#pragma unroll
    for (int cc = 0; cc < 7; cc++) emit(14 + cc, st.Minv[cc]);
}

}  // namespace pddp
