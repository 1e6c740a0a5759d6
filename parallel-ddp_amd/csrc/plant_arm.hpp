// KUKA iiwa14 rigid-body dynamics q'' = M(q)^-1 (u - C(q,q') - 0.5 q') and its analytic gradient,
// one wavefront per evaluation.
//
// Replaces the reference plug-ins  dynamics<T> (plants/dynamics_arm.cuh:2097-2163)  and
// dynamicsGradient<T> (:2167-2289); output layouts are the reference's: qdd[7], and
// dqdd[col*7 + row] with columns (dq[7], dqd[7], du[7]) (utils/integrators.cuh:17).
//
// The forward pass keeps the reference's world-frame composite-rigid-body formulation (world inertias
// Iw = X^T I X, composite inertias, joint-space forces F = Ic S, M_ij = S_min . F_max, unpivoted
// Gauss-Jordan) so that rounding behaves alike.  The gradient is NOT the reference's dense
// dT/dTA/dIw tensor chain (21 KB of scratch, ~0.4 Mflop): it is re-derived with the world-frame
// identities, valid for joint j at or below link i on a serial chain,
//     dS_i/dq_j  = crm(S_j) S_i            (j <  i)
//     dIw_i/dq_j = crf(S_j) Iw_i - Iw_i crm(S_j)   (j <= i)
//     dv_i/dq_j  = crm(S_j) (v_i - v_j),   dv_i/dqd_j = S_j
// which turn every tensor contraction into a handful of 6-vector cross products and 6x6 mat-vecs
// (~10x fewer flops, ~9 KB of LDS), and remove the serial recursion over links.
#pragma once

#include "pddp_common.hpp"

namespace pddp {

constexpr int kArmNB = 7;

template <typename T>
struct ArmModel {            // read-only robot data (device global memory)
    T I[kArmNB * 36];        // link spatial inertias about the joint frame, col-major 6x6, [ang; lin]
    T F[kArmNB * 16];        // fixed joint frames F_i (col-major 4x4); link transform Tb_i = F_i * Rz(q_i)
    T grav;                  // 9.81, or 0 when gravity is compensated by the robot (MPC_MODE)
};

// EE_TYPE 0 / 2 (plants/dynamics_arm.cuh:50-65): with the default URDF link 7's spatial inertia is the base values x INERTIA_MODIFIER (1 / 3 / 5) and its mass
// 1.2 + WEIGHT_MODIFIER (0 / 0.03 / 0.5) (initI, :338-347); the tables hold EE_TYPE 1.  (The WAFR model's last link is fixed; there EE_TYPE only moves the tool point.)
template <typename T>
inline void arm_model_apply_ee_type(ArmModel<T>& m, int wafr_urdf, int ee_type) {
    if (wafr_urdf || ee_type == 1) return;
    const double im = ee_type == 0 ? 1.0 : 5.0, wm = ee_type == 0 ? 0.0 : 0.5;
    T* S = m.I + 36 * 6;
    S[0] = (T)(0.0055 * im); S[7] = (T)(0.0055 * im); S[14] = (T)(0.005 * im);
    S[4] = (T)(-0.024 * im); S[24] = (T)(-0.024 * im); S[9] = (T)(0.024 * im); S[19] = (T)(0.024 * im);
    S[21] = (T)(1.2 + wm); S[28] = (T)(1.2 + wm); S[35] = (T)(1.2 + wm);
}

template <typename T>
struct ArmScratch {          // per-wave LDS
    T I[kArmNB * 36];
    T F[kArmNB * 16];
    T grav;
    T sc[2 * kArmNB];
    T Tb[kArmNB * 16];
    T Tw[kArmNB * 16];
    T TA[kArmNB * 36];
    T S[kArmNB * 6];
    T ITA[kArmNB * 36];
    T Iw[kArmNB * 36];
    T Ic[kArmNB * 36];
    T v[kArmNB * 6];
    T cvs[kArmNB * 6];
    T JdV[kArmNB * 6];
    T t1[kArmNB * 6], t2[kArmNB * 6], Fj[kArmNB * 6], Wb[kArmNB * 6], Wn[kArmNB * 6];
    T MI[kArmNB * 2 * kArmNB];
    T gjC[kArmNB], gjR[kArmNB + 1];
    T tau[kArmNB];
};

template <typename T>
struct ArmGradScratch {      // additional per-wave LDS for the gradient
    T dS[49 * 6];            // dS_i/dq_j           [i][j]
    T dvq[49 * 6];           // dv_i/dq_j           [i][j]
    T tmpM[49 * 6];          // (dIc_i/dq_k) S_i + Ic_i dS_i/dq_k   [i][k]
    T term[49 * 6];          // per-link increment of d(JdotV)/dq, then its prefix sum over links
    T dJq[49 * 6];           // d(JdotV_i)/dqd_j    [i][j]
    T dWb[kArmNB * 14 * 6];  // d(Wb_i)/d(q,qd)_j   [i][jj], then suffix-summed over i
    T dTau[14 * kArmNB];     // [jj][i]
    T dM[kArmNB * 49];       // dM/dq_k             [k][c][r]
};

template <typename T>
PDDP_HD void arm_load_model(const Wave& w, ArmScratch<T>& s, const ArmModel<T>* mdl) {
    PDDP_FOR(e, kArmNB * 36) s.I[e] = mdl->I[e];
    PDDP_FOR(e, kArmNB * 16) s.F[e] = mdl->F[e];
    if (w.lane == 0) s.grav = mdl->grav;
    wsync();
}

// Forward dynamics.  x = [q; qd] (14), u (7) and qdd (7) live in LDS (or host memory).
template <typename T>
PDDP_HD void arm_dynamics(const Wave& w, ArmScratch<T>& s, T* qdd, const T* x, const T* u) {
    constexpr int NB = kArmNB;
    const T* qd = x + NB;
    PDDP_FOR(b, NB) { s.sc[b] = tsin<T>(x[b]); s.sc[NB + b] = tcos<T>(x[b]); }
    wsync();
    PDDP_FOR(e, NB * 16) {                       // link transforms Tb_i = F_i * Rz(q_i)
        const int b = e >> 4, col = (e >> 2) & 3, r = e & 3;
        const T sn = s.sc[b], cs = s.sc[NB + b];
        const T* F = &s.F[16 * b];
        T val;
        if (col == 0) val = cs * F[r] + sn * F[4 + r];
        else if (col == 1) val = -sn * F[r] + cs * F[4 + r];
        else val = F[4 * col + r];
        s.Tb[e] = val;
    }
    wsync();
    PDDP_FOR(e, 16) s.Tw[e] = s.Tb[e];            // world transforms, serial chain
    wsync();
    for (int b = 1; b < NB; b++) {
        PDDP_FOR(e, 16) {
            const int ky = e >> 2, kx = e & 3;
            T val = 0;
            for (int i = 0; i < 4; i++) val += s.Tw[16 * (b - 1) + kx + 4 * i] * s.Tb[16 * b + ky * 4 + i];
            s.Tw[16 * b + e] = val;
        }
        wsync();
    }
    PDDP_FOR(e, NB * 9) {                        // world->link Pluecker transforms TA_i = [R' 0; skew(-R'p) R'  R']
        const int b = e / 9, row = e % 3, col = (e % 9) / 3;
        const T* Tw = &s.Tw[16 * b];
        T t[3];
        for (int k = 0; k < 3; k++) t[k] = -(Tw[4 * k] * Tw[12] + Tw[4 * k + 1] * Tw[13] + Tw[4 * k + 2] * Tw[14]);
        const T rt = Tw[col + 4 * row];          // R'(row, col) = R(col, row)
        T* TA = &s.TA[36 * b];
        TA[col * 6 + row] = rt;
        TA[(col + 3) * 6 + row + 3] = rt;
        TA[(col + 3) * 6 + row] = 0;
        // (skew(t) R')(row, col) = sum_i skew(t)(row,i) R(col,i)
        const int i1 = (row + 1) % 3, i2 = (row + 2) % 3;
        TA[col * 6 + row + 3] = -t[i2] * Tw[col + 4 * i1] + t[i1] * Tw[col + 4 * i2];
    }
    PDDP_FOR(b, NB) {                            // joint axes S_i = [z_i; p_i x z_i]
        const T* Tw = &s.Tw[16 * b];
        T* S = &s.S[6 * b];
        S[0] = Tw[8]; S[1] = Tw[9]; S[2] = Tw[10];
        cross3(S + 3, Tw + 12, Tw + 8);
    }
    wsync();
    PDDP_FOR(e, NB * 36) {                       // I_i * TA_i
        const int b = e / 36, c = (e % 36) / 6, r = e % 6;
        T val = 0;
        for (int i = 0; i < 6; i++) val += s.I[36 * b + r + 6 * i] * s.TA[36 * b + c * 6 + i];
        s.ITA[e] = val;
    }
    PDDP_FOR(e, 6) {                             // link twists v_i = sum_{l<=i} S_l qd_l
        T run = 0;
        for (int b = 0; b < NB; b++) { run = s.S[6 * b + e] * qd[b] + run; s.v[6 * b + e] = run; }
    }
    wsync();
    PDDP_FOR(e, NB * 36) {                       // world inertias Iw_i = TA_i' (I_i TA_i)
        const int b = e / 36, c = (e % 36) / 6, r = e % 6;
        T val = 0;
        for (int i = 0; i < 6; i++) val += s.TA[36 * b + r * 6 + i] * s.ITA[36 * b + c * 6 + i];
        s.Iw[e] = val;
    }
    PDDP_FOR(b, NB) {                            // velocity-product accelerations, per link
        T o[6];
        crm_mul(o, &s.v[6 * b], &s.S[6 * b]);
        for (int i = 0; i < 6; i++) s.cvs[6 * b + i] = qd[b] * o[i];
    }
    wsync();
    PDDP_FOR(e, 36) {                            // composite inertias (suffix sums)
        T run = 0;
        for (int b = NB - 1; b >= 0; b--) { run += s.Iw[36 * b + e]; s.Ic[36 * b + e] = run; }
    }
    PDDP_FOR(e, 6) {                             // JdotV_i (prefix sums)
        T run = 0;
        for (int b = 0; b < NB; b++) { run = s.cvs[6 * b + e] + run; s.JdV[6 * b + e] = run; }
    }
    wsync();
    PDDP_FOR(e, 3 * NB * 6) {                    // Iw v,  Iw (JdotV + g),  Ic S
        const int which = e / (NB * 6), b = (e % (NB * 6)) / 6, r = e % 6;
        T val = 0;
        if (which == 0) { for (int i = 0; i < 6; i++) val += s.Iw[36 * b + r + 6 * i] * s.v[6 * b + i]; s.t1[6 * b + r] = val; }
        else if (which == 1) { for (int i = 0; i < 6; i++) val += s.Iw[36 * b + r + 6 * i] * (s.JdV[6 * b + i] + (i == 5 ? s.grav : T(0))); s.t2[6 * b + r] = val; }
        else { for (int i = 0; i < 6; i++) val += s.Ic[36 * b + r + 6 * i] * s.S[6 * b + i]; s.Fj[6 * b + r] = val; }
    }
    wsync();
    PDDP_FOR(b, NB) {                            // body wrenches Wb_i = crf(v_i) Iw_i v_i + Iw_i (JdotV_i + g)
        T o[6];
        crf_mul(o, &s.v[6 * b], &s.t1[6 * b]);
        for (int i = 0; i < 6; i++) s.Wb[6 * b + i] = o[i] + s.t2[6 * b + i];
    }
    PDDP_FOR(e, NB * NB) {                       // [M | I]
        const int b = e / NB, kx = e % NB;
        const int jI = kx <= b ? kx : b, iI = kx <= b ? b : kx;
        s.MI[b * NB + kx] = dot6(&s.S[6 * jI], &s.Fj[6 * iI]);
        s.MI[(b + NB) * NB + kx] = T(kx == b ? 1 : 0);
    }
    wsync();
    PDDP_FOR(e, 6) {                             // net wrenches (suffix sums)
        T run = 0;
        for (int b = NB - 1; b >= 0; b--) { run += s.Wb[6 * b + e]; s.Wn[6 * b + e] = run; }
    }
    wsync();
    PDDP_FOR(b, NB) s.tau[b] = u[b] - (dot6(&s.S[6 * b], &s.Wn[6 * b]) + T(0.5) * qd[b]);
    for (int piv = 0; piv < NB; piv++) {         // unpivoted Gauss-Jordan on [M | I] -> [I | Minv]
        PDDP_FOR(kr, NB) s.gjC[kr] = s.MI[kr + piv * NB];
        PDDP_FOR(kc, NB + 1) s.gjR[kc] = s.MI[piv + (piv + kc) * NB];
        wsync();
        PDDP_FOR(e, NB * (NB + 1)) {
            const int kr = e % NB, kc = e / NB;
            const T inv = T(1) / s.gjR[0];
            T& a = s.MI[kr + (kc + piv) * NB];
            if (kr == piv) a *= inv; else a -= s.gjC[kr] * inv * s.gjR[kc];
        }
        wsync();
    }
    PDDP_FOR(r, NB) {
        T val = 0;
        for (int i = 0; i < NB; i++) val += s.MI[NB * NB + r + NB * i] * s.tau[i];
        qdd[r] = val;
    }
    wsync();
}

// Gradient.  Runs the forward pass first (qdd is an output too).  dqdd: 7 x 21 col-major.
template <typename T>
PDDP_HD void arm_dynamics_gradient(const Wave& w, ArmScratch<T>& s, ArmGradScratch<T>& g, T* dqdd, T* qdd, const T* x,
                                   const T* u) {
    constexpr int NB = kArmNB;
    const T* qd = x + NB;
    arm_dynamics(w, s, qdd, x, u);
    const T* Minv = &s.MI[NB * NB];
    PDDP_FOR(e, NB * NB) {                       // G1: dS, dv/dq, and the dM building block
        const int i = e / NB, j = e % NB;
        T* dS = &g.dS[6 * e]; T* dv = &g.dvq[6 * e]; T* tm = &g.tmpM[6 * e];
        if (j < i) {
            T dlt[6];
            crm_mul(dS, &s.S[6 * j], &s.S[6 * i]);
            for (int c = 0; c < 6; c++) dlt[c] = s.v[6 * i + c] - s.v[6 * j + c];
            crm_mul(dv, &s.S[6 * j], dlt);
        } else {
            for (int c = 0; c < 6; c++) { dS[c] = 0; dv[c] = 0; }
        }
        // tmpM[i][k=j]: k <= i -> crf(S_k) F_i ;  k > i -> crf(S_k)(Ic_k S_i) - Ic_k (crm(S_k) S_i)
        if (j <= i) crf_mul(tm, &s.S[6 * j], &s.Fj[6 * i]);
        else {
            T a[6], b2[6], c2[6];
            mat6_mul(a, &s.Ic[36 * j], &s.S[6 * i]);
            crf_mul(tm, &s.S[6 * j], a);
            crm_mul(b2, &s.S[6 * j], &s.S[6 * i]);
            mat6_mul(c2, &s.Ic[36 * j], b2);
            for (int c = 0; c < 6; c++) tm[c] -= c2[c];
        }
    }
    wsync();
    PDDP_FOR(e, NB * NB) {                       // G2: increments of d(JdotV)/dq and closed form d(JdotV)/dqd
        const int l = e / NB, j = e % NB;
        T* tr = &g.term[6 * e]; T* dq = &g.dJq[6 * e];
        if (j < l) {
            T a[6], b2[6];
            crm_mul(a, &g.dvq[6 * e], &s.S[6 * l]);
            crm_mul(b2, &s.v[6 * l], &g.dS[6 * e]);
            for (int c = 0; c < 6; c++) tr[c] = (a[c] + b2[c]) * qd[l];
        } else for (int c = 0; c < 6; c++) tr[c] = 0;
        if (j <= l) {
            T a[6];
            crm_mul(a, &s.v[6 * j], &s.S[6 * j]);
            for (int c = 0; c < 6; c++) dq[c] = g.dvq[6 * e + c] + a[c];
        } else for (int c = 0; c < 6; c++) dq[c] = 0;
    }
    wsync();
    PDDP_FOR(e, NB * 6) {                        // G3: prefix over links -> d(JdotV_i)/dq_j (in place)
        const int j = e / 6, c = e % 6;
        T run = 0;
        for (int l = 0; l < NB; l++) { run += g.term[6 * (l * NB + j) + c]; g.term[6 * (l * NB + j) + c] = run; }
    }
    wsync();
    PDDP_FOR(e, NB * 2 * NB) {                   // G4: body-wrench derivatives
        const int i = e / (2 * NB), jj = e % (2 * NB);
        T* out = &g.dWb[6 * e];
        const int j = jj < NB ? jj : jj - NB;
        if (j > i) { for (int c = 0; c < 6; c++) out[c] = 0; continue; }
        const T* Iw = &s.Iw[36 * i]; const T* Sj = &s.S[6 * j]; const T* vi = &s.v[6 * i];
        if (jj < NB) {
            const int ij = i * NB + j;
            T a[6], t[6], r1[6], r2[6], r3[6], in1[6], in2[6], r4[6];
            for (int c = 0; c < 6; c++) a[c] = s.JdV[6 * i + c] + (c == 5 ? s.grav : T(0));
            crm_mul(t, Sj, a);
            for (int c = 0; c < 6; c++) t[c] = g.term[6 * ij + c] - t[c];
            mat6_mul(r1, Iw, t);                                  // Iw (dJdV - crm(S_j)(JdV + g))
            crf_mul(r2, Sj, &s.t2[6 * i]);                        // crf(S_j) Iw (JdV + g)
            crf_mul(r3, &g.dvq[6 * ij], &s.t1[6 * i]);            // crf(dv) Iw v
            crm_mul(t, Sj, vi);
            for (int c = 0; c < 6; c++) t[c] = g.dvq[6 * ij + c] - t[c];
            mat6_mul(in1, Iw, t);
            crf_mul(in2, Sj, &s.t1[6 * i]);
            for (int c = 0; c < 6; c++) in1[c] += in2[c];
            crf_mul(r4, vi, in1);                                 // crf(v)(dIw v + Iw dv)
            for (int c = 0; c < 6; c++) out[c] = r1[c] + r2[c] + r3[c] + r4[c];
        } else {
            T r1[6], r2[6], t[6], r3[6];
            mat6_mul(r1, Iw, &g.dJq[6 * (i * NB + j)]);
            crf_mul(r2, Sj, &s.t1[6 * i]);
            mat6_mul(t, Iw, Sj);
            crf_mul(r3, vi, t);
            for (int c = 0; c < 6; c++) out[c] = r1[c] + r2[c] + r3[c];
        }
    }
    wsync();
    PDDP_FOR(e, 2 * NB * 6) {                    // G5: suffix over links (net wrench derivatives, in place)
        const int jj = e / 6, c = e % 6;
        T run = 0;
        for (int i = NB - 1; i >= 0; i--) { run += g.dWb[6 * (i * 2 * NB + jj) + c]; g.dWb[6 * (i * 2 * NB + jj) + c] = run; }
    }
    PDDP_FOR(e, NB * NB * NB) {                  // dM/dq_k  [k][c][r]
        const int k = e / (NB * NB), c = (e % (NB * NB)) / NB, r = e % NB;
        const int jI = r <= c ? r : c, iI = r <= c ? c : r;
        g.dM[e] = dot6(&g.dS[6 * (jI * NB + k)], &s.Fj[6 * iI]) + dot6(&s.S[6 * jI], &g.tmpM[6 * (iI * NB + k)]);
    }
    wsync();
    PDDP_FOR(e, 2 * NB * NB) {                   // G6: dtau - (dM/dq_k) qdd
        const int jj = e / NB, i = e % NB;
        T val = dot6(&s.S[6 * i], &g.dWb[6 * (i * 2 * NB + jj)]);
        if (jj < NB) val += dot6(&g.dS[6 * (i * NB + jj)], &s.Wn[6 * i]);
        T rhs = -(val + T(jj - NB == i ? 0.5 : 0));
        if (jj < NB) {
            T mq = 0;
            for (int c = 0; c < NB; c++) mq += g.dM[NB * NB * jj + c * NB + i] * qdd[c];
            rhs -= mq;
        }
        g.dTau[e] = rhs;
    }
    wsync();
    PDDP_FOR(e, 3 * NB * NB) {                   // G7: dqdd = Minv [dtau_q - dM qdd | dtau_qd | I]
        const int jj = e / NB, r = e % NB;
        if (jj < 2 * NB) {
            T val = 0;
            for (int i = 0; i < NB; i++) val += Minv[r + NB * i] * g.dTau[jj * NB + i];
            dqdd[e] = val;
        } else dqdd[e] = Minv[(jj - 2 * NB) * NB + r];
    }
    wsync();
}

}  // namespace pddp
