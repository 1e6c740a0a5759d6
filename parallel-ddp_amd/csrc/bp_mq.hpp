// Backward (Riccati-like) pass of the 12-state / 4-control closed-form plants (the quadrotor, BASELINE configs[4]) on the MATRIX CORES (round 5): n + m = 16 is exactly one
// 16 x 16 tile, so every dense product of a knot is a short chain of v_mfma_f32_16x16x4_f32 (float) / v_mfma_f64_16x16x4_f64 (double handles: the parity instantiation),
// one wavefront per (problem, block of knots).  Same function as bp_block() (bp.hpp) / bp_cl_block() (bp_cl.hpp), which restate backPassKern and its inner routines
// (DDPHelpers/bpHelpers.cuh:18-420) for this size: linearXfrmOrLoad, backprop with the regulariser on the B columns of AB2 (so rho reaches Hxu and Huu, not the Hux block the
// gains are computed from), invHuu_dim4 -- the 4 x 4 ADJUGATE inverse with its det > 0 test (:132-188), which can fail: err[block] = 1 and the line-search kernel raises
// rho --, computeKTdu, computeCTG, computeFSVars, computeExpRed.  The tile algebra is bp_mfma.hpp's (mfma(X, Y) = X'Y on "RB tiles": lane (g, c) keeps rows (g, r = 0..3) of
// column c); what is specific to this size is the index order:
//     register r < 3 of lane group g  <->  state 3 g + r          register 3 of lane group g  <->  control g (control-row tiles) / the vector column's row (unused)
//     column c = (cg, cr) likewise: cr < 3 a state column, cr = 3 the control cg (control-column tiles B, W_u, Huu) or, for (3, 3), the VECTOR column of the state-column
//     tiles (p next to P, g_x next to Hxx, du next to K ...)
// so a sum over the 12 states is instructions r = 0, 1, 2 and a sum over the 4 controls is instruction r = 3 ALONE: 11 matrix instructions per knot (two state
// contractions x 3 -- P'[A | B] on ONE tile and [A B]'W, which holds all four blocks of the Hessian: Hxu and the vector column [A B]'p are read out of those two tiles
// through LDS transpositions (round 6, PDDP_MQ_LDS_T; until then two more contractions, 17 in all; the first cut issued one contraction per block, 23: same sums,
// same bits every time, tools/quad_mq_equal.py / quad_equal.py) --, K, K'Huu, B K, and two for the new cost-to-go; + 4 for the segment's sweep map, FUSE) where the
// lane-per-column kernel issues ~1250 vector instructions.  A knot's operands are requested one knot ahead (PDDP_MQ_PREFETCH).  Only the 4 x 4 inverse runs on the
// vector ALU (16 cofactors, one per lane, through 48 words of LDS -- the operations and their order are bp_cl_block's, i.e. the reference's).
// Float results agree with the oracle within the float32 bar (tests/test_fp32_bar.py), not bit for bit: sums over the state index run in the matrix core's order.
// The cost Hessian of the running knots is taken as diag(P::weight) where it is the plant's own (closed-form cost files); a Hessian overridden through the API and plug-in
// costs take the instantiation that reads H_k.  k_bp_cl stays selectable (pddp_config.kernels.cf_bp = cl): it is bit for bit the cooperative kernel.
#pragma once

#include "bp_mfma.hpp"
#include "plants.hpp"

namespace pddp {

template <typename T> __device__ __forceinline__ mx4t<T> mq_states(const mx4t<T>& X, const mx4t<T>& Y, mx4t<T> acc) {      // sum over the 12 states
    acc = Mx<T>::mfma(X[0], Y[0], acc); acc = Mx<T>::mfma(X[1], Y[1], acc); acc = Mx<T>::mfma(X[2], Y[2], acc);
    return acc;
}
template <typename T> __device__ __forceinline__ mx4t<T> mq_controls(const mx4t<T>& X, const mx4t<T>& Y, mx4t<T> acc) { return Mx<T>::mfma(X[3], Y[3], acc); }   // sum over the 4 controls

#ifndef PDDP_MQ_PREFETCH
#define PDDP_MQ_PREFETCH 1    // a knot's operands requested one knot ahead (measured: tools/quad_bp_ab.py, profiles/r05_quad_mfma.md)
#endif
#ifndef PDDP_MQ_FUSE_PREFETCH
#define PDDP_MQ_FUSE_PREFETCH 0   // ... for the instantiations that compose the sweep maps: without the eight prefetch registers they fit SIX waves per SIMD (78 registers, no scratch) --
                                  // 1.71-1.75 ms against 1.79-1.81 with the prefetch at five waves; [A B] alone ahead (2) spills at six: 2.06-2.20 (profiles/r06_quad.md)
#endif
#ifndef PDDP_MQ_STAGE
#define PDDP_MQ_STAGE 0       // 1: float handles send [P | p] and [A - B K | B du] of a knot through LDS in memory order as 16-byte pieces -- built, same bits, measured SLOWER (round 6,
                              // profiles/r06_quad.md: 2.11 -> 2.24 ms mean of four alternating handles each); the product keeps the dwordx3 stores in tile order
#endif
#ifndef PDDP_MQ_LDS_T
#define PDDP_MQ_LDS_T 1       // Hxu' and the vector column [A B]'p are read out of products the knot forms anyway -- row (vector column) of W = [P | p]'[A | B] and the control columns
                              // of [A B]'W -- through two LDS transpositions instead of two more three-instruction products (W'[A B], [A B]'V): 21 -> 15 matrix instructions per knot
                              // with the maps, the same sums in the same order, the same bits (tools/quad_equal.py); 0: the products (round 5's second cut)
#endif
constexpr int kMqStage = 160;                  // one staged pair: a 12 x 12 block (144) + its 12-vector right behind it, padded to whole 16-byte pieces
constexpr int kMqLds = 64 + 2 * kMqStage + 64; // elements per wave: Huu 16 | cofactors 16 | inverse 16 | boundary p 16 | staging [P | p] | staging [A - B K | B du] | [A B]'p 16 | Hxu 48

// Per-knot memory operations of the loop.  float: BUFFER instructions -- a wave-uniform resource per array, the knot's position a scalar byte offset, the lane's share a
// loop-invariant 32-bit vector offset (no 64-bit vector address arithmetic per access: as in bp_mfma.hpp, that arithmetic was a third of the first cut's vector
// instructions); three consecutive elements are one dwordx3.  double (parity instantiation): plain pointers.
typedef unsigned mq_u3 __attribute__((ext_vector_type(3)));
template <typename T> struct MqMem;
template <> struct MqMem<float> {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ explicit MqMem(const float* base) : r(mx_rsrc(base)) {}
    __device__ __forceinline__ void ld3(float* o, unsigned velem, unsigned selem) const {
        const mq_u3 w = __builtin_amdgcn_raw_buffer_load_b96(r, 4u * velem, 4u * selem, 0);
        o[0] = __uint_as_float(w[0]); o[1] = __uint_as_float(w[1]); o[2] = __uint_as_float(w[2]);
    }
    __device__ __forceinline__ float ld1(unsigned velem, unsigned selem) const { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 4u * velem, 4u * selem, 0)); }
    __device__ __forceinline__ void st3(float a, float b_, float c, unsigned velem, unsigned selem) const {
        mq_u3 w; w[0] = __float_as_uint(a); w[1] = __float_as_uint(b_); w[2] = __float_as_uint(c);
        __builtin_amdgcn_raw_buffer_store_b96(w, r, 4u * velem, 4u * selem, 0);
    }
    __device__ __forceinline__ void st1(float a, unsigned velem, unsigned selem) const { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), r, 4u * velem, 4u * selem, 0); }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ void st4(const float* lds16, unsigned velem, unsigned selem) const {     // 16 bytes from a 16-byte aligned LDS address
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u4*>(lds16), r, 4u * velem, 4u * selem, 0);
    }
};
template <> struct MqMem<double> {
    double* p;
    __device__ __forceinline__ explicit MqMem(const double* base) : p(const_cast<double*>(base)) {}
    __device__ __forceinline__ void ld3(double* o, unsigned velem, unsigned selem) const { const double* q = p + (size_t)selem + velem; o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; }
    __device__ __forceinline__ double ld1(unsigned velem, unsigned selem) const { return p[(size_t)selem + velem]; }
    __device__ __forceinline__ void st3(double a, double b_, double c, unsigned velem, unsigned selem) const { double* q = p + (size_t)selem + velem; q[0] = a; q[1] = b_; q[2] = c; }
    __device__ __forceinline__ void st1(double a, unsigned velem, unsigned selem) const { p[(size_t)selem + velem] = a; }
};

// One (problem, block of knots).  FS: M > 1 (write A - B K, B du for the forward sweep).  DIAGH: the running knots' cost Hessian is diag(P::weight) (the closed-form cost
// files; taken from the weights, H is not read) -- otherwise the four blocks of H_k are read (a Hessian overridden through pddp_set_array("H"), plug-in costs, and the
// executed-reference fixtures of tests/test_fixtures_direct.py, whose H is dense).  Returns through b.err[block]: 1 = Huu not invertible with a positive determinant.
// FUSE (with FS; round 6): the block composes its segment's forward-sweep map instead of writing A - B K | B du of every knot -- Psi <- Psi G_k with
// G_k = [A - B K, B du; 0, 1] (13 x 13: the tile's spare row / column (lane group 3, register 3 | the vector column) is the homogeneous coordinate), FOUR more matrix
// instructions per knot against 624 of the knot's ~2400 bytes of memory traffic and the whole per-knot linear sweep (k_sweep_cf: it read those bytes back);
// Psi' goes to b.segmap[problem][block] and k_sweep_maps_cf (kernels.hpp) finishes.  The last block's segment has no boundary behind it: it composes nothing.
// Same mathematics as bp_mfma.hpp's kMxFuseSweep (the arm's production path since round 3); A - B K | B du stay available through pddp_refresh_reference_views.
template <typename P, typename T, bool FS, bool DIAGH, bool FUSE = false>
__device__ void mq_bp_block(T* lds, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, int pb, int blk) {
    static_assert(!FUSE || FS, "segment maps exist with M > 1 only");
    static_assert(P::NX == 12 && P::NU == 4, "matrix-core backward pass of the 12-state / 4-control plants");
    using X = Mx<T>;
    using mx4 = mx4t<T>;
    constexpr int NX = 12, NU = 4, NM = 16, SZP = NX * NX, SZAB = NX * NM, SZH = NM * NM;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int cg = X::g_of(c), cr = X::r_of(c);
    const bool cx = cr < 3, cu = cr == 3, cv = (cg == 3 && cr == 3);       // state column / control column (control cg) / vector column
    const int sc = cx ? 3 * cg + cr : 0;                                     // this lane's state column
    const SolverState<T>& st = b.state[pb];
    if (st.done) return;
    const int N = dm.N, NBk = dm.NB;
    const T rho = st.rho;
    const size_t halfP = (size_t)(b.Pp - b.P), halfp = (size_t)(b.pp - b.p), knot0 = (size_t)pb * N;
    T* Pw = b.P + (st.pw ? halfP : 0) + knot0 * SZP; const T* Pr = b.P + (st.pw ? 0 : halfP) + knot0 * SZP;
    T* pw = b.p + (st.pw ? halfp : 0) + knot0 * NX;  const T* pr = b.p + (st.pw ? 0 : halfp) + knot0 * NX;
    const T* AB = b.AB + knot0 * SZAB; const T* H = b.H + knot0 * SZH; const T* gg = b.g + knot0 * NM;
    T* KT = b.KT + knot0 * (NX * NU); T* du = b.du + knot0 * NU; T* ApBK = b.ApBK + knot0 * SZP; T* Bdu = b.Bdu + knot0 * NX;
    const T* dcur = b.dcur + knot0 * NX;
    const T* xc = b.xb + ((size_t)pb * 2 + st.cur) * N * NX; const T* xp2 = b.xb + ((size_t)pb * 2 + st.cur2) * N * NX;
    T* ldsU = lds; T* ldsA = lds + 16; T* ldsI = lds + 32; T* ldsL = lds + 48;
    // the running knots' cost Hessian: diag(weight) -- this lane's state column's entry and its lane group's control's
    const T wx = cx ? P::weight(cw, sc, 0, N) : T(0), wu = P::weight(cw, NX + g, 0, N);
    const mx4 zero = {T(0), T(0), T(0), T(0)};

    int ks = NBk * (blk + 1) - 1, iterCount;
    mx4 Pa = zero;                                                           // [P | p]: P(3g + r, sc), p in the vector column
    const MqMem<T> mAB(AB), mH(H), mG(gg), mKT(KT), mDu(du), mF(ApBK), mBd(Bdu), mP(Pw), mp(pw);
    // loop-invariant element offsets of this lane inside a knot's blocks
    const unsigned oCol = (unsigned)((cx ? sc : NX + cg) * NX + 3 * g), oBt = (unsigned)((NX + g) * NX + sc), oG3 = (unsigned)(3 * g), oGu = (unsigned)(NX + g);
    const unsigned oKT = (unsigned)(sc + NX * g), oP = (unsigned)(sc * NX + 3 * g), oHcol = (unsigned)((cx ? sc : NX + cg) * NM), oHxu = (unsigned)((NX + g) * NM + sc);
    // A knot's 12 x 12 block + its 12-vector (P | p, A - B K | B du): three consecutive elements per lane in TILE order -- 48 lanes x 12 bytes.  Float handles send them
    // through LDS instead (the wave's staging area holds the pair in MEMORY order) and store 16-byte pieces in address order: 36 lanes for the block, 3 for the vector
    // (the step that was worth 18 % on the arm's [P | p], bp_mfma.hpp; LDS operations of one wave retire in order: no barrier, and the next knot's staging writes
    // cannot overtake these reads).
    constexpr bool STG = PDDP_MQ_STAGE && sizeof(T) == 4;
    auto store_pair = [&](T* stg, const MqMem<T>& mBlock, const MqMem<T>& mVec, int slot, const mx4& t) {
        if constexpr (STG) {
            if (cx) { stg[oP] = t[0]; stg[oP + 1] = t[1]; stg[oP + 2] = t[2]; }
            else if (cv) { stg[SZP + oG3] = t[0]; stg[SZP + oG3 + 1] = t[1]; stg[SZP + oG3 + 2] = t[2]; }
            wsync();
            if (lane < SZP / 4) mBlock.st4(reinterpret_cast<const float*>(stg) + 4 * lane, 4u * (unsigned)lane, (unsigned)slot * SZP);
            else if (lane < SZP / 4 + NX / 4) mVec.st4(reinterpret_cast<const float*>(stg) + 4 * lane, 4u * (unsigned)(lane - SZP / 4), (unsigned)slot * NX);
        } else {
            if (cx) mBlock.st3(t[0], t[1], t[2], oP, (unsigned)slot * SZP);
            else if (cv) mVec.st3(t[0], t[1], t[2], oG3, (unsigned)slot * NX);
        }
    };
    T* const stgP = lds + 64; T* const stgF = lds + 64 + kMqStage;
    auto store_ctg = [&](int slot, const mx4& t) { store_pair(stgP, mP, mp, slot, t); };                    // P, p of one knot
    if (ks == N - 1) {                                                       // last block: the final cost (bpHelpers.cuh:362-367)
        const T* Hf = H + (size_t)ks * SZH; const T* gf = gg + (size_t)ks * NM;
        const T* q = cx ? Hf + sc * NM + 3 * g : gf + 3 * g;
        if (cx || cv) { Pa[0] = q[0]; Pa[1] = q[1]; Pa[2] = q[2]; }
        store_ctg(ks - 1, Pa);
        ks--; iterCount = NBk - 2;
    } else {                                                                 // boundary cost-to-go of the previous iteration + linear transform (:18-34), + P d (:73)
        iterCount = NBk - 1;
        const T* bP = Pr + (size_t)ks * SZP;
        if (cx) { const T* q = bP + sc * NX + 3 * g; Pa[0] = q[0]; Pa[1] = q[1]; Pa[2] = q[2]; }
        if (lane < NX) {
            T dot = T(0), val = T(0);
            for (int j = 0; j < NX; j++) {
                const T pj = bP[lane + NX * j];
                dot += pj * (xc[NX * (ks + 1) + j] - xp2[NX * (ks + 1) + j]);
                val += dcur[(size_t)ks * NX + j] * pj;
            }
            ldsL[lane] = (dot + pr[(size_t)ks * NX + lane]) + val;
        }
        wsync();
        if (cv) { Pa[0] = ldsL[3 * g]; Pa[1] = ldsL[3 * g + 1]; Pa[2] = ldsL[3 * g + 2]; }
        wsync();
    }
    T dJ0 = T(0), dJ1 = T(0);                                                // lanes (g, vector column): control g's partial sums of the expected reduction
    const bool fuse = FUSE && blk < dm.M - 1;
    mx4 PsiT = zero;                                                         // Psi'(i, j) = Psi(j, i): starts as the identity of the 13 x 13 augmented map
    if (FUSE) {
#pragma unroll
        for (int r = 0; r < 3; r++) PsiT[r] = (cx && 3 * g + r == sc) ? T(1) : T(0);
        PsiT[3] = (cv && g == 3) ? T(1) : T(0);
    }
    // A knot's operands as they come from memory: this lane's three elements of its column of [A B], B(sc, g), g_x, g_u (and, read from H_k, its column's pieces).
    // PDDP_MQ_PREFETCH: requested ONE KNOT AHEAD -- the loads of knot k - 1 are in flight while knot k's products run (the compiler cannot hoist them itself past the
    // knot's stores); without it every knot begins with a round trip to memory that only the other resident waves hide.
    struct Ops { T v[3], bt, gx[3], gu, hc[3], hu, hxu; };
    constexpr int PF = FUSE ? PDDP_MQ_FUSE_PREFETCH : PDDP_MQ_PREFETCH;    // 1: everything one knot ahead; 2: [A B] one knot ahead, the gradient at the knot; 0: nothing ahead
    auto fetch_g = [&](int k, Ops& o) { const unsigned sG = (unsigned)k * NM; mG.ld3(o.gx, oG3, sG); o.gu = mG.ld1(oGu, sG); };
    auto fetch = [&](int k, Ops& o) {
        const unsigned sAB = (unsigned)k * SZAB;                             // (wave-uniform element offsets of the knot)
        mAB.ld3(o.v, oCol, sAB);
        o.bt = FS ? mAB.ld1(oBt, sAB) : T(0);                                // B(sc, g)   (state-column lanes)
        if (PF != 2) fetch_g(k, o);
        if (!DIAGH) {                                                        // Hcost(kx, ky) | Hcost(12 + b, kx) | Hcost(kx, 12 + b) | Hcost(12 + a, 12 + b): column-major H_k
            const unsigned sH = (unsigned)k * SZH;
            mH.ld3(o.hc, oHcol + oG3, sH);                                   // this lane's column of H_k
            o.hu = mH.ld1(oHcol + oGu, sH); o.hxu = mH.ld1(oHxu, sH);
        }
    };
    Ops nxt;
    if (PF && iterCount >= 0) fetch(ks, nxt);
    for (int iter = iterCount; iter >= 0; iter--, ks--) {
        // ---- operands: A(3g + r, sc) | B(3g + r, control cg) | B(sc, control g) | g_x, g_u in the vector column
        Ops o;
        if (PF) { o = nxt; if (PF == 2) fetch_g(ks, o); if (iter > 0) fetch(ks - 1, nxt); } else fetch(ks, o);
        // ONE tile holds [A | B]: A(3g + r, sc) in the state columns, B(3g + r, control cg) in the control columns (column 15 is control 3 here -- the vector
        // column of the state-column tiles is kept in a tile of its own, V, below)
        mx4 AB = {o.v[0], o.v[1], o.v[2], T(0)}, CXX = zero;
        const T bt = o.bt;
        const T gx0 = o.gx[0], gx1 = o.gx[1], gx2 = o.gx[2], gu = o.gu;
        mx4 CUX = {T(0), T(0), T(0), cv ? gu : T(0)}, CUU = zero, CXU = zero;
        if (DIAGH) {
            if (cx) { CXX[0] = (3 * g == sc) ? wx : T(0); CXX[1] = (3 * g + 1 == sc) ? wx : T(0); CXX[2] = (3 * g + 2 == sc) ? wx : T(0); }
            CUU[3] = (cu && g == cg) ? wu : T(0);
        } else {
            if (cx) { CXX[0] = o.hc[0]; CXX[1] = o.hc[1]; CXX[2] = o.hc[2]; CUX[3] = o.hu; CXU[3] = o.hxu; }
            else CUU[3] = o.hu;
        }
        if (cv) { CXX[0] = gx0; CXX[1] = gx1; CXX[2] = gx2; }
        // ---- W = P'[A | B]; the B columns take rho B (backprop :39-64)
        mx4 Wr = mq_states<T>(Pa, AB, zero);
        if (cu) { Wr[0] += rho * AB[0]; Wr[1] += rho * AB[1]; Wr[2] += rho * AB[2]; }
        const mx4 V = {cv ? Pa[0] : T(0), cv ? Pa[1] : T(0), cv ? Pa[2] : T(0), T(0)};      // p in the vector column
        // ---- H blocks (:66-93) as THREE products instead of one per block (round 5, second cut: 12 matrix instructions where the first cut issued 18) -- the sums, their
        // operands and their order are the per-block products' (a matrix instruction sums over the four lane groups' rows only: the other blocks' entries never meet):
        //   Hm  = [A B]' Wr:  rows = columns of [A B]; state rows: Hxx(kx, ky) in the state columns; control rows: Hux(b, kx) (state columns: no rho) | Huu(b, a) (with rho)
        //   HmT = Wr' [A B]:  control rows: Hxu(kx, b) as [b][kx] (with rho)
        //   Hv  = [A B]' V:   vector column: A'p (state rows) and B'p (control rows)
        mx4 Hm, Hv = zero; T hxu = T(0);
        if constexpr (PDDP_MQ_LDS_T) {
            // W's row of the vector column IS p'[A | B] (the rho term above touches the state rows only): Hv(j) = sum_s [A B](s, j) p(s), the products and their order those of
            // [A B]'V.  It sits in lane group 3's register 3, one column per lane; the vector-column lanes want it as rows: through LDS.  Likewise Hxu(kx, b) =
            // sum_s A(s, kx) W(s, control b) is entry (kx, control column b) of Hm = [A B]'W -- the sum W'[A B] forms for its entry (b, kx), multiplication commuted.
            T* ldsV = lds + 64 + 2 * kMqStage; T* ldsX = ldsV + 16;
            if (g == 3) ldsV[cx ? sc : NX + cg] = Wr[3];
            Hm = mq_states<T>(AB, Wr, zero);
            if (cu) { ldsX[cg * NX + 3 * g] = Hm[0]; ldsX[cg * NX + 3 * g + 1] = Hm[1]; ldsX[cg * NX + 3 * g + 2] = Hm[2]; }
            wsync();
            if (cv) { Hv[0] = ldsV[3 * g]; Hv[1] = ldsV[3 * g + 1]; Hv[2] = ldsV[3 * g + 2]; Hv[3] = ldsV[NX + g]; }
            if (cx) hxu = ldsX[g * NX + sc];
            wsync();
        } else {
            Hm = mq_states<T>(AB, Wr, zero);
            const mx4 HmT = mq_states<T>(Wr, AB, zero);
            Hv = mq_states<T>(AB, V, zero);
            hxu = HmT[3];
        }
        mx4 Hxx, Hux = CUX, HxuT = CXU, Huu = CUU;
#pragma unroll
        for (int r = 0; r < 3; r++) Hxx[r] = (cx ? Hm[r] : cv ? Hv[r] : T(0)) + CXX[r];          // Hxx(kx, ky) | g_x
        Hxx[3] = T(0);
        Hux[3] = (cx ? Hm[3] : cv ? Hv[3] : T(0)) + CUX[3];                                       // Hux(b, kx)  | g_u     (no rho)
        HxuT[3] = (cx ? hxu : T(0)) + CXU[3];                                                     // Hxu(kx, b) as [b][kx]  (with rho)
        Huu[3] = (cu ? Hm[3] : T(0)) + CUU[3];                                                    // Huu(a, b)              (with rho)
        // ---- Huu^-1: 4 x 4 adjugate with a det > 0 test (invHuu_dim4 :132-188; the operations of bp_cl_block): A2[row + 4 col]
        if (cu) ldsU[g + 4 * cg] = Huu[3];
        wsync();
        {
            const int e = lane & 15, ky = e / 4, kx = e % 4;
            const int r0 = (kx + 1) % 4, c0 = (ky + 1) % 4, r1 = (r0 + 1) % 4, c1 = (c0 + 1) % 4, r2 = (r1 + 1) % 4, c2 = (c1 + 1) % 4;
            const T* A2 = ldsU;
            const T f0 = A2[c0 * 4 + r0], f1 = A2[c0 * 4 + r1], f2 = A2[c0 * 4 + r2];
            const T f3 = A2[c1 * 4 + r0], f4 = A2[c1 * 4 + r1], f5 = A2[c1 * 4 + r2];
            const T f6 = A2[c2 * 4 + r0], f7 = A2[c2 * 4 + r1], f8 = A2[c2 * 4 + r2];
            const T cdet = f0 * f4 * f8 + f3 * f7 * f2 + f6 * f1 * f5 - f2 * f4 * f6 - f5 * f7 * f0 - f8 * f1 * f3;
            const T mine = ((kx + ky) % 2 ? T(-1) : T(1)) * cdet;
            ldsA[ky * 4 + kx] = mine;                                        // (the four lane groups write the same sixteen values)
            wsync();
            const T val = T(1) / (ldsA[0] * A2[0] + ldsA[1] * A2[1] + ldsA[2] * A2[2] + ldsA[3] * A2[3]);
            if (val <= T(0)) { if (lane == 0) b.err[(size_t)pb * dm.M + blk] = 1; return; }      // (uniform over the wave)
            ldsI[kx * 4 + ky] = val * mine;
            wsync();
        }
        mx4 InvT = zero;                                                     // [b][a] = Huu^-1(a, b): lane (g = b, control column a)
        if (cu) InvT[3] = ldsI[cg + 4 * g];
        wsync();
        // ---- gains (computeKTdu :208-220): K(a, kx) | du(a), row a = lane group
        const mx4 Kp = mq_controls<T>(InvT, Hux, zero);
        if (cx) mKT.st1(Kp[3], oKT, (unsigned)ks * (NX * NU));
        else if (cv) mDu.st1(Kp[3], (unsigned)g, (unsigned)ks * NU);
        // T1(kx, b) = sum_a K(a, kx) Huu(a, b) - Hxu(kx, b) as [b][kx]; its vector column is Huu' du
        const mx4 T1t = mq_controls<T>(Huu, Kp, zero) - HxuT;
        dJ0 += Kp[3] * Hux[3]; dJ1 += Kp[3] * T1t[3];                        // (only the vector-column lanes' sums are used: computeExpRed :317-334)
        if (FS && (!FUSE || fuse)) {                                         // A - B K | B du  (computeFSVars :281-312)
            const mx4 BT = {T(0), T(0), T(0), cx ? bt : T(0)};               // [b][kx] = B(kx, b)
            const mx4 BK = mq_controls<T>(BT, Kp, zero);
            if constexpr (!FUSE) {
                const mx4 Gt = cv ? BK : AB - BK;                             // (stored from the state-column lanes and the vector column only)
                store_pair(stgF, mF, mBd, ks, Gt);
            } else {
                mx4 Gt = cx ? AB - BK : cv ? BK : zero;                       // the control columns are not part of the map
                Gt[3] = (cv && g == 3) ? T(1) : T(0);                         // G(12, 12) = 1: the homogeneous coordinate
                PsiT = mq_controls<T>(Gt, PsiT, mq_states<T>(Gt, PsiT, zero));
            }
        }
        if (iter != 0 || blk != 0) {                                         // new cost-to-go (computeCTG :225-276): P(kx, ky) | p(kx); the one in front of knot 0 is never used (:396)
            mx4 val = mq_controls<T>(T1t, Kp, zero);
            val = mq_controls<T>(-Kp, Hux, val);
            mx4 Pn = Hxx + val;
            Pn[3] = T(0);                                                    // (the control / padding rows carry by-products: keep the tile clean)
            store_ctg(ks - 1, Pn);
            Pa = Pn;
        }
    }
    if (FUSE && fuse) {                                                      // Psi' of this segment: o[c * 16 + l] = Psi(l, c), c = 12 the affine part
        T* o = b.segmap + ((size_t)pb * dm.M + blk) * 256;
        if (cx || cv) {
            const int j = cx ? sc : NX;
#pragma unroll
            for (int r = 0; r < 3; r++) o[(3 * g + r) * 16 + j] = PsiT[r];
            if (g == 3) o[NX * 16 + j] = PsiT[3];
        }
    }
    // dJexp[2 blk], [2 blk + 1]: the four per-control partial sums in order (the vector column's lanes 15, 31, 47, 63 for float; the same lanes by g for double)
    {
        const int lv = X::q_of(3, 3);                                        // the vector column's lane inside a lane group
        T a0 = X::readlane(dJ0, lv), a1 = X::readlane(dJ1, lv);
        a0 += X::readlane(dJ0, 16 + lv); a1 += X::readlane(dJ1, 16 + lv);
        a0 += X::readlane(dJ0, 32 + lv); a1 += X::readlane(dJ1, 32 + lv);
        a0 += X::readlane(dJ0, 48 + lv); a1 += X::readlane(dJ1, 48 + lv);
        if (lane == 0) {
            T* dJexp = b.dJexp + (size_t)pb * 2 * dm.M;
            dJexp[2 * blk] = a0; dJexp[2 * blk + 1] = a1;
            b.err[(size_t)pb * dm.M + blk] = 0;
        }
    }
}

}  // namespace pddp
