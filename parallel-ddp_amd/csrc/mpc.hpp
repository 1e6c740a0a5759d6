// MPC warm start: what loadVarsGPU_MPC and storeVarsGPU_MPC do around the iLQR loop (DDPHelpers/MPCHelpers.cuh:602-655, 755-774;
// joint-space cost).  One wavefront per problem.
//
//   load   shift the previous solution by `shift` knots -- x, d, P, p, Pp, pp hold their last knot, u and KT are zero-filled
//          (shiftAndCopy :427-465, FLAG) and their knots N-2, N-1 are not touched, exactly like the reference --, or clear
//          u, KT, P, p, Pp, pp; zero du, dmax, err and AB at knot N-2; then roll the trajectory out open loop from the MEASURED state
//          with the shifted controls (rolloutMPC :525-560; FULL_ROLLOUT = the whole horizon, otherwise the first shooting segment,
//          plus rolloutMPC2 :570-598 for the last `shift` knots with feedback around the shifted previous trajectory).
//          The shifted previous solution is kept (x_old, u_old, KT_old) as the fall-back.
//   store  a solve counts as successful when an accepted iteration used a step-size index > 0 (sic, :986-991); otherwise the
//          current trajectory, controls and gains fall back to the shifted previous solution.
// Data-movement difference: the reference shifts the winner's candidate slot and copies into d_xp; here the current trajectory is
// one half of xb, and the load leaves the rolled-out trajectory in half 0 and the shifted previous one in half 1.
#pragma once

#include "fp.hpp"
#include "fp_pipe.hpp"
#include "plant_arm_lg.hpp"
#include "solver_state.hpp"

namespace pddp {

template <typename T>
struct MpcBuffers {      // extra arrays of a handle that has been used for MPC: [B][N][.]
    T *x_old, *u_old, *KT_old;
};

// dst[k] = src[min(k + shift, DIM_N - 1)] (or 0 beyond the data when zero_fill) for k < DIM_N - 1.  dst2 (optional) receives the same values.  When
// copy_last, knot DIM_N - 1 of src is also copied to dst/dst2 (needed when dst is not src).  The array is walked as ONE flat run in ascending order in
// pieces of 8 elements per lane: a piece is read completely before it is written and only ever reads at or above what it writes, so shifting in place is
// safe -- and the loads of a piece are in flight together (one element per lane and knot at a time, as a first version did, spends one memory latency per
// knot: 0.3 ms of a 1.2 ms control cycle for the two cost-to-go arrays).
template <typename T>
PDDP_HD void mpc_shift(const Wave& w, T* dst, T* dst2, const T* src, int per_knot, int DIM_N, int shift, bool zero_fill, bool copy_last) {
    constexpr int U = 8;
    const int total = (DIM_N - 1) * per_knot, last = (DIM_N - 1) * per_knot, off = shift * per_knot;
    for (int base = 0; base < total; base += w.nlanes * U) {
        T v[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int e = base + j * w.nlanes + w.lane;
            if (e < total) {
                const int es = e + off;
                v[j] = es < last ? src[es] : (zero_fill ? T(0) : src[last + (e % per_knot)]);
            }
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int e = base + j * w.nlanes + w.lane;
            if (e < total) { dst[e] = v[j]; if (dst2) dst2[e] = v[j]; }
        }
    }
    if (copy_last) PDDP_FOR(i, per_knot) { const T val = src[last + i]; dst[last + i] = val; if (dst2) dst2[last + i] = val; }
}
// workgroup-wide stage boundary of the load kernel (its waves share the shifting; one "wave" on the host)
PDDP_HD void mpc_block_sync(int nwaves) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (nwaves > 1) __syncthreads(); else wsync();
#else
    (void)nwaves;
#endif
}

template <typename P, typename T>
struct MpcScratch {
    typename P::Scratch plant;
    IntegScratch<P, T> integ;
    T x[P::NX], xn[P::NX], u[P::NU], dx[P::NX];
};

template <typename P, int INTEG, typename T, int V = -1>
PDDP_HD void mpc_load_body(const Wave& w, MpcScratch<P, T>& s, const Buffers<T>& b, const MpcBuffers<T>& mb, const Dims& dm, T dt, int pb,
                           const T* xActual, int shift, int clear_vars, int full_rollout, int wave_id = 0, int nwaves = 1, float* pipe = nullptr) {
    // wave_id / nwaves: the workgroup's waves share the shifting and the fall-back copies (task t runs on wave t % nwaves); the rollout is wave 0's
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU;
    const int N = dm.N;
    const int cur = b.state[pb].cur;
    T* x0 = b.xb + ((size_t)pb * 2 + 0) * N * NX;          // after the load: the rolled-out trajectory
    T* x1 = b.xb + ((size_t)pb * 2 + 1) * N * NX;          // after the load: the shifted previous trajectory (the reference's d_xp)
    T* xsrc = cur == 0 ? x0 : x1;
    T* u = b.ucur + (size_t)pb * N * NU; T* d = b.dcur + (size_t)pb * N * NX;
    T* KT = b.KT + (size_t)pb * N * NX * NU;
    T* Pm = b.P + (size_t)pb * N * NX * NX; T* Pp = b.Pp + (size_t)pb * N * NX * NX; T* pv = b.p + (size_t)pb * N * NX; T* pp = b.pp + (size_t)pb * N * NX;
    T* x_old = mb.x_old + (size_t)pb * N * NX; T* u_old = mb.u_old + (size_t)pb * N * NU; T* KT_old = mb.KT_old + (size_t)pb * N * NX * NU;
    bool piped = false;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (P::PLANT == 4 && INTEG == 1 && V >= 0 && sizeof(T) == 4) if (pipe && nwaves >= 8) {
        // The arm in float with a built-in robot model, eight waves: the serial open-loop rollout (63 steps: the longest single item of a control cycle) runs as the pipeline
        // of fp_pipe.hpp from the FIRST cycle of the kernel -- it needs the measured state and the shifted controls, nothing else -- while other waves shift the cost-to-go,
        // the gains and the defects and save the fall-back copies (before: all of that first, ~80 us, then the rollout):
        //   wave 0     x shift, u shift, fall-back copies of x and u, then the chain (Newton-Euler bias, solve, Euler step; stores the last two states)
        //   waves 1, 2 factors of the mass matrix of alternate steps, one step ahead; store the states they pick up (lane 0; every lane carries the same rollout)
        //   wave 3 P, p    wave 4 Pp, pp    wave 5 KT and its fall-back copy (posts counter 7: the closed-loop tail without FULL_ROLLOUT reads KT)    wave 6 d, du, flags
        piped = true;
        const TlPipeLds pl = tl_pipe_lds(pipe, false);
        const int n_roll = full_rollout ? N : dm.NB;
        if (wave_id == 1 || wave_id == 2) {
            T xx[NX];
#pragma unroll
            for (int i = 0; i < NX; i++) xx[i] = xActual[i];
            tl_pipe_factor_wave<V>(pl, wave_id - 1, n_roll - 1, xx, dt, w.lane, x0);
            return;
        }
        if (wave_id == 3) {
            if (clear_vars) { PDDP_FOR(e, N * NX * NX) Pm[e] = 0; PDDP_FOR(e, N * NX) pv[e] = 0; }
            else if (shift > 0) { mpc_shift<T>(w, Pm, nullptr, Pm, NX * NX, N, shift, false, false); mpc_shift<T>(w, pv, nullptr, pv, NX, N, shift, false, false); }
            return;
        }
        if (wave_id == 4) {
            if (clear_vars) { PDDP_FOR(e, N * NX * NX) Pp[e] = 0; PDDP_FOR(e, N * NX) pp[e] = 0; }
            else if (shift > 0) { mpc_shift<T>(w, Pp, nullptr, Pp, NX * NX, N, shift, false, false); mpc_shift<T>(w, pp, nullptr, pp, NX, N, shift, false, false); }
            return;
        }
        if (wave_id == 5) {
            if (clear_vars) { PDDP_FOR(e, N * NX * NU) KT[e] = 0; }
            else if (shift > 0) mpc_shift<T>(w, KT, nullptr, KT, NX * NU, N - 1, shift, true, false);
            mpc_shift<T>(w, KT_old, nullptr, KT, NX * NU, N + 1, 0, false, false);            // plain copy of all N knots, 8 loads in flight per lane
            tl_pipe_post(pl.flag + 7, 1);
            return;
        }
        if (wave_id == 6) {
            if (shift > 0) mpc_shift<T>(w, d, nullptr, d, NX, N, shift, false, false);
            PDDP_FOR(e, N * NU) b.du[(size_t)pb * N * NU + e] = 0;
            PDDP_FOR(e, dm.A) b.dmax[(size_t)pb * dm.A + e] = 0;
            PDDP_FOR(e, dm.M) b.err[(size_t)pb * dm.M + e] = 0;
            PDDP_FOR(e, NX * NM) b.AB[((size_t)pb * N + N - 2) * NX * NM + e] = 0;
            return;
        }
        if (wave_id != 0) return;
        mpc_shift<T>(w, cur == 0 ? x0 : x1, cur == 0 ? x1 : x0, xsrc, NX, N, shift, false, true);
        if (clear_vars) { PDDP_FOR(e, N * NU) u[e] = 0; }
        else if (shift > 0) mpc_shift<T>(w, u, nullptr, u, NU, N - 1, shift, true, false);
        wsync();
        PDDP_FOR(e, N * NU) u_old[e] = u[e];
        PDDP_FOR(e, N * NX) x_old[e] = x1[e];
        wsync();
        const T grav = reinterpret_cast<const ArmModel<T>*>(b.model)->grav;
        T xx[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) xx[i] = xActual[i];
        if (w.lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; i++) x0[i] = xx[i];
        }
        T un[NU];
#pragma unroll
        for (int i = 0; i < NU; i++) un[i] = u[i];
        for (int k = 0; k < n_roll - 1; k++) {
            T uk[NU];
#pragma unroll
            for (int i = 0; i < NU; i++) uk[i] = un[i];
            if (k + 1 < n_roll - 1) {
#pragma unroll
                for (int i = 0; i < NU; i++) un[i] = u[NU * (k + 1) + i];
            }
            tl_pipe_chain_step<V, false>(pl, k, xx, uk, dt, grav, w.lane);
            if (w.lane == 0 && k + 1 >= n_roll - 2) {                 // the last two states (no factor wave picks them up)
#pragma unroll
                for (int i = 0; i < NX; i++) x0[NX * (k + 1) + i] = xx[i];
            }
        }
        __threadfence_block();
        if (!full_rollout) tl_pipe_wait(pl.flag + 7, 1);                  // the closed-loop tail reads the shifted gains
    }
#endif
    if (!piped) {
    // ---- shift (shift == 0 degenerates to plain copies of the current values, which is what the reference's buffers hold then)
    const auto mine = [&](int task) { return task % nwaves == wave_id; };
    if (mine(0)) {
        mpc_shift<T>(w, cur == 0 ? x0 : x1, cur == 0 ? x1 : x0, xsrc, NX, N, shift, false, true);
        if (shift > 0) mpc_shift<T>(w, d, nullptr, d, NX, N, shift, false, false);
        PDDP_FOR(e, N * NU) b.du[(size_t)pb * N * NU + e] = 0;
        PDDP_FOR(e, dm.A) b.dmax[(size_t)pb * dm.A + e] = 0;
        PDDP_FOR(e, dm.M) b.err[(size_t)pb * dm.M + e] = 0;
        PDDP_FOR(e, NX * NM) b.AB[((size_t)pb * N + N - 2) * NX * NM + e] = 0;
    }
    if (clear_vars) {
        if (mine(1)) { PDDP_FOR(e, N * NX * NX) Pm[e] = 0; PDDP_FOR(e, N * NX) pv[e] = 0; }
        if (mine(2)) { PDDP_FOR(e, N * NX * NX) Pp[e] = 0; PDDP_FOR(e, N * NX) pp[e] = 0; }
        if (mine(3)) { PDDP_FOR(e, N * NU) u[e] = 0; PDDP_FOR(e, N * NX * NU) KT[e] = 0; }
    } else if (shift > 0) {
        if (mine(1)) { mpc_shift<T>(w, Pm, nullptr, Pm, NX * NX, N, shift, false, false); mpc_shift<T>(w, pv, nullptr, pv, NX, N, shift, false, false); }
        if (mine(2)) { mpc_shift<T>(w, Pp, nullptr, Pp, NX * NX, N, shift, false, false); mpc_shift<T>(w, pp, nullptr, pp, NX, N, shift, false, false); }
        if (mine(3)) { mpc_shift<T>(w, u, nullptr, u, NU, N - 1, shift, true, false); mpc_shift<T>(w, KT, nullptr, KT, NX * NU, N - 1, shift, true, false); }
    }
    mpc_block_sync(nwaves);
    // the fall-back: shifted previous trajectory / controls / gains (u_old is the reference's d_up: knots N-2, N-1 keep whatever they held).  The controls
    // are saved by wave 0 before its rollout touches them (rolloutMPC2 rewrites the last `shift` knots); nobody writes x1 or KT from here on.
    if (mine(0)) PDDP_FOR(e, N * NU) u_old[e] = u[e];
    if (mine(1)) PDDP_FOR(e, N * NX) x_old[e] = x1[e];
    if (mine(2) || (nwaves > 3 && wave_id == 3)) {
        const int part = nwaves > 3 ? (wave_id == 3 ? 1 : 0) : -1, half = (N * NX * NU) / 2;     // two waves share the largest copy
        const int e0 = part == 1 ? half : 0, e1 = part == 0 ? half : N * NX * NU;
        for (int e = e0 + w.lane; e < e1; e += w.nlanes) KT_old[e] = KT[e];
    }
    }   // !piped
    if (wave_id != 0) return;
    wsync();
    // ---- open-loop rollout from the measured state (rolloutMPC)
    P::load_model(w, s.plant, reinterpret_cast<const typename P::Model*>(b.model));
    PDDP_FOR(i, NX) { const T v = xActual[i]; s.x[i] = v; x0[i] = v; }
    wsync();
    const int n_roll = full_rollout ? N : dm.NB;
    bool rolled = piped;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (P::PLANT == 4 && INTEG == 1) if (!rolled) {
        // the arm: this serial rollout is a third of an MPC control cycle; one lane group runs it with the register-resident dynamics of the forward
        // pass (plant_arm_lg.hpp: the same numbers as P::dynamics, bit for bit), about half the time per step of the wave-cooperative evaluation
        if (w.lane < 7) {                       // lane 7 and the rest of the wave stay out of the lane-group code (lanegroup.hpp)
            using L = LgDevice<T>;
            ArmLgConst<L> c; c.Itab = s.plant.I; c.Ftab = s.plant.F; c.grav = s.plant.grav;
            ArmLgState<L> st;
            const int l = w.lane;
            T q = xActual[l], qd = xActual[l + 7];
            for (int k = 0; k < n_roll - 1; k++) {
                const T qdd = arm_lg_dynamics<L, true>(c, st, q, qd, u[NU * k + l]);
                const T qn = q + dt * qd, qdn = qd + dt * qdd;       // Euler (utils/integrators.cuh:24-36)
                x0[NX * (k + 1) + l] = qn; x0[NX * (k + 1) + l + 7] = qdn;
                q = qn; qd = qdn;
            }
        }
        wsync();
        rolled = true;
    }
#endif
    if (!rolled) for (int k = 0; k < n_roll - 1; k++) {
        PDDP_FOR(i, NU) s.u[i] = u[NU * k + i];
        wsync();
        integrator_step<P, INTEG>(w, s.plant, s.integ, s.xn, s.x, s.u, dt);
        PDDP_FOR(i, NX) { const T v = s.xn[i]; x0[NX * (k + 1) + i] = v; s.x[i] = v; }
        wsync();
    }
    // ---- the last `shift` knots with feedback around the shifted previous trajectory (rolloutMPC2; only without FULL_ROLLOUT)
    if (!full_rollout && dm.M > 1 && shift > 0) {
        const int ks = N - 1 - shift;
        PDDP_FOR(i, NX) s.x[i] = x0[NX * ks + i];
        wsync();
        for (int k = 0; k < shift; k++) {
            const int kn = ks + k;
            PDDP_FOR(i, NX) s.dx[i] = s.x[i] - x1[NX * kn + i];
            wsync();
            PDDP_FOR(r, NU) {
                const T* KTk = KT + NX * NU * kn;
                T Kdx = 0;
                for (int c = 0; c < NX; c++) Kdx += KTk[c + r * NX] * s.dx[c];
                const T uv = u[NU * kn + r] - Kdx;
                s.u[r] = uv; u[NU * kn + r] = uv;
            }
            wsync();
            integrator_step<P, INTEG>(w, s.plant, s.integ, s.xn, s.x, s.u, dt);
            PDDP_FOR(i, NX) { const T v = s.xn[i]; x0[NX * (kn + 1) + i] = v; s.x[i] = v; }
            wsync();
        }
    }
}

// after the loop: fall back to the shifted previous solution when the solve did not take a step
template <typename P, typename T>
PDDP_HD void mpc_store_body(const Wave& w, const Buffers<T>& b, const MpcBuffers<T>& mb, const Dims& dm, int pb) {
    constexpr int NX = P::NX, NU = P::NU;
    const int N = dm.N;
    const SolverState<T>& st = b.state[pb];
    if (st.took_step) return;
    T* xc = b.xb + ((size_t)pb * 2 + st.cur) * N * NX;
    PDDP_FOR(e, N * NX) xc[e] = mb.x_old[(size_t)pb * N * NX + e];
    PDDP_FOR(e, N * NU) b.ucur[(size_t)pb * N * NU + e] = mb.u_old[(size_t)pb * N * NU + e];
    PDDP_FOR(e, N * NX * NU) b.KT[(size_t)pb * N * NX * NU + e] = mb.KT_old[(size_t)pb * N * NX * NU + e];
}

}  // namespace pddp
