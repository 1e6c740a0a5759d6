// MPCHelpers.hpp -- the reference's receding-horizon (MPC) wrapper, same struct and function names, over libpddp.so.
//
//   algTrace / trajVars / GPUVars / matDimms / costParams, loadCost       DDPHelpers/MPCHelpers.cuh:51-135
//   allocateTrajVars / freeTrajVars                                       :138-155
//   allocateMemory_GPU_MPC / freeMemory_GPU_MPC                           :156-272
//   runiLQR_MPC_GPU                                                       :862-1045 (with loadVarsGPU_MPC :602-655, storeVarsGPU_MPC :755-774)
//   getHardwareControls                                                   :819-858
//   loadTraj (trajVars and GPUVars overloads)                             utils/exampleUtils.cuh:49-58, 72-79
//
// so that the reference's lock-step loop (examples/WAFR_MPC_examples.cu:160-238, `testMPC_lockstep`) reads the same here.
// Included by hostapi/config.hpp when MPC_MODE is 1.  With EE_COST 0 the goal gv->xGoal is a full state, with EE_COST 1 its first six
// entries are the tool-point goal (x, y, z, roll, pitch, yaw) and gv->xTarget is the nominal-state target (zeros unless set).
//
// As in DDPWrappers.hpp, GPUVars' device pointers are the solver handle's own arrays (see there for which ones keep the
// reference's layout); the handle lives in GPUVars::d_P's registry entry.  Bookkeeping kept exactly as the reference has it:
//   * shift = floor((tActual_plant - tv->t0_plant) / TIME_STEP_LENGTH_IN_us)                                    (:875)
//   * variables are cleared when clear_vars or when more than SOLVES_TO_RESET solves in a row failed             (:610)
//   * a solve is "successful" when some accepted iteration used a step-size INDEX > 0 (sic, :986-991); it resets
//     tv->last_successful_solve, which storeVarsGPU_MPC then increments; only when it is 1 afterwards is the new
//     trajectory copied into trajVars (under tv->lock), otherwise the device falls back to the shifted previous solution
//   * tv->t0_sys / t0_plant always advance to this call's clock values                                           (:756)
//   * the time budget: with USE_MAX_SOLVER_TIME and a finite budget the exit flags and the clock are polled after EVERY sweep (the
//     overshoot is at most one iteration; the reference checks three times per iteration, :919,:941,:1001); without a budget the
//     sweeps are enqueued in groups of PDDP_POLL_EVERY
//   * algTrace: J and alpha receive entries 0..iter of every solve, tTime the wall time of the call, initTime 0; the per-phase
//     vectors stay empty unless the caller fills them from pddp_solve_ex (the sweeps of an MPC solve are graph replays)
//   * use_cost_shift only acts on the end-effector cost (plants/cost_arm.cuh:212): final weights from knot N-1-shift on.
#ifndef PDDP_HOSTAPI_MPCHELPERS_HPP
#define PDDP_HOSTAPI_MPCHELPERS_HPP

#include <cmath>
#include <cstdint>
#include <mutex>
#include <vector>

#define TIME_STEP_LENGTH_IN_ms (TIME_STEP * 1000.0)                                                     // MPCHelpers.cuh:29-33
#define TIME_STEP_LENGTH_IN_us (TIME_STEP_LENGTH_IN_ms * 1000.0)
#define get_time_us_i64(time) (static_cast<int64_t>(std::ceil(get_time_us(time))))
#define get_time_steps_us_d(start, end) (static_cast<double>((end) - (start)) / TIME_STEP_LENGTH_IN_us)
#define get_time_steps_us_f(start, end) (static_cast<int>(std::floor(get_time_steps_us_d(start, end))))
#ifndef SOLVES_TO_RESET
#define SOLVES_TO_RESET 10                                                                              // :34-36
#endif
#ifndef FULL_ROLLOUT
#define FULL_ROLLOUT 1                                                                                  // :37-39
#endif
#ifndef TRAJ_RUNNER_TIME_STEPS
#define TRAJ_RUNNER_TIME_STEPS NUM_TIME_STEPS                                                           // :43-45
#endif
#ifndef USE_FEEDBACK_IN_TRAJ_RUNNER
#define USE_FEEDBACK_IN_TRAJ_RUNNER 1                                                                   // :46-48
#endif
#ifndef PD_GAINS_ON_STATE
#define PD_GAINS_ON_STATE 0                                                                             // :40-42
#endif
#ifndef MAX_SOLVER_TIME
#define MAX_SOLVER_TIME 10000.0                                                                         // config.cuh:84
#endif
#ifndef USE_MAX_SOLVER_TIME
#define USE_MAX_SOLVER_TIME 1                                                                           // config.cuh:190-192
#endif
#ifndef USE_ALG_TRACE
#define USE_ALG_TRACE 1                                                                                 // config.cuh:65-67
#endif

template <typename T>
struct algTrace {
    std::vector<T> J; std::vector<int> alpha;
    std::vector<double> tTime, simTime, sweepTime, initTime, bpTime, nisTime;
};

template <typename T>
struct trajVars {                       // what the trajectory runner reads (under `lock`)
    T *x, *u, *KT;
    int ld_x, ld_u, ld_KT;
    int64_t t0_plant, t0_sys;
    std::mutex* lock;
    bool first_pass;
    int last_successful_solve;
};

template <typename T>
struct GPUVars {
    T **d_x, **h_d_x, *d_xp, *d_xp2, *d_x_old;
    T **d_u, **h_d_u, *d_up, *d_u_old;
    T *d_P, *d_p, *d_Pp, *d_pp;
    T *d_AB, *d_H, *d_g, *d_KT, *d_KT_old, *d_du;
    T **d_d, **h_d_d, *d_dT, *d_dp, *d_dM, *d;
    T *d_ApBK, *d_Bdu;
    T *d_JT, *J, *dJexp, *d_dJexp;
    T *alpha, *d_alpha; int *alphaIndex, *err, *d_err;
    T *d_I, *d_Tbody;
    T *xGoal, *d_xGoal, *xActual, *d_xActual;
    pddpStream_t* streams;
    std::mutex* lock;
    T *xTarget, *d_xTarget;
};

struct matDimms { int ld_x, ld_u, ld_P, ld_p, ld_AB, ld_H, ld_g, ld_KT, ld_du, ld_d, ld_A; };

template <typename T>
struct costParams {
    T Q_EE1, Q_EE2, QF_EE1, QF_EE2, Q_EEV1, Q_EEV2, QF_EEV1, QF_EEV2, Q_xdEE, QF_xdEE, Q_xEE, QF_xEE, R_EE;
    T Q1, Q2, QF1, QF2, R;
};

template <typename T>
void loadCost(costParams<T>* c, T Q_EE1 = _Q_EE1, T Q_EE2 = _Q_EE2, T QF_EE1 = _QF_EE1, T QF_EE2 = _QF_EE2, T Q_EEV1 = _Q_EEV1, T Q_EEV2 = _Q_EEV2,
              T QF_EEV1 = _QF_EEV1, T QF_EEV2 = _QF_EEV2, T R_EE = _R_EE, T Q_xdEE = _Q_xdEE, T QF_xdEE = _QF_xdEE, T Q_xEE = _Q_xEE,
              T QF_xEE = _QF_xEE, T Q1 = _Q1, T Q2 = _Q2, T R = _R, T QF1 = _QF1, T QF2 = _QF2) {
    const costParams<T> v = {Q_EE1, Q_EE2, QF_EE1, QF_EE2, Q_EEV1, Q_EEV2, QF_EEV1, QF_EEV2, Q_xdEE, QF_xdEE, Q_xEE, QF_xEE, R_EE, Q1, Q2, QF1, QF2, R};
    *c = v;
}
template <typename T> void loadCost(costParams<T>* dst, costParams<T>* src) { *dst = *src; }

template <typename T>
void allocateTrajVars(trajVars<T>* tv, matDimms* md) {
    tv->x = static_cast<T*>(std::calloc((size_t)md->ld_x * NUM_TIME_STEPS, sizeof(T)));
    tv->u = static_cast<T*>(std::calloc((size_t)md->ld_u * NUM_TIME_STEPS, sizeof(T)));
    tv->KT = static_cast<T*>(std::calloc((size_t)md->ld_KT * DIM_KT_c * NUM_TIME_STEPS, sizeof(T)));
    tv->ld_x = md->ld_x; tv->ld_u = md->ld_u; tv->ld_KT = md->ld_KT;
    tv->t0_plant = 0; tv->t0_sys = 0; tv->first_pass = true; tv->last_successful_solve = 0;
    tv->lock = new std::mutex;
}
template <typename T>
void freeTrajVars(trajVars<T>* tv) { std::free(tv->x); std::free(tv->u); std::free(tv->KT); delete tv->lock; }

template <typename T>
void allocateMemory_GPU_MPC(GPUVars<T>* gv, matDimms* md, trajVars<T>* tv) {
    using namespace pddp_hostapi;
    allocateMemory_GPU<T>(&gv->d_x, &gv->h_d_x, &gv->d_xp, &gv->d_xp2, &gv->d_u, &gv->h_d_u, &gv->d_up, &gv->d_xGoal, &gv->xGoal, &gv->d_P, &gv->d_Pp,
                          &gv->d_p, &gv->d_pp, &gv->d_AB, &gv->d_H, &gv->d_g, &gv->d_KT, &gv->d_du, &gv->d_d, &gv->h_d_d, &gv->d_dp, &gv->d_dT, &gv->d_dM,
                          &gv->d, &gv->d_ApBK, &gv->d_Bdu, &gv->d_JT, &gv->J, &gv->d_dJexp, &gv->dJexp, &gv->alpha, &gv->d_alpha, &gv->alphaIndex,
                          &gv->d_err, &gv->err, &md->ld_x, &md->ld_u, &md->ld_P, &md->ld_p, &md->ld_AB, &md->ld_H, &md->ld_g, &md->ld_KT, &md->ld_du,
                          &md->ld_d, &md->ld_A, &gv->streams, &gv->d_I, &gv->d_Tbody);
    pddp_handle h = find(gv->d_P)->h;
    gv->d_x_old = dev<T>(h, "x_old"); gv->d_u_old = dev<T>(h, "u_old"); gv->d_KT_old = dev<T>(h, "KT_old");
    gv->d_xActual = dev<T>(h, "xActual"); gv->xActual = static_cast<T*>(std::calloc(STATE_SIZE, sizeof(T)));
    gv->xTarget = static_cast<T*>(std::calloc(md->ld_x, sizeof(T))); gv->d_xTarget = nullptr;    // nominal-state target: end-effector cost only
    gv->lock = new std::mutex;
    allocateTrajVars<T>(tv, md);
}

template <typename T>
void freeMemory_GPU_MPC(GPUVars<T>* gv) {
    std::free(gv->xActual); std::free(gv->xTarget); delete gv->lock;
    freeMemory_GPU<T>(gv->d_x, gv->h_d_x, gv->d_xp, gv->d_xp2, gv->d_u, gv->h_d_u, gv->d_up, gv->xGoal, gv->d_xGoal, gv->d_P, gv->d_Pp, gv->d_p, gv->d_pp,
                      gv->d_AB, gv->d_H, gv->d_g, gv->d_KT, gv->d_du, gv->d_d, gv->h_d_d, gv->d_dp, gv->d_dM, gv->d_dT, gv->d, gv->d_ApBK, gv->d_Bdu,
                      gv->d_JT, gv->J, gv->d_dJexp, gv->dJexp, gv->alpha, gv->d_alpha, gv->alphaIndex, gv->d_err, gv->err, gv->streams, gv->d_I,
                      gv->d_Tbody);
}

// utils/exampleUtils.cuh:49-58: a constant initial trajectory (state xInit or 0, control uInit or 0.01), zero gains
template <typename T>
void loadTraj(trajVars<T>* tv, matDimms* md, T* xInit = nullptr, T* uInit = nullptr) {
    for (int k = 0; k < NUM_TIME_STEPS; k++) {
        for (int i = 0; i < STATE_SIZE; i++) tv->x[(size_t)k * md->ld_x + i] = xInit ? xInit[i] : T(0);
        for (int i = 0; i < CONTROL_SIZE; i++) tv->u[(size_t)k * md->ld_u + i] = uInit ? uInit[i] : T(0.01);
    }
    std::memset(tv->KT, 0, (size_t)md->ld_KT * DIM_KT_c * NUM_TIME_STEPS * sizeof(T));
}
// utils/exampleUtils.cuh:72-79: ... and the same trajectory as the device's current one, measured state = its first knot
template <typename T>
void loadTraj(GPUVars<T>* gv, trajVars<T>* tv, matDimms* md, T* xInit = nullptr, T* uInit = nullptr) {
    using namespace pddp_hostapi;
    loadTraj<T>(tv, md, xInit, uInit);
    std::memcpy(gv->xActual, tv->x, STATE_SIZE * sizeof(T));
    pddp_handle h = find(gv->d_P)->h;
    pddp_state st;
    check(pddp_get_state(h, &st), "pddp_get_state");
    std::vector<T> xb(2 * (size_t)NUM_TIME_STEPS * STATE_SIZE);
    std::memcpy(xb.data(), tv->x, xb.size() / 2 * sizeof(T));
    std::memcpy(xb.data() + xb.size() / 2, tv->x, xb.size() / 2 * sizeof(T));
    check(pddp_set_array(h, "xb", xb.data(), xb.size() * sizeof(T)), "xb");
    check(pddp_set_array(h, "ucur", tv->u, (size_t)NUM_TIME_STEPS * CONTROL_SIZE * sizeof(T)), "ucur");
    check(pddp_set_array(h, "KT", tv->KT, (size_t)NUM_TIME_STEPS * STATE_SIZE * CONTROL_SIZE * sizeof(T)), "KT");
}

template <typename T>
void runiLQR_MPC_GPU(trajVars<T>* tv, GPUVars<T>* gv, matDimms* md, algTrace<T>* data, costParams<T>* cst, int64_t tActual_sys, int64_t tActual_plant,
                     int ignoreFirstDefectFlag, int max_iter = MAX_ITER, double time_budget = MAX_SOLVER_TIME, int clear_vars = 0,
                     bool use_cost_shift = 0) {
    using namespace pddp_hostapi;
    (void)md; (void)use_cost_shift;
    struct timeval start, end;
    gettimeofday(&start, NULL);
    Context* ctx = find(gv->d_P);
    pddp_handle h = ctx->h;
    apply_cost<T>(ctx, cst->Q1, cst->Q2, cst->R, cst->QF1, cst->QF2, cst->Q_EE1, cst->Q_EE2, cst->QF_EE1, cst->QF_EE2, cst->R_EE, cst->Q_xEE, cst->QF_xEE,
                  cst->Q_xdEE, cst->QF_xdEE);
#if EE_COST
    check(pddp_set_ee_cost_shift(h, use_cost_shift ? 1 : 0), "pddp_set_ee_cost_shift");                       // finalCostShift, :876
    check(pddp_set_array(h, "xTarget", gv->xTarget, STATE_SIZE * sizeof(T)), "xTarget");                       // d_xTarget, :612
#endif
    int shift = get_time_steps_us_f(tv->t0_plant, tActual_plant);
    if (shift < 0) shift = 0;
    if (shift > NUM_TIME_STEPS - 2) shift = NUM_TIME_STEPS - 2;
    const int clear = (tv->last_successful_solve > SOLVES_TO_RESET || clear_vars) ? 1 : 0;                // :610
    if (max_iter > MAX_ITER) max_iter = MAX_ITER;
    std::vector<T> x((size_t)NUM_TIME_STEPS * STATE_SIZE), u((size_t)NUM_TIME_STEPS * CONTROL_SIZE), KT((size_t)NUM_TIME_STEPS * STATE_SIZE * CONTROL_SIZE);
    T* Jtmp = reinterpret_cast<T*>(ctx->Jtmp.data());
    int success = 0, iter = 0;
    check(pddp_mpc_solve(h, gv->xActual, gv->xGoal, &shift, clear, FULL_ROLLOUT, ignoreFirstDefectFlag, max_iter, USE_MAX_SOLVER_TIME ? time_budget : 0.0,
                         (USE_MAX_SOLVER_TIME && time_budget > 0) ? 1 : PDDP_POLL_EVERY, x.data(), u.data(), KT.data(), Jtmp, ctx->atmp.data(), &success, &iter), "runiLQR_MPC_GPU");
    pddp_state st;
    check(pddp_get_state(h, &st), "pddp_get_state");
    *gv->alphaIndex = st.alphaIndex;
    // storeVarsGPU_MPC (:755-774)
    if (success) tv->last_successful_solve = 0;                                                           // :986-991
    tv->lock->lock();
    tv->t0_sys = tActual_sys; tv->t0_plant = tActual_plant;
    tv->last_successful_solve++;
    if (tv->last_successful_solve == 1) {
        std::memcpy(tv->x, x.data(), x.size() * sizeof(T)); std::memcpy(tv->u, u.data(), u.size() * sizeof(T));
        std::memcpy(tv->KT, KT.data(), KT.size() * sizeof(T));
    }
    tv->lock->unlock();
#if USE_ALG_TRACE
    if (data) {
        for (int i = 0; i <= iter && i <= MAX_ITER; i++) { data->alpha.push_back(ctx->atmp[i]); data->J.push_back(Jtmp[i]); }
        gettimeofday(&end, NULL);
        data->initTime.push_back(0.0);
        data->tTime.push_back(time_delta_ms(start, end));
    }
#else
    (void)data; (void)end;
#endif
}

// What the trajectory runner sends to the robot at time tActual (MPCHelpers.cuh:819-858): zero-order hold on u and K, first-order
// hold on the nominal state; returns 1 when tActual is outside the trajectory.  Host-only, serial.
template <typename T>
int getHardwareControls(double* q_out, double* u_out, T* x, T* u, T* KT, double t0, const double* qActual, const double* qdActual, double tActual,
                        int ld_x, int ld_u, int ld_KT, double* q_prev = nullptr, double* u_prev = nullptr, double alpha = 0) {
    const double steps = get_time_steps_us_d(t0, tActual);
    const int k = static_cast<int>(steps);
    const double frac = steps - static_cast<double>(k);
    if (k >= TRAJ_RUNNER_TIME_STEPS - 2 || k < 0) return 1;
    const T* uk = u + (size_t)k * ld_u;
    if (USE_FEEDBACK_IN_TRAJ_RUNNER) {
        const T* KTk = KT + (size_t)k * ld_KT * DIM_KT_c;
        const T *x_lo = x + (size_t)k * ld_x, *x_hi = x + (size_t)(k + 1) * ld_x;
        T dx[STATE_SIZE];
        for (int i = 0; i < STATE_SIZE; i++) {
            const T nominal = static_cast<T>(1.0 - frac) * x_lo[i] + static_cast<T>(frac) * x_hi[i];
            dx[i] = static_cast<T>(i < NUM_POS ? qActual[i] : qdActual[i - NUM_POS]) - nominal;
            if (PD_GAINS_ON_STATE && i < NUM_POS) q_out[i] = static_cast<double>(nominal);
        }
        for (int r = 0; r < CONTROL_SIZE; r++) {
            T val = uk[r];
            for (int c = 0; c < STATE_SIZE; c++) val -= KTk[c + r * ld_KT] * dx[c];
            u_out[r] = static_cast<double>(val);
        }
    } else {
        for (int r = 0; r < CONTROL_SIZE; r++) u_out[r] = static_cast<double>(uk[r]);
    }
    if (!PD_GAINS_ON_STATE) for (int i = 0; i < NUM_POS; i++) q_out[i] = qActual[i];
    if (q_prev != nullptr && u_prev != nullptr && alpha > 0)
        for (int r = 0; r < CONTROL_SIZE; r++) { u_out[r] = (1 - alpha) * u_out[r] + alpha * u_prev[r]; u_prev[r] = u_out[r]; }
    return 0;
}

#endif
