// DDPWrappers.hpp -- the reference's solver entry points, same names and argument order, over libpddp.so.
//
//   allocateMemory_GPU<T>   DDPHelpers/nisInitHelpers.cuh:768-861
//   runiLQR_GPU<T>          DDPHelpers/DDPWrappers.cuh:10-138
//   freeMemory_GPU<T>       DDPHelpers/nisInitHelpers.cuh:865-882
//
// so that a caller written against the reference (examples/WAFR_iLQR_examples.cu:303-361, `testGPU`) recompiles after
// replacing `#include "config.cuh"` by `#include "hostapi/config.hpp"` and `cudaStream_t` by `pddpStream_t`.
//
// What the arguments mean here.  The reference makes the CALLER hold every device buffer and hands them back on each
// call.  This implementation keeps one solver handle (include/pddp.h) behind them: allocateMemory_GPU creates the handle
// and fills the caller's variables with the handle's own device arrays, runiLQR_GPU finds the handle back from `d_P`.
// Buffers with the reference's layout and meaning (readable/writable by the caller between solves, e.g. for warm starts
// or an MPC shift): d_P d_p d_Pp d_pp d_AB d_H d_g d_KT d_du d_ApBK d_Bdu d_JT d_dJexp d_err d_alpha d_xGoal d_up d_dp
// and the per-alpha tables d_x/h_d_x, d_u/h_d_u, d_d/h_d_d.  Differences, all consequences of copies the reference makes
// and this design does not:
//   * d_P / d_Pp (d_p / d_pp) are the two halves of the cost-to-go double buffer: "Pp <- P" is a flip of an index
//     (pddp_state.pw), so after a solve the last cost-to-go is in d_P or in d_Pp (pddp_get_state tells which);
//   * d_xp / d_xp2 are the two halves of one double buffer whose roles (current trajectory / trajectory the stored
//     boundary cost-to-go belongs to) alternate with every accepted iteration instead of being copied;
//   * the per-alpha slots are pure outputs of the forward pass and d_ApBK / d_Bdu are not written by production sweeps (the forward sweep's operands are
//     composed inside the backward pass): runiLQR_GPU rebuilds all of them before it returns (pddp_refresh_reference_views: the accepted trajectory in every
//     alpha slot like memcpyCurrAKern leaves it, A - B K / B du of the last backward pass), unless compiled with -DPDDP_REFERENCE_VIEWS=0;
//   * d_I / d_Tbody point to the library's robot constants (spatial inertias / fixed joint frames), not to the
//     reference's 36-float-per-link scratch layout; treat them as opaque;
//   * `streams` has NUM_STREAMS entries that all alias the solver's single stream (one stream is all it needs).
// Errors: like the reference's gpuErrchk (utils/cudaUtils.cu:31-37), a device/runtime error prints and exit()s; numerical
// failure stays in band (alphaOut[iter] = -1, loop exit on RHO_MAX unless IGNORE_MAX_ROX_EXIT).
#ifndef PDDP_HOSTAPI_DDPWRAPPERS_HPP
#define PDDP_HOSTAPI_DDPWRAPPERS_HPP

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#ifndef PDDP_PHASE_TIMERS
#define PDDP_PHASE_TIMERS 1   // fill bpTime/simTime/nisTime per iteration (HIP events; sweeps are then not replayed from a hipGraph)
#endif
#ifndef PDDP_POLL_EVERY
#define PDDP_POLL_EVERY 8     // sweeps enqueued between two polls of the device-side exit flags
#endif
#ifndef PDDP_DEVICE
#define PDDP_DEVICE 0
#endif
#ifndef PDDP_REFERENCE_VIEWS
#define PDDP_REFERENCE_VIEWS 1   // after every runiLQR_GPU: d_ApBK / d_Bdu rebuilt and the accepted trajectory copied into every alpha slot, like the reference leaves them
#endif                           // (one small launch per solve; 0 for callers that never read those buffers)
#ifndef PDDP_EE_INITIAL_COST_FIX
#define PDDP_EE_INITIAL_COST_FIX 0   // 1: pddp_config.ee_initial_cost_fix (include/pddp.h) -- NOT the reference's behaviour
#endif

namespace pddp_hostapi {

struct Context {
    pddp_handle h;
    double cost[14];              // the weights the handle currently holds: Q1 Q2 R QF1 QF2 | Q_EE1 Q_EE2 QF_EE1 QF_EE2 R_EE Q_xEE QF_xEE Q_xdEE QF_xdEE
    std::vector<double> phase;    // [5][MAX_ITER+2]: bp, sweep + rollouts, line search, setup, the linear sweep alone (pddp_solve_ex)
    std::vector<char> Jtmp;       // [MAX_ITER+2] elements of T
    std::vector<int> atmp;
};
// handle registry, keyed by the d_P the caller hands back on every call; guarded: allocate / run / free may come from different host threads (the reference's MPC loop
// solves on one thread while another publishes, MPCHelpers.cuh:62,756-773)
inline std::map<const void*, Context*>& registry() { static std::map<const void*, Context*> r; return r; }
inline std::mutex& registry_lock() { static std::mutex m; return m; }

inline void check(int rc, const char* what) {
    if (rc == 0) return;
    std::fprintf(stderr, "GPUassert: %s: %s (code %d)\n", what, pddp_last_error(), rc);
    std::exit(rc < 0 ? -rc : rc);
}
template <typename T> T* dev(pddp_handle h, const char* name) {
    void* p = nullptr; size_t nb = 0;
    check(pddp_array_ptr(h, name, &p, &nb), name);
    return static_cast<T*>(p);
}
// the reference takes the cost weights on every call (DDPWrappers.cuh:17-21): forward them when they changed
template <typename T>
void apply_cost(Context* ctx, T Q1, T Q2, T R, T QF1, T QF2, T Q_EE1, T Q_EE2, T QF_EE1, T QF_EE2, T R_EE, T Q_xEE, T QF_xEE, T Q_xdEE, T QF_xdEE) {
    const double v[14] = {(double)Q1, (double)Q2, (double)R, (double)QF1, (double)QF2, (double)Q_EE1, (double)Q_EE2, (double)QF_EE1, (double)QF_EE2,
                          (double)R_EE, (double)Q_xEE, (double)QF_xEE, (double)Q_xdEE, (double)QF_xdEE};
    bool joint = false, ee = false;
    for (int i = 0; i < 5; i++) joint |= (T)v[i] != (T)ctx->cost[i];
    for (int i = 5; i < 14; i++) ee |= (T)v[i] != (T)ctx->cost[i];
    if (joint) check(pddp_set_cost(ctx->h, v[0], v[1], v[2], v[3], v[4]), "pddp_set_cost");
    if (ee && EE_COST) check(pddp_set_cost_ee(ctx->h, v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13]), "pddp_set_cost_ee");
    for (int i = 0; i < 14; i++) ctx->cost[i] = v[i];
}
inline Context* find(const void* d_P) {
    std::lock_guard<std::mutex> guard(registry_lock());
    auto it = registry().find(d_P);
    if (it == registry().end()) { std::fprintf(stderr, "GPUassert: buffers were not obtained from allocateMemory_GPU\n"); std::exit(1); }
    return it->second;
}

}  // namespace pddp_hostapi

template <typename T>
void allocateMemory_GPU(T*** d_x, T*** h_d_x, T** d_xp, T** d_xp2, T*** d_u, T*** h_d_u, T** d_up, T** d_xGoal, T** xGoal, T** d_P, T** d_Pp,
                        T** d_p, T** d_pp, T** d_AB, T** d_H, T** d_g, T** d_KT, T** d_du, T*** d_d, T*** h_d_d, T** d_dp, T** d_dT, T** d_dM,
                        T** d, T** d_ApBK, T** d_Bdu, T** d_JT, T** J, T** d_dJexp, T** dJexp, T** alpha, T** d_alpha, int** alphaIndex,
                        int** d_err, int** err, int* ld_x, int* ld_u, int* ld_P, int* ld_p, int* ld_AB, int* ld_H, int* ld_g, int* ld_KT,
                        int* ld_du, int* ld_d, int* ld_A, pddpStream_t** streams, T** d_I = nullptr, T** d_Tbody = nullptr) {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "algType must be float or double");
    using namespace pddp_hostapi;
    pddp_config c;
    check(pddp_default_config(&c, PLANT), "pddp_default_config");
    c.dtype = std::is_same<T, double>::value ? 1 : 0;
    c.N = NUM_TIME_STEPS; c.M = M_BLOCKS; c.A = NUM_ALPHA; c.integrator = INTEGRATOR; c.batch = 1; c.max_iter = MAX_ITER;
    c.wafr_urdf = USE_WAFR_URDF; c.mpc_mode = MPC_MODE; c.ignore_max_rho_exit = IGNORE_MAX_ROX_EXIT; c.device = PDDP_DEVICE;
    c.use_graph = 1;   // sweeps replay from a hipGraph; with PDDP_PHASE_TIMERS runiLQR_GPU asks pddp_solve_ex for per-phase times, which launches kernel by kernel
    c.total_time = TOTAL_TIME; c.alpha_base = ALPHA_BASE; c.rho_init = RHO_INIT; c.max_defect = MAX_DEFECT_SIZE; c.tol_cost = TOL_COST;
    c.exp_red_min = EXP_RED_MIN; c.exp_red_max = EXP_RED_MAX;
    c.Q1 = _Q1; c.Q2 = _Q2; c.R = _R; c.QF1 = _QF1; c.QF2 = _QF2;
    c.ee_cost = EE_COST; c.Q_EE1 = _Q_EE1; c.Q_EE2 = _Q_EE2; c.QF_EE1 = _QF_EE1; c.QF_EE2 = _QF_EE2; c.R_EE = _R_EE;
    c.Q_xEE = _Q_xEE; c.QF_xEE = _QF_xEE; c.Q_xdEE = _Q_xdEE; c.QF_xdEE = _QF_xdEE; c.ee_on_link_z = EE_ON_LINK_Z; c.ee_type = EE_TYPE;
    c.ee_initial_cost_fix = PDDP_EE_INITIAL_COST_FIX;
    c.use_finite_diff = USE_FINITE_DIFF; c.finite_diff_epsilon = FINITE_DIFF_EPSILON;   // config.cuh:68-71
    c.use_limits = USE_LIMITS_FLAG; c.use_smooth_abs = USE_SMOOTH_ABS; c.smooth_abs_alpha = SMOOTH_ABS_ALPHA;   // config.cuh:171-176
    Context* ctx = new Context();
    check(pddp_create(&c, &ctx->h), "allocateMemory_GPU");
    pddp_handle h = ctx->h;
    const double w0[14] = {_Q1, _Q2, _R, _QF1, _QF2, _Q_EE1, _Q_EE2, _QF_EE1, _QF_EE2, _R_EE, _Q_xEE, _QF_xEE, _Q_xdEE, _QF_xdEE};
    for (int i = 0; i < 14; i++) ctx->cost[i] = w0[i];
    ctx->phase.assign(5 * (MAX_ITER + 2), 0.0);
    ctx->Jtmp.assign(sizeof(T) * (MAX_ITER + 2), 0);
    ctx->atmp.assign(MAX_ITER + 2, 0);

    *ld_x = DIM_x_r; *ld_u = DIM_u_r; *ld_P = DIM_P_r; *ld_p = DIM_p_r; *ld_AB = DIM_AB_r; *ld_H = DIM_H_r; *ld_g = DIM_g_r;
    *ld_KT = DIM_KT_r; *ld_du = DIM_du_r; *ld_d = DIM_d_r; *ld_A = DIM_A_r;
    const size_t N = NUM_TIME_STEPS;
    T* xs = dev<T>(h, "xs"); T* us = dev<T>(h, "us"); T* ds = dev<T>(h, "ds");
    *h_d_x = static_cast<T**>(std::malloc(NUM_ALPHA * sizeof(T*)));
    *h_d_u = static_cast<T**>(std::malloc(NUM_ALPHA * sizeof(T*)));
    *h_d_d = static_cast<T**>(std::malloc(NUM_ALPHA * sizeof(T*)));
    for (int a = 0; a < NUM_ALPHA; a++) {
        (*h_d_x)[a] = xs + a * N * STATE_SIZE; (*h_d_u)[a] = us + a * N * CONTROL_SIZE; (*h_d_d)[a] = ds + a * N * STATE_SIZE;
    }
    *d_x = reinterpret_cast<T**>(dev<void*>(h, "xs_ptrs")); *d_u = reinterpret_cast<T**>(dev<void*>(h, "us_ptrs"));
    *d_d = reinterpret_cast<T**>(dev<void*>(h, "ds_ptrs"));
    *d_xp = dev<T>(h, "xb"); *d_xp2 = dev<T>(h, "xb") + N * STATE_SIZE; *d_up = dev<T>(h, "ucur"); *d_dp = dev<T>(h, "dcur");
    *d_xGoal = dev<T>(h, "xGoal"); *xGoal = static_cast<T*>(std::calloc(STATE_SIZE, sizeof(T)));
    *d_P = dev<T>(h, "P"); *d_Pp = dev<T>(h, "Pp"); *d_p = dev<T>(h, "p"); *d_pp = dev<T>(h, "pp");
    *d_AB = dev<T>(h, "AB"); *d_H = dev<T>(h, "H"); *d_g = dev<T>(h, "g"); *d_KT = dev<T>(h, "KT"); *d_du = dev<T>(h, "du");
    *d_dT = dev<T>(h, "dmax"); *d_dM = dev<T>(h, "dmax"); *d = static_cast<T*>(std::calloc(NUM_ALPHA, sizeof(T)));
    *d_ApBK = dev<T>(h, "ApBK"); *d_Bdu = dev<T>(h, "Bdu");
    *d_JT = dev<T>(h, "J"); *J = static_cast<T*>(std::calloc(NUM_ALPHA, sizeof(T)));
    *d_dJexp = dev<T>(h, "dJexp"); *dJexp = static_cast<T*>(std::calloc(2 * M_BLOCKS_B, sizeof(T)));
    *d_alpha = dev<T>(h, "alpha"); *alpha = static_cast<T*>(std::calloc(NUM_ALPHA, sizeof(T)));
    check(pddp_get_array(h, "alpha", *alpha, NUM_ALPHA * sizeof(T)), "alpha");      // alpha[i] = ALPHA_BASE^i (:829)
    *alphaIndex = static_cast<int*>(std::calloc(1, sizeof(int)));
    *d_err = dev<int>(h, "err"); *err = static_cast<int*>(std::calloc(M_BLOCKS_B, sizeof(int)));
    void* st = nullptr;
    check(pddp_stream(h, &st), "pddp_stream");
    *streams = static_cast<pddpStream_t*>(std::malloc(NUM_STREAMS * sizeof(pddpStream_t)));
    for (int i = 0; i < NUM_STREAMS; i++) (*streams)[i] = static_cast<pddpStream_t>(st);
    if (d_I) *d_I = (PLANT == 4) ? dev<T>(h, "model_I") : nullptr;
    if (d_Tbody) *d_Tbody = (PLANT == 4) ? dev<T>(h, "model_F") : nullptr;
    { std::lock_guard<std::mutex> guard(registry_lock()); registry()[*d_P] = ctx; }
}

template <typename T>
void runiLQR_GPU(T* x0, T* u0, T* KT0, T* P0, T* p0, T* d0, T* xGoal, T* Jout, int* alphaOut, int forwardRolloutFlag, int clearVarsFlag,
                 int ignoreFirstDefectFlag, double* tTime, double* simTime, double* sweepTime, double* bpTime, double* nisTime, double* initTime,
                 pddpStream_t* streams, T** d_x, T** h_d_x, T* d_xp, T* d_xp2, T** d_u, T** h_d_u, T* d_up, T* d_P, T* d_p, T* d_Pp, T* d_pp, T* d_AB,
                 T* d_H, T* d_g, T* d_KT, T* d_du, T** d_d, T** h_d_d, T* d_dp, T* d_dT, T* d, T* d_ApBK, T* d_Bdu, T* d_dM, T* alpha, T* d_alpha,
                 int* alphaIndex, T* d_JT, T* J, T* dJexp, T* d_dJexp, T* d_xGoal, int* err, int* d_err, int ld_x, int ld_u, int ld_P, int ld_p,
                 int ld_AB, int ld_H, int ld_g, int ld_KT, int ld_du, int ld_d, int ld_A, T* d_I = nullptr, T* d_Tbody = nullptr,
                 T Q_EE1 = _Q_EE1, T Q_EE2 = _Q_EE2, T QF_EE1 = _QF_EE1, T QF_EE2 = _QF_EE2, T Q_EEV1 = _Q_EEV1, T Q_EEV2 = _Q_EEV2,
                 T QF_EEV1 = _QF_EEV1, T QF_EEV2 = _QF_EEV2, T R_EE = _R_EE, T Q_xdEE = _Q_xdEE, T QF_xdEE = _QF_xdEE, T Q_xEE = _Q_xEE,
                 T QF_xEE = _QF_xEE, T Q1 = _Q1, T Q2 = _Q2, T R = _R, T QF1 = _QF1, T QF2 = _QF2) {
    using namespace pddp_hostapi;
    (void)streams; (void)d_x; (void)h_d_x; (void)d_xp; (void)d_xp2; (void)d_u; (void)h_d_u; (void)d_up; (void)d_p; (void)d_Pp; (void)d_pp;
    (void)d_AB; (void)d_H; (void)d_g; (void)d_KT; (void)d_du; (void)d_d; (void)h_d_d; (void)d_dp; (void)d_dT; (void)d_ApBK; (void)d_Bdu; (void)d_dM;
    (void)alpha; (void)d_alpha; (void)d_JT; (void)d_dJexp; (void)d_xGoal; (void)d_err; (void)ld_x; (void)ld_u; (void)ld_P; (void)ld_p; (void)ld_AB;
    (void)ld_H; (void)ld_g; (void)ld_KT; (void)ld_du; (void)ld_d; (void)ld_A; (void)d_I; (void)d_Tbody;
    (void)Q_EEV1; (void)Q_EEV2; (void)QF_EEV1; (void)QF_EEV2;
    Context* ctx = find(d_P);
    pddp_handle h = ctx->h;
    apply_cost<T>(ctx, Q1, Q2, R, QF1, QF2, Q_EE1, Q_EE2, QF_EE1, QF_EE2, R_EE, Q_xEE, QF_xEE, Q_xdEE, QF_xdEE);
    double times[2] = {0, 0};
    int sweeps = 0;
    T* Jtmp = reinterpret_cast<T*>(ctx->Jtmp.data());
    check(pddp_solve_ex(h, x0, u0, xGoal, KT0, P0, p0, d0, Jtmp, ctx->atmp.data(), forwardRolloutFlag, clearVarsFlag, ignoreFirstDefectFlag,
                        PDDP_POLL_EVERY, times, PDDP_PHASE_TIMERS ? ctx->phase.data() : nullptr, &sweeps), "runiLQR_GPU");
    std::memcpy(Jout, Jtmp, (MAX_ITER + 1) * sizeof(T));                 // the reference's arrays hold MAX_ITER+1 entries
    std::memcpy(alphaOut, ctx->atmp.data(), (MAX_ITER + 1) * sizeof(int));
#if PDDP_REFERENCE_VIEWS
    check(pddp_refresh_reference_views(h), "pddp_refresh_reference_views");      // d_ApBK / d_Bdu and the winner in every alpha slot, as the reference leaves them
#endif
    pddp_state st;
    check(pddp_get_state(h, &st), "pddp_get_state");
    const int iter = st.iter;
    *alphaIndex = st.alphaIndex;
    check(pddp_get_array(h, "J", J, NUM_ALPHA * sizeof(T)), "J");
    check(pddp_get_array(h, "dmax", d, NUM_ALPHA * sizeof(T)), "dmax");
    check(pddp_get_array(h, "dJexp", dJexp, 2 * M_BLOCKS_B * sizeof(T)), "dJexp");
    check(pddp_get_array(h, "err", err, M_BLOCKS_B * sizeof(int)), "err");
    if (tTime) *tTime = times[0];
    if (initTime) *initTime = times[1];
    const int stride = MAX_ITER + 2;
    for (int k = 0; k < iter && k < MAX_ITER; k++) {                      // bpTime[iter-1] ... (DDPWrappers.cuh:65,78,89,100)
        const double* ph = ctx->phase.data();
        if (bpTime) bpTime[k] = PDDP_PHASE_TIMERS ? ph[0 * stride + k] : 0.0;
        if (sweepTime) sweepTime[k] = PDDP_PHASE_TIMERS ? ph[4 * stride + k] : 0.0;      // forwardSweepKern's part (k_sweep_maps & co.; DDPWrappers.cuh:77)
        if (simTime) simTime[k] = PDDP_PHASE_TIMERS ? ph[1 * stride + k] - ph[4 * stride + k] + ph[2 * stride + k] : 0.0;   // rollouts + line search (:89)
        if (nisTime) nisTime[k] = PDDP_PHASE_TIMERS ? ph[3 * stride + k] : 0.0;
    }
    std::printf("GPU (MI355X) Parallel blocks:[%d] t:[%f] with FP[%f], FS[%f], BP[%f], NIU[%f] Xf:[%.4f, %.4f] iters:[%d] cost:[%f] max_d[%f]\n",
                M_BLOCKS_B, times[0], simTime ? *simTime : 0.0, sweepTime ? *sweepTime : 0.0, bpTime ? *bpTime : 0.0, nisTime ? *nisTime : 0.0,
                (double)x0[DIM_x_r * (NUM_TIME_STEPS - 1)], (double)x0[DIM_x_r * (NUM_TIME_STEPS - 1) + 1], iter, st.prevJ,
                (double)d[st.alphaIndex]);
}

template <typename T>
void freeMemory_GPU(T** d_x, T** h_d_x, T* d_xp, T* d_xp2, T** d_u, T** h_d_u, T* d_up, T* xGoal, T* d_xGoal, T* d_P, T* d_Pp, T* d_p, T* d_pp,
                    T* d_AB, T* d_H, T* d_g, T* d_KT, T* d_du, T** d_d, T** h_d_d, T* d_dp, T* d_dM, T* d_dT, T* d, T* d_ApBK, T* d_Bdu, T* d_JT,
                    T* J, T* d_dJexp, T* dJexp, T* alpha, T* d_alpha, int* alphaIndex, int* d_err, int* err, pddpStream_t* streams,
                    T* d_I = nullptr, T* d_Tbody = nullptr) {
    using namespace pddp_hostapi;
    (void)d_x; (void)d_xp; (void)d_xp2; (void)d_u; (void)d_up; (void)d_xGoal; (void)d_Pp; (void)d_p; (void)d_pp; (void)d_AB; (void)d_H; (void)d_g;
    (void)d_KT; (void)d_du; (void)d_d; (void)d_dp; (void)d_dM; (void)d_dT; (void)d_ApBK; (void)d_Bdu; (void)d_JT; (void)d_dJexp; (void)d_alpha;
    (void)d_err; (void)d_I; (void)d_Tbody;
    Context* ctx = find(d_P);
    { std::lock_guard<std::mutex> guard(registry_lock()); registry().erase(d_P); }
    check(pddp_destroy(ctx->h), "freeMemory_GPU");
    delete ctx;
    std::free(h_d_x); std::free(h_d_u); std::free(h_d_d); std::free(xGoal); std::free(d); std::free(J); std::free(dJexp); std::free(alpha);
    std::free(alphaIndex); std::free(err); std::free(streams);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The reference's CPU twins (DDPHelpers/nisInitHelpers.cuh:886-925 allocateMemory_CPU, :950-958 freeMemory_CPU; DDPHelpers/DDPWrappers.cuh:142-248
// runiLQR_CPU), same names and argument order, over libpddp_cpu.so (include/pddp_cpu.h): the caller owns plain host buffers, exactly as upstream.
// Compiled in when the translation unit defines PDDP_WITH_CPU_PATH (link with -lpddp_cpu); examples/WAFR_iLQR_examples.cu:231-299 (`testCPU`, both
// its serial and its parallel line-search branch) then recompiles unchanged.  runiLQR_GPU never routes here, and this path never touches the GPU library.
#ifdef PDDP_WITH_CPU_PATH
#include "../../include/pddp_cpu.h"
#include <cmath>
#include <thread>

template <typename T>
void allocateMemory_CPU(T** x, T** xp, T** xp2, T** u, T** up, T** xGoal, T** P, T** Pp, T** p, T** pp, T** AB, T** H, T** g, T** KT, T** du, T** d, T** dp,
                        T** ApBK, T** Bdu, T** JT, T** dJexp, T** alpha, int** err, int* ld_x, int* ld_u, int* ld_P, int* ld_p, int* ld_AB, int* ld_H,
                        int* ld_g, int* ld_KT, int* ld_du, int* ld_d, int* ld_A, T** I = nullptr, T** Tbody = nullptr) {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "algType must be float or double");
    *ld_x = DIM_x_r; *ld_u = DIM_u_r; *ld_AB = DIM_AB_r; *ld_P = DIM_P_r; *ld_p = DIM_p_r; *ld_H = DIM_H_r; *ld_g = DIM_g_r; *ld_KT = DIM_KT_r;
    *ld_du = DIM_du_r; *ld_d = DIM_d_r; *ld_A = DIM_A_r;
    const size_t N = NUM_TIME_STEPS;
    auto arr = [](size_t count) { return static_cast<T*>(std::calloc(count, sizeof(T))); };
    *x = arr(DIM_x_r * N); *xp = arr(DIM_x_r * N); *xp2 = arr(DIM_x_r * N); *u = arr(DIM_u_r * N); *up = arr(DIM_x_r * N);
    *xGoal = arr(EE_COST ? 6 : STATE_SIZE);
    *P = arr(DIM_P_r * DIM_P_c * N); *Pp = arr(DIM_P_r * DIM_P_c * N); *p = arr(DIM_p_r * N); *pp = arr(DIM_p_r * N);
    *AB = arr(DIM_AB_r * DIM_AB_c * N); *H = arr(DIM_H_r * DIM_H_c * N); *g = arr(DIM_g_r * N); *KT = arr(DIM_KT_r * DIM_KT_c * N); *du = arr(DIM_du_r * N);
    *d = arr(DIM_d_r * N); *dp = arr(DIM_d_r * N); *Bdu = arr(DIM_d_r * N); *ApBK = arr(DIM_A_r * DIM_A_c * N);
    int bp_t = 1, fsim_t = 1, cost_t = 1, integ_t = 1;
    pddp_cpu_thread_counts(M_BLOCKS_B, 0, &bp_t, &fsim_t, &cost_t, &integ_t);
    *JT = arr(fsim_t > cost_t ? fsim_t : cost_t);          // max(FSIM_THREADS, COST_THREADS) partial sums (:917)
    *dJexp = arr(2 * (M_BLOCKS_B > bp_t ? M_BLOCKS_B : bp_t));
    *alpha = arr(NUM_ALPHA);
    for (int i = 0; i < NUM_ALPHA; i++) (*alpha)[i] = (T)std::pow((double)ALPHA_BASE, (double)i);   // :920
    *err = static_cast<int*>(std::calloc(M_BLOCKS_B > bp_t ? M_BLOCKS_B : bp_t, sizeof(int)));
    if (I) *I = arr(36 * NUM_POS);                          // opaque here: the robot constants live inside the library (wafr_urdf / mpc_mode select them)
    if (Tbody) *Tbody = arr(36 * NUM_POS);
}

template <typename T>
inline pddp_config pddp_cpu_config_from_macros(T Q1, T Q2, T R, T QF1, T QF2) {
    pddp_config c;
    std::memset(&c, 0, sizeof(c));
    c.plant = PLANT; c.dtype = std::is_same<T, double>::value ? 1 : 0;
    c.N = NUM_TIME_STEPS; c.M = M_BLOCKS; c.A = NUM_ALPHA; c.integrator = INTEGRATOR; c.batch = 1; c.max_iter = MAX_ITER;
    c.wafr_urdf = USE_WAFR_URDF; c.mpc_mode = MPC_MODE; c.ignore_max_rho_exit = IGNORE_MAX_ROX_EXIT;
    c.total_time = TOTAL_TIME; c.alpha_base = ALPHA_BASE; c.rho_init = RHO_INIT; c.max_defect = MAX_DEFECT_SIZE; c.tol_cost = TOL_COST;
    c.exp_red_min = EXP_RED_MIN; c.exp_red_max = EXP_RED_MAX; c.Q1 = Q1; c.Q2 = Q2; c.R = R; c.QF1 = QF1; c.QF2 = QF2; c.ee_cost = EE_COST; c.ee_type = EE_TYPE; c.ee_on_link_z = EE_ON_LINK_Z;
    c.use_finite_diff = USE_FINITE_DIFF; c.finite_diff_epsilon = FINITE_DIFF_EPSILON;
    c.use_limits = USE_LIMITS_FLAG; c.use_smooth_abs = USE_SMOOTH_ABS; c.smooth_abs_alpha = SMOOTH_ABS_ALPHA;
    return c;
}

template <typename T>
void runiLQR_CPU(T* x0, T* u0, T* KT0, T* P0, T* p0, T* d0, T* xGoal, T* Jout, int* alphaOut, int forwardRolloutFlag, int clearVarsFlag, int ignoreFirstDefectFlag,
                 double* tTime, double* simTime, double* sweepTime, double* bpTime, double* nisTime, double* initTime, T* x, T* xp, T* xp2, T* u, T* up, T* P,
                 T* p, T* Pp, T* pp, T* AB, T* H, T* g, T* KT, T* du, T* d, T* dp, T* ApBK, T* Bdu, T* alpha, T* JT, T* dJexp, int* err, int ld_x, int ld_u,
                 int ld_P, int ld_p, int ld_AB, int ld_H, int ld_g, int ld_KT, int ld_du, int ld_d, int ld_A, T* I = nullptr, T* Tbody = nullptr,
                 T Q_EE1 = _Q_EE1, T Q_EE2 = _Q_EE2, T QF_EE1 = _QF_EE1, T QF_EE2 = _QF_EE2, T Q_EEV1 = _Q_EEV1, T Q_EEV2 = _Q_EEV2, T QF_EEV1 = _QF_EEV1,
                 T QF_EEV2 = _QF_EEV2, T R_EE = _R_EE, T Q_xdEE = _Q_xdEE, T QF_xdEE = _QF_xdEE, T Q_xEE = _Q_xEE, T QF_xEE = _QF_xEE, T Q1 = _Q1, T Q2 = _Q2,
                 T R = _R, T QF1 = _QF1, T QF2 = _QF2) {
    (void)ld_x; (void)ld_u; (void)ld_P; (void)ld_p; (void)ld_AB; (void)ld_H; (void)ld_g; (void)ld_KT; (void)ld_du; (void)ld_d; (void)ld_A; (void)I; (void)Tbody;
    (void)Q_EE1; (void)Q_EE2; (void)QF_EE1; (void)QF_EE2; (void)Q_EEV1; (void)Q_EEV2; (void)QF_EEV1; (void)QF_EEV2; (void)R_EE; (void)Q_xdEE; (void)QF_xdEE;
    (void)Q_xEE; (void)QF_xEE;
    pddp_config c = pddp_cpu_config_from_macros<T>(Q1, Q2, R, QF1, QF2);
    pddp_cpu_buffers b = {x, xp, xp2, u, up, P, p, Pp, pp, AB, H, g, KT, du, d, dp, ApBK, Bdu, alpha, JT, dJexp, err, nullptr, nullptr, nullptr, nullptr};
    int iter = 0;
    const int rc = pddp_cpu_run_ilqr(&c, &b, x0, u0, KT0, P0, p0, d0, xGoal, Jout, alphaOut, forwardRolloutFlag, clearVarsFlag, ignoreFirstDefectFlag, tTime,
                                     simTime, sweepTime, bpTime, nisTime, initTime, 0, &iter);
    if (rc) { std::fprintf(stderr, "runiLQR_CPU: %s (code %d)\n", pddp_cpu_last_error(), rc); std::exit(rc < 0 ? -rc : rc); }
    std::printf("CPU Parallel blocks:[%d] t:[%f] with FP[%f], FS[%f], BP[%f], NIU[%f] Xf:[%.4f, %.4f] iters:[%d] cost:[%f] max_d[%f]\n", M_BLOCKS_B, *tTime,
                *simTime, *sweepTime, *bpTime, *nisTime, (double)x0[DIM_x_r * (NUM_TIME_STEPS - 1)], (double)x0[DIM_x_r * (NUM_TIME_STEPS - 1) + 1], iter,
                (double)Jout[iter], 0.0);
}

// The parallel-line-search variants (nisInitHelpers.cuh:927-948 allocateMemory_CPU2, :959-966 freeMemory_CPU2; DDPWrappers.cuh:252-363 runiLQR_CPU2): one
// trajectory slot (xs, us, ds) and one partial-sum array (JTs) per step size; `testCPU(0)` of examples/WAFR_iLQR_examples.cu:231-299 runs this path.
template <typename T>
void allocateMemory_CPU2(T*** xs, T** x, T** xp, T** xp2, T*** us, T** u, T** up, T** xGoal, T** P, T** Pp, T** p, T** pp, T** AB, T** H, T** g, T** KT, T** du,
                         T*** ds, T** d, T** dp, T** ApBK, T** Bdu, T*** JTs, T** dJexp, T** alpha, int** err, int* ld_x, int* ld_u, int* ld_P, int* ld_p,
                         int* ld_AB, int* ld_H, int* ld_g, int* ld_KT, int* ld_du, int* ld_d, int* ld_A, T** I = nullptr, T** Tbody = nullptr) {
    *xs = static_cast<T**>(std::malloc(NUM_ALPHA * sizeof(T*))); *us = static_cast<T**>(std::malloc(NUM_ALPHA * sizeof(T*)));
    *ds = static_cast<T**>(std::malloc(NUM_ALPHA * sizeof(T*))); *JTs = static_cast<T**>(std::malloc(NUM_ALPHA * sizeof(T*)));
    allocateMemory_CPU<T>(x, xp, xp2, u, up, xGoal, P, Pp, p, pp, AB, H, g, KT, du, d, dp, ApBK, Bdu, &((*JTs)[0]), dJexp, alpha, err, ld_x, ld_u, ld_P, ld_p, ld_AB,
                          ld_H, ld_g, ld_KT, ld_du, ld_d, ld_A, I, Tbody);
    int bp_t = 1, fsim_t = 1, cost_t = 1, integ_t = 1;
    pddp_cpu_thread_counts(M_BLOCKS_B, 0, &bp_t, &fsim_t, &cost_t, &integ_t);
    const size_t N = NUM_TIME_STEPS, nj = fsim_t > cost_t ? fsim_t : cost_t;
    for (int i = 0; i < NUM_ALPHA; i++) {
        (*xs)[i] = static_cast<T*>(std::calloc(DIM_x_r * N, sizeof(T))); (*us)[i] = static_cast<T*>(std::calloc(DIM_u_r * N, sizeof(T)));
        (*ds)[i] = static_cast<T*>(std::calloc(DIM_d_r * N, sizeof(T)));
        if (i > 0) (*JTs)[i] = static_cast<T*>(std::calloc(nj, sizeof(T)));      // slot 0 is allocateMemory_CPU's JT (upstream allocates it twice and leaks the first)
    }
}

template <typename T>
void runiLQR_CPU2(T* x0, T* u0, T* KT0, T* P0, T* p0, T* d0, T* xGoal, T* Jout, int* alphaOut, int forwardRolloutFlag, int clearVarsFlag, int ignoreFirstDefectFlag,
                  double* tTime, double* simTime, double* sweepTime, double* bpTime, double* nisTime, double* initTime, T** xs, T* x, T* xp, T* xp2, T** us, T* u,
                  T* up, T* P, T* p, T* Pp, T* pp, T* AB, T* H, T* g, T* KT, T* du, T** ds, T* d, T* dp, T* ApBK, T* Bdu, T* alphas, T** JTs, T* dJexp, int* err,
                  int ld_x, int ld_u, int ld_P, int ld_p, int ld_AB, int ld_H, int ld_g, int ld_KT, int ld_du, int ld_d, int ld_A, T* I = nullptr, T* Tbody = nullptr,
                  T Q_EE1 = _Q_EE1, T Q_EE2 = _Q_EE2, T QF_EE1 = _QF_EE1, T QF_EE2 = _QF_EE2, T Q_EEV1 = _Q_EEV1, T Q_EEV2 = _Q_EEV2, T QF_EEV1 = _QF_EEV1,
                  T QF_EEV2 = _QF_EEV2, T R_EE = _R_EE, T Q_xdEE = _Q_xdEE, T QF_xdEE = _QF_xdEE, T Q_xEE = _Q_xEE, T QF_xEE = _QF_xEE, T Q1 = _Q1, T Q2 = _Q2,
                  T R = _R, T QF1 = _QF1, T QF2 = _QF2) {
    (void)ld_x; (void)ld_u; (void)ld_P; (void)ld_p; (void)ld_AB; (void)ld_H; (void)ld_g; (void)ld_KT; (void)ld_du; (void)ld_d; (void)ld_A; (void)I; (void)Tbody;
    (void)Q_EE1; (void)Q_EE2; (void)QF_EE1; (void)QF_EE2; (void)Q_EEV1; (void)Q_EEV2; (void)QF_EEV1; (void)QF_EEV2; (void)R_EE; (void)Q_xdEE; (void)QF_xdEE;
    (void)Q_xEE; (void)QF_xEE;
    pddp_config c = pddp_cpu_config_from_macros<T>(Q1, Q2, R, QF1, QF2);
    pddp_cpu_buffers b = {x, xp, xp2, u, up, P, p, Pp, pp, AB, H, g, KT, du, d, dp, ApBK, Bdu, alphas, JTs[0], dJexp, err,
                          reinterpret_cast<void**>(xs), reinterpret_cast<void**>(us), reinterpret_cast<void**>(ds), reinterpret_cast<void**>(JTs)};
    int iter = 0;
    const int rc = pddp_cpu_run_ilqr2(&c, &b, x0, u0, KT0, P0, p0, d0, xGoal, Jout, alphaOut, forwardRolloutFlag, clearVarsFlag, ignoreFirstDefectFlag, tTime,
                                      simTime, sweepTime, bpTime, nisTime, initTime, 0, &iter);
    if (rc) { std::fprintf(stderr, "runiLQR_CPU2: %s (code %d)\n", pddp_cpu_last_error(), rc); std::exit(rc < 0 ? -rc : rc); }
    std::printf("CPU Parallel blocks:[%d] t:[%f] with FP[%f], FS[%f], BP[%f], NIU[%f] Xf:[%.4f, %.4f] iters:[%d] cost:[%f] max_d[%f]\n", M_BLOCKS_B, *tTime,
                *simTime, *sweepTime, *bpTime, *nisTime, (double)x0[DIM_x_r * (NUM_TIME_STEPS - 1)], (double)x0[DIM_x_r * (NUM_TIME_STEPS - 1) + 1], iter,
                (double)Jout[iter], 0.0);
}

template <typename T>
void freeMemory_CPU(T* x, T* xp, T* xp2, T* u, T* up, T* P, T* Pp, T* p, T* pp, T* AB, T* H, T* g, T* KT, T* du, T* d, T* dp, T* Bdu, T* ApBK, T* dJexp, int* err,
                    T* alpha, T* JT, T* xGoal, T* I = nullptr, T* Tbody = nullptr) {
    std::free(x); std::free(xp); std::free(xp2); std::free(u); std::free(up); std::free(P); std::free(Pp); std::free(p); std::free(pp); std::free(AB); std::free(H);
    std::free(g); std::free(KT); std::free(du); std::free(d); std::free(dp); std::free(Bdu); std::free(ApBK); std::free(dJexp); std::free(err); std::free(alpha);
    std::free(JT); std::free(xGoal); if (I) std::free(I); if (Tbody) std::free(Tbody);
}
template <typename T>
void freeMemory_CPU2(T** xs, T* x, T* xp, T* xp2, T** us, T* u, T* up, T* P, T* Pp, T* p, T* pp, T* AB, T* H, T* g, T* KT, T* du, T** ds, T* d, T* dp, T* Bdu, T* ApBK,
                     T* dJexp, int* err, T* alpha, T** JTs, T* xGoal, T* I = nullptr, T* Tbody = nullptr) {
    freeMemory_CPU<T>(x, xp, xp2, u, up, P, Pp, p, pp, AB, H, g, KT, du, d, dp, Bdu, ApBK, dJexp, err, alpha, JTs[0], xGoal, I, Tbody);
    for (int i = 0; i < NUM_ALPHA; i++) { std::free(xs[i]); std::free(us[i]); std::free(ds[i]); if (i > 0) std::free(JTs[i]); }
    std::free(xs); std::free(us); std::free(ds); std::free(JTs);
}
#endif   // PDDP_WITH_CPU_PATH

#endif
