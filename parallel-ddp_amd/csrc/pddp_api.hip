// libpddp.so: host side of the C ABI declared in include/pddp.h.  gfx950 only; no CPU fallback.
#include "solver_base.hpp"

static thread_local std::string g_err;
extern "C" const char* pddp_last_error(void) { return g_err.c_str(); }
int pddp_internal_fail(int code, const std::string& msg) { g_err = msg; return code; }      // every translation unit of the library reports through this (solver_base.hpp)
#ifdef PDDP_HAS_USER_PLANT
static constexpr int kMaxPlant = 5;
static int user_nx() { return pddp_user_plant_state_size(); }
static int user_nu() { return pddp_user_plant_control_size(); }
#else
static constexpr int kMaxPlant = 4;
static int user_nx() { return -1; }
static int user_nu() { return -1; }
#endif
extern "C" int pddp_state_size(int plant) { return plant == 1 ? 2 : plant == 2 ? 4 : plant == 3 ? 12 : plant == 4 ? 14 : plant == 5 ? user_nx() : -1; }
extern "C" int pddp_control_size(int plant) { return plant == 1 ? 1 : plant == 2 ? 1 : plant == 3 ? 4 : plant == 4 ? 7 : plant == 5 ? user_nu() : -1; }

// Reference defaults: config.cuh:24-61 per plant, :78-136 algorithm, plants/cost_arm.cuh:97-103 weights.
extern "C" int pddp_default_config(pddp_config* c, int plant) {
    if (!c || plant < 1 || plant > kMaxPlant) return fail(PDDP_EINVAL, "plant must be 1..4 (5: the user plant of a `make user PLANT_POLICY=...` build)");
    std::memset(c, 0, sizeof(*c));
    c->plant = plant; c->dtype = 0;
    c->N = plant == 4 ? 64 : 128; c->M = 4;                      // plant 5 (a user plant) starts from the pendulum's defaults
    c->A = (plant == 3 || plant == 4) ? 16 : 32;
    c->integrator = plant == 4 ? 1 : 3;
    c->batch = 1; c->max_iter = 100; c->ignore_max_rho_exit = 1;
    c->total_time = plant == 4 ? 0.5 : 4.0;
    c->alpha_base = (plant == 3 || plant == 4) ? 0.5 : 0.75;
    c->rho_init = plant == 4 ? 12.5 : (plant == 3 ? 1.0 : 10.0);
    c->max_defect = plant == 2 ? 0.75 : 1.0;
    c->tol_cost = 0.0001; c->exp_red_min = 0.05; c->exp_red_max = 1.25;
    c->Q1 = 0.1; c->Q2 = 0.001; c->R = 0.0001; c->QF1 = 1000.0; c->QF2 = 1000.0;
    c->Q_EE1 = 0.1; c->Q_EE2 = 0.0; c->QF_EE1 = 1000.0; c->QF_EE2 = 0.0; c->R_EE = 0.0001; c->Q_xEE = 0.0; c->QF_xEE = 0.0; c->Q_xdEE = 0.1; c->QF_xdEE = 1000.0;
    c->ee_on_link_z = 0.0635;   // plants/cost_arm.cuh:104-115, dynamics_arm.cuh:57-58 (EE_TYPE 1)
    c->use_finite_diff = 0; c->finite_diff_epsilon = 0.00001;   // config.cuh:68-71
    c->use_limits = 0;                                          // config.cuh:171-173
    c->use_smooth_abs = 0; c->smooth_abs_alpha = 0.2;           // config.cuh:174-176, cost_arm.cuh:116-118
    c->ee_type = 1;                                             // dynamics_arm.cuh:50-52
    std::memset(&c->kernels, 0, sizeof(c->kernels));           // the library's own kernel selection
    return 0;
}

static SolverBase* make_solver(const pddp_config& c) {
    switch (c.plant) {
    case 1: return pddp_make_solver_pend(c);
    case 2: return pddp_make_solver_cart(c);
    case 3: return pddp_make_solver_quad(c);
    case 4: return pddp_make_solver_arm(c);      // the arm is Euler-only, as config.cuh:58
#ifdef PDDP_HAS_USER_PLANT
    case 5: return pddp_make_solver_user(c);
#endif
    }
    return nullptr;
}

// HBM counter calibration (profiling tool): streams `count` floats from src to dst with the access shape the sweep kernels use (one dword per lane, consecutive lanes on
// consecutive addresses).  Its byte count is known exactly, which calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for this access width (MI355X_MICROARCH.md, "HBM").
__global__ __launch_bounds__(256) void k_hbm_calib_dword(const float* __restrict__ src, float* __restrict__ dst, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int pddp_create(const pddp_config* cfg, pddp_handle* out) {
    if (!cfg || !out) return fail(PDDP_EINVAL, "null argument");
    const pddp_config& c = *cfg;
    if (c.plant < 1 || c.plant > kMaxPlant) return fail(PDDP_EINVAL, "plant must be 1..4 (5: the user plant of a `make user PLANT_POLICY=...` build)");
    if (c.N < 4 || (c.N & (c.N - 1)) || c.N > 1024) return fail(PDDP_EINVAL, "N must be a power of two in [4,1024] (the reference's tree reductions assume it)");
    if (c.M < 1 || c.N % c.M || c.N / c.M < 2 || c.M > 16) return fail(PDDP_EINVAL, "M must divide N, N/M >= 2, M <= 16");
    if (c.A < 1 || c.A > 64 || c.batch < 1 || c.max_iter < 1) return fail(PDDP_EINVAL, "A in [1,64], batch >= 1, max_iter >= 1");
    if (c.ee_cost && c.plant != 4) return fail(PDDP_EINVAL, "ee_cost: the end-effector cost family belongs to the KUKA arm (plant 4)");
    if (c.ee_type < 0 || c.ee_type > 2) return fail(PDDP_EINVAL, "ee_type: EE_TYPE is 0 (no end effector), 1 (flange) or 2 (flange + peg) (dynamics_arm.cuh:50-65)");
    if (c.use_limits && c.plant != 4) return fail(PDDP_EINVAL, "use_limits: USE_LIMITS_FLAG belongs to the KUKA arm's cost files (plant 4)");
    if (c.use_smooth_abs && !(c.plant == 4 && c.ee_cost && c.smooth_abs_alpha > 0.0)) return fail(PDDP_EINVAL, "use_smooth_abs: USE_SMOOTH_ABS belongs to the end-effector cost (plant 4, ee_cost = 1, smooth_abs_alpha > 0)");
    if (c.use_finite_diff && (c.integrator != 1 || c.ee_cost || !(c.finite_diff_epsilon > 0.0)))
        return fail(PDDP_EINVAL, "use_finite_diff: the finite-difference [A B] is the Euler rule's (finiteDiffInner, nisInitHelpers.cuh:138-166), with the joint-space cost and a positive finite_diff_epsilon");
    if (c.plant == 4 && ((c.A > 8 && c.A % 8 == 0) ? 8 : c.A) * c.M > 128)
        return fail(PDDP_EINVAL, "KUKA arm: (candidates per workgroup) * M must not exceed 128 -- a workgroup rolls out 8 candidates when A is a multiple of 8, otherwise all A");
    if (c.plant == 4 && (double)c.batch * c.N * (c.A * 14 > 441 ? c.A * 14 : 441) >= 4294967296.0)
        return fail(PDDP_EINVAL, "KUKA arm: batch * N too large for the 32-bit element offsets of the lane-group kernels (split the batch over several handles)");
    SolverBase* s = make_solver(c);
    if (!s) return fail(PDDP_EINVAL, "unsupported plant / integrator / dtype combination (the arm supports Euler only)");
    s->cfg = c;
    int rc = s->init();
    if (rc) { delete s; return rc; }
    *out = new pddp_solver{s};
    return 0;
}
extern "C" int pddp_destroy(pddp_handle h) { if (h) { delete h->impl; delete h; } return 0; }
#define IMPL(h) if (!(h)) return fail(PDDP_EINVAL, "null handle"); SolverBase* s = (h)->impl
extern "C" int pddp_load(pddp_handle h, const void* x0, const void* u0, const void* xg, int clear, int ifd) { IMPL(h); return s->load(x0, u0, xg, nullptr, nullptr, nullptr, nullptr, 0, clear, ifd); }
extern "C" int pddp_load_ex(pddp_handle h, const void* x0, const void* u0, const void* xg, const void* KT0, const void* P0, const void* p0, const void* d0,
                            int rollout, int clear, int ifd) {
    IMPL(h);
    if (!x0 || !u0 || !xg) return fail(PDDP_EINVAL, "pddp_load: null trajectory or goal");
    return s->load(x0, u0, xg, KT0, P0, p0, d0, rollout, clear, ifd);
}
extern "C" int pddp_iterate(pddp_handle h, int sweeps) { IMPL(h); return s->iterate(sweeps); }
extern "C" int pddp_sync(pddp_handle h) { IMPL(h); return s->sync(); }
extern "C" int pddp_status(pddp_handle h, int* done, int* iters) { IMPL(h); return s->status(done, iters); }
extern "C" int pddp_store(pddp_handle h, void* x, void* u, void* KT, void* Jout, int* alphaOut, void* dmax) { IMPL(h); return s->store(x, u, KT, Jout, alphaOut, dmax); }
extern "C" int pddp_time_sweeps(pddp_handle h, int sweeps, float* ms_total, float* ms_phase) { IMPL(h); return s->time_sweeps(sweeps, ms_total, ms_phase); }
extern "C" int pddp_time_kernels(pddp_handle h, int sweeps, float* ms6, char* names, int name_stride) {
    IMPL(h); if (sweeps < 1 || !ms6 || (names && name_stride < 8)) return fail(PDDP_EINVAL, "pddp_time_kernels: bad arguments");
    return s->time_kernels(sweeps, ms6, names, name_stride);
}
extern "C" int pddp_set_cost(pddp_handle h, double Q1, double Q2, double R, double QF1, double QF2) { IMPL(h); return s->set_cost(Q1, Q2, R, QF1, QF2); }
extern "C" int pddp_simulate(pddp_handle h, const void* x, const void* u, const void* KT, double t0_us, double elapsed_us, int substeps, const void* goal_xyz,
                             void* xActual_inout, double* avg_err, int* failed) {
    IMPL(h); if (!x || !u || !KT || !xActual_inout) return fail(PDDP_EINVAL, "null argument");
    return s->simulate(x, u, KT, t0_us, elapsed_us, substeps, goal_xyz, xActual_inout, avg_err, failed);
}
extern "C" int pddp_ee_pos(pddp_handle h, int count, const void* x, void* eePos) { IMPL(h); if (!x || !eePos) return fail(PDDP_EINVAL, "null argument"); return s->ee_pos(count, x, eePos); }
extern "C" int pddp_set_ee_cost_shift(pddp_handle h, int on) { IMPL(h); s->cfg.ee_cost_shift = on ? 1 : 0; return 0; }
extern "C" int pddp_set_cost_ee(pddp_handle h, double Q_EE1, double Q_EE2, double QF_EE1, double QF_EE2, double R_EE, double Q_xEE, double QF_xEE,
                                double Q_xdEE, double QF_xdEE) {
    IMPL(h); const double v[9] = {Q_EE1, Q_EE2, QF_EE1, QF_EE2, R_EE, Q_xEE, QF_xEE, Q_xdEE, QF_xdEE}; return s->set_cost_ee(v);
}
extern "C" int pddp_set_benchmark_mode(pddp_handle h, int on) { IMPL(h); s->bench_mode = on ? 1 : 0; return 0; }
extern "C" int pddp_array_bytes(pddp_handle h, const char* name, size_t* bytes) { IMPL(h); void* p; return s->array(name, &p, bytes); }
extern "C" int pddp_refresh_reference_views(pddp_handle h) { IMPL(h); return s->reference_views(3); }
extern "C" int pddp_array_ptr(pddp_handle h, const char* name, void** ptr, size_t* bytes) { IMPL(h); if (!ptr || !bytes) return fail(PDDP_EINVAL, "null argument"); return s->array(name, ptr, bytes); }
extern "C" int pddp_set_array(pddp_handle h, const char* name, const void* host, size_t bytes) {
    IMPL(h); void* p; size_t cap; int rc = s->array(name, &p, &cap); if (rc) return rc;
    if (bytes > cap) return fail(PDDP_EINVAL, "set_array: too many bytes");
    HIPCHK(hipStreamSynchronize(s->stream));
    if (std::strcmp(name, "AB") == 0 && bytes < cap && (rc = s->ab_view(0))) return rc;      // a partial write lands on the current values
    // the first write into "H" of a handle that keeps the compact end-effector position block: bring the reference-layout array up to date FIRST (a partial write
    // then lands on the current values, and the caller's data is not overwritten by the expansion afterwards)
    if (std::strcmp(name, "H") == 0 && !s->h_overridden && (rc = s->h_view())) return rc;
    // A - B K / B du of a handle whose production sweeps compose maps instead of writing them: materialise both from the last sweep FIRST (the one that is not being
    // written must not stay stale, and the phase hook's rollouts must not rebuild over the caller's data afterwards)
    if ((std::strcmp(name, "ApBK") == 0 || std::strcmp(name, "Bdu") == 0) && s->fs_vars_stale && (rc = s->reference_views(1))) return rc;
    const bool cand = std::strcmp(name, "xs") == 0 || std::strcmp(name, "us") == 0;
    if (cand && s->cand_stale && (rc = s->cand_view(0))) return rc;                          // (the other of the two arrays must hold the records' values before both go back)
    HIPCHK(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    if (cand) return s->cand_view(1);                                                         // handles whose setup adopts from the knot-major records: keep them in step
    if (std::strncmp(name, "model_", 6) == 0) return s->model_changed();
    if (std::strcmp(name, "AB") == 0) return s->ab_view(1);
    if (std::strcmp(name, "H") == 0 && !s->h_overridden) { s->h_overridden = true; s->drop_graph(); return s->ab_keep_reference_layout(); }
    return 0;
}
extern "C" int pddp_get_array(pddp_handle h, const char* name, void* host, size_t bytes) {
    IMPL(h); void* p; size_t cap; int rc = s->array(name, &p, &cap); if (rc) return rc;
    if (bytes > cap) return fail(PDDP_EINVAL, "get_array: too many bytes");
    HIPCHK(hipStreamSynchronize(s->stream));
    if (std::strcmp(name, "AB") == 0 && (rc = s->ab_view(0))) return rc;
    if (std::strcmp(name, "H") == 0 && (rc = s->h_view())) return rc;
    if ((std::strcmp(name, "ApBK") == 0 || std::strcmp(name, "Bdu") == 0) && s->fs_vars_stale && (rc = s->reference_views(1))) return rc;
    if ((std::strcmp(name, "xs") == 0 || std::strcmp(name, "us") == 0) && s->cand_stale && (rc = s->cand_view(0))) return rc;      // (handles whose rollouts keep records)
    HIPCHK(hipMemcpy(host, p, bytes, hipMemcpyDeviceToHost)); return 0;
}
extern "C" int pddp_get_state(pddp_handle h, pddp_state* out) { IMPL(h); return s->get_state(out); }
extern "C" int pddp_set_state(pddp_handle h, const pddp_state* in) { IMPL(h); return s->set_state(in); }
extern "C" int pddp_run_phase(pddp_handle h, int phase) { IMPL(h); return s->run_phase(phase); }
extern "C" int pddp_plant_eval(pddp_handle h, int what, int count, const void* x, const void* u, void* out) { IMPL(h); return s->plant_eval(what, count, x, u, out); }

extern "C" int pddp_hbm_calibration(int device, size_t bytes, int reps) {
    if (bytes < 4096 || reps < 1) return fail(PDDP_EINVAL, "hbm_calibration: bad arguments");
    HIPCHK(hipSetDevice(device));
    float *a = nullptr, *b2 = nullptr;
    HIPCHK(hipMalloc((void**)&a, bytes)); HIPCHK(hipMalloc((void**)&b2, bytes));
    HIPCHK(hipMemset(a, 1, bytes)); HIPCHK(hipMemset(b2, 0, bytes));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_hbm_calib_dword, dim3(256 * 16), dim3(256), 0, 0, a, b2, bytes / sizeof(float));
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    hipFree(a); hipFree(b2);
    return 0;
}

extern "C" int pddp_mpc_solve(pddp_handle h, const void* xActual, const void* xGoal, const int* shift, int clear_vars, int full_rollout, int ifd,
                              int max_iter, double time_budget_ms, int poll_every, void* x, void* u, void* KT, void* Jout, int* alphaOut, int* success, int* iters) {
    IMPL(h);
    if (!xActual || !xGoal || !shift) return fail(PDDP_EINVAL, "pddp_mpc_solve: null argument");
    return s->mpc_solve(xActual, xGoal, shift, clear_vars, full_rollout, ifd, max_iter, time_budget_ms, poll_every, x, u, KT, Jout, alphaOut, success, iters);
}
extern "C" int pddp_get_config(pddp_handle h, pddp_config* out) { IMPL(h); if (!out) return fail(PDDP_EINVAL, "null argument"); *out = s->cfg; return 0; }
extern "C" int pddp_stream(pddp_handle h, void** hip_stream) { IMPL(h); if (!hip_stream) return fail(PDDP_EINVAL, "null argument"); *hip_stream = (void*)s->stream; return 0; }

// runiLQR_GPU (DDPWrappers.cuh:10-138) for the batch.
extern "C" int pddp_solve_ex(pddp_handle h, void* x0, void* u0, const void* xGoal, const void* KT0, const void* P0, const void* p0, const void* d0,
                             void* Jout, int* alphaOut, int rollout, int clear, int ifd, int poll_every, double* times_ms, double* phase_ms,
                             int* sweeps_out) {
    IMPL(h);
    if (!x0 || !u0 || !xGoal) return fail(PDDP_EINVAL, "pddp_solve: null trajectory or goal");
    const double t0 = now_ms();
    int rc = s->load(x0, u0, xGoal, KT0, P0, p0, d0, rollout, clear, ifd);
    if (rc) return rc;
    double t_init = now_ms() - t0;
    std::vector<int> done(s->cfg.batch);
    const int chunk = poll_every > 0 ? poll_every : 8;         // sweeps enqueued between two polls of the exit flags
    const int stride = s->cfg.max_iter + 2;
    int sweeps = 0;
    for (int guard = 0; guard < 1000000; guard++) {
        if ((rc = phase_ms ? s->iterate_traced(chunk, phase_ms, sweeps, stride) : s->iterate(chunk))) return rc;
        sweeps += chunk;
        if ((rc = s->status(done.data(), nullptr))) return rc;
        bool all = true;
        for (int d : done) all &= (d != 0);
        if (all) break;
    }
    const double t1 = now_ms();
    if ((rc = s->store(x0, u0, nullptr, Jout, alphaOut, nullptr))) return rc;
    const double t2 = now_ms();
    if (times_ms) { times_ms[0] = t2 - t0; times_ms[1] = t_init + (t2 - t1); }
    if (sweeps_out) *sweeps_out = sweeps;
    return 0;
}
extern "C" int pddp_solve(pddp_handle h, void* x0, void* u0, const void* xGoal, void* Jout, int* alphaOut, int clear, int ifd, double* times_ms) {
    return pddp_solve_ex(h, x0, u0, xGoal, nullptr, nullptr, nullptr, nullptr, Jout, alphaOut, 0, clear, ifd, 8, times_ms, nullptr, nullptr);
}
