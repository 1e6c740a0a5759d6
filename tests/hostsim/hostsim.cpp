// TEST TOOL -- NOT PRODUCT CODE.
// libpddp_hostsim.so: the C ABI of include/pddp.h implemented by running the kernel BODIES (parallel-ddp_amd/csrc/
// bodies.hpp and below, the very headers the HIP kernels are compiled from) on the host with a 1-lane "wave".
// It exists so that the arithmetic and indexing of the kernels can be checked against the oracle on a machine
// without a GPU (`pytest -m "not gpu"`).  It cannot detect cross-lane races; the GPU tests do that.
// Only tests/ loads it; the product binding (parallel-ddp_amd/pyddp) refuses to.
#include "hostsim_impl.hpp"

static thread_local std::string g_err;
int hostsim_fail(int code, const std::string& m) { g_err = m; return code; }
extern "C" const char* pddp_last_error(void) { return g_err.c_str(); }
#ifdef PDDP_HOSTSIM_HAS_USER_PLANT
static constexpr int kMaxPlant = 5;
static int user_nx() { return hostsim_user_state_size(); }
static int user_nu() { return hostsim_user_control_size(); }
#else
static constexpr int kMaxPlant = 4;
static int user_nx() { return -1; }
static int user_nu() { return -1; }
#endif
extern "C" int pddp_state_size(int plant) { return plant == 1 ? 2 : plant == 2 ? 4 : plant == 3 ? 12 : plant == 4 ? 14 : plant == 5 ? user_nx() : -1; }
extern "C" int pddp_control_size(int plant) { return plant == 1 ? 1 : plant == 2 ? 1 : plant == 3 ? 4 : plant == 4 ? 7 : plant == 5 ? user_nu() : -1; }
extern "C" int pddp_default_config(pddp_config* c, int plant) {
    std::memset(c, 0, sizeof(*c));
    c->plant = plant; c->N = plant == 4 ? 64 : 128; c->M = 4; c->A = (plant == 3 || plant == 4) ? 16 : 32;   /* plant 5 (user plant): the pendulum's defaults */
    c->integrator = plant == 4 ? 1 : 3; c->batch = 1; c->max_iter = 100; c->ignore_max_rho_exit = 1;
    c->total_time = plant == 4 ? 0.5 : 4.0; c->alpha_base = (plant == 3 || plant == 4) ? 0.5 : 0.75;
    c->rho_init = plant == 4 ? 12.5 : (plant == 3 ? 1.0 : 10.0); c->max_defect = plant == 2 ? 0.75 : 1.0;
    c->tol_cost = 0.0001; c->exp_red_min = 0.05; c->exp_red_max = 1.25;
    c->Q1 = 0.1; c->Q2 = 0.001; c->R = 0.0001; c->QF1 = 1000.0; c->QF2 = 1000.0;
    c->Q_EE1 = 0.1; c->Q_EE2 = 0.0; c->QF_EE1 = 1000.0; c->QF_EE2 = 0.0; c->R_EE = 0.0001; c->Q_xEE = 0.0; c->QF_xEE = 0.0; c->Q_xdEE = 0.1; c->QF_xdEE = 1000.0;
    c->ee_on_link_z = 0.0635;   // plants/cost_arm.cuh:104-115, dynamics_arm.cuh:57-58 (EE_TYPE 1)
    c->use_finite_diff = 0; c->finite_diff_epsilon = 0.00001;
    c->use_limits = 0; c->use_smooth_abs = 0; c->smooth_abs_alpha = 0.2;
    c->ee_type = 1;
    return 0;
}

static Base* mk(const pddp_config& c) {
    switch (c.plant) {
    case 1: return hostsim_make_pend(c);
    case 2: return hostsim_make_cart(c);
    case 3: return hostsim_make_quad(c);
    case 4: return hostsim_make_arm(c);
#ifdef PDDP_HOSTSIM_HAS_USER_PLANT
    case 5: return hostsim_make_user(c);
#endif
    }
    return nullptr;
}
extern "C" int pddp_create(const pddp_config* cfg, pddp_handle* out) {
    const pddp_config& c = *cfg;
    if (c.plant < 1 || c.plant > kMaxPlant) return fail(PDDP_EINVAL, "plant must be 1..4 (5: the user plant of a `make user PLANT_POLICY=...` build)");
    if (c.N < 4 || (c.N & (c.N - 1)) || c.N > 1024) return fail(PDDP_EINVAL, "N must be a power of two in [4,1024]");
    if (c.M < 1 || c.N % c.M || c.N / c.M < 2 || c.M > 16) return fail(PDDP_EINVAL, "M must divide N, N/M >= 2, M <= 16");
    if (c.A < 1 || c.A > 64 || c.batch < 1 || c.max_iter < 1) return fail(PDDP_EINVAL, "A in [1,64], batch >= 1, max_iter >= 1");
    Base* s = mk(c);
    if (!s) return fail(PDDP_EINVAL, "unsupported plant / integrator / dtype combination");
#ifdef PDDP_HOSTSIM_HAS_USER_PLANT
    if (c.plant == 5) { const std::string complaint = hostsim_user_setup(c); if (!complaint.empty()) { delete s; return fail(PDDP_EINVAL, complaint); } }
#endif
    s->bp_default_coop = (c.kernels.bp == 3);     // kernels.bp = coop: same override as the library
    *out = new pddp_solver{s};
    return 0;
}
extern "C" int pddp_destroy(pddp_handle h) { if (h) { delete h->impl; delete h; } return 0; }
extern "C" int pddp_load(pddp_handle h, const void* x0, const void* u0, const void* xg, int clear, int ifd) { return h->impl->load(x0, u0, xg, nullptr, nullptr, nullptr, nullptr, 0, clear, ifd); }
extern "C" int pddp_load_ex(pddp_handle h, const void* x0, const void* u0, const void* xg, const void* KT0, const void* P0, const void* p0, const void* d0, int rollout, int clear, int ifd) {
    return h->impl->load(x0, u0, xg, KT0, P0, p0, d0, rollout, clear, ifd);
}
extern "C" int pddp_hbm_calibration(int, size_t, int) { return fail(PDDP_ENODEVICE, "host emulation has no HBM"); }
extern "C" int pddp_iterate(pddp_handle h, int sweeps) { return h->impl->iterate(sweeps); }
extern "C" int pddp_sync(pddp_handle) { return 0; }
extern "C" int pddp_status(pddp_handle h, int* done, int* iters) { return h->impl->status(done, iters); }
extern "C" int pddp_store(pddp_handle h, void* x, void* u, void* KT, void* Jout, int* alphaOut, void* dmax) { return h->impl->store(x, u, KT, Jout, alphaOut, dmax); }
// multi-GPU exchanges: the host emulation is a single "rank" without a device; the entry points exist so that the symbol check passes
extern "C" int pddp_comm_unique_id(void*) { return fail(PDDP_ENODEVICE, "host emulation: no RCCL"); }
extern "C" int pddp_comm_init(pddp_comm_handle*, int, int, const void*, int) { return fail(PDDP_ENODEVICE, "host emulation: no RCCL"); }
extern "C" int pddp_comm_destroy(pddp_comm_handle) { return 0; }
extern "C" int pddp_comm_ranks(pddp_comm_handle, int*, int*) { return fail(PDDP_ENODEVICE, "host emulation: no RCCL"); }
extern "C" int pddp_comm_all_done(pddp_comm_handle, pddp_handle, int*) { return fail(PDDP_ENODEVICE, "host emulation: no RCCL"); }
extern "C" int pddp_comm_allgather_costs(pddp_comm_handle, pddp_handle, double*) { return fail(PDDP_ENODEVICE, "host emulation: no RCCL"); }
extern "C" int pddp_comm_allreduce_max(pddp_comm_handle, double*) { return fail(PDDP_ENODEVICE, "host emulation: no RCCL"); }
extern "C" int pddp_get_config(pddp_handle h, pddp_config* out) { *out = h->impl->cfg; return 0; }
extern "C" int pddp_time_kernels(pddp_handle h, int sweeps, float* ms6, char* names, int stride) { h->impl->iterate(sweeps); for (int i = 0; i < 6; i++) { ms6[i] = 0; if (names) names[(size_t)i * stride] = 0; } return 0; }
extern "C" int pddp_time_sweeps(pddp_handle h, int sweeps, float* t, float* ph) { h->impl->iterate(sweeps); if (t) *t = 0; if (ph) for (int i = 0; i < 4; i++) ph[i] = 0; return 0; }
extern "C" int pddp_set_benchmark_mode(pddp_handle h, int on) { h->impl->bench = on ? 1 : 0; return 0; }
extern "C" int pddp_array_bytes(pddp_handle h, const char* name, size_t* bytes) { void* p; return h->impl->array(name, &p, bytes); }
extern "C" int pddp_array_ptr(pddp_handle h, const char* name, void** ptr, size_t* bytes) { return h->impl->array(name, ptr, bytes); }
extern "C" int pddp_set_array(pddp_handle h, const char* name, const void* host, size_t bytes) {
    void* p; size_t cap; int rc = h->impl->array(name, &p, &cap); if (rc) return rc;
    if (bytes > cap) return fail(PDDP_EINVAL, "too many bytes"); std::memcpy(p, host, bytes); return 0;
}
extern "C" int pddp_get_array(pddp_handle h, const char* name, void* host, size_t bytes) {
    void* p; size_t cap; int rc = h->impl->array(name, &p, &cap); if (rc) return rc;
    if (bytes > cap) return fail(PDDP_EINVAL, "too many bytes"); std::memcpy(host, p, bytes); return 0;
}
extern "C" int pddp_get_state(pddp_handle h, pddp_state* out) { return h->impl->get_state(out); }
extern "C" int pddp_set_state(pddp_handle h, const pddp_state* in) { return h->impl->set_state(in); }
extern "C" int pddp_run_phase(pddp_handle h, int phase) { return h->impl->run_phase(phase); }
extern "C" int pddp_plant_eval(pddp_handle h, int what, int count, const void* x, const void* u, void* out) { return h->impl->plant_eval(what, count, x, u, out); }
extern "C" int pddp_solve_ex(pddp_handle h, void* x0, void* u0, const void* xGoal, const void* KT0, const void* P0, const void* p0, const void* d0,
                             void* Jout, int* alphaOut, int rollout, int clear, int ifd, int, double* times_ms, double* phase_ms, int* sweeps_out) {
    Base* s = h->impl; s->load(x0, u0, xGoal, KT0, P0, p0, d0, rollout, clear, ifd);
    std::vector<int> done(s->cfg.batch);
    int sweeps = 0;
    for (int guard = 0; guard < 100000; guard++) {
        s->iterate(1); sweeps++; s->status(done.data(), nullptr);
        bool all = true; for (int d : done) all &= (d != 0);
        if (all) break;
    }
    s->store(x0, u0, nullptr, Jout, alphaOut, nullptr);
    if (times_ms) { times_ms[0] = 0; times_ms[1] = 0; }
    if (phase_ms) std::memset(phase_ms, 0, sizeof(double) * 5 * (s->cfg.max_iter + 2));      // [5][max_iter + 2] (include/pddp.h)
    if (sweeps_out) *sweeps_out = sweeps;
    return 0;
}
extern "C" int pddp_solve(pddp_handle h, void* x0, void* u0, const void* xGoal, void* Jout, int* alphaOut, int clear, int ifd, double* times_ms) {
    return pddp_solve_ex(h, x0, u0, xGoal, nullptr, nullptr, nullptr, nullptr, Jout, alphaOut, 0, clear, ifd, 1, times_ms, nullptr, nullptr);
}
extern "C" int pddp_mpc_solve(pddp_handle h, const void* xActual, const void* xGoal, const int* shift, int clear_vars, int full_rollout, int ifd,
                              int max_iter, double, int, void* x, void* u, void* KT, void* Jout, int* alphaOut, int* success, int* iters) {
    return h->impl->mpc_solve(xActual, xGoal, shift, clear_vars, full_rollout, ifd, max_iter, x, u, KT, Jout, alphaOut, success, iters);
}
extern "C" int pddp_simulate(pddp_handle h, const void* x, const void* u, const void* KT, double t0_us, double elapsed_us, int substeps, const void* goal, void* xa,
                             double* avg_err, int* failed) { return h->impl->simulate(x, u, KT, t0_us, elapsed_us, substeps, goal, xa, avg_err, failed); }
extern "C" int pddp_ee_pos(pddp_handle h, int count, const void* x, void* out) { return h->impl->ee_pos(count, x, out); }
extern "C" int pddp_stream(pddp_handle, void** st) { if (st) *st = nullptr; return 0; }
