// The part of a handle that the C ABI (pddp_api.hip) sees: no kernels, no plant -- the Solver<Plant, Integrator, T> template behind it (solver_impl.hpp) is instantiated in one
// translation unit per plant (pddp_plant_{pend,cart,quad,arm,user}.hip), so that a change to one family's kernels recompiles one unit and the units build in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <sys/time.h>
#include <vector>

#include "../../include/pddp.h"

#if defined(PDDP_USER_PLANT_HEADER) || defined(PDDP_REF_PLANT_FILE)
#define PDDP_HAS_USER_PLANT 1      /* a `make user` build: plant 5 exists (policy header or reference-form plug-in) */
#endif

int pddp_internal_fail(int code, const std::string& msg);      // sets pddp_last_error (pddp_api.hip); the library's other translation units report through it
static inline int fail(int code, const std::string& msg) { return pddp_internal_fail(code, msg); }
#define HIPCHK(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return fail(PDDP_ENODEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
static double now_ms() { timeval t; gettimeofday(&t, nullptr); return t.tv_sec * 1e3 + t.tv_usec * 1e-3; }

struct SolverBase {
    pddp_config cfg;
    virtual ~SolverBase() {}
    virtual int init() = 0;
    virtual int load(const void* x0, const void* u0, const void* xg, const void* KT0, const void* P0, const void* p0, const void* d0, int rollout, int clear, int ignore_first_defect) = 0;
    virtual int iterate(int sweeps) = 0;
    virtual int sync() = 0;
    virtual int status(int* done, int* iters) = 0;
    virtual int store(void* x, void* u, void* KT, void* Jout, int* alphaOut, void* dmax) = 0;
    virtual int time_sweeps(int sweeps, float* ms_total, float* ms_phase) = 0;
    virtual int time_kernels(int sweeps, float* ms, char* names, int name_stride) = 0;
    virtual int array(const char* name, void** ptr, size_t* bytes) = 0;
    virtual int get_state(pddp_state* out) = 0;
    virtual int set_state(const pddp_state* in) = 0;
    virtual int run_phase(int phase) = 0;
    virtual int plant_eval(int what, int count, const void* x, const void* u, void* out) = 0;
    virtual int model_changed() = 0;
    virtual int iterate_traced(int sweeps, double* phase_ms, int first_sweep, int stride) = 0;
    virtual int simulate(const void* x, const void* u, const void* KT, double t0_us, double elapsed_us, int substeps, const void* goal, void* xActual,
                         double* avg_err, int* failed) = 0;
    virtual int ee_pos(int count, const void* x, void* out) = 0;
    virtual int set_cost(double Q1, double Q2, double R, double QF1, double QF2) = 0;
    virtual int set_cost_ee(const double* v) = 0;
    virtual int mpc_solve(const void* xActual, const void* xGoal, const int* shift, int clear_vars, int full_rollout, int ifd, int max_iter, double budget_ms,
                          int poll_every, void* x, void* u, void* KT, void* Jout, int* alphaOut, int* success, int* iters) = 0;
    int bench_mode = 0;
    bool h_overridden = false;     // pddp_set_array("H"): the cost Hessian is no longer known to be the plant's own (diagonal for the joint-space cost)
    virtual void drop_graph() = 0;
    virtual int ab_view(int to_compact) = 0;       // compact [A B] handles (ab_compact.hpp): refresh the reference-layout array "AB" from the compact one (0) or the reverse (1)
    virtual int h_view() = 0;                      // handles with the compact end-effector Hessian block: refresh the reference-layout array "H"
    virtual int cand_view(int to_records) = 0;     // closed-form handles whose production rollouts keep knot-major records (k_fp_cf): refresh xs / us from them (0) or the reverse (1)
    bool cand_stale = false;                       // production rollouts wrote records since xs / us were last written
    virtual int reference_views(int what) = 0;     // rebuild d_ApBK / d_Bdu (1) and the winner in every step-size slot (2) from the state the last sweep left
    bool fs_vars_stale = false;                    // fused sweeps ran since A - B K / B du were last written (pddp_get_array materialises them first)
    virtual int ab_keep_reference_layout() = 0;   // leave the compact mode for good (the cost Hessian was overridden: the backward pass reads the reference layout then)
    hipStream_t stream = nullptr;
};
struct pddp_solver { SolverBase* impl; };

// one factory per plant unit: a Solver<Plant<T>, integrator, T> for the configuration's element type and integrator, or nullptr
SolverBase* pddp_make_solver_pend(const pddp_config& c);
SolverBase* pddp_make_solver_cart(const pddp_config& c);
SolverBase* pddp_make_solver_quad(const pddp_config& c);
SolverBase* pddp_make_solver_arm(const pddp_config& c);
#ifdef PDDP_HAS_USER_PLANT
SolverBase* pddp_make_solver_user(const pddp_config& c);     // plant 5 of a `make user` build (pddp_plant_user.hip)
int pddp_user_plant_state_size();
int pddp_user_plant_control_size();
#endif
