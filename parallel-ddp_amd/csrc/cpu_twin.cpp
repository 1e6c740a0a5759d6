// libpddp_cpu.so: the reference's CPU iLQR path (runiLQR_CPU, DDPHelpers/DDPWrappers.cuh:142-248) -- see include/pddp_cpu.h.
// Host C++ only (g++); the numerical bodies are the headers the HIP kernels are compiled from, instantiated with a 1-lane "wave".
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <sys/time.h>
#include <thread>
#include <vector>

#include "../../include/pddp_cpu.h"
#include "bp.hpp"
#include "fp.hpp"
#include "nis.hpp"
#include "solver_state.hpp"
#include "iiwa14_model_data.h"

using namespace pddp;

static thread_local std::string g_cpu_err;
static int cpu_fail(int code, const std::string& m) { g_cpu_err = m; return code; }
extern "C" const char* pddp_cpu_last_error(void) { return g_cpu_err.c_str(); }

static double wall_ms() { timeval t; gettimeofday(&t, nullptr); return t.tv_sec * 1e3 + t.tv_usec * 1e-3; }

extern "C" int pddp_cpu_thread_counts(int M, int cores, int* bp, int* fsim, int* cost, int* integ) {
    if (cores <= 0) cores = (int)std::thread::hardware_concurrency();
    if (cores <= 0) cores = 1;
    if (bp) *bp = std::max(std::min(M, cores), 1);          // config.cuh:158-159 (USE_HYPER_THREADING 0)
    if (fsim) *fsim = std::max(std::min(M, cores), 1);
    if (cost) *cost = std::max(cores / 2, 1);               // :156-157
    if (integ) *integ = std::max(cores / 2, 1);
    return 0;
}

namespace {

template <typename T> void fill_model(ArmModel<T>& m, const pddp_config& c) {
    const int v = c.wafr_urdf ? 1 : 0;
    for (int b = 0; b < 7; b++) {
        for (int i = 0; i < 36; i++) m.I[36 * b + i] = (T)IIWA14_SPATIAL_INERTIA[v][b][i];
        for (int i = 0; i < 16; i++) m.F[16 * b + i] = (T)IIWA14_JOINT_FRAME[v][b][i];
    }
    m.grav = (T)(c.mpc_mode ? 0.0 : 9.81);                  // plants/dynamics_arm.cuh:42-46
    arm_model_apply_ee_type(m, c.wafr_urdf, c.ee_type);
}
void fill_model(EmptyModel& m, const pddp_config&) { m.unused = 0; }

// threads of one phase: the reference creates them per phase and joins them (e.g. fpHelpers.cuh:425-441)
struct Phase {
    std::vector<std::thread> th;
    void go(std::function<void()> f) { th.emplace_back(std::move(f)); }
    void join() { for (auto& t : th) t.join(); th.clear(); }
};
// compute_reps (utils/threadUtils.h:19): thread tid of dim handles items tid, tid + dim, ...
inline int reps_of(int tid, int dim, int total) { return total / dim + (tid < total % dim ? 1 : 0); }

template <typename P, int INTEG, typename T>
int run_cpu(const pddp_config& c, const pddp_cpu_buffers& B, T* x0, T* u0, const T* KT0, const T* P0, const T* p0, const T* d0, const T* xGoal, T* Jout,
            int* alphaOut, int rollout, int clearVars, int ignoreFirstDefect, double* tTime, double* simTime, double* sweepTime, double* bpTime,
            double* nisTime, double* initTime, int cores, int* iters_out, int parallel_ls) {
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU;
    const int N = c.N, M = c.M, A = c.A;
    Dims dm; dm.N = N; dm.M = M; dm.A = A; dm.NB = N / M;
    CostWeights<T> cw{}; cw.Q1 = (T)c.Q1; cw.Q2 = (T)c.Q2; cw.R = (T)c.R; cw.QF1 = (T)c.QF1; cw.QF2 = (T)c.QF2; cw.ee = 0; cw.limits = c.use_limits;
    cw.fd_eps = c.use_finite_diff ? c.finite_diff_epsilon : 0.0;      // USE_FINITE_DIFF: integratorGradientThreaded's other definition (nisInitHelpers.cuh:185-201)
    const T dt = (T)(c.total_time / (N - 1));                // TIME_STEP, config.cuh:136
    typename P::Model model; fill_model(model, c);
#ifdef PDDP_REF_PLANT_FILE
    if constexpr (P::PLANT == 5) { const std::string complaint = ref_plugin_setup<T>(N); if (!complaint.empty()) return cpu_fail(PDDP_EINVAL, complaint); }
#endif
    int BP_T, FSIM_T, COST_T, INT_T;
    pddp_cpu_thread_counts(M, cores, &BP_T, &FSIM_T, &COST_T, &INT_T);
    int cores_eff = cores > 0 ? cores : (int)std::thread::hardware_concurrency();
    if (cores_eff <= 0) cores_eff = 1;
    T *x = (T*)B.x, *xp = (T*)B.xp, *xp2 = (T*)B.xp2, *u = (T*)B.u, *up = (T*)B.up, *Pm = (T*)B.P, *pv = (T*)B.p, *Pp = (T*)B.Pp, *pp = (T*)B.pp;
    T *AB = (T*)B.AB, *H = (T*)B.H, *g = (T*)B.g, *KT = (T*)B.KT, *du = (T*)B.du, *d = (T*)B.d, *dp = (T*)B.dp, *ApBK = (T*)B.ApBK, *Bdu = (T*)B.Bdu;
    T *alpha = (T*)B.alpha, *JT = (T*)B.JT, *dJexp = (T*)B.dJexp; int* err = B.err;
    const size_t szx = (size_t)NX * N, szu = (size_t)NU * N, szP = (size_t)NX * NX * N;
    const Wave w = this_wave();
    Phase ph;
    const double t_start = wall_ms();
    double t2 = t_start;

    // ---- phase bodies (each thread owns private scratch; the global arrays are the caller's)
    auto sim_segments = [&](int tid, T al, T* x, T* u, T* d) {                 // forwardSim (fpHelpers.cuh:305-328): segments tid, tid + FSIM_T, ...
        SimScratch<P, T> s; std::vector<T> segx((size_t)NX * M), dnorm(M);
        P::load_model(w, s.plant, &model);
        FpArgs<T> a{}; a.x = x; a.u = u; a.d = d; a.xcur = xp; a.ucur = up; a.dcur = dp; a.KT = KT; a.du = du; a.ApBK = ApBK; a.Bdu = Bdu; a.alpha = al; a.dt = dt;
        a.segx = segx.data(); a.dnorm = dnorm.data();
        // the sweep left the start states of segments 1 .. M - 1 in x.  Segment 0 starts from the current trajectory (forward_sim_segment never reads segx[0..NX)) and its
        // thread WRITES x[0..NX) while the others copy: reading knot 0 here was a data race (ThreadSanitizer, round 6: profiles/r06_sanitizers.log) -- on a value nobody used
        for (int b = 1; b < M; b++) for (int i = 0; i < NX; i++) segx[(size_t)NX * b + i] = x[(size_t)NX * b * dm.NB + i];
        const int r = reps_of(tid, FSIM_T, M);
        for (int i = 0; i < r; i++) forward_sim_segment<P, INTEG, T>(w, s, dm, a, tid + i * FSIM_T, cw, xGoal, nullptr);
    };
    auto cost_part = [&](int tid, const T* x, const T* u, T* JT) {                          // costThreaded (fpHelpers.cuh:156-163): knots tid, tid + COST_T, ...
        T acc = 0;
        const int r = reps_of(tid, COST_T, N);
        for (int i = 0; i < r; i++) { const int k = tid + i * COST_T; acc += P::cost(cw, x + (size_t)k * NX, u + (size_t)k * NU, xGoal, k, N); }
        JT[tid] = acc;
    };
    auto derivs = [&](int tid, int dim, bool cost_part_, bool dyn_part, const T* x, const T* u) {   // costGradientHessianThreaded / integratorGradientThreaded (nisInitHelpers.cuh:97-136, 225-242)
        NisScratch<P, INTEG, T> s;
        P::load_model(w, s.plant, &model);
        const int total = cost_part_ ? N : N - 1, r = reps_of(tid, dim, total);
        for (int i = 0; i < r; i++) {
            const int k = tid + i * dim;
            const T* xk = x + (size_t)k * NX; const T* uk = u + (size_t)k * NU;
            if (cost_part_) {
                T* Hk = H + (size_t)k * NM * NM; T* gk = g + (size_t)k * NM;
                if constexpr (P::kPluginCost) P::cost_grad(cw, Hk, gk, xk, uk, xGoal, k, N);      // a cost file in the reference's form (ref_plugin.hpp): the user's costGrad
                else {
                for (int e = 0; e < NM * NM; e++) { const int ii = e / NM, jj = e % NM; Hk[e] = (ii == jj) ? P::weight(cw, ii, k, N) : T(0); }
                for (int ii = 0; ii < NM; ii++) gk[ii] = P::weight(cw, ii, k, N) * (ii < NX ? (xk[ii] - xGoal[ii]) : uk[ii - NX]);
                }
                if constexpr (P::PLANT == 4) { if (cw.limits) { const int nl = (k == N - 1) ? NX : NM; for (int ii = 0; ii < nl; ii++) gk[ii] += arm_limit_term<T>(xk, uk, ii, 1); } }
            }
            if (dyn_part) {
                for (int ii = 0; ii < NX; ii++) s.x[ii] = xk[ii];
                for (int ii = 0; ii < NU; ii++) s.u[ii] = uk[ii];
                if (INTEG == 1 && cw.fd_eps > 0.0) integrator_gradient_fd<P, T>(w, s.plant, s.fd, AB + (size_t)k * NX * NM, s.x, s.u, dt, cw.fd_eps);
                else integrator_gradient<P, INTEG>(w, s.plant, s.pgrad, s.integ, AB + (size_t)k * NX * NM, s.x, s.u, dt);
            }
        }
    };
    auto back_blocks = [&](int tid, T rho, T* x, T* d) {                 // backPassThreaded (bpHelpers.cuh:424-481): blocks tid, tid + BP_T, ...
        BpScratch<P, T> s;
        BpArgs<T> a{}; a.AB = AB; a.Pm = Pm; a.pv = pv; a.Pp = Pp; a.pp = pp; a.H = H; a.g = g; a.KT = KT; a.du = du; a.dcur = d; a.ApBK = ApBK; a.Bdu = Bdu;
        // the expected reduction is per THREAD in the reference (dJexp[2 tid] is zeroed once and collects every block the thread owns, bpHelpers.cuh:431,467); bp_block
        // leaves one pair per block, added up here in the thread's block order (with fewer threads than blocks -- cores < M -- a thread owns several)
        std::vector<T> dj(2 * (size_t)M, T(0));
        a.xcur = x; a.xprev2 = xp2; a.dJexp = dj.data(); a.rho = rho; a.Hrw = H; a.grw = g;       // CPU path: H, g accumulate in place (bpHelpers.cuh:90-91)
        std::vector<int> e(M, 0); a.err = e.data();
        int any = 0;
        const int r = reps_of(tid, BP_T, M);
        T d0 = 0, d1 = 0;
        for (int i = 0; i < r; i++) { const int blk = tid + i * BP_T; any |= bp_block<P, T>(w, s, dm, blk, a); d0 += dj[2 * blk]; d1 += dj[2 * blk + 1]; }
        dJexp[2 * tid] = d0; dJexp[2 * tid + 1] = d1;
        err[tid] = any;
    };

    // ---- loadVarsCPU (nisInitHelpers.cuh:656-736)
    std::memcpy(x, x0, szx * sizeof(T)); std::memcpy(u, u0, szu * sizeof(T)); std::memcpy(xp, x0, szx * sizeof(T)); std::memcpy(up, u0, szu * sizeof(T));
    if (clearVars) {
        std::memset(Pm, 0, szP * sizeof(T)); std::memset(Pp, 0, szP * sizeof(T)); std::memset(pv, 0, szx * sizeof(T)); std::memset(pp, 0, szx * sizeof(T));
        std::memset(KT, 0, (size_t)NX * NU * N * sizeof(T)); std::memset(d, 0, szx * sizeof(T));
    } else {
        if (!P0 || !p0 || !KT0 || !d0) return cpu_fail(PDDP_EINVAL, "runiLQR_CPU: clearVarsFlag = 0 needs KT0, P0, p0, d0");
        std::memcpy(Pm, P0, szP * sizeof(T)); std::memcpy(Pp, P0, szP * sizeof(T)); std::memcpy(pv, p0, szx * sizeof(T)); std::memcpy(pp, p0, szx * sizeof(T));
        std::memcpy(KT, KT0, (size_t)NX * NU * N * sizeof(T)); std::memcpy(d, d0, szx * sizeof(T));
    }
    std::memset(du, 0, szu * sizeof(T));
    for (int i = 0; i < BP_T; i++) err[i] = 0;
    std::memset(AB + (size_t)NX * NM * (N - 2), 0, (size_t)NX * NM * sizeof(T));
    std::memcpy(dp, d, szx * sizeof(T));                     // (the rollout reads the defects through dcur; backwardPassCPU copies d -> dp again)
    if (rollout) { for (int t = 0; t < FSIM_T; t++) ph.go([&, t] { sim_segments(t, alpha[0], x, u, d); }); ph.join(); }
    // ---- initAlgCPU (:401-457)
    alphaOut[0] = rollout ? 0 : -1;
    for (int t = 0; t < COST_T; t++) ph.go([&, t] { cost_part(t, x, u, JT); });
    for (int t = 0; t < COST_T; t++) ph.go([&, t] { derivs(t, COST_T, true, false, x, u); });
    for (int t = 0; t < INT_T; t++) ph.go([&, t] { derivs(t, INT_T, false, true, x, u); });
    ph.go([&] { std::memcpy(xp, x, szx * sizeof(T)); }); ph.go([&] { std::memcpy(xp2, x, szx * sizeof(T)); }); ph.go([&] { std::memcpy(up, u, szu * sizeof(T)); });
    ph.join();
    T prevJ = 0;
    for (int t = 0; t < COST_T; t++) prevJ += JT[t];
    Jout[0] = prevJ; prevJ *= (T)(1 + 2 * c.tol_cost);        // :456 (the GPU path ADDS 2 TOL_COST, :393)
    T dJ = 0, J = 0, z = 0, maxd = 0; int iter = 1; T rho = (T)c.rho_init, drho = (T)1.0; int alphaIndex = 0;
    (void)z;
    *initTime = wall_ms() - t2;

    if (parallel_ls) {
        // ================================================================== runiLQR_CPU2 (DDPWrappers.cuh:252-363): the parallel line search.
        // Every candidate owns a trajectory slot; a chunk of FSIM_ALPHA_THREADS = max(CPU_CORES / M_BLOCKS_F, 1) step sizes (config.cuh:160) is swept, rolled out and
        // costed concurrently (forwardSweep2, forwardSimCPU2: fpHelpers.cuh:81-92, 489-571), the best acceptable candidate of the chunk wins (ALPHA_BEST_SWITCH 1),
        // otherwise the next chunk is tried while its start is below NUM_ALPHA - FSIM_ALPHA_THREADS.  As the reference has it: no cost-tolerance exit
        // (acceptRejectTrajCPU2, nisInitHelpers.cuh:586), a rejected iteration restores every slot, an accepted one copies the winner into every other slot
        // after the derivatives (nextIterationSetupCPU2 :329-351); backward pass and final store read the winner's slot.
        T** xs = (T**)B.xs; T** us = (T**)B.us; T** ds = (T**)B.ds; T** JTs = (T**)B.JTs;
        const int FAT = std::max(cores_eff / M, 1);
        for (int a = 0; a < A; a++) { std::memcpy(xs[a], x, szx * sizeof(T)); std::memcpy(us[a], u, szu * sizeof(T)); if (M > 1) std::memcpy(ds[a], d, szx * sizeof(T)); }   // initAlgCPU2 :471-484
        *initTime = wall_ms() - t_start;
        std::vector<T> Js(A);
        while (true) {
            t2 = wall_ms();
            while (true) {
                for (int t = 0; t < BP_T; t++) ph.go([&, t] { back_blocks(t, rho, xs[alphaIndex], ds[alphaIndex]); });
                if (M > 1) ph.go([&] { std::memcpy(dp, ds[alphaIndex], szx * sizeof(T)); });
                ph.join();
                int fail = 0;
                for (int t = 0; t < BP_T; t++) fail |= err[t];
                if (!fail) break;
                drho = std::max(drho * (T)kRhoFactor, (T)kRhoFactor); rho = std::min(rho * drho, (T)kRhoMax);
                if (rho == (T)kRhoMax && !c.ignore_max_rho_exit) break;
                std::memcpy(Pm, Pp, szP * sizeof(T)); std::memcpy(pv, pp, szx * sizeof(T));
            }
            bpTime[iter - 1] = wall_ms() - t2;
            dJ = (T)-1.0; alphaIndex = 0; sweepTime[iter - 1] = 0; simTime[iter - 1] = 0;
            while (true) {
                const int start = alphaIndex, cnt = std::min(FAT, A - start);
                t2 = wall_ms();
                if (M > 1) {
                    for (int ai = 0; ai < cnt; ai++) ph.go([&, ai] {
                        const int a = start + ai;
                        SweepScratch<P, T> ss; std::vector<T> segx((size_t)NX * M);
                        FpArgs<T> fa{}; fa.x = xs[a]; fa.xcur = xp; fa.dcur = ds[a]; fa.ApBK = ApBK; fa.Bdu = Bdu; fa.alpha = alpha[a]; fa.segx = segx.data();
                        forward_sweep<P, T>(w, ss, dm, fa);
                    });
                    ph.join();
                }
                sweepTime[iter - 1] += wall_ms() - t2;
                t2 = wall_ms();
                for (int ai = 0; ai < cnt; ai++) for (int t = 0; t < FSIM_T; t++) ph.go([&, ai, t] { const int a = start + ai; sim_segments(t, alpha[a], xs[a], us[a], ds[a]); });
                std::thread cpy;
                if (start == 0) {
                    cpy = std::thread([&] { std::memcpy(xp2, xp, szx * sizeof(T)); });
                    for (int i = 1; i < BP_T; i++) { dJexp[0] += dJexp[2 * i]; dJexp[1] += dJexp[2 * i + 1]; }
                }
                ph.join();
                for (int ai = 0; ai < cnt; ai++) for (int t = 0; t < COST_T; t++) ph.go([&, ai, t] { const int a = start + ai; cost_part(t, xs[a], us[a], JTs[a]); });
                ph.join();
                for (int ai = 0; ai < cnt; ai++) { Js[ai] = 0; for (int t = 0; t < COST_T; t++) Js[ai] += JTs[start + ai][t]; }
                if (start == 0) cpy.join();
                int found = -1;
                for (int ai = 0; ai < cnt; ai++) {
                    const int a = start + ai; const T al = alpha[a];
                    const T cdJ = prevJ - Js[ai]; const bool JFlag = cdJ >= (T)0 && cdJ > dJ;
                    const T cz = cdJ / (al * dJexp[0] + (T)0.5 * al * al * dJexp[1]);
                    const bool zFlag = (T)c.exp_red_min < cz && cz < (T)c.exp_red_max;
                    const T cd = 0;                                   // defectComp never updates its maximum (fpHelpers.cuh:123)
                    const bool dFlag = (M > 1) ? cd < (T)c.max_defect : true;
                    if (JFlag && zFlag && dFlag) {
                        if (ignoreFirstDefect && cd < (T)c.max_defect) ignoreFirstDefect = 0;
                        found = a; dJ = cdJ; z = cz; J = Js[ai]; maxd = cd;
                    }
                }
                simTime[iter - 1] += wall_ms() - t2;
                if (found == -1) { if (alphaIndex < A - FAT) { alphaIndex += FAT; continue; } alphaIndex = -1; break; }
                alphaIndex = found; break;
            }
            t2 = wall_ms();
            bool exit_now = false;
            if (alphaIndex == -1) {
                drho = std::max(drho * (T)kRhoFactor, (T)kRhoFactor); rho = std::min(rho * drho, (T)kRhoMax);
                alphaOut[iter] = -1; Jout[iter] = prevJ;
                for (int a = 0; a < A; a++) { std::memcpy(xs[a], xp, szx * sizeof(T)); std::memcpy(us[a], up, szu * sizeof(T)); if (M > 1) std::memcpy(ds[a], dp, szx * sizeof(T)); }
                if (rho == (T)kRhoMax && !c.ignore_max_rho_exit) exit_now = true;
            } else {
                drho = std::min(drho / (T)kRhoFactor, (T)(1.0 / kRhoFactor)); rho = std::max(rho * drho, (T)kRhoMin);
                dJ = dJ / prevJ; prevJ = J; alphaOut[iter] = alphaIndex; Jout[iter] = J;
            }
            if (!exit_now) { if (iter == c.max_iter) exit_now = true; else iter += 1; }
            if (exit_now) { nisTime[iter - 1] = wall_ms() - t2; if (alphaIndex == -1) alphaIndex = 0; break; }
            int flag = 1;
            if (alphaIndex == -1) { alphaIndex = 0; flag = 0; }
            {
                T* xw = xs[alphaIndex]; T* uw = us[alphaIndex];
                for (int t = 0; t < COST_T; t++) ph.go([&, t] { derivs(t, COST_T, true, false, xw, uw); });
                for (int t = 0; t < INT_T; t++) ph.go([&, t] { derivs(t, INT_T, false, true, xw, uw); });
                ph.go([&] { std::memcpy(Pp, Pm, szP * sizeof(T)); }); ph.go([&] { std::memcpy(pp, pv, szx * sizeof(T)); });
                ph.go([&] { std::memcpy(xp, xw, szx * sizeof(T)); }); ph.go([&] { std::memcpy(up, uw, szu * sizeof(T)); });
                ph.join();
            }
            if (flag) for (int a = 0; a < A; a++) if (a != alphaIndex) {
                std::memcpy(xs[a], xs[alphaIndex], szx * sizeof(T)); std::memcpy(us[a], us[alphaIndex], szu * sizeof(T)); if (M > 1) std::memcpy(ds[a], ds[alphaIndex], szx * sizeof(T));
            }
            nisTime[iter - 2] = wall_ms() - t2;
        }
        t2 = wall_ms();
        std::memcpy(x0, xs[alphaIndex], szx * sizeof(T)); std::memcpy(u0, us[alphaIndex], szu * sizeof(T));
        const double t_end2 = wall_ms();
        *initTime += t_end2 - t2;
        *tTime = t_end2 - t_start;
        if (iters_out) *iters_out = iter;
        return 0;
    }
    while (true) {
        // ---- backwardPassCPU (bpHelpers.cuh:522-566); its "rho maxed out" return value is ignored by the caller (DDPWrappers.cuh:177), as here
        t2 = wall_ms();
        while (true) {
            for (int t = 0; t < BP_T; t++) ph.go([&, t] { back_blocks(t, rho, x, d); });
            if (M > 1) ph.go([&] { std::memcpy(dp, d, szx * sizeof(T)); });
            ph.join();
            int fail = 0;
            for (int t = 0; t < BP_T; t++) fail |= err[t];
            if (!fail) break;
            drho = std::max(drho * (T)kRhoFactor, (T)kRhoFactor); rho = std::min(rho * drho, (T)kRhoMax);
            if (rho == (T)kRhoMax && !c.ignore_max_rho_exit) break;
            std::memcpy(Pm, Pp, szP * sizeof(T)); std::memcpy(pv, pp, szx * sizeof(T));
        }
        bpTime[iter - 1] = wall_ms() - t2;
        // ---- serial line search (DDPWrappers.cuh:187-207) over forwardSimCPU (fpHelpers.cuh:413-486)
        dJ = (T)-1.0; alphaIndex = 0; sweepTime[iter - 1] = 0; simTime[iter - 1] = 0;
        while (true) {
            const T al = alpha[alphaIndex];
            t2 = wall_ms();
            if (M > 1) {                                     // forwardSweep: the segment start states of this candidate, into x
                SweepScratch<P, T> ss; std::vector<T> segx((size_t)NX * M);
                FpArgs<T> a{}; a.x = x; a.xcur = xp; a.dcur = d; a.ApBK = ApBK; a.Bdu = Bdu; a.alpha = al; a.segx = segx.data();
                forward_sweep<P, T>(w, ss, dm, a);
            }
            sweepTime[iter - 1] += wall_ms() - t2;
            t2 = wall_ms();
            for (int t = 0; t < FSIM_T; t++) ph.go([&, t] { sim_segments(t, al, x, u, d); });
            const bool first = (al == (T)1);
            std::thread cpy;
            if (first) {                                     // xp2 <- xp while the rollouts run; the expected reduction is summed over the backward-pass threads (sic, :438)
                cpy = std::thread([&] { std::memcpy(xp2, xp, szx * sizeof(T)); });
                for (int i = 1; i < BP_T; i++) { dJexp[0] += dJexp[2 * i]; dJexp[1] += dJexp[2 * i + 1]; }
            }
            ph.join();
            for (int t = 0; t < COST_T; t++) ph.go([&, t] { cost_part(t, x, u, JT); });
            ph.join();
            J = 0;
            for (int t = 0; t < COST_T; t++) J += JT[t];
            if (first) cpy.join();
            dJ = prevJ - J;
            const bool JFlag = dJ >= (T)0;
            z = dJ / (al * dJexp[0] + (T)0.5 * al * al * dJexp[1]);
            const bool zFlag = (T)c.exp_red_min < z && z < (T)c.exp_red_max;
            bool dFlag = true;
            if (M > 1) { maxd = 0; dFlag = maxd < (T)c.max_defect; }   // defectComp never updates its maximum (fpHelpers.cuh:123): always 0 on this path
            bool failed;
            if (JFlag && zFlag && dFlag) { if (ignoreFirstDefect && maxd < (T)c.max_defect) ignoreFirstDefect = 0; failed = false; }
            else {
                std::memcpy(x, xp, szx * sizeof(T)); std::memcpy(u, up, szu * sizeof(T)); if (M > 1) std::memcpy(d, dp, szx * sizeof(T));
                failed = true;
            }
            simTime[iter - 1] += wall_ms() - t2;
            if (failed) { if (alphaIndex < A - 1) { alphaIndex++; continue; } alphaIndex = -1; }
            break;
        }
        // ---- acceptRejectTrajCPU (nisInitHelpers.cuh:520-555)
        t2 = wall_ms();
        bool exit_now = false;
        if (alphaIndex == -1) {
            drho = std::max(drho * (T)kRhoFactor, (T)kRhoFactor); rho = std::min(rho * drho, (T)kRhoMax);
            alphaOut[iter] = -1; Jout[iter] = prevJ;
            std::memcpy(x, xp, szx * sizeof(T)); std::memcpy(u, up, szu * sizeof(T)); if (M > 1) std::memcpy(d, dp, szx * sizeof(T));
            if (rho == (T)kRhoMax && !c.ignore_max_rho_exit) exit_now = true;
        } else {
            drho = std::min(drho / (T)kRhoFactor, (T)(1.0 / kRhoFactor)); rho = std::max(rho * drho, (T)kRhoMin);
            dJ = dJ / prevJ; prevJ = J; alphaOut[iter] = alphaIndex; Jout[iter] = J;
            if (dJ < (T)c.tol_cost) exit_now = true;
        }
        if (!exit_now) { if (iter == c.max_iter) exit_now = true; else iter += 1; }
        if (exit_now) { nisTime[iter - 1] = wall_ms() - t2; break; }
        // ---- nextIterationSetupCPU (:281-325)
        for (int t = 0; t < COST_T; t++) ph.go([&, t] { derivs(t, COST_T, true, false, x, u); });
        for (int t = 0; t < INT_T; t++) ph.go([&, t] { derivs(t, INT_T, false, true, x, u); });
        ph.go([&] { std::memcpy(Pp, Pm, szP * sizeof(T)); }); ph.go([&] { std::memcpy(pp, pv, szx * sizeof(T)); });
        ph.go([&] { std::memcpy(xp, x, szx * sizeof(T)); }); ph.go([&] { std::memcpy(up, u, szu * sizeof(T)); });
        ph.join();
        nisTime[iter - 2] = wall_ms() - t2;
    }
    // ---- storeVarsCPU (:752-764)
    t2 = wall_ms();
    std::memcpy(x0, x, szx * sizeof(T)); std::memcpy(u0, u, szu * sizeof(T));
    const double t_end = wall_ms();
    *initTime += t_end - t2;
    *tTime = t_end - t_start;
    if (iters_out) *iters_out = iter;
    return 0;
}

template <template <typename> class PT, typename T, typename... Args>
int by_integrator(int integ, Args&&... args) {
    switch (integ) {
    case 1: return run_cpu<PT<T>, 1, T>(std::forward<Args>(args)...);
    case 2: return run_cpu<PT<T>, 2, T>(std::forward<Args>(args)...);
    case 3: return run_cpu<PT<T>, 3, T>(std::forward<Args>(args)...);
    }
    return cpu_fail(PDDP_EINVAL, "integrator must be 1 (Euler), 2 (midpoint) or 3 (RK3)");
}
template <typename T>
int by_plant(const pddp_config& c, const pddp_cpu_buffers& B, void* x0, void* u0, const void* KT0, const void* P0, const void* p0, const void* d0, const void* xg,
             void* Jout, int* alphaOut, int rollout, int clear, int ifd, double* tT, double* sT, double* swT, double* bT, double* nT, double* iT, int cores, int* it, int par) {
#define PDDP_CPU_ARGS c, B, (T*)x0, (T*)u0, (const T*)KT0, (const T*)P0, (const T*)p0, (const T*)d0, (const T*)xg, (T*)Jout, alphaOut, rollout, clear, ifd, tT, sT, swT, bT, nT, iT, cores, it, par
    switch (c.plant) {
    case 1: return by_integrator<PendPlant, T>(c.integrator, PDDP_CPU_ARGS);
    case 2: return by_integrator<CartPlant, T>(c.integrator, PDDP_CPU_ARGS);
    case 3: return by_integrator<QuadPlant, T>(c.integrator, PDDP_CPU_ARGS);
    case 4: if (c.integrator != 1) return cpu_fail(PDDP_EINVAL, "the arm is Euler-only (config.cuh:58)");
            return run_cpu<ArmPlant<T>, 1, T>(PDDP_CPU_ARGS);
#ifdef PDDP_USER_PLANT_HEADER
    case 5: return by_integrator<UserPlant, T>(c.integrator, PDDP_CPU_ARGS);
#endif
    }
#undef PDDP_CPU_ARGS
    return cpu_fail(PDDP_EINVAL, "plant must be 1..4 (5: the user plant of a `make user PLANT_POLICY=...` build)");
}

}  // namespace

static int cpu_run_common(int parallel_ls, const pddp_config* cfg, const pddp_cpu_buffers* buf, void* x0, void* u0, const void* KT0, const void* P0, const void* p0,
                                 const void* d0, const void* xGoal, void* Jout, int* alphaOut, int forwardRolloutFlag, int clearVarsFlag,
                                 int ignoreFirstDefectFlag, double* tTime, double* simTime, double* sweepTime, double* bpTime, double* nisTime,
                                 double* initTime, int cores, int* iters_out) {
    if (!cfg || !buf || !x0 || !u0 || !xGoal || !Jout || !alphaOut || !tTime || !simTime || !sweepTime || !bpTime || !nisTime || !initTime)
        return cpu_fail(PDDP_EINVAL, "pddp_cpu_run_ilqr: null argument");
    const pddp_config& c = *cfg;
    if (parallel_ls && (!buf->xs || !buf->us || !buf->ds || !buf->JTs)) return cpu_fail(PDDP_EINVAL, "pddp_cpu_run_ilqr2: the per-candidate arrays xs, us, ds, JTs of allocateMemory_CPU2 are missing");
    if (c.ee_cost) return cpu_fail(PDDP_EINVAL, "pddp_cpu_run_ilqr: joint-space cost only (EE_COST 0)");
    if (c.use_finite_diff && (c.integrator != 1 || !(c.finite_diff_epsilon > 0.0))) return cpu_fail(PDDP_EINVAL, "pddp_cpu_run_ilqr: use_finite_diff needs the Euler rule and a positive finite_diff_epsilon");
    if (c.N < 4 || (c.N & (c.N - 1)) || c.M < 1 || c.N % c.M || c.N / c.M < 2 || c.A < 1 || c.A > 64 || c.max_iter < 1)
        return cpu_fail(PDDP_EINVAL, "pddp_cpu_run_ilqr: N a power of two >= 4, M dividing N with N/M >= 2, 1 <= A <= 64, max_iter >= 1");
    return c.dtype == 1 ? by_plant<double>(c, *buf, x0, u0, KT0, P0, p0, d0, xGoal, Jout, alphaOut, forwardRolloutFlag, clearVarsFlag, ignoreFirstDefectFlag,
                                           tTime, simTime, sweepTime, bpTime, nisTime, initTime, cores, iters_out, parallel_ls)
                        : by_plant<float>(c, *buf, x0, u0, KT0, P0, p0, d0, xGoal, Jout, alphaOut, forwardRolloutFlag, clearVarsFlag, ignoreFirstDefectFlag,
                                          tTime, simTime, sweepTime, bpTime, nisTime, initTime, cores, iters_out, parallel_ls);
}

#define PDDP_CPU_RUN_ARGS cfg, buf, x0, u0, KT0, P0, p0, d0, xGoal, Jout, alphaOut, forwardRolloutFlag, clearVarsFlag, ignoreFirstDefectFlag, tTime, simTime, sweepTime, bpTime, nisTime, initTime, cores, iters_out
extern "C" int pddp_cpu_run_ilqr(const pddp_config* cfg, const pddp_cpu_buffers* buf, void* x0, void* u0, const void* KT0, const void* P0, const void* p0,
                                 const void* d0, const void* xGoal, void* Jout, int* alphaOut, int forwardRolloutFlag, int clearVarsFlag,
                                 int ignoreFirstDefectFlag, double* tTime, double* simTime, double* sweepTime, double* bpTime, double* nisTime,
                                 double* initTime, int cores, int* iters_out) { return cpu_run_common(0, PDDP_CPU_RUN_ARGS); }
extern "C" int pddp_cpu_run_ilqr2(const pddp_config* cfg, const pddp_cpu_buffers* buf, void* x0, void* u0, const void* KT0, const void* P0, const void* p0,
                                  const void* d0, const void* xGoal, void* Jout, int* alphaOut, int forwardRolloutFlag, int clearVarsFlag,
                                  int ignoreFirstDefectFlag, double* tTime, double* simTime, double* sweepTime, double* bpTime, double* nisTime,
                                  double* initTime, int cores, int* iters_out) { return cpu_run_common(1, PDDP_CPU_RUN_ARGS); }

// a plant file + cost file in the reference's own form (make user PLANT_FILE=... COST_FILE=...): included LAST, so that what those files #define stays out of the library
#include "ref_plugin.hpp"
