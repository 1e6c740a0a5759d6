// exampleUtils.hpp -- the pieces of the reference's lock-step MPC experiment that sit either side of the solve, same names and argument
// meaning, over libpddp.so:
//
//   loadInitialState                      utils/exampleUtils.cuh:40-46
//   evNorm                                utils/exampleUtils.cuh:84-93            (tool point through pddp_ee_pos)
//   loadFig8Goal                          examples/WAFR_MPC_examples.cu:93-104     (interpolation over a goal table; see Fig8Goals)
//   simulateForward<T, SUBSTEPS>          examples/WAFR_MPC_examples.cu:105-139    (the simulated robot, through pddp_simulate)
//   fig8Simulate                          examples/WAFR_MPC_examples.cu:140-184
//   printAllTimingStats                   (shape of) utils/exampleUtils.cuh        median / mean of the algTrace vectors
//
// Needs MPC_MODE 1 (for trajVars / GPUVars) and, for evNorm / the figure-eight logic, EE_COST 1 -- exactly as upstream.
// The goal table is DATA: the reference compiles 200 (x, y, z) points into loadFig8Goal.  Here the table is loaded at run time
// (Fig8Goals::load, a csv of x,y,z lines -- tests/golden/fig8_goals.csv holds the reference's 200 points) or generated
// (Fig8Goals::lemniscate) when no file is given.
#ifndef PDDP_HOSTAPI_EXAMPLEUTILS_HPP
#define PDDP_HOSTAPI_EXAMPLEUTILS_HPP

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#ifndef PI
#define PI 3.14159                    // utils/cudaUtils.h
#endif

template <typename T>
void loadInitialState(T* xInit, int mode = 0) {   // 0 vertical, 1 centre of the workspace, 2 "patrick" pose
    for (int i = 0; i < STATE_SIZE; i++) xInit[i] = 0;
    if (mode == 1) { xInit[1] = (T)(PI / 4.0); xInit[3] = (T)(-PI / 4.0); xInit[5] = (T)(PI / 4.0); }
    if (mode == 2) {
        xInit[0] = (T)(PI / 2.0); xInit[1] = (T)(-PI / 6.0); xInit[2] = (T)(-PI / 3.0); xInit[3] = (T)(-PI / 2.0); xInit[4] = (T)(3.0 * PI / 4.0);
        xInit[5] = (T)(-PI / 4.0); xInit[6] = 0;
    }
}

struct Fig8Goals {
    std::vector<double> x, y, z;
    static Fig8Goals& table() { static Fig8Goals g; return g; }
    bool load(const char* csv) {
        std::FILE* f = std::fopen(csv, "r");
        if (!f) return false;
        x.clear(); y.clear(); z.clear();
        char line[512];
        while (std::fgets(line, sizeof(line), f)) {
            double a, b, c;
            if (line[0] != '#' && std::sscanf(line, "%lf,%lf,%lf", &a, &b, &c) == 3) { x.push_back(a); y.push_back(b); z.push_back(c); }
        }
        std::fclose(f);
        return x.size() >= 2;
    }
    void lemniscate(int n = 200) {   // a figure eight in the y-z plane in front of the robot, about the size of the reference's
        x.assign(n, 0.6556285); y.resize(n); z.resize(n);
        for (int i = 0; i < n; i++) { const double ph = 2.0 * 3.14159265358979323846 * i / (n - 1); y[i] = 0.30 * std::sin(ph + 0.47); z[i] = 0.49 - 0.195 * std::sin(2.0 * (ph + 0.47)); }
    }
};

// goal[0..2] = the figure-eight point at `time` of a figure that takes totalTime (same units), linearly interpolated; goal[3..5] = 0.
// Returns how many whole figures have been completed.
template <typename T>
int loadFig8Goal(T* goal, double time, double totalTime) {
    Fig8Goals& g = Fig8Goals::table();
    if (g.x.empty()) g.lemniscate();
    const int numGoals = (int)g.x.size();
    const double tstep = totalTime / (numGoals - 1), goalNum = time / tstep;
    const T fraction = static_cast<T>(goalNum - std::floor(goalNum));
    const int rep = static_cast<int>(std::floor(goalNum)) / numGoals;
    const int rd = static_cast<int>(std::floor(goalNum)) % numGoals, ru = static_cast<int>(std::ceil(goalNum)) % numGoals;
    goal[0] = ((T)1 - fraction) * (T)g.x[rd] + fraction * (T)g.x[ru]; goal[3] = 0;
    goal[1] = ((T)1 - fraction) * (T)g.y[rd] + fraction * (T)g.y[ru]; goal[4] = 0;
    goal[2] = ((T)1 - fraction) * (T)g.z[rd] + fraction * (T)g.z[ru]; goal[5] = 0;
    return rep;
}

#if EE_COST
// |tool point - goal| and |qd| of a state
template <typename T>
void evNorm(GPUVars<T>* gv, T* xActual, T* xGoal, T* eNorm, T* vNorm, T* eePos) {
    using namespace pddp_hostapi;
    check(pddp_ee_pos(find(gv->d_P)->h, 1, xActual, eePos), "pddp_ee_pos");
    *eNorm = 0; for (int i = 0; i < 3; i++) { const T d = eePos[i] - xGoal[i]; *eNorm += d * d; } *eNorm = static_cast<T>(std::sqrt(*eNorm));
    *vNorm = 0; for (int i = 0; i < NUM_POS; i++) *vNorm += xActual[NUM_POS + i] * xActual[NUM_POS + i]; *vNorm = static_cast<T>(std::sqrt(*vNorm));
}
template <typename T>
void evNorm(GPUVars<T>* gv, T* xActual, T* xGoal, T* eNorm, T* vNorm) { T eePos[6]; evNorm(gv, xActual, xGoal, eNorm, vNorm, eePos); }
#endif

// The simulated robot: follows tvars' plan from xActual for elapsedTime microseconds in SUBSTEPS steps (plant in double), accumulating
// the distance of the tool point to the figure-eight goal at goalTime; returns the average (0 when the plan ran out: "CRITICAL FAILURE").
template <typename T, int SUBSTEPS>
T simulateForward(GPUVars<T>* gv, trajVars<T>* tvars, T* xActual, double elapsedTime, double goalTime, double totalTime) {
    using namespace pddp_hostapi;
    T goal[6];
    loadFig8Goal<T>(goal, goalTime, totalTime);
    double err = 0; int failed = 0;
    check(pddp_simulate(find(gv->d_P)->h, tvars->x, tvars->u, tvars->KT, static_cast<double>(tvars->t0_plant), elapsedTime, SUBSTEPS, EE_COST ? goal : nullptr,
                        xActual, &err, &failed), "pddp_simulate");
    if (failed) { std::printf("CRITICAL FAILURE ERROR ABORT MISSION\n"); return 0; }
    return static_cast<T>(err);
}

// One lock-step turn after a solve: simulate the robot for the time the solve took, then move the goal (figure-eight phase) or test for
// the initial convergence that starts the figure.  Returns 1 when one whole figure has been tracked.
template <typename T>
int fig8Simulate(GPUVars<T>* gv, T* xActual, T* xGoal, trajVars<T>* tvars, T* error, double* goalTime, double* timePrint, int* counter,
                 int* initial_convergence_flag, double elapsedTime_us, double totalTime_us, T eNormLim, T vNormLim, int ld_x, int doFig8, int debugMode = 1) {
    (void)timePrint; (void)ld_x;
    tvars->t0_plant = 0; tvars->t0_sys = 0;
    *error += simulateForward<T, 150>(gv, tvars, xActual, elapsedTime_us, *goalTime, totalTime_us);
#if EE_COST
    if (doFig8 || debugMode == 1) {
        T eePos[6], eNorm = 0, vNorm = 0;
        evNorm<T>(gv, xActual, xGoal, &eNorm, &vNorm, eePos);
        if (debugMode == 1)
            std::printf("[[%f,%f,%f],[%f,%f,%f],%f,%f,%f],\n", (double)eePos[0], (double)eePos[1], (double)eePos[2], (double)xGoal[0], (double)xGoal[1], (double)xGoal[2],
                        (double)eNorm, (double)((*error) / (*counter)), (double)vNorm);
        if (doFig8) {
            if (*initial_convergence_flag) { *goalTime += elapsedTime_us; if (loadFig8Goal<T>(xGoal, *goalTime, totalTime_us) > 0) return 1; }
            else if (eNorm < eNormLim && vNorm < vNormLim) { *initial_convergence_flag = 1; *error = 0; *counter = 0; }
        }
    }
#else
    (void)xGoal; (void)goalTime; (void)counter; (void)initial_convergence_flag; (void)totalTime_us; (void)eNormLim; (void)vNormLim; (void)doFig8; (void)debugMode;
#endif
    return 0;
}

// The reference's own argument lists (no solver object: its helpers are free functions over host data).  They use the one solver the
// process has allocated; with several, call the overloads above.
namespace pddp_hostapi {
template <typename T> GPUVars<T>* only_solver_vars() {
    static GPUVars<T> gv;
    if (registry().size() != 1) { std::fprintf(stderr, "GPUassert: exactly one allocateMemory_GPU[_MPC] must be live for the argument lists without GPUVars\n"); std::exit(1); }
    gv.d_P = static_cast<T*>(const_cast<void*>(registry().begin()->first));
    return &gv;
}
}  // namespace pddp_hostapi
#if EE_COST
template <typename T> void evNorm(T* xActual, T* xGoal, T* eNorm, T* vNorm, T* eePos) { evNorm(pddp_hostapi::only_solver_vars<T>(), xActual, xGoal, eNorm, vNorm, eePos); }
template <typename T> void evNorm(T* xActual, T* xGoal, T* eNorm, T* vNorm) { T eePos[6]; evNorm(pddp_hostapi::only_solver_vars<T>(), xActual, xGoal, eNorm, vNorm, eePos); }
#endif
template <typename T, int SUBSTEPS>
T simulateForward(trajVars<T>* tvars, T* xActual, double elapsedTime, double goalTime, double totalTime) {
    return simulateForward<T, SUBSTEPS>(pddp_hostapi::only_solver_vars<T>(), tvars, xActual, elapsedTime, goalTime, totalTime);
}
template <typename T>
int fig8Simulate(T* xActual, T* xGoal, trajVars<T>* tvars, T* error, double* goalTime, double* timePrint, int* counter, int* initial_convergence_flag,
                 double elapsedTime_us, double totalTime_us, T eNormLim, T vNormLim, int ld_x, int doFig8, int debugMode = 1) {
    return fig8Simulate<T>(pddp_hostapi::only_solver_vars<T>(), xActual, xGoal, tvars, error, goalTime, timePrint, counter, initial_convergence_flag, elapsedTime_us,
                           totalTime_us, eNormLim, vNormLim, ld_x, doFig8, debugMode);
}

template <typename T>
void printAllTimingStats(algTrace<T>* data) {
    auto stat = [](std::vector<double> v, const char* name) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        double mean = 0; for (double e : v) mean += e; mean /= v.size();
        std::printf("%s: median %.4f ms, mean %.4f ms, max %.4f ms over %zu\n", name, v[v.size() / 2], mean, v.back(), v.size());
    };
    stat(data->tTime, "tTime"); stat(data->initTime, "initTime"); stat(data->bpTime, "bpTime"); stat(data->simTime, "simTime"); stat(data->nisTime, "nisTime");
}

#endif
