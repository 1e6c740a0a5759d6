// MPC_examples.cpp -- the reference's lock-step receding-horizon loop (examples/WAFR_MPC_examples.cu:160-238, `testMPC_lockstep`,
// GPU branch) written against the MI355X-native drop-in: same structs, same entry points, joint-space goal.
//
// The loop: solve to convergence once to warm start; then per control cycle (a) advance the clock by the cycle time, (b) take the
// measured state -- here the plan's own state at that time through the trajectory runner's interpolation plus a seeded
// disturbance, since the robot-side simulator is outside this path --, (c) runiLQR_MPC_GPU with an iteration cap and a time budget.
//
// build:  g++ -O2 -std=c++11 [-DEE_COST=1] examples/MPC_examples.cpp -Llib -lpddp -Wl,-rpath,'$ORIGIN/../lib' -o examples/MPC_examples[_ee]
// run:    examples/MPC_examples [cycles] [iterations per cycle] [budget ms] [cycle time in knots] [disturbance std]
#define USE_WAFR_URDF 1
#ifndef EE_COST
#define EE_COST 0      // -DEE_COST=1: the end-effector cost of examples/WAFR_MPC_examples.cu:4-37 (goal = a tool-point position that moves along a lemniscate)
#endif
#define MPC_MODE 1
#define IGNORE_MAX_ROX_EXIT 0
#define TOL_COST 0.00001
#define PLANT 4
#ifndef NUM_ALPHA
#define NUM_ALPHA 8
#endif
#define _Q1 0.1
#define _Q2 0.001
#define _R 0.0001
#define _QF1 1000.0
#define _QF2 1000.0
#include "../hostapi/config.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>

int main(int argc, char** argv) {
    const int cycles = argc > 1 ? std::atoi(argv[1]) : 20;
    const int itersToDo = argc > 2 ? std::atoi(argv[2]) : 10;
    const double timeLimit = argc > 3 ? std::atof(argv[3]) : 1000.0;
    const double cycleKnots = argc > 4 ? std::atof(argv[4]) : 1.0;
    const double noiseStd = argc > 5 ? std::atof(argv[5]) : 0.001;
    typedef algType T;
    trajVars<T>* tvars = new trajVars<T>; matDimms* dimms = new matDimms; algTrace<T>* atrace = new algTrace<T>;
    costParams<T>* cst = new costParams<T>; loadCost(cst);
    GPUVars<T>* algvars = new GPUVars<T>; allocateMemory_GPU_MPC<T>(algvars, dimms, tvars);

    T xInit[STATE_SIZE] = {0}; xInit[1] = (T)(PI / 4.0); xInit[3] = (T)(-PI / 4.0); xInit[5] = (T)(PI / 4.0);   // loadInitialState mode 1
#if EE_COST
    // a slow lemniscate in the y-z plane in front of the robot; (roll, pitch, yaw) goals are 0 and carry no weight (_Q_EE2 = 0)
    const double period_us = 4.0e6;
    auto setGoal = [&](double t_us) {
        const double ph = 2.0 * 3.14159265358979 * t_us / period_us;
        algvars->xGoal[0] = (T)0.55; algvars->xGoal[1] = (T)(0.20 * std::sin(ph)); algvars->xGoal[2] = (T)(0.45 + 0.12 * std::sin(2.0 * ph));
        algvars->xGoal[3] = algvars->xGoal[4] = algvars->xGoal[5] = (T)0;
    };
    setGoal(0.0);
#else
    const double goal[7] = {0.5, 0.6, -0.3, -0.9, 0.2, 0.7, 0.1};
    for (int i = 0; i < STATE_SIZE; i++) algvars->xGoal[i] = i < NUM_POS ? (T)goal[i] : (T)0;
#endif
    loadTraj<T>(algvars, tvars, dimms, xInit, nullptr);
    runiLQR_MPC_GPU<T>(tvars, algvars, dimms, atrace, cst, 0, 0, 1);                                          // warm start to convergence
    std::printf("warm start: %zu iterations, J %.4f -> %.4f, %.3f ms\n", atrace->J.size() - 1, (double)atrace->J.front(), (double)atrace->J.back(),
                atrace->tTime.back());

    std::mt19937 rng(7);
    std::normal_distribution<double> noise(0.0, noiseStd);
    std::vector<double> cycle_ms;
    int64_t clock_us = 0;
    for (int c = 0; c < cycles; c++) {
        const int64_t elapsed_us = static_cast<int64_t>(std::ceil(cycleKnots * TIME_STEP_LENGTH_IN_us));
        clock_us += elapsed_us;
        // the measured state: the plan at the new time (first-order hold, as getHardwareControls interpolates it) + disturbance
        const double steps = get_time_steps_us_d(tvars->t0_plant, clock_us);
        const int k = std::min(static_cast<int>(steps), NUM_TIME_STEPS - 2);
        const double frac = std::min(steps - k, 1.0);
        for (int i = 0; i < STATE_SIZE; i++)
            algvars->xActual[i] = (T)((1.0 - frac) * tvars->x[k * tvars->ld_x + i] + frac * tvars->x[(k + 1) * tvars->ld_x + i] + noise(rng));
        const size_t traced = atrace->J.size();
#if EE_COST
        setGoal((double)clock_us);
#endif
        runiLQR_MPC_GPU<T>(tvars, algvars, dimms, atrace, cst, clock_us, clock_us, 0, itersToDo, timeLimit);
        cycle_ms.push_back(atrace->tTime.back());
        std::printf("cycle %3d  shift %d  iterations %2zu  J %.4f -> %.4f  last_successful_solve %d  %.3f ms\n", c, k, atrace->J.size() - traced - 1,
                    (double)atrace->J[traced], (double)atrace->J.back(), tvars->last_successful_solve, cycle_ms.back());
    }
    std::sort(cycle_ms.begin(), cycle_ms.end());
    if (!cycle_ms.empty()) std::printf("median cycle %.3f ms (max %.3f) for <= %d iterations\n", cycle_ms[cycle_ms.size() / 2], cycle_ms.back(), itersToDo);
    freeMemory_GPU_MPC<T>(algvars); freeTrajVars<T>(tvars);
    delete algvars; delete atrace; delete tvars; delete dimms; delete cst;
    return 0;
}
