// MPC_lockstep.cpp -- the reference's lock-step figure-eight tracking experiment (examples/WAFR_MPC_examples.cu:185-238 `testMPC_lockstep`, GPU branch;
// the experiment behind test/WAFR_fig8.py:5-12) against the MI355X-native drop-in: solve -> simulate the robot for as long as the solve took ->
// move the goal along the figure -> solve again, until one whole figure has been tracked; prints the average tracking error.
//
// build:  g++ -O2 -std=c++11 examples/MPC_lockstep.cpp -Llib -lpddp -Wl,-rpath,'$ORIGIN/../lib' -o examples/MPC_lockstep
// run:    examples/MPC_lockstep [iterations per solve = 4] [time budget ms = 10] [seconds per figure = 10] [goals.csv] [fixed cycle us]
//         goals.csv: x,y,z lines (tests/golden/fig8_goals.csv = the reference's 200 points); default: a generated lemniscate.
//         fixed cycle us > 0: simulate that long per cycle instead of the measured solve time (reproducible runs).
#define USE_WAFR_URDF 1
#define EE_COST 1
#define MPC_MODE 1
#define IGNORE_MAX_ROX_EXIT 0
#define TOL_COST 0.00001
#define PLANT 4
#include "../hostapi/config.hpp"

int main(int argc, char** argv) {
    typedef algType T;
    const int itersToDo = argc > 1 ? std::atoi(argv[1]) : 4;
    const double timeLimit = argc > 2 ? std::atof(argv[2]) : 10.0;
    const double totalTime_us = 1000000.0 * (argc > 3 ? std::atof(argv[3]) : 10.0);
    if (argc > 4 && argv[4][0] && !Fig8Goals::table().load(argv[4])) { std::fprintf(stderr, "cannot read goals from %s\n", argv[4]); return 1; }
    const double fixed_us = argc > 5 ? std::atof(argv[5]) : 0.0;
    const int maxCycles = 200000;
    T eNormLim = (T)0.05, vNormLim = (T)0.05;
    double goalTime = 0, timePrint = 0; int initial_convergence_flag = 0, counter = 0; T error = 0; struct timeval start, end;
    trajVars<T>* tvars = new trajVars<T>; matDimms* dimms = new matDimms; algTrace<T>* atrace = new algTrace<T>; costParams<T>* cst = new costParams<T>; loadCost(cst);
    GPUVars<T>* algvars = new GPUVars<T>; allocateMemory_GPU_MPC<T>(algvars, dimms, tvars);
    T xInit[STATE_SIZE]; loadInitialState<T>(xInit, 1);
    loadTraj<T>(algvars, tvars, dimms, xInit, nullptr); loadFig8Goal<T>(algvars->xGoal, goalTime, totalTime_us);
    runiLQR_MPC_GPU<T>(tvars, algvars, dimms, atrace, cst, 0, 0, 1); tvars->t0_plant = 0; double elapsedTime_us = 0;
    int cycles = 0, figure_done = 0;
    while (cycles < maxCycles) {
        counter++; cycles++;
        gettimeofday(&start, NULL);
        runiLQR_MPC_GPU<T>(tvars, algvars, dimms, atrace, cst, 0, static_cast<int64_t>(elapsedTime_us), 0, itersToDo, timeLimit);
        gettimeofday(&end, NULL);
        elapsedTime_us = fixed_us > 0 ? fixed_us : time_delta_us(start, end);
        if (fig8Simulate<T>(algvars->xActual, algvars->xGoal, tvars, &error, &goalTime, &timePrint, &counter, &initial_convergence_flag, elapsedTime_us, totalTime_us,
                            eNormLim, vNormLim, dimms->ld_x, 1, 0)) { figure_done = 1; break; }
    }
    std::printf("\n\ncycles: %d  figure completed: %d  reached the start of the figure: %d\n", cycles, figure_done, initial_convergence_flag);
    std::printf("Average tracking error: [%f]\n", (double)(error / (counter > 0 ? counter : 1)));
    printAllTimingStats(atrace);
    freeMemory_GPU_MPC<T>(algvars); freeTrajVars<T>(tvars);
    delete algvars; delete atrace; delete tvars; delete dimms; delete cst;
    return 0;
}
