// iLQR_examples.cpp -- the reference's benchmark driver shape (examples/WAFR_iLQR_examples.cu: loadXU :69-121, testGPU
// :303-361, statistics :123-227) written against the MI355X-native drop-in: same macros, same entry points, same outputs.
//
// build (host-only C++, no HIP headers needed):
//   g++ -O2 -std=c++11 -DPLANT=4 -DNUM_TIME_STEPS=128 -DNUM_ALPHA=8 examples/iLQR_examples.cpp -Llib -lpddp -Wl,-rpath,'$ORIGIN/../lib' -o examples/iLQR_examples
// run:  examples/iLQR_examples [solves] [seed]
#ifndef PLANT
#define PLANT 4
#endif
#define EE_COST 0
#define TOL_COST 0.0
#define USE_WAFR_URDF 1
#define _Q1 0.1      // q
#define _Q2 0.001    // qd
#define _R 0.0001
#define _QF1 1000.0  // q
#define _QF2 1000.0  // qd
#if PLANT == 4
#ifndef NUM_TIME_STEPS
#define NUM_TIME_STEPS 128
#endif
#ifndef NUM_ALPHA
#define NUM_ALPHA 8
#endif
#endif
#define PDDP_WITH_CPU_PATH 1     // also the reference's CPU entry points (allocateMemory_CPU / runiLQR_CPU / freeMemory_CPU over libpddp_cpu.so)
#include "../hostapi/config.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#define ROLLOUT_FLAG 0
static const double kNoiseStd = 0.001;

// initial trajectory and goal of the reference example for each plant (WAFR_iLQR_examples.cu:19-53,69-121)
template <typename T>
static void loadXU(T* x, T* u, T* xGoal, int ld_x, int ld_u, std::mt19937& rng) {
    std::normal_distribution<double> noise(0.0, kNoiseStd);
    const double PI = 3.14159; (void)PI;
    for (int k = 0; k < NUM_TIME_STEPS; k++) {
        T* xk = x + k * ld_x; T* uk = u + k * ld_u;
#if PLANT == 1
        xk[0] = 0; xk[1] = (T)noise(rng); uk[0] = (T)0.01;
#elif PLANT == 2
        xk[0] = 0; xk[1] = 0; xk[2] = (T)noise(rng); xk[3] = (T)noise(rng); uk[0] = (T)0.01;
#elif PLANT == 3
        for (int i = 0; i < 12; i++) xk[i] = i == 2 ? (T)0.5 : (i < 6 ? (T)0 : (T)noise(rng));
        for (int i = 0; i < 4; i++) uk[i] = (T)1.22625;
#else
        const double q0[7] = {-0.5 * PI, 0.25 * PI, 0.167 * PI, -0.167 * PI, 0.125 * PI, 0.167 * PI, 0.5 * PI};
        const double u0[7] = {0.0, -102.9832, 11.1968, 47.0724, 2.5993, -7.0290, -0.0907};   // gravity compensation at q0 (WAFR URDF)
        for (int i = 0; i < 7; i++) { xk[i] = (T)q0[i]; xk[i + 7] = (T)noise(rng); uk[i] = (T)u0[i]; }
#endif
    }
#if PLANT == 1
    xGoal[0] = (T)3.1416; xGoal[1] = 0;
#elif PLANT == 2
    xGoal[0] = 0; xGoal[1] = (T)3.1416; xGoal[2] = 0; xGoal[3] = 0;
#elif PLANT == 3
    for (int i = 0; i < 12; i++) xGoal[i] = 0;
    xGoal[0] = 7; xGoal[1] = 10; xGoal[2] = (T)0.5;
#else
    const double g[7] = {0, 0, 0, -0.25 * PI, 0, 0.25 * PI, 0.5 * PI};
    for (int i = 0; i < 14; i++) xGoal[i] = i < 7 ? (T)g[i] : (T)0;
#endif
}

static double median(std::vector<double> v) {
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v.size() % 2 ? v[v.size() / 2] : 0.5 * (v[v.size() / 2 - 1] + v[v.size() / 2]);
}

template <typename T>
static void testGPU(int solves, unsigned seed) {
    int ld_x, ld_u, ld_P, ld_p, ld_AB, ld_H, ld_g, ld_KT, ld_du, ld_d, ld_A;
    pddpStream_t* streams;
    T *alpha, *d_alpha; int* alphaIndex;
    T *d_P, *d_p, *d_Pp, *d_pp, *d_AB, *d_H, *d_g, *d_KT, *d_du;
    T **d_x, **d_u, **h_d_x, **h_d_u, *d_xp, *d_xp2, *d_up, *d_JT, *J;
    T **d_d, **h_d_d, *d_dp, *d_dT, *d, *d_ApBK, *d_Bdu, *d_dM;
    int *err, *d_err; T *dJexp, *d_dJexp; T *xGoal, *d_xGoal; T *d_I, *d_Tbody;
    allocateMemory_GPU<T>(&d_x, &h_d_x, &d_xp, &d_xp2, &d_u, &h_d_u, &d_up, &d_xGoal, &xGoal, &d_P, &d_Pp, &d_p, &d_pp, &d_AB, &d_H, &d_g, &d_KT,
                          &d_du, &d_d, &h_d_d, &d_dp, &d_dT, &d_dM, &d, &d_ApBK, &d_Bdu, &d_JT, &J, &d_dJexp, &dJexp, &alpha, &d_alpha, &alphaIndex,
                          &d_err, &err, &ld_x, &ld_u, &ld_P, &ld_p, &ld_AB, &ld_H, &ld_g, &ld_KT, &ld_du, &ld_d, &ld_A, &streams, &d_I, &d_Tbody);
    std::vector<T> x0(ld_x * NUM_TIME_STEPS), u0(ld_u * NUM_TIME_STEPS);
    std::vector<T> Jout((size_t)solves * (MAX_ITER + 1)); std::vector<int> alphaOut((size_t)solves * (MAX_ITER + 1), -2);
    std::vector<double> tTime(solves), initTime(solves), fsim((size_t)solves * MAX_ITER), fsweep((size_t)solves * MAX_ITER),
        bp((size_t)solves * MAX_ITER), nis((size_t)solves * MAX_ITER);
    std::mt19937 rng(seed);
    for (int i = 0; i < solves; i++) {
        std::printf("<<<TESTING GPU %d/%d>>>\n", i + 1, solves);
        loadXU<T>(x0.data(), u0.data(), xGoal, ld_x, ld_u, rng);
        runiLQR_GPU<T>(x0.data(), u0.data(), nullptr, nullptr, nullptr, nullptr, xGoal, &Jout[(size_t)i * (MAX_ITER + 1)],
                       &alphaOut[(size_t)i * (MAX_ITER + 1)], ROLLOUT_FLAG, 1, 1, &tTime[i], &fsim[(size_t)i * MAX_ITER], &fsweep[(size_t)i * MAX_ITER],
                       &bp[(size_t)i * MAX_ITER], &nis[(size_t)i * MAX_ITER], &initTime[i], streams, d_x, h_d_x, d_xp, d_xp2, d_u, h_d_u, d_up, d_P,
                       d_p, d_Pp, d_pp, d_AB, d_H, d_g, d_KT, d_du, d_d, h_d_d, d_dp, d_dT, d, d_ApBK, d_Bdu, d_dM, alpha, d_alpha, alphaIndex, d_JT, J,
                       dJexp, d_dJexp, d_xGoal, err, d_err, ld_x, ld_u, ld_P, ld_p, ld_AB, ld_H, ld_g, ld_KT, ld_du, ld_d, ld_A, d_I, d_Tbody);
    }
    std::printf("Final state:\n");
    for (int i = 0; i < STATE_SIZE; i++) std::printf("%15.5f ", (double)x0[(NUM_TIME_STEPS - 2) * ld_x + i]);
    std::printf("\n");
    // per-iteration medians over the solves: J trace, cumulative time trace, phase times (WAFR_iLQR_examples.cu:123-227)
    std::printf("Median J trace / cumulative ms / alpha of solve 0:\n");
    double cum = median(initTime);
    for (int it = 0; it <= MAX_ITER; it++) {
        std::vector<double> Js, ts;
        for (int i = 0; i < solves; i++) {
            if (alphaOut[(size_t)i * (MAX_ITER + 1) + it] == -2) continue;
            Js.push_back((double)Jout[(size_t)i * (MAX_ITER + 1) + it]);
            if (it > 0) ts.push_back(fsim[(size_t)i * MAX_ITER + it - 1] + fsweep[(size_t)i * MAX_ITER + it - 1] + bp[(size_t)i * MAX_ITER + it - 1] +
                                     nis[(size_t)i * MAX_ITER + it - 1]);
        }
        if (Js.empty()) break;
        cum += median(ts);
        if (it < 12 || it % 10 == 0) std::printf("  iter %3d  J %14.6f  t %9.4f ms  alpha %d\n", it, median(Js), cum, alphaOut[it]);
    }
    std::vector<double> loop(solves);
    for (int i = 0; i < solves; i++) loop[i] = tTime[i] - initTime[i];
    std::printf("Median total %.3f ms, init %.3f ms, loop %.3f ms; phase medians of iteration 1: BP %.4f FP %.4f NIS %.4f ms\n", median(tTime),
                median(initTime), median(loop), bp[0], fsim[0], nis[0]);
    freeMemory_GPU<T>(d_x, h_d_x, d_xp, d_xp2, d_u, h_d_u, d_up, xGoal, d_xGoal, d_P, d_Pp, d_p, d_pp, d_AB, d_H, d_g, d_KT, d_du, d_d, h_d_d, d_dp, d_dM,
                      d_dT, d, d_ApBK, d_Bdu, d_JT, J, d_dJexp, dJexp, alpha, d_alpha, alphaIndex, d_err, err, streams, d_I, d_Tbody);
}

// testCPU of the reference example (examples/WAFR_iLQR_examples.cu:231-299): serialAlphas = 1 ('CS') runs allocateMemory_CPU / runiLQR_CPU / freeMemory_CPU,
// serialAlphas = 0 ('C', the reference's default for the CPU) the parallel line search allocateMemory_CPU2 / runiLQR_CPU2 / freeMemory_CPU2 -- the same call
// sequences as upstream.
template <typename T>
static void testCPU(int serialAlphas, int solves, unsigned seed) {
    int ld_x, ld_u, ld_P, ld_p, ld_AB, ld_H, ld_g, ld_KT, ld_du, ld_d, ld_A;
    T *alpha, *P, *p, *Pp, *pp, *AB, *H, *g, *KT, *du, *x, *u, *xp, *xp2, *up, *JT = nullptr, *d, *dp, *ApBK, *Bdu, *dJexp, *xGoal, *I, *Tbody;
    T **xs = nullptr, **us = nullptr, **ds = nullptr, **JTs = nullptr;
    int* err;
    if (serialAlphas) allocateMemory_CPU<T>(&x, &xp, &xp2, &u, &up, &xGoal, &P, &Pp, &p, &pp, &AB, &H, &g, &KT, &du, &d, &dp, &ApBK, &Bdu, &JT, &dJexp, &alpha, &err, &ld_x,
                                            &ld_u, &ld_P, &ld_p, &ld_AB, &ld_H, &ld_g, &ld_KT, &ld_du, &ld_d, &ld_A, &I, &Tbody);
    else allocateMemory_CPU2<T>(&xs, &x, &xp, &xp2, &us, &u, &up, &xGoal, &P, &Pp, &p, &pp, &AB, &H, &g, &KT, &du, &ds, &d, &dp, &ApBK, &Bdu, &JTs, &dJexp, &alpha, &err,
                                &ld_x, &ld_u, &ld_P, &ld_p, &ld_AB, &ld_H, &ld_g, &ld_KT, &ld_du, &ld_d, &ld_A, &I, &Tbody);
    std::vector<T> x0(ld_x * NUM_TIME_STEPS), u0(ld_u * NUM_TIME_STEPS);
    std::vector<T> Jout((size_t)solves * (MAX_ITER + 1)); std::vector<int> alphaOut((size_t)solves * (MAX_ITER + 1), -2);
    std::vector<double> tTime(solves), initTime(solves), fsim((size_t)solves * MAX_ITER), fsweep((size_t)solves * MAX_ITER), bp((size_t)solves * MAX_ITER),
        nis((size_t)solves * MAX_ITER);
    std::mt19937 rng(seed);
    for (int i = 0; i < solves; i++) {
        std::printf(serialAlphas ? "<<<TESTING CPU %d/%d>>>\n" : "<<<TESTING CPU-P %d/%d>>>\n", i + 1, solves);
        loadXU<T>(x0.data(), u0.data(), xGoal, ld_x, ld_u, rng);
        if (serialAlphas)
            runiLQR_CPU<T>(x0.data(), u0.data(), nullptr, nullptr, nullptr, nullptr, xGoal, &Jout[(size_t)i * (MAX_ITER + 1)], &alphaOut[(size_t)i * (MAX_ITER + 1)],
                           ROLLOUT_FLAG, 1, 1, &tTime[i], &fsim[(size_t)i * MAX_ITER], &fsweep[(size_t)i * MAX_ITER], &bp[(size_t)i * MAX_ITER], &nis[(size_t)i * MAX_ITER],
                           &initTime[i], x, xp, xp2, u, up, P, p, Pp, pp, AB, H, g, KT, du, d, dp, ApBK, Bdu, alpha, JT, dJexp, err, ld_x, ld_u, ld_P, ld_p, ld_AB, ld_H,
                           ld_g, ld_KT, ld_du, ld_d, ld_A, I, Tbody);
        else
            runiLQR_CPU2<T>(x0.data(), u0.data(), nullptr, nullptr, nullptr, nullptr, xGoal, &Jout[(size_t)i * (MAX_ITER + 1)], &alphaOut[(size_t)i * (MAX_ITER + 1)],
                            ROLLOUT_FLAG, 1, 1, &tTime[i], &fsim[(size_t)i * MAX_ITER], &fsweep[(size_t)i * MAX_ITER], &bp[(size_t)i * MAX_ITER], &nis[(size_t)i * MAX_ITER],
                            &initTime[i], xs, x, xp, xp2, us, u, up, P, p, Pp, pp, AB, H, g, KT, du, ds, d, dp, ApBK, Bdu, alpha, JTs, dJexp, err, ld_x, ld_u, ld_P, ld_p,
                            ld_AB, ld_H, ld_g, ld_KT, ld_du, ld_d, ld_A, I, Tbody);
    }
    std::printf("Final state:\n");
    for (int i = 0; i < STATE_SIZE; i++) std::printf("%15.5f ", (double)x0[(NUM_TIME_STEPS - 2) * ld_x + i]);
    std::printf("\n");
    std::vector<double> loop(solves);
    for (int i = 0; i < solves; i++) loop[i] = tTime[i] - initTime[i];
    int its = 0;
    while (its < MAX_ITER && alphaOut[its + 1] != -2) its++;
    std::printf("CPU median total %.3f ms, init %.3f ms, loop %.3f ms; solve 0: %d iterations, J %.6f -> %.6f\n", median(tTime), median(initTime), median(loop), its,
                (double)Jout[0], (double)Jout[its]);
    if (serialAlphas) freeMemory_CPU<T>(x, xp, xp2, u, up, P, Pp, p, pp, AB, H, g, KT, du, d, dp, Bdu, ApBK, dJexp, err, alpha, JT, xGoal, I, Tbody);
    else freeMemory_CPU2<T>(xs, x, xp, xp2, us, u, up, P, Pp, p, pp, AB, H, g, KT, du, ds, d, dp, Bdu, ApBK, dJexp, err, alpha, JTs, xGoal, I, Tbody);
}

// usage: iLQR_examples [G|C|CS] [solves] [seed]   (the reference's main: 'G' GPU, 'C' CPU with the parallel line search, 'CS' CPU serial, 'S' SLQ -- WAFR_iLQR_examples.cu:425-438).  A leading number keeps
// the old form "iLQR_examples solves seed" = GPU.
int main(int argc, char** argv) {
    int a = 1;
    char hardware = 'G';
    if (argc > 1 && (argv[1][0] == 'G' || argv[1][0] == 'C')) { hardware = argv[1][0]; a = 2; }
    const int solves = argc > a ? std::atoi(argv[a]) : 10;
    const unsigned seed = argc > a + 1 ? (unsigned)std::atoi(argv[a + 1]) : 1u;
    if (hardware == 'C') testCPU<algType>((int)(argv[1][1] == 'S'), solves, seed); else testGPU<algType>(solves, seed);
    return 0;
}
