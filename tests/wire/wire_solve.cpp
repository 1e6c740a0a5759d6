// TEST TOOL (-m gpu): a trajectory SOLVED on the GPU through the struct facade (allocateMemory_GPU_MPC / loadTraj / runiLQR_MPC_GPU, hostapi/MPCHelpers.hpp), packed into the
// trajectory message the reference's MPC loop publishes (LCM_MPCLoop_Handler::handleStatus, DDPHelpers/LCMHelpers.cuh:239-262 -> hostapi/LCMHelpers.hpp trajectoryMessage)
// and written out twice: the encoded bytes (argv[1]) and the raw plan as the solver left it in trajVars (argv[2]: utime, then x, u, KT as native floats).
// tests/test_wire_format.py decodes the bytes with its own reader and compares bit for bit.
#define USE_WAFR_URDF 1
#define EE_COST 0
#define MPC_MODE 1
#define IGNORE_MAX_ROX_EXIT 0
#define TOL_COST 0.00001
#define PLANT 4
#define NUM_ALPHA 8
#include "../../parallel-ddp_amd/hostapi/config.hpp"
#include <cstdio>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    typedef algType T;
    trajVars<T>* tvars = new trajVars<T>; matDimms* dimms = new matDimms; algTrace<T>* atrace = new algTrace<T>;
    costParams<T>* cst = new costParams<T>; loadCost(cst);
    GPUVars<T>* algvars = new GPUVars<T>; allocateMemory_GPU_MPC<T>(algvars, dimms, tvars);
    T xInit[STATE_SIZE] = {0}; xInit[1] = (T)(PI / 4.0); xInit[3] = (T)(-PI / 4.0); xInit[5] = (T)(PI / 4.0);
    const double goal[7] = {0.5, 0.6, -0.3, -0.9, 0.2, 0.7, 0.1};
    for (int i = 0; i < STATE_SIZE; i++) algvars->xGoal[i] = i < NUM_POS ? (T)goal[i] : (T)0;
    loadTraj<T>(algvars, tvars, dimms, xInit, nullptr);
    runiLQR_MPC_GPU<T>(tvars, algvars, dimms, atrace, cst, 0, 0, 1);
    const int64_t clock_us = (int64_t)(1.5 * TIME_STEP_LENGTH_IN_us);
    for (int i = 0; i < STATE_SIZE; i++) algvars->xActual[i] = tvars->x[1 * tvars->ld_x + i];
    runiLQR_MPC_GPU<T>(tvars, algvars, dimms, atrace, cst, clock_us, clock_us, 0, 4, 1000.0);       // one control cycle on top of the warm start
    std::printf("iterations %zu J0 %.6f J %.6f\n", atrace->J.size() - 1, (double)atrace->J.front(), (double)atrace->J.back());
    pddp_wire::lcmt_trajectory<T> m = trajectoryMessage<T>(tvars, dimms);
    const std::vector<uint8_t> wire = m.encode();
    FILE* f = std::fopen(argv[1], "wb"); if (!f) return 3;
    std::fwrite(wire.data(), 1, wire.size(), f); std::fclose(f);
    f = std::fopen(argv[2], "wb"); if (!f) return 3;
    const int64_t ut = tvars->t0_plant;
    std::fwrite(&ut, sizeof(ut), 1, f);
    std::fwrite(tvars->x, sizeof(T), (size_t)dimms->ld_x * NUM_TIME_STEPS, f);
    std::fwrite(tvars->u, sizeof(T), (size_t)dimms->ld_u * NUM_TIME_STEPS, f);
    std::fwrite(tvars->KT, sizeof(T), (size_t)dimms->ld_KT * DIM_KT_c * NUM_TIME_STEPS, f);
    std::fclose(f);
    std::printf("wire_len %zu sizes %d %d %d utime %lld\n", wire.size(), m.x_size, m.u_size, m.KT_size, (long long)m.utime);
    freeMemory_GPU_MPC<T>(algvars);
    return 0;
}
