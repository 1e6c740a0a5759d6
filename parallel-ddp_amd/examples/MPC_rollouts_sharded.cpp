// MPC_rollouts_sharded.cpp -- BASELINE configs[3] from the C/C++ host layer: 64 independent Kuka MPC rollouts (N = 64, 8 alphas x 4 segments, end-effector
// cost, MPC_MODE gravity 0; examples/WAFR_MPC_examples.cu:4-37) sharded round-robin over the GPUs of one node, one process per GPU, with the two exchanges
// of include/pddp.h "multi-GPU" (RCCL all-reduce of the exit flag per poll, all-gather of the cost table at the end) -- no Python, no torch.
//
// build:  g++ -O2 -std=c++11 examples/MPC_rollouts_sharded.cpp -Llib -lpddp -Wl,-rpath,'$ORIGIN/../lib' -o examples/MPC_rollouts_sharded
// run:    one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK in the environment (torchrun, mpirun -x, or a shell loop), e.g.
//         for r in 0 1 2 3 4 5 6 7; do RANK=$r WORLD_SIZE=8 LOCAL_RANK=$r PDDP_RENDEZVOUS=/tmp/pddp.id examples/MPC_rollouts_sharded & done; wait
//         A single process (no environment) is a world of one.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pddp.h"

static int env_int(const char* k, int dflt) { const char* v = std::getenv(k); return v ? std::atoi(v) : dflt; }
#define CHECK(call) do { int rc_ = (call); if (rc_) { std::fprintf(stderr, "%s: %s (code %d)\n", #call, pddp_last_error(), rc_); return 1; } } while (0)

int main() {
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", 0);
    const int total = 64, N = 64, n = 14, m = 7;
    if (total % world) { std::fprintf(stderr, "the 64 rollouts must split evenly over the ranks\n"); return 2; }
    const int B = total / world;
    // ---- rendezvous: rank 0 publishes the communicator id in a file
    unsigned char id[PDDP_COMM_ID_BYTES];
    const std::string path = std::getenv("PDDP_RENDEZVOUS") ? std::getenv("PDDP_RENDEZVOUS") : "/tmp/pddp_rollouts_sharded.id";
    if (rank == 0) {
        CHECK(pddp_comm_unique_id(id));
        if (world > 1) { FILE* f = std::fopen((path + ".tmp").c_str(), "wb"); std::fwrite(id, 1, sizeof(id), f); std::fclose(f); std::rename((path + ".tmp").c_str(), path.c_str()); }
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 6000 && !(f = std::fopen(path.c_str(), "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) { std::fprintf(stderr, "rank %d: no rendezvous file %s\n", rank, path.c_str()); return 3; }
        std::fclose(f);
    }
    pddp_comm_handle comm;
    CHECK(pddp_comm_init(&comm, rank, world, id, device));
    // ---- this rank's handle: the problems {r : r % world == rank}
    pddp_config c;
    CHECK(pddp_default_config(&c, 4));
    c.N = N; c.M = 4; c.A = 8; c.batch = B; c.device = device; c.use_graph = 1; c.max_iter = 100;
    c.wafr_urdf = 1; c.mpc_mode = 1; c.tol_cost = 1e-5; c.total_time = 0.5; c.ignore_max_rho_exit = 0; c.ee_cost = 1;
    pddp_handle h;
    CHECK(pddp_create(&c, &h));
    std::vector<float> x0((size_t)B * N * n, 0.f), u0((size_t)B * N * m, 0.01f), goal((size_t)B * n, 0.f);
    for (int l = 0; l < B; l++) {
        const int r = l * world + rank;                                   // global rollout id
        const double ph = 2 * M_PI * r / total;
        for (int k = 0; k < N; k++) {
            float* x = &x0[((size_t)l * N + k) * n];
            x[1] = 0.7f + 0.01f * (float)std::sin(7.0 * r); x[3] = -0.8f + 0.01f * (float)std::cos(5.0 * r); x[5] = 0.75f;     // the MPC example's start pose, perturbed per rollout
        }
        goal[(size_t)l * n + 0] = 0.55f; goal[(size_t)l * n + 1] = 0.20f * (float)std::sin(ph); goal[(size_t)l * n + 2] = 0.45f + 0.12f * (float)std::sin(2 * ph);   // a point of the figure
    }
    CHECK(pddp_load(h, x0.data(), u0.data(), goal.data(), 1, 1));
    double zero = 0; CHECK(pddp_comm_allreduce_max(comm, &zero));         // barrier: every rank is loaded
    const auto t0 = std::chrono::steady_clock::now();
    int all_done = 0, sweeps = 0;
    while (!all_done && sweeps < 800) { CHECK(pddp_iterate(h, 8)); sweeps += 8; CHECK(pddp_comm_all_done(comm, h, &all_done)); }
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CHECK(pddp_comm_allreduce_max(comm, &ms));
    std::vector<double> costs((size_t)total * 2);
    CHECK(pddp_comm_allgather_costs(comm, h, costs.data()));
    if (rank == 0) {
        int best = 0;
        for (int g = 1; g < total; g++) if (costs[2 * g + 1] < costs[2 * best + 1]) best = g;
        int w = 0; pddp_comm_ranks(comm, nullptr, &w);
        std::printf("rccl_ranks_seen %d: %d rollouts (%d per GPU), %d sweeps until every rollout on every rank exited, %.3f ms (max over ranks)\n", w, total, B, sweeps, ms);
        std::printf("best rollout %d: J %.6f -> %.6f; rollout 0: %.6f -> %.6f\n", best, costs[2 * best], costs[2 * best + 1], costs[0], costs[1]);
    }
    pddp_destroy(h); pddp_comm_destroy(comm);
    if (rank == 0 && world > 1) std::remove(path.c_str());
    return 0;
}
