// __global__ entry points: one DDP sweep = k_bp -> k_fp -> k_ls -> k_nis on one stream, no host sync.
// grid.y indexes the independent problems of the batch; every workgroup is made of whole 64-lane waves, each wave
// owning one unit of work (see pddp_common.hpp).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include "bodies.hpp"
#include "fp_lg.hpp"
#include "nis_lg.hpp"
#include "bp_lg.hpp"
#include "mpc.hpp"
#include "sim.hpp"

namespace pddp {

// backward pass: grid (M, B), block 64.  Replaces backPassKern<<<M_BLOCKS_B,(8,7)>>> (bpHelpers.cuh:492).
template <typename P, typename T>
__global__ __launch_bounds__(64) void k_bp(Buffers<T> b, Dims dm) {
    __shared__ BpScratch<P, T> s;
    bp_body<P, T>(this_wave(), s, b, dm, blockIdx.x, blockIdx.y);
}

// the same with a whole 256-thread workgroup per block of knots: every stage has 98..441 independent outputs, so when only a few
// problems are in flight (one MPC solve: the latency case) four waves per block cut the per-knot time
template <typename P, typename T>
__global__ __launch_bounds__(256) void k_bp_wide(Buffers<T> b, Dims dm) {
    __shared__ BpScratch<P, T> s;
    bp_body<P, T>(this_block(), s, b, dm, blockIdx.x, blockIdx.y);
}

// forward pass: grid (A, B), block M*64, dynamic LDS.  One workgroup per (candidate alpha, problem): wave 0 runs the
// linear sweep, then wave b rolls out segment b, then wave 0 reduces cost and defect.  Replaces forwardSweepKern<<<A,14>>>,
// forwardSimKern<<<(M,A),(8,7)>>>, costKern<<<A,N>>> and defectKern<<<A,N>>> (DDPWrappers.cuh:73, fpHelpers.cuh:366,383,388).
template <typename P, typename T>
struct FpLds {
    static PDDP_HD size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
    static size_t bytes(int M, int N) {
        return align16(sizeof(SweepScratch<P, T>)) + (size_t)M * align16(sizeof(SimScratch<P, T>)) +
               align16(sizeof(T) * (size_t)N) + align16(sizeof(T) * (size_t)M * P::NX) + 2 * align16(sizeof(T) * (size_t)M);
    }
};

// init_rollout = 1: the optional initial rollout of loadVarsGPU (forwardRolloutFlag, nisInitHelpers.cuh:642-648): launched with
// grid (1, B); no sweep, every segment starts from the loaded state x0[b*NB], alpha = alpha[0], candidate slot 0.
template <typename P, int INTEG, typename T>
__global__ void k_fp(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int init_rollout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int a_idx = blockIdx.x, pb = blockIdx.y, M = dm.M;
    if (init_rollout != 1 && !fp_active<T>(b, dm, pb)) return;
    using L = FpLds<P, T>;
    unsigned char* ptr = lds_raw;
    SweepScratch<P, T>& sw = *reinterpret_cast<SweepScratch<P, T>*>(ptr); ptr += L::align16(sizeof(SweepScratch<P, T>));
    const int wave_id = threadIdx.x / kWave;
    SimScratch<P, T>& sim = *reinterpret_cast<SimScratch<P, T>*>(ptr + (size_t)wave_id * L::align16(sizeof(SimScratch<P, T>)));
    ptr += (size_t)M * L::align16(sizeof(SimScratch<P, T>));
    T* cost_k = reinterpret_cast<T*>(ptr); ptr += L::align16(sizeof(T) * (size_t)dm.N);
    T* segx = reinterpret_cast<T*>(ptr); ptr += L::align16(sizeof(T) * (size_t)M * P::NX);
    T* dnorm = reinterpret_cast<T*>(ptr); ptr += L::align16(sizeof(T) * (size_t)M);
    T* segJ = reinterpret_cast<T*>(ptr);
    const Wave w = this_wave();
    const FpArgs<T> a = fp_args<P, T>(b, dm, pb, a_idx, dt, segx, dnorm, segJ);
    if (init_rollout == 1) {
        rollout_seed_segment<P, T>(w, dm, a, wave_id);
        __syncthreads();
    } else if (init_rollout == 2) {                      // PDDP_PHASE_ROLLOUT: the candidates' segment start states as they stand in xs, no sweep
        rollout_seed_from_candidate<P, T>(w, dm, a, wave_id);
        __syncthreads();
    } else if (M > 1) {
        if (wave_id == 0) forward_sweep<P, T>(w, sw, dm, a);
        __syncthreads();
    }
    P::load_model(w, sim.plant, reinterpret_cast<const typename P::Model*>(b.model));
    forward_sim_segment<P, INTEG, T>(w, sim, dm, a, wave_id, cw, b.xGoal + (size_t)pb * P::NX, cost_k);
    __syncthreads();
    bool ee = false;
    if constexpr (P::PLANT == 4) ee = cw.ee != 0;
    if (wave_id == 0) fp_reduce<T>(w, b, dm, pb, a_idx, cost_k, dnorm, ee ? segJ : nullptr);
}

// ---------------------------------------------------------------------------------------------- KUKA arm: lane-group forward pass
// (fp_lg.hpp).  k_sweep_lg: grid (ceil(A/8), B), block 64 -- one 8-lane group per candidate alpha.
template <typename T>
__global__ __launch_bounds__(64) void k_sweep_lg(Buffers<T> b, Dims dm, T dt) {
    const int pb = blockIdx.y, a_idx = blockIdx.x * kLgPerWave + (threadIdx.x >> 3);
    if (!fp_active<T>(b, dm, pb) || a_idx >= dm.A || LgDevice<T>::lane() == 7) return;   // lane 7 of every group stays inactive (lanegroup.hpp)
    const FpLgArgs<T> a = fp_lg_args<T>(b, dm, pb, a_idx, dt, nullptr);
    arm_lg_forward_sweep<LgDevice<T>, T>(dm, a);
}
// k_fp_lg: grid (B, C), block 64 * ceil(Ac*M/8) with Ac = A / C candidates per workgroup: group i of the block rolls out segment i / Ac of candidate
// a0 + i % Ac (the 8 groups of a wave are 8 candidates of one segment: they read the same gains, which the memory pipeline coalesces); per-knot
// costs meet in LDS and are tree-summed per candidate in the reference's pairing.  Dynamic LDS: Ac*(N+M) elements.  C > 1 (A a multiple of 8 above 8:
// one workgroup per 8 candidates) spreads one problem over C compute units -- the same waves in total, half the latency per sweep for A = 16.
template <typename T, int MAXT, bool EE = false>
__global__ __launch_bounds__(MAXT, 1) void k_fp_lg(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int init_rollout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int pb = blockIdx.x;
    if (!init_rollout && !fp_active<T>(b, dm, pb)) return;
    const int A_all = init_rollout ? 1 : dm.A, A_eff = A_all / (int)gridDim.y, a0 = (int)blockIdx.y * A_eff, n_inst = A_eff * dm.M;
    __shared__ ArmModel<T> lds_model;
    T* cost_k = reinterpret_cast<T*>(lds_raw);              // [A_eff][N]
    T* dnorm = cost_k + (size_t)A_eff * dm.N;               // [A_eff][M]
    {
        const T* src = reinterpret_cast<const T*>(b.model); T* dst = reinterpret_cast<T*>(&lds_model);
        for (int e = threadIdx.x; e < (int)(sizeof(ArmModel<T>) / sizeof(T)); e += blockDim.x) dst[e] = src[e];
    }
    __syncthreads();
    const int inst = threadIdx.x >> 3;
    if (inst < n_inst && LgDevice<T>::lane() < 7) {        // lane 7 of every group stays inactive (lanegroup.hpp)
        const int a_loc = inst % A_eff, seg = inst / A_eff;
        ArmLgConst<LgDevice<T>> c;
        arm_lg_load_const<LgDevice<T>, T>(c, &lds_model);
        const FpLgArgs<T> a = fp_lg_args<T>(b, dm, pb, a0 + a_loc, dt, dnorm + a_loc * dm.M);
        if constexpr (EE) arm_lg_rollout_segment_ee<LgDevice<T>, T>(c, dm, a, seg, cw, cost_k + (size_t)a_loc * dm.N, init_rollout != 0);   // segment sums in cost_k[a][0..M)
        else arm_lg_rollout_segment<LgDevice<T>, T>(c, dm, a, seg, cw, cost_k + (size_t)a_loc * dm.N, init_rollout != 0);
    }
    __syncthreads();
    const Wave w = this_wave();
    const int wave_id = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
    for (int a_loc = wave_id; a_loc < A_eff; a_loc += nwaves)
        fp_reduce<T>(w, b, dm, pb, a0 + a_loc, cost_k + (size_t)a_loc * dm.N, dnorm + a_loc * dm.M, EE ? cost_k + (size_t)a_loc * dm.N : nullptr);
}
// k_bp_lg: grid (ceil(B*M/8)), block 64 -- one 8-lane group per (problem, block of knots) (bp_lg.hpp); 8 x 2.4 KB of LDS.
template <typename T>
__global__ __launch_bounds__(64) void k_bp_lg(Buffers<T> b, Dims dm, int batch) {
    __shared__ __attribute__((aligned(16))) T lds[kLgPerWave * kBpLgFloats];
    const int inst = blockIdx.x * kLgPerWave + (threadIdx.x >> 3);
    if (inst >= batch * dm.M || LgDevice<T>::lane() == 7) return;    // lane 7 of every group stays inactive (lanegroup.hpp)
    arm_lg_bp_body<LgDevice<T>, T>(lds + (threadIdx.x >> 3) * kBpLgFloats, b, dm, inst % dm.M, inst / dm.M, LgDevice<T>::lane() == 6);
}

// k_nis_lg: grid (ceil(N/32), B), block 256 -- one 8-lane group per knot, 32 knots per workgroup (nis_lg.hpp).
template <typename T, bool EE = false>
__global__ __launch_bounds__(256, 2) void k_nis_lg(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int mode) {
    __shared__ ArmModel<T> lds_model;
    {
        const T* src = reinterpret_cast<const T*>(b.model); T* dst = reinterpret_cast<T*>(&lds_model);
        for (int e = threadIdx.x; e < (int)(sizeof(ArmModel<T>) / sizeof(T)); e += blockDim.x) dst[e] = src[e];
    }
    __syncthreads();
    const int k = blockIdx.x * 32 + (threadIdx.x >> 3), pb = blockIdx.y;
    if (k >= dm.N || LgDevice<T>::lane() == 7) return;       // lane 7 of every group stays inactive (lanegroup.hpp)
    ArmLgConst<LgDevice<T>> c;
    arm_lg_load_const<LgDevice<T>, T>(c, &lds_model);
    arm_lg_nis_body<LgDevice<T>, T, EE>(c, b, dm, cw, dt, mode, k, pb);
}

// forward dynamics (grad = 0: out qdd[count][7]) or its gradient (grad = 1: out dqdd[count][7*21]) of `count` (x,u) samples,
// one lane group each (tests, micro-benchmarks)
template <typename T>
__global__ __launch_bounds__(64) void k_plant_eval_lg(const void* model, int count, const T* x, const T* u, T* out, int grad) {
    using L = LgDevice<T>;
    ArmLgConst<L> c;
    arm_lg_load_const<L, T>(c, reinterpret_cast<const ArmModel<T>*>(model));
    ArmLgState<L> st;
    if (L::lane() == 7) return;                              // lane 7 of every group stays inactive (lanegroup.hpp)
    for (int i = blockIdx.x * kLgPerWave + (threadIdx.x >> 3); i < count; i += gridDim.x * kLgPerWave) {
        const T q = x[(size_t)i * 14 + L::lane()], qd = x[(size_t)i * 14 + 7 + L::lane()], uu = u[(size_t)i * 7 + L::lane()];
        if (grad == 2) { out[(size_t)i * 7 + L::lane()] = arm_lg_dynamics<L, true>(c, st, q, qd, uu); continue; }   // packed variant (forward pass)
        const T qdd = arm_lg_dynamics<L>(c, st, q, qd, uu);
        if (!grad) out[(size_t)i * 7 + L::lane()] = qdd;
        else { T* o = out + (size_t)i * 147 + L::lane(); arm_lg_gradient<L>(c, st, qd, qdd, [o](int jj, T val) { o[7 * jj] = val; }); }
    }
}

// teacher-forcing hook only (pddp_run_phase(FP)): the line-search kernel normally adds the thread-lane forward pass's per-segment partial sums
template <typename T>
__global__ __launch_bounds__(64) void k_reduce_parts(Buffers<T> b, Dims dm, int batch) {
    const int pb = blockIdx.x * 64 + threadIdx.x;
    if (pb < batch && b.parts_fresh && b.parts_fresh[pb]) { tl_reduce_parts<T>(b, dm, pb); b.parts_fresh[pb] = 0; }
}
// line search + accept/reject: grid (B), block 64; one lane takes the decision the reference takes on the host
// (fpHelpers.cuh:395-408, nisInitHelpers.cuh:489-518).
template <typename T>
__global__ __launch_bounds__(64) void k_ls(Buffers<T> b, Dims dm, SolverParams sp, int freeze_exit) {
    const int pb = blockIdx.x;
    // thread-lane forward pass: lane a adds candidate a's per-segment partial sums (tl_reduce_parts, the same order) -- the loads of the A x M parts are in
    // flight together instead of one after the other on the deciding lane
    if (b.parts_fresh && b.parts_fresh[pb]) {
        const int a = threadIdx.x;
        if (a < dm.A) {
            const size_t slot = (size_t)pb * dm.A + a;
            T J = T(0), mx = T(0);
            for (int s = 0; s < dm.M; s++) { J += b.Jpart[slot * dm.M + s]; mx = tmax(mx, b.dpart[slot * dm.M + s]); }
            b.J[slot] = J; b.dmax[slot] = mx;
        }
        __syncthreads();                                    // (A <= 64: one wave) the sums are visible to lane 0
        if (threadIdx.x == 0) b.parts_fresh[pb] = 0;
        __threadfence_block();
    }
    if (threadIdx.x == 0) ls_body<T>(b, dm, sp, pb, freeze_exit);
}

// The same decision with MANY problems in flight: grid ceil(B / 64), block 64, one THREAD per problem (ls_body adds the thread-lane rollouts' partial sums itself, in the
// same order) -- 256 waves instead of 16384 one-wave workgroups whose launch is all they cost.
template <typename T>
__global__ __launch_bounds__(64) void k_ls_many(Buffers<T> b, Dims dm, SolverParams sp, int freeze_exit, int batch) {
    const int pb = blockIdx.x * 64 + threadIdx.x;
    if (pb < batch) ls_body<T>(b, dm, sp, pb, freeze_exit);
}

// next-iteration setup: grid (N, B), block 64 (see nis_body).
template <typename P, int INTEG, typename T>
__global__ __launch_bounds__(64) void k_nis(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int mode) {
    __shared__ NisScratch<P, INTEG, T> s;
    nis_body<P, INTEG, T>(this_wave(), s, b, dm, cw, dt, mode, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------- closed-form plants, many problems in flight: thread-serial kernels
// The pendulum, cart-pole, quadrotor (and user plants) have 2..12 states: a whole 64-lane wave per unit of work (k_bp, k_fp, k_nis above: the reference's launch
// shape re-mapped) leaves most lanes idle and pays a wave's instruction stream for a 4x4 product.  With many problems in flight one THREAD owns the unit instead:
// the SAME body, instantiated with a one-lane "wave" (Wave{0, 1, 0}: every PDDP_FOR loop runs serially in index order -- the arithmetic of the host emulation the
// CPU suite holds against the oracle, and bit for bit what the cooperative kernels compute, whose lanes each own whole output elements), its stage scratch in
// the thread's private memory.  64 independent units per wave, no LDS, no barriers.
//   k_bp_ts : thread = (problem, block of knots)      replaces backPassKern<<<M,(8,7)>>>                      (bpHelpers.cuh:339-420)
//   k_fp_ts : thread = (problem, candidate)           sweep, the M rollouts in turn, cost tree, defect max     (fpHelpers.cuh:57-63, 279-301, 134-152, 96-111)
//   k_nis_ts: thread = (problem, knot)                derivatives + winner adoption                            (nisInitHelpers.cuh:247-279)
PDDP_HD Wave serial_wave() { return Wave{0, 1, 0}; }
template <typename P, typename T>
__global__ __launch_bounds__(64) void k_bp_ts(Buffers<T> b, Dims dm, int batch) {
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= batch * dm.M) return;
    BpScratch<P, T> s;
    bp_body<P, T>(serial_wave(), s, b, dm, inst % dm.M, inst / dm.M);
}
constexpr int kTsMaxN = 256, kTsMaxM = 16;        // longest horizon / most segments the thread-serial forward pass keeps per-knot costs / hand-off states for
template <typename P, int INTEG, typename T>
__global__ __launch_bounds__(64) void k_fp_ts(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int batch, int no_sweep = 0) {
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= batch * dm.A) return;
    const int pb = inst / dm.A, a_idx = inst - pb * dm.A;
    if (!fp_active<T>(b, dm, pb)) return;
    SweepScratch<P, T> sw; SimScratch<P, T> sim;
    T cost_k[kTsMaxN], segx[kTsMaxM * P::NX], dnorm[kTsMaxM], segJ[kTsMaxM];
    const Wave w = serial_wave();
    const FpArgs<T> a = fp_args<P, T>(b, dm, pb, a_idx, dt, segx, dnorm, segJ);
    if (no_sweep) { for (int bInd = 0; bInd < dm.M; bInd++) rollout_seed_from_candidate<P, T>(w, dm, a, bInd); }
    else if (dm.M > 1) forward_sweep<P, T>(w, sw, dm, a);
    P::load_model(w, sim.plant, reinterpret_cast<const typename P::Model*>(b.model));
    for (int bInd = 0; bInd < dm.M; bInd++) forward_sim_segment<P, INTEG, T>(w, sim, dm, a, bInd, cw, b.xGoal + (size_t)pb * P::NX, cost_k);
    fp_reduce<T>(w, b, dm, pb, a_idx, cost_k, dnorm, nullptr);
}
template <typename P, int INTEG, typename T>
__global__ __launch_bounds__(64) void k_nis_ts(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int mode, int batch) {
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= batch * dm.N) return;
    NisScratch<P, INTEG, T> s;
    nis_body<P, INTEG, T>(serial_wave(), s, b, dm, cw, dt, mode, inst % dm.N, inst / dm.N);
}

// In between for the larger closed-form plants (quadrotor: 12 states, 16 columns of [A B]): G lanes per unit, 64 / G units per wavefront, the SAME bodies with a
// G-lane "wave" (every PDDP_FOR loop strides by G; stage scratch per unit in LDS).  One thread per unit keeps 12 x 16 stage matrices in private memory (scratch
// traffic), a whole wave per unit runs the plug-in's scalar closed-form gradient on one lane of 64 and a 192-entry matrix chain with 3 entries per lane.
//   k_nis_gl: G lanes = (problem, knot)        k_bp_gl: G lanes = (problem, block of knots)
// (Batching the plug-in's scalar code of 16 knots onto one wavefront of a 256-thread block -- 48 lanes active instead of 12 -- was slower: 7.2 -> 8.9 ms at 16384
// problems; three of four waves wait at the barriers around it.)
template <typename P, int INTEG, typename T, int G>
__global__ __launch_bounds__(64) void k_nis_gl(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int mode, int batch) {
    constexpr int U = 64 / G;
    __shared__ NisScratch<P, INTEG, T> s[U];
    const int grp = threadIdx.x / G, inst = blockIdx.x * U + grp;
    if (inst >= batch * dm.N) return;
    nis_body<P, INTEG, T>(Wave{(int)threadIdx.x & (G - 1), G, 0}, s[grp], b, dm, cw, dt, mode, inst % dm.N, inst / dm.N);
}
template <typename P, typename T, int G>
__global__ __launch_bounds__(64) void k_bp_gl(Buffers<T> b, Dims dm, int batch) {
    constexpr int U = 64 / G;
    __shared__ BpScratch<P, T> s[U];
    const int grp = threadIdx.x / G, inst = blockIdx.x * U + grp;
    if (inst >= batch * dm.M) return;
    bp_body<P, T>(Wave{(int)threadIdx.x & (G - 1), G, 0}, s[grp], b, dm, inst % dm.M, inst / dm.M);
}

// API view of the compact end-effector Hessian block (Buffers::Hc): H_k of every running knot in the reference layout = Jee' Jee (+ Qx on its diagonal, already in the
// block) in the position rows / columns, Qxd and R_EE on the rest of the diagonal.  The final knot's block is written in full by the setup kernel.  thread = knot.
template <typename T>
__global__ __launch_bounds__(64) void k_hc_expand(Buffers<T> b, int knots, int N, T Qxd, T Ru) {
    const int G = blockIdx.x * 64 + threadIdx.x;
    if (G >= knots || (G % N) == N - 1) return;
    T* H = b.H + (size_t)G * 441; const T* hc = b.Hc + (size_t)G * 49;
    for (int e = 0; e < 441; e++) {
        const int c = e / 21, r = e % 21;
        H[e] = (r < 7 && c < 7) ? hc[c * 7 + r] : (r != c ? T(0) : (r < 14 ? Qxd : Ru));
    }
}

// after the initial rollout: candidate slot 0 becomes the current trajectory (initAlgGPU copies slot 0 to xp, up, dp,
// nisInitHelpers.cuh:378-381).  grid (N, B), block 64.
template <typename P, typename T>
__global__ __launch_bounds__(64) void k_adopt_slot0(Buffers<T> b, Dims dm) {
    constexpr int NX = P::NX, NU = P::NU;
    const int k = blockIdx.x, pb = blockIdx.y, N = dm.N;
    const Wave w = this_wave();
    const size_t slot = (size_t)pb * dm.A;
    PDDP_FOR(i, NX) b.xb[((size_t)pb * 2 * N + k) * NX + i] = b.xs[(slot * N + k) * NX + i];
    PDDP_FOR(i, NU) b.ucur[((size_t)pb * N + k) * NU + i] = b.us[(slot * N + k) * NU + i];
    PDDP_FOR(i, NX) b.dcur[((size_t)pb * N + k) * NX + i] = b.ds[(slot * N + k) * NX + i];
}

// MPC warm start / fall-back (mpc.hpp): grid (B), block 256 -- four waves share the shifting of the previous solution, wave 0 rolls out; block 512 for the arm in float with a
// built-in robot model (V >= 0): the rollout is a three-wave pipeline that starts at once, the other waves shift and copy beside it.
template <typename P, int INTEG, typename T, int V = -1>
__global__ __launch_bounds__(512) void k_mpc_load(Buffers<T> b, MpcBuffers<T> mb, Dims dm, T dt, const T* xActual, const int* shift, int clear_vars, int full_rollout, const T* goal_in, int tshift_mode) {
    __shared__ MpcScratch<P, T> s;
    // the cycle's goals and final-cost shifts arrive in the same transfer as the measured states: put them where the sweeps read them (nothing in this kernel does)
    if (threadIdx.x < P::NX) b.xGoal[(size_t)blockIdx.x * P::NX + threadIdx.x] = goal_in[(size_t)blockIdx.x * P::NX + threadIdx.x];
    if (threadIdx.x == 0) b.tshift[blockIdx.x] = tshift_mode ? shift[blockIdx.x] : 0;
    float* pipe = nullptr;
    if constexpr (P::PLANT == 4 && INTEG == 1 && V >= 0 && sizeof(T) == 4) {       // the warm-start rollout as a pipeline over three waves (fp_pipe.hpp)
        __shared__ __attribute__((aligned(16))) float pipe_lds[kPipeLdsOpenLoop / 4];
        pipe = pipe_lds;
        if (threadIdx.x < kPipeFlags) tl_pipe_lds(pipe, false).flag[threadIdx.x] = 0;
        __syncthreads();                                                                     // the counters are zero before any wave takes its role
    }
    mpc_load_body<P, INTEG, T, V>(this_wave(), s, b, mb, dm, dt, blockIdx.x, xActual + (size_t)blockIdx.x * P::NX, shift[blockIdx.x], clear_vars, full_rollout,
                                  (int)threadIdx.x >> 6, (int)blockDim.x >> 6, pipe);
}
// out != null: the problem's results packed into one record of rec_bytes for a single transfer back -- [SolverState, padded to 16 bytes][x: current half][u][K]
// [Jout: out_stride][alphaOut: out_stride ints]; a problem that took no step packs the fall-back arrays it has just been restored from (the same values).
template <typename P, typename T>
__global__ __launch_bounds__(256) void k_mpc_store(Buffers<T> b, MpcBuffers<T> mb, Dims dm, int only_exited, unsigned char* out, int rec_bytes, int out_stride) {
    constexpr int NX = P::NX, NU = P::NU;
    const int pb = blockIdx.x, N = dm.N;
    const SolverState<T> st = b.state[pb];
    const bool skip = only_exited && !st.done;                  // (uniform) a problem that is still iterating keeps its trajectory: the caller goes on with it
    if (!skip && threadIdx.x < 64) mpc_store_body<P, T>(this_wave(), b, mb, dm, pb);          // (wave 0; blocks of 256 threads only pack faster)
    if (!out) return;
    const int nt = blockDim.x;
    unsigned char* r = out + (size_t)pb * rec_bytes;
    if (threadIdx.x == 0) *reinterpret_cast<SolverState<T>*>(r) = st;
    if (skip) return;
    T* rx = reinterpret_cast<T*>(r + (sizeof(SolverState<T>) + 15) / 16 * 16);
    const bool old = !st.took_step;
    const T* xs = old ? mb.x_old + (size_t)pb * N * NX : b.xb + ((size_t)pb * 2 + st.cur) * N * NX;
    const T* us = old ? mb.u_old + (size_t)pb * N * NU : b.ucur + (size_t)pb * N * NU;
    const T* ks = old ? mb.KT_old + (size_t)pb * N * NX * NU : b.KT + (size_t)pb * N * NX * NU;
    for (int e = threadIdx.x; e < N * NX; e += nt) rx[e] = xs[e];
    rx += N * NX;
    for (int e = threadIdx.x; e < N * NU; e += nt) rx[e] = us[e];
    rx += N * NU;
    for (int e = threadIdx.x; e < N * NX * NU; e += nt) rx[e] = ks[e];
    rx += N * NX * NU;
    for (int e = threadIdx.x; e < out_stride; e += nt) { rx[e] = b.Jout[(size_t)pb * out_stride + e]; reinterpret_cast<int*>(rx + out_stride)[e] = b.alphaOut[(size_t)pb * out_stride + e]; }
}

// initial cost + solver state: grid (B), block 64, dynamic LDS N*sizeof(T).
template <typename P, typename T>
__global__ __launch_bounds__(64) void k_init_cost(Buffers<T> b, Dims dm, CostWeights<T> cw, SolverParams sp, int ignore_first_defect, int rollout,
                                                  int stage, int keep_alpha) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    init_cost_body<P, T>(this_wave(), reinterpret_cast<T*>(lds_raw), b, dm, cw, sp, ignore_first_defect, rollout, blockIdx.x, stage, keep_alpha);
}
// ---------------------------------------------------------------------------------------------- HBM counter calibration (profiling tool)
// Streams `count` floats from src to dst with the access shape the sweep kernels use (one dword per lane, consecutive
// lanes on consecutive addresses).  Its byte count is known exactly, which calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE
// for this access width (MI355X_MICROARCH.md, "HBM": only the 16 B/lane case is documented).
__global__ __launch_bounds__(256) void k_hbm_calib_dword(const float* __restrict__ src, float* __restrict__ dst, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------- plant evaluation (tests / tools)
template <typename P, int INTEG, typename T>
__global__ __launch_bounds__(64) void k_plant_eval(const void* model, int what, int count, const T* x, const T* u, T* out, T dt) {
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU, NP = P::NPOS;
    __shared__ NisScratch<P, INTEG, T> s;
    __shared__ IntegScratch<P, T> is;
    __shared__ T xn[NX], qdd[NP], dq[NP * NM];
    const Wave w = this_wave();
    P::load_model(w, s.plant, reinterpret_cast<const typename P::Model*>(model));
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        PDDP_FOR(e, NX) s.x[e] = x[(size_t)i * NX + e];
        PDDP_FOR(e, NU) s.u[e] = u[(size_t)i * NU + e];
        wsync();
        if (what == 0) { P::dynamics(w, s.plant, qdd, s.x, s.u); PDDP_FOR(e, NP) out[(size_t)i * NP + e] = qdd[e]; }
        else if (what == 1) { P::gradient(w, s.plant, s.pgrad, dq, qdd, s.x, s.u); PDDP_FOR(e, NP * NM) out[(size_t)i * NP * NM + e] = dq[e]; }
        else if (what == 2) { integrator_step<P, INTEG, T>(w, s.plant, is, xn, s.x, s.u, dt); PDDP_FOR(e, NX) out[(size_t)i * NX + e] = xn[e]; }
        else { integrator_gradient<P, INTEG, T>(w, s.plant, s.pgrad, s.integ, out + (size_t)i * NX * NM, s.x, s.u, dt); }
        wsync();
    }
}

// ---------------------------------------------------------------------------------------------- lock-step plant simulator (sim.hpp): one wave
template <typename PD, int INTEG, typename T>
__global__ __launch_bounds__(64) void k_plant_sim(const void* model, PlantSimArgs<T> a) {
    __shared__ PlantSimScratch<PD, T> s;
    plant_sim_body<PD, INTEG, T>(this_wave(), s, model, a);
}
// tool point of `count` states: grid (count), block 64
template <typename P, typename T>
__global__ __launch_bounds__(64) void k_ee_pos(const void* model, T ee_z, const T* x, T* out) {
    __shared__ typename P::Scratch plant;
    __shared__ EeScratch<T> ee;
    __shared__ T xs[P::NX], us[P::NU], qdd[P::NPOS];
    ee_pos_body<P, T>(this_wave(), plant, ee, xs, us, qdd, model, ee_z, x + (size_t)blockIdx.x * P::NX, out + (size_t)blockIdx.x * 6);
}

}  // namespace pddp
