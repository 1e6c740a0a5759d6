// __global__ entry points: one DDP sweep = k_bp -> k_fp -> k_ls -> k_nis on one stream, no host sync.
// grid.y indexes the independent problems of the batch; every workgroup is made of whole 64-lane waves, each wave
// owning one unit of work (see pddp_common.hpp).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include "bodies.hpp"
#include "fp_lg.hpp"
#include "nis_lg.hpp"
#include "bp_lg.hpp"
#include "bp_cl.hpp"
#include "bp_mq.hpp"
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
#define PDDP_UNROLL _Pragma("unroll")
// Forward pass of the scalar closed-form plants with the device full, second form (BASELINE configs[4], the quadrotor).  k_fp_ts above keeps thread = (problem, step
// size) and the reference's arithmetic, but (counters in profiles/r04_quad.md: waves spend 55 % of their cycles in s_waitcnt, 22 % issuing)
//   * every thread fetches its problem's K_k, (A - B K)_k, x_k, u_k, du_k, (B du)_k itself: ~90 vector-memory instructions per step, the same 16 addresses A times over;
//   * each step's loads are issued right after the previous step's stores, and gfx9's single in-order vmcnt makes the wait for the loads a wait for the store
//     acknowledgements as well -- one write round trip on every step's critical path.
// Here the wavefront = 64 / A problems x A step sizes fetches each knot's operands ONCE, coalesced, ONE STEP AHEAD of their use (issued before the step's arithmetic,
// parked in registers, written to a double-buffered LDS stage after it), and the lanes read them back as broadcasts; the step's stores come last, so that what the
// next step waits for is a load that was issued a whole step earlier.  Two loops as before -- since round 6 two LAUNCHES (fp_cf_body PART 0 / 1: k_sweep_cf, k_fp_cf) --:
// the linear sweep (forward_sweep, fp.hpp) for the segments' start states, which go to the candidates' records and are read back from there, then the rollouts
// (forward_sim_segment).  Per thread the arithmetic is that of those two functions, operation by operation.  No barriers: a one-wave block, LDS operations of a wave
// retire in order (wsync).
template <typename P, typename T, int A>
struct FpCfStage {
    static constexpr int PW = 64 / A, NX = P::NX, NU = P::NU;
    union {
        struct { T M[PW][NX * NX], xp[PW][NX], Bdu[PW][NX], d[PW][NX]; } sw;
        struct { T K[PW][NX * NU], xp[PW][NX], up[PW][NU], du[PW][NU]; } ro;
    };
};
template <typename T, int PW, int L>
struct FpCfRegs { static constexpr int R = (PW * L + 63) / 64; T v[R]; };
// lane's share of the PW problems' L-vectors of a knot: element idx = lane + 64 j  ->  problem idx / L, entry idx % L.  The arrays are [problem][knot][L], so relative to the
// wave's first problem at the knot -- a wave-uniform address the scalar unit forms -- a lane's element sits at a loop-invariant BYTE offset that fits 32 bits: one register per
// fetch and no 64-bit vector arithmetic in the step (round 6: the per-lane 64-bit pointers of every fetch were what spilled when the kernel was asked to fit three or four waves).
template <typename T, int PW, int L>
struct FpCfLane {
    static constexpr int R = (PW * L + 63) / 64;
    unsigned off[R];
    __device__ __forceinline__ void init(size_t per_problem, int pb0, int batch, int lane) {
        PDDP_UNROLL for (int j = 0; j < R; j++) {
            const int idx = lane + 64 * j, p = idx / L, e = idx - p * L;
            const int q = ((pb0 + p < batch) ? pb0 + p : batch - 1) - pb0;
            off[j] = (idx < PW * L) ? (unsigned)(((size_t)q * per_problem + e) * sizeof(T)) : 0u;
        }
    }
};
// BUFFER instructions (as in bp_mq.hpp / bp_mfma.hpp): a wave-uniform resource on the wave's first problem, the knot's position a scalar byte offset, the lane's share the
// 32-bit vector offset -- unconditional (a lane without a share reads offset 0 and drops the value: no exec-mask juggling around every load)
template <typename T, int PW, int L>
__device__ __forceinline__ void fp_cf_fetch(FpCfRegs<T, PW, L>& r, const FpCfLane<T, PW, L>& ln, __amdgpu_buffer_rsrc_t rs, int k) {
    const unsigned so = (unsigned)k * (unsigned)(L * sizeof(T));
    PDDP_UNROLL for (int j = 0; j < FpCfRegs<T, PW, L>::R; j++) r.v[j] = mx_bld<T>(rs, ln.off[j], so);
}
// n consecutive values at (resource, lane byte offset, scalar byte offset) as 16-byte pieces
typedef unsigned cf_u4 __attribute__((ext_vector_type(4)));
template <typename T, int n>
__device__ __forceinline__ void cf_bst_vec(__amdgpu_buffer_rsrc_t rs, unsigned vbyte, unsigned sbyte, const T* v) {
    constexpr int W = 16 / sizeof(T);
    if constexpr (n % W == 0) {
        PDDP_UNROLL for (int j = 0; j < n / W; j++) {
            T q[W];
            PDDP_UNROLL for (int e = 0; e < W; e++) q[e] = v[j * W + e];
            cf_u4 w; __builtin_memcpy(&w, q, 16);
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, vbyte + 16u * j, sbyte, 0);
        }
    } else {                                                                 // (the cart-pole's 5-element records: element by element)
        PDDP_UNROLL for (int j = 0; j < n; j++) mx_bst<T>(rs, v[j], vbyte + (unsigned)(j * sizeof(T)), sbyte);
    }
}
template <typename T, int n>
__device__ __forceinline__ void cf_bld_vec(T* v, __amdgpu_buffer_rsrc_t rs, unsigned vbyte, unsigned sbyte) {
    constexpr int W = 16 / sizeof(T);
    if constexpr (n % W == 0) {
        PDDP_UNROLL for (int j = 0; j < n / W; j++) {
            const cf_u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, vbyte + 16u * j, sbyte, 0);
            T q[W]; __builtin_memcpy(q, &w, 16);
            PDDP_UNROLL for (int e = 0; e < W; e++) v[j * W + e] = q[e];
        }
    } else {
        PDDP_UNROLL for (int j = 0; j < n; j++) v[j] = mx_bld<T>(rs, vbyte + (unsigned)(j * sizeof(T)), sbyte);
    }
}
template <typename T, int PW, int L>
__device__ __forceinline__ void fp_cf_put(const FpCfRegs<T, PW, L>& r, T* dst, int lane) {
    PDDP_UNROLL for (int j = 0; j < FpCfRegs<T, PW, L>::R; j++) { const int idx = lane + 64 * j; if (idx < PW * L) dst[idx] = r.v[j]; }
}
// n consecutive values to a 16-byte aligned address, as 16-byte stores (the compiler cannot see the alignment of base + k * n: the buffers come from hipMalloc and
// n * sizeof(T) is a multiple of 16 for the plants this kernel serves)
template <typename T, int n>
__device__ __forceinline__ void cf_store_vec(T* dst, const T* v) {
    if constexpr ((n * sizeof(T)) % 16 == 0) {
        typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int W = 16 / sizeof(T);
        PDDP_UNROLL for (int j = 0; j < n / W; j++) { V q; PDDP_UNROLL for (int e = 0; e < W; e++) q[e] = v[j * W + e]; reinterpret_cast<V*>(dst)[j] = q; }
    } else {
        PDDP_UNROLL for (int j = 0; j < n; j++) dst[j] = v[j];
    }
}
template <typename T, int n>
__device__ __forceinline__ void cf_load_vec(T* v, const T* src) {
    if constexpr ((n * sizeof(T)) % 16 == 0) {
        typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int W = 16 / sizeof(T);
        PDDP_UNROLL for (int j = 0; j < n / W; j++) { const V q = reinterpret_cast<const V*>(src)[j]; PDDP_UNROLL for (int e = 0; e < W; e++) v[j * W + e] = q[e]; }
    } else {
        PDDP_UNROLL for (int j = 0; j < n; j++) v[j] = src[j];
    }
}
// integrator_step (integrators.hpp) for one thread with everything in registers
template <typename P, int INTEG, typename T>
__device__ __forceinline__ void cf_integrator_step(T* xn, const T* x, const T* u, T dt) {
    constexpr int NP = P::NPOS, NX = P::NX;
    T q1[NP];
    P::dynamics_eval(q1, x, u);
    if constexpr (INTEG == 1) {
        PDDP_UNROLL for (int i = 0; i < NP; i++) { xn[i] = x[i] + dt * x[i + NP]; xn[i + NP] = x[i + NP] + dt * q1[i]; }
    } else if constexpr (INTEG == 2) {
        T x2[NX];
        PDDP_UNROLL for (int i = 0; i < NP; i++) { x2[i] = x[i] + T(0.5) * dt * x[i + NP]; x2[i + NP] = x[i + NP] + T(0.5) * dt * q1[i]; }
        P::dynamics_eval(q1, x2, u);
        PDDP_UNROLL for (int i = 0; i < NP; i++) { xn[i] = x[i] + dt * x[i + NP]; xn[i + NP] = x[i + NP] + dt * q1[i]; }
    } else {
        T x2[NX], x3[NX], q2[NP], q3[NP];
        PDDP_UNROLL for (int i = 0; i < NP; i++) { x2[i] = x[i] + T(0.5) * dt * x[i + NP]; x2[i + NP] = x[i + NP] + T(0.5) * dt * q1[i]; }
        P::dynamics_eval(q2, x2, u);
        PDDP_UNROLL for (int i = 0; i < NP; i++) {
            x3[i] = x[i] + dt * (T(2) * x2[i + NP] - x[i + NP]);
            x3[i + NP] = x[i + NP] + dt * (T(2) * q2[i] - q1[i]);
        }
        P::dynamics_eval(q3, x3, u);
        PDDP_UNROLL for (int i = 0; i < NP; i++) {
            xn[i] = x[i] + (dt / T(6)) * (x[i + NP] + T(4) * x2[i + NP] + x3[i + NP]);
            xn[i + NP] = x[i + NP] + (dt / T(6)) * (q1[i] + T(4) * q2[i] + q3[i]);
        }
    }
}
#ifndef PDDP_SWEEPCF_PIN
#define PDDP_SWEEPCF_PIN 1     // columns of the map whose reads may be in flight together in k_sweep_cf's product loop (1: every column's products pinned behind its reads)
#endif
// PART 0: the linear sweep alone (k_sweep_cf), PART 1: the rollouts alone (k_fp_cf).  Until round 5 one kernel ran both loops; the register allocation of the two together
// is what kept it at two waves per SIMD -- asked for three or four, the compiler spilled inside the SWEEP loop (the rollout loop fits 128 registers) -- so they are two
// launches now, each with its own occupancy; the sweep leaves the segments' start states in the candidates' records, where the rollouts read them as before.
template <typename P, int INTEG, typename T, int A, int PART>
__device__ __forceinline__ void fp_cf_body(FpCfStage<P, T, A>* stage, T (*goal_s)[P::NX], Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, int batch) {
    constexpr int NX = P::NX, NU = P::NU, PW = 64 / A;
    static_assert(64 % A == 0 && PW * NX <= 64, "a wavefront holds whole problems; one fetch per lane for the state");
    const int lane = threadIdx.x, grp = lane / A, a_idx = lane - grp * A, pb0 = blockIdx.x * PW, N = dm.N;
    const int pb_raw = pb0 + grp, pb = pb_raw < batch ? pb_raw : batch - 1;
    const bool live = pb_raw < batch && fp_active<T>(b, dm, pb);
    const size_t slot = (size_t)pb * A + a_idx;
    // candidates leave as RECORDS state | control in the knot-major array xw[problem][knot][step size][NX + NU] (solver_state.hpp; the arm's thread-lane rollouts keep theirs
    // the same way): the A lanes of a problem fill one contiguous run of A records per step -- whole cache lines -- where the candidate-major slots of xs / us got 48- and
    // 16-byte pieces 12 KB apart (counters: 8.7 GB written per launch for 4.3 GB of candidates).  The setup kernels adopt the accepted candidate from the records; the
    // phase hook runs k_fp_ts (candidate-major xs / us, the reference's arrays) and copies them into xw (k_cand_to_xw).
    constexpr int REC = NX + NU;
    // addresses: a wave-uniform buffer resource on the wave's first problem + the knot's scalar byte offset + this lane's loop-invariant 32-bit byte offset
    const __amdgpu_buffer_rsrc_t r_rec = mx_rsrc(b.xw + (size_t)pb0 * N * A * REC), r_d = mx_rsrc(b.ds + (size_t)pb0 * A * N * NX);
    const unsigned rec_off = (unsigned)(((size_t)(pb - pb0) * N * A + a_idx) * REC * sizeof(T));      // record of knot k of this lane: + k * A * REC elements
    constexpr unsigned rstr = (unsigned)(A * REC * sizeof(T));
    const unsigned d_off = (unsigned)(((size_t)(pb - pb0) * A + a_idx) * N * NX * sizeof(T));         // boundary defect of knot k: + k * NX elements
    // the goal of the wave's problems lives in LDS (round 6): the cost of every step read it from memory -- twelve loads per step queued behind the previous step's stores
    // (one in-order vmcnt), and a dozen registers' worth of addresses and values kept alive across the three dynamics evaluations
    { const int p = lane / NX, q = (pb0 + p < batch) ? pb0 + p : batch - 1; if (lane < PW * NX) (&goal_s[0][0])[lane] = b.xGoal[(size_t)q * NX + (lane - p * NX)]; }
    const T* xg = goal_s[grp];
    const T alpha = b.alpha[a_idx];
    // the current trajectory sits in one half of xb per PROBLEM (state.cur): this lane's share of the state fetch is entry e of problem p
    const __amdgpu_buffer_rsrc_t r_xcur = mx_rsrc(b.xb + (size_t)pb0 * 2 * N * NX);
    unsigned xcur_off;
    { const int p = lane / NX, q = (pb0 + p < batch) ? pb0 + p : batch - 1; xcur_off = (unsigned)((((size_t)(q - pb0) * 2 + b.state[q].cur) * N * NX + (lane - p * NX)) * sizeof(T)); }
    const bool xlane = lane < PW * NX;
    T xstart[NX];
    // ---- the linear sweep: x_{k+1} = xcur_{k+1} + (A - B K)_k (x_k - xcur_k) - alpha (B du)_k + [boundary] d_k ----
    if constexpr (PART == 0) {
        if (dm.M <= 1) return;
        // (tried and measured, round 6 -- profiles/r06_quad.md: the operands requested TWO knots ahead instead of one: 0.86 -> 0.85 ms, nothing; the map kept in registers across
        // the 16 lanes of a problem's DPP row with v_mul_f32_dpp row_newbcast instead of LDS broadcasts: same bits, 0.86 -> 0.95 ms)
        FpCfRegs<T, PW, NX * NX> rM; FpCfRegs<T, PW, NX> rBdu, rd; T rxp;
        FpCfLane<T, PW, NX * NX> lM; FpCfLane<T, PW, NX> lV;
        lM.init((size_t)N * NX * NX, pb0, batch, lane); lV.init((size_t)N * NX, pb0, batch, lane);
        const __amdgpu_buffer_rsrc_t r_M = mx_rsrc(b.ApBK + (size_t)pb0 * N * NX * NX), r_Bdu = mx_rsrc(b.Bdu + (size_t)pb0 * N * NX), r_dc = mx_rsrc(b.dcur + (size_t)pb0 * N * NX);
        auto fetch = [&](int k) {
            fp_cf_fetch<T, PW, NX * NX>(rM, lM, r_M, k);
            rxp = mx_bld<T>(r_xcur, xcur_off, (unsigned)k * (unsigned)(NX * sizeof(T)));
            fp_cf_fetch<T, PW, NX>(rBdu, lV, r_Bdu, k);
            fp_cf_fetch<T, PW, NX>(rd, lV, r_dc, k);
        };
        auto put = [&](FpCfStage<P, T, A>& sg) {
            fp_cf_put<T, PW, NX * NX>(rM, &sg.sw.M[0][0], lane);
            if (xlane) (&sg.sw.xp[0][0])[lane] = rxp;
            fp_cf_put<T, PW, NX>(rBdu, &sg.sw.Bdu[0][0], lane); fp_cf_put<T, PW, NX>(rd, &sg.sw.d[0][0], lane);
        };
        fetch(0); put(stage[0]); wsync();
        T xk[NX];
        PDDP_UNROLL for (int i = 0; i < NX; i++) xk[i] = stage[0].sw.xp[grp][i];
        for (int k = 0; k < N - 1; k++) {
            const FpCfStage<P, T, A>& sc = stage[k & 1];
            FpCfStage<P, T, A>& sn = stage[(k + 1) & 1];
            fetch(k + 1);
            T val[NX];
            PDDP_UNROLL for (int r = 0; r < NX; r++) val[r] = 0;
            // column i of the map as explicit 16-byte LDS reads with its products pinned behind them: left to itself the compiler merges the 144 scalar reads into 36 vector
            // reads and places ALL of them in front of the first product (144 live registers: what kept this loop at two waves per SIMD)
            PDDP_UNROLL for (int i = 0; i < NX; i++) {
                const T dxs = xk[i] - sc.sw.xp[grp][i];
                if constexpr ((NX * sizeof(T)) % 16 == 0) {
                    typedef T MV __attribute__((ext_vector_type(16 / sizeof(T))));
                    constexpr int W = 16 / sizeof(T);
                    const MV* col = reinterpret_cast<const MV*>(&sc.sw.M[grp][NX * i]);
                    PDDP_UNROLL for (int q = 0; q < NX / W; q++) { const MV m = col[q]; PDDP_UNROLL for (int e = 0; e < W; e++) val[q * W + e] += m[e] * dxs; }
                    // (the sums are only used behind the next LDS stores, in another basic block: the optimiser SINKS all 288 operations there and leaves the reads here --
                    // an empty statement that "uses" the partial sums keeps every column's products next to its reads)
                    if (i % PDDP_SWEEPCF_PIN == PDDP_SWEEPCF_PIN - 1) { PDDP_UNROLL for (int r = 0; r < NX; r++) asm volatile("" : "+v"(val[r])); }
                } else {
                    PDDP_UNROLL for (int r = 0; r < NX; r++) val[r] += sc.sw.M[grp][r + NX * i] * dxs;
                }
            }
            put(sn); wsync();
            const bool bnd = dm.on_defect_boundary(k);
            PDDP_UNROLL for (int r = 0; r < NX; r++) {
                T xv = sn.sw.xp[grp][r];
                xv += -alpha * sc.sw.Bdu[grp][r] + val[r] + (bnd ? sc.sw.d[grp][r] : T(0));
                xk[r] = xv;
            }
            if (bnd && live) cf_bst_vec<T, NX>(r_rec, rec_off, (unsigned)(k + 1) * rstr, xk);
            wsync();
        }
    }
    // ---- the rollouts ----
    if constexpr (PART == 1) {
    FpCfRegs<T, PW, NX * NU> rK; FpCfRegs<T, PW, NU> rup, rdu; T rxp;
    FpCfLane<T, PW, NX * NU> lK; FpCfLane<T, PW, NU> lU;
    lK.init((size_t)N * NX * NU, pb0, batch, lane); lU.init((size_t)N * NU, pb0, batch, lane);
    const __amdgpu_buffer_rsrc_t r_K = mx_rsrc(b.KT + (size_t)pb0 * N * NX * NU), r_up = mx_rsrc(b.ucur + (size_t)pb0 * N * NU), r_du = mx_rsrc(b.du + (size_t)pb0 * N * NU);
    auto fetch = [&](int k) {
        fp_cf_fetch<T, PW, NX * NU>(rK, lK, r_K, k);
        rxp = mx_bld<T>(r_xcur, xcur_off, (unsigned)k * (unsigned)(NX * sizeof(T)));
        fp_cf_fetch<T, PW, NU>(rup, lU, r_up, k);
        fp_cf_fetch<T, PW, NU>(rdu, lU, r_du, k);
    };
    auto put = [&](FpCfStage<P, T, A>& sg) {
        fp_cf_put<T, PW, NX * NU>(rK, &sg.ro.K[0][0], lane);
        if (xlane) (&sg.ro.xp[0][0])[lane] = rxp;
        fp_cf_put<T, PW, NU>(rup, &sg.ro.up[0][0], lane); fp_cf_put<T, PW, NU>(rdu, &sg.ro.du[0][0], lane);
    };
    fetch(0); put(stage[0]); wsync();
    T cost_k[kTsMaxN];
    T x[NX], u[NU];
    PDDP_UNROLL for (int i = 0; i < NX; i++) x[i] = stage[0].ro.xp[grp][i];
    T dmx = 0;
    for (int k = 0; k < N - 1; k++) {
        const FpCfStage<P, T, A>& sc = stage[k & 1];
        fetch(k + 1);
        const bool bnd = dm.on_defect_boundary(k);
        if (bnd) cf_bld_vec<T, NX>(xstart, r_rec, rec_off, (unsigned)(k + 1) * rstr);      // the next segment's start state, from the sweep above
        T dx[NX], xn[NX];
        PDDP_UNROLL for (int i = 0; i < NX; i++) dx[i] = x[i] - sc.ro.xp[grp][i];
        PDDP_UNROLL for (int r = 0; r < NU; r++) {                     // u = ucur - alpha du - K (x - xcur)      (computeControlKT)
            T Kdx = 0;
            PDDP_UNROLL for (int c = 0; c < NX; c++) Kdx += sc.ro.K[grp][c + r * NX] * dx[c];
            T uv = sc.ro.up[grp][r];
            uv -= alpha * sc.ro.du[grp][r] + Kdx;
            u[r] = uv;
        }
        cost_k[k] = P::cost(cw, x, u, xg, k, N);
        cf_integrator_step<P, INTEG, T>(xn, x, u, dt);
        put(stage[(k + 1) & 1]); wsync();
        T rec[REC];                                         // this knot's record: the state the step started from and its control
        PDDP_UNROLL for (int i = 0; i < NX; i++) rec[i] = x[i];
        PDDP_UNROLL for (int r = 0; r < NU; r++) rec[NX + r] = u[r];
        if (live) cf_bst_vec<T, REC>(r_rec, rec_off, (unsigned)k * rstr, rec);
        if (bnd) {                                          // last step of a non-final segment: defect against the next start state, which the next segment starts from
            T sdef = 0, dv[NX];
            PDDP_UNROLL for (int i = 0; i < NX; i++) { dv[i] = xn[i] - xstart[i]; sdef += tabs(dv[i]); x[i] = xstart[i]; }
            dmx = tmax(dmx, sdef);
            if (live) cf_bst_vec<T, NX>(r_d, d_off, (unsigned)k * (unsigned)(NX * sizeof(T)), dv);
        } else {
            PDDP_UNROLL for (int i = 0; i < NX; i++) x[i] = xn[i];
        }
    }
    {                                                       // final knot: terminal cost, and its (unused) control is carried along
        const FpCfStage<P, T, A>& sc = stage[(N - 1) & 1];
        PDDP_UNROLL for (int r = 0; r < NU; r++) u[r] = sc.ro.up[grp][r];
        T rec[REC];
        PDDP_UNROLL for (int i = 0; i < NX; i++) rec[i] = x[i];
        PDDP_UNROLL for (int r = 0; r < NU; r++) rec[NX + r] = u[r];
        if (live) cf_bst_vec<T, REC>(r_rec, rec_off, (unsigned)(N - 1) * rstr, rec);
        cost_k[N - 1] = P::cost(cw, x, u, xg, N - 1, N);
        dmx = tmax(dmx, T(0));
    }
    if (!live) return;
    const T J = tree_sum<T>(serial_wave(), cost_k, N);
    b.J[slot] = J; b.dmax[slot] = dmx;
    }
}
#ifndef PDDP_FPCF_WAVES
#define PDDP_FPCF_WAVES 4      // resident waves per SIMD the float rollout kernel is compiled for (128 registers: the step loop holds no spill)
#endif
#ifndef PDDP_SWEEPCF_WAVES
#define PDDP_SWEEPCF_WAVES 2
#endif
template <typename P, int INTEG, typename T, int A>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? PDDP_FPCF_WAVES : 1))) void k_fp_cf(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int batch) {
    __shared__ FpCfStage<P, T, A> stage[2];
    __shared__ T goal_s[64 / A][P::NX];
    fp_cf_body<P, INTEG, T, A, 1>(stage, goal_s, b, dm, cw, dt, batch);
}
template <typename P, int INTEG, typename T, int A>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? PDDP_SWEEPCF_WAVES : 1))) void k_sweep_cf(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int batch) {
    __shared__ FpCfStage<P, T, A> stage[2];
    __shared__ T goal_s[64 / A][P::NX];
    fp_cf_body<P, INTEG, T, A, 0>(stage, goal_s, b, dm, cw, dt, batch);
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

// Knot-batched setup for the scalar closed-form plants with the RK3 Jacobian (BASELINE configs[4], the quadrotor).  k_nis_gl above runs the plug-in's scalar code on
// 3 of every 16 lanes (4.4e9 vector instructions per sweep at 16384 problems, counters in profiles/r04_quad.md: the kernel is instruction-issue bound) and assembles
// [A B] entry by entry with run-time indices.  Here one wavefront takes KB consecutive knots in two phases:
//   1  the three stage gradients -> LDS (dqdd of the three stages, entry-major and knot-minor; conflict-free for the writes of phase 1 AND for the column reads of
//      phase 2); g_k and the winner's copies by the same lanes.
//      Round 4: lane = knot, the stages one after the other (each call also returns the stage's qdd, from which the next stage's state follows) -- 16 of 64 lanes live, because
//      3 x 96 words of LDS per knot set the number of resident knots, and with it the occupancy (KB = 16 beat 32 and 64: 2.15 / 2.6 / 3.6 ms).
//      Round 6: lane = (stage, knot), 3 KB <= 64: stage s's lane first forms ITS stage state with s evaluations of the plug-in's dynamics() -- the same functions on the same
//      operands as rk3_stage_chain (integrators.hpp), and a scalar plug-in's dynamics() returns the numbers its dynamicsGradient() returns as qdd (plants.hpp
//      scalar_plugin_qdd_is_dynamics) -- then all three stage gradients run as ONE pass of the gradient code: 2 dynamics + 1 gradient per wavefront instead of 3 gradients,
//      48 (KB = 16) or 60 (KB = 20) lanes live.  The same bits in every output (tests/test_closed_form_serial.py).
//   2  16 lanes = one knot, lane = COLUMN of [A B], four knots per pass: the column of T1, T2 (rk3_assemble's sums, in its order) stays in registers, every index but
//      the column is a compile-time constant, the factors (0.5 dt d1 + delta), (2 dt T1 - dt d1 + delta) are formed once per column instead of once per row
// Columns of dqdd that are zero for EVERY state of the plant (GradZeroCols: the quadrotor's accelerations do not depend on its position or its linear velocity -- 6 of 16) are
// neither staged nor multiplied: their products are exact zeros, adding one changes no finite sum (LDS per knot 288 -> 180 words: three resident waves per SIMD instead of two).
// No barrier inside a phase; one between them (a one-wave block).
template <typename P> struct GradZeroCols { static constexpr unsigned value = 0u; };                                    // bit c: column c of dynamicsGradient's dqdd is identically zero
template <typename T> struct GradZeroCols<QuadPlant<T>> { static constexpr unsigned value = 0x1C7u; };                  // x y z and their rates (plants/dynamics_quad.cuh:75-169 never writes them; tests/test_closed_form_pins.py)
template <typename P> struct GradCols {
    static constexpr int NM = P::NX + P::NU;
    static constexpr unsigned zero = GradZeroCols<P>::value;
    static constexpr bool live(int c) { return !((zero >> c) & 1u); }
    static constexpr int index(int c) { int n = 0; for (int i = 0; i < c; i++) n += live(i) ? 1 : 0; return n; }     // position of column c among the staged ones
    static constexpr int count = index(NM);
};
template <typename P, typename T, int KB>
struct NisKbLds { T d[GradCols<P>::count * P::NPOS][3][KB + 1]; };
template <typename P, typename T, int KB>
__global__ __launch_bounds__(64) void k_nis_kb(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, int mode, int batch) {
    constexpr int NP = P::NPOS, NX = P::NX, NU = P::NU, NM = NX + NU, ND = NP * NM;
    using GC = GradCols<P>;
    constexpr bool kStageLanes = 3 * KB <= 64;          // phase 1 with one lane per (stage, knot)
    static_assert(NM <= 16 && KB % 4 == 0 && KB <= 64, "one column of [A B] per lane of a 16-lane group");
    __shared__ NisKbLds<P, T, KB> lds;
    const int lane = threadIdx.x, N = dm.N, total = batch * N;
    // ---- phase 1 ----
    {
        const int stage = kStageLanes ? lane / KB : 0, j = kStageLanes ? lane - stage * KB : lane;
        const int inst = blockIdx.x * KB + j;
        if (lane < (kStageLanes ? 3 * KB : KB) && inst < total) {
            const int pb = inst / N, k = inst - pb * N;
            const SolverState<T>& st = b.state[pb];
            const bool moved = mode == 1 || st.accepted == 1;
            if (moved) {
                T* xc = b.xb + ((size_t)pb * 2 + st.cur) * N * NX + (size_t)k * NX;
                T* uc = b.ucur + ((size_t)pb * N + k) * NU;
                // the winner's copies and the cost gradient: the work of ONE lane per knot, spread over the knot's three stage lanes
                const bool copy_x = stage == 0, copy_u = !kStageLanes || stage == 1, do_g = !kStageLanes || stage == 2;
                T x[NX], u[NU];
                if (mode == 0) {
                    const size_t slot = (size_t)pb * dm.A + st.alphaIndex;
                    const T* xw = b.xw ? b.xw + (((size_t)pb * N + k) * dm.A + st.alphaIndex) * (NX + NU) : b.xs + (slot * N + k) * NX;      // the accepted candidate's record, or its slots of xs / us
                    const T* uw = b.xw ? xw + NX : b.us + (slot * N + k) * NU;
                    // (16-byte pieces: every knot block of these arrays starts on a 16-byte boundary when its size is a multiple of 16 bytes -- cf_load_vec / cf_store_vec)
                    cf_load_vec<T, NX>(x, xw); cf_load_vec<T, NU>(u, uw);
                    if (copy_x) cf_store_vec<T, NX>(xc, x);
                    if (copy_u) {
                        cf_store_vec<T, NU>(uc, u);
                        if (dm.M > 1 && dm.on_defect_boundary(k)) {
                            const T* dw = b.ds + (slot * N + k) * NX; T* dc = b.dcur + ((size_t)pb * N + k) * NX;
                            T dv[NX]; cf_load_vec<T, NX>(dv, dw); cf_store_vec<T, NX>(dc, dv);
                        }
                    }
                } else {
                    cf_load_vec<T, NX>(x, xc); cf_load_vec<T, NU>(u, uc);
                }
                if (mode == 1 || !st.done) {
                    if (do_g) {
                        const T* xg = b.xGoal + (size_t)pb * NX;
                        T* gk = b.g + ((size_t)pb * N + k) * NM;
                        if constexpr (P::kPluginCost) P::cost_grad(cw, b.H + ((size_t)pb * N + k) * NM * NM, gk, x, u, xg, k, N);
                        else {
                            T gv[NM], xgv[NX];
                            cf_load_vec<T, NX>(xgv, xg);
                            PDDP_UNROLL for (int i = 0; i < NM; i++) gv[i] = P::weight(cw, i, k, N) * (i < NX ? (x[i] - xgv[i]) : u[i - NX]);
                            cf_store_vec<T, NM>(gk, gv);
                        }
                    }
                    if (k < N - 1) {
                        auto stage_to_lds = [&](int s_, const T* dd) {
                            PDDP_UNROLL for (int c = 0; c < NM; c++) {
                                if (!GC::live(c)) continue;
                                PDDP_UNROLL for (int r = 0; r < NP; r++) lds.d[GC::index(c) * NP + r][s_][j] = dd[c * NP + r];
                            }
                        };
                        T dd[ND], q1[NP], q2[NP], q3[NP], xm[NX];
                        if constexpr (kStageLanes) {
                            // this lane's stage state, from the plug-in's dynamics (rk3_stage_chain's expressions, the reference's stage-state quirk included:
                            // integrators.cuh:182,190-191)
                            PDDP_UNROLL for (int i = 0; i < NX; i++) xm[i] = x[i];
                            if (stage >= 1) {
                                P::dynamics_eval(q1, x, u);
                                if (stage == 1) {
                                    PDDP_UNROLL for (int i = 0; i < NP; i++) { xm[i] = x[i] + T(0.5) * dt * x[i + NP]; xm[i + NP] = x[i] + T(0.5) * dt * q1[i]; }
                                } else {
                                    T x1[NX];
                                    PDDP_UNROLL for (int i = 0; i < NP; i++) { x1[i] = x[i] + T(0.5) * dt * x[i + NP]; x1[i + NP] = x[i] + T(0.5) * dt * q1[i]; }
                                    P::dynamics_eval(q2, x1, u);
                                    PDDP_UNROLL for (int i = 0; i < NP; i++) {
                                        const T v1 = x1[i + NP];
                                        xm[i + NP] = x[i] + dt * q1[i] + T(2) * dt * q2[i];
                                        xm[i] = x[i] + dt * x[i + NP] + T(2) * dt * v1;
                                    }
                                }
                            }
                            P::gradient_eval(dd, q3, xm, u);
                            stage_to_lds(stage, dd);
                        } else {
                            P::gradient_eval(dd, q1, x, u);
                            stage_to_lds(0, dd);
                            PDDP_UNROLL for (int i = 0; i < NP; i++) { xm[i] = x[i] + T(0.5) * dt * x[i + NP]; xm[i + NP] = x[i] + T(0.5) * dt * q1[i]; }
                            P::gradient_eval(dd, q2, xm, u);
                            stage_to_lds(1, dd);
                            PDDP_UNROLL for (int i = 0; i < NP; i++) {                                  // xm2 from xm1's velocity half BEFORE it is overwritten (rk3_stage_chain)
                                const T v1 = xm[i + NP];
                                xm[i + NP] = x[i] + dt * q1[i] + T(2) * dt * q2[i];
                                xm[i] = x[i] + dt * x[i + NP] + T(2) * dt * v1;
                            }
                            P::gradient_eval(dd, q3, xm, u);
                            stage_to_lds(2, dd);
                        }
                    }
                }
            }
        }
    }
    wsync();
    // ---- phase 2: 16 lanes = knot, lane = column ----
    const int grp = lane >> 4, ky = lane & 15;
    const int c2 = ky < NP ? ky : (ky < NX ? ky - NP : ky);
    // where this lane's two columns of dqdd sit among the staged ones (-1: a column that is identically zero)
    int ci_ky = -1, ci_c2 = -1;
    PDDP_UNROLL for (int c = 0; c < NM; c++) { if (GC::live(c)) { if (ky == c) ci_ky = GC::index(c); if (c2 == c) ci_c2 = GC::index(c); } }
    const int ri_ky = (ci_ky < 0 ? 0 : ci_ky) * NP, ri_c2 = (ci_c2 < 0 ? 0 : ci_c2) * NP;
    for (int rnd = 0; rnd < KB / 4; rnd++) {
        const int j = rnd * 4 + grp, inst = blockIdx.x * KB + j;
        if (inst >= total || ky >= NM) continue;
        const int pb = inst / N, k = inst - pb * N;
        const SolverState<T>& st = b.state[pb];
        if (!(mode == 1 || (st.accepted == 1 && !st.done))) continue;
        if constexpr (!P::kPluginCost) {
            if (mode == 1) {                                                            // the constant Hessian: written when the problem is loaded (nis_knot, nis.hpp)
                T* Hk = b.H + ((size_t)pb * N + k) * NM * NM + (size_t)ky * NM;
                T hv[NM];
                PDDP_UNROLL for (int i = 0; i < NM; i++) hv[i] = (i == ky) ? P::weight(cw, i, k, N) : T(0);
                cf_store_vec<T, NM>(Hk, hv);
            }
        }
        if (k >= N - 1) continue;
        const T hdt = T(0.5) * dt, dt2 = T(2) * dt;
        T d1c[NP], T1[NX], T2[NX];
        PDDP_UNROLL for (int r = 0; r < NP; r++) { const T v = lds.d[ri_ky + r][0][j]; d1c[r] = ci_ky < 0 ? T(0) : v; }
        const T lead = ky < NP ? (hdt * T(0) + T(1)) : (hdt * T(1) + T(0));
        T f[NP];
        PDDP_UNROLL for (int i = 0; i < NP; i++) f[i] = hdt * d1c[i] + T(ky == i + NP ? 1 : 0);
        PDDP_UNROLL for (int kx = 0; kx < NP; kx++) T1[kx] = T(1) * (hdt * d1c[kx] + T(ky == kx + NP ? 1 : 0)) + T(0);
        PDDP_UNROLL for (int r = 0; r < NP; r++) {
            const T selv = lds.d[ri_c2 + r][1][j];
            const T sel = ci_c2 < 0 ? T(0) : selv;
            T val = ky < NX ? sel * lead : T(0);
            PDDP_UNROLL for (int i = 0; i < NP; i++) { if (GC::live(i + NP)) val += lds.d[GC::index(i + NP) * NP + r][1][j] * f[i]; }      // (a zero column adds an exact zero)
            T1[r + NP] = val + (ky < NX ? T(0) : sel);
        }
        T gq[NX];
        PDDP_UNROLL for (int i = 0; i < NP; i++) gq[i] = dt2 * T1[i] - dt * T(i + NP == ky ? 1 : 0) + T(ky == i ? 1 : 0);
        PDDP_UNROLL for (int i = NP; i < NX; i++) gq[i] = dt2 * T1[i] - dt * d1c[i - NP] + T(ky == i ? 1 : 0);
        PDDP_UNROLL for (int kx = 0; kx < NP; kx++) T2[kx] = T(1) * (dt2 * T1[kx + NP] - dt * d1c[kx] + T(ky == kx + NP ? 1 : 0)) + T(0);
        PDDP_UNROLL for (int r = 0; r < NP; r++) {
            T val = 0;
            PDDP_UNROLL for (int i = 0; i < NX; i++) { if (GC::live(i)) val += lds.d[GC::index(i) * NP + r][2][j] * gq[i]; }
            const T own = lds.d[ri_ky + r][2][j];
            T2[r + NP] = val + (ky < NX ? T(0) : (ci_ky < 0 ? T(0) : own));
        }
        T* ABc = b.AB + ((size_t)pb * N + k) * NX * NM + (size_t)ky * NX;
        T abv[NX];
        PDDP_UNROLL for (int kx = 0; kx < NX; kx++) {
            const T dx = kx < NP ? T(kx + NP == ky ? 1 : 0) : d1c[kx - NP];
            abv[kx] = (dt / T(6)) * dx + (dt2 / T(3)) * T1[kx] + (dt / T(6)) * T2[kx] + T(kx == ky ? 1 : 0);
        }
        cf_store_vec<T, NX>(ABc, abv);                                       // this lane's column: 16-byte pieces; a knot's 16 lanes fill one contiguous run
    }
}

// phase hook of the handles whose production rollouts keep records (k_fp_cf): the candidate-major xs / us that k_fp_ts just wrote -> xw.  thread = (problem, knot, step size)
template <typename P, typename T>
__global__ __launch_bounds__(256) void k_cand_to_xw(Buffers<T> b, Dims dm, int batch) {
    constexpr int NX = P::NX, NU = P::NU, REC = NX + NU;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)batch * dm.N * dm.A;
    if (id >= total) return;
    const int a = (int)(id % dm.A), k = (int)((id / dm.A) % dm.N), pb = (int)(id / ((size_t)dm.A * dm.N));
    const size_t slot = (size_t)pb * dm.A + a;
    T* r = b.xw + id * REC;
    for (int i = 0; i < NX; i++) r[i] = b.xs[(slot * dm.N + k) * NX + i];
    for (int i = 0; i < NU; i++) r[NX + i] = b.us[(slot * dm.N + k) * NU + i];
}

// API view the other way (pddp_get_array("xs" / "us") on such a handle after production sweeps): the records of the last rollouts -> the candidate-major arrays
template <typename P, typename T>
__global__ __launch_bounds__(256) void k_xw_to_cand(Buffers<T> b, Dims dm, int batch) {
    constexpr int NX = P::NX, NU = P::NU, REC = NX + NU;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)batch * dm.N * dm.A;
    if (id >= total) return;
    const int a = (int)(id % dm.A), k = (int)((id / dm.A) % dm.N), pb = (int)(id / ((size_t)dm.A * dm.N));
    const size_t slot = (size_t)pb * dm.A + a;
    const T* r = b.xw + id * REC;
    for (int i = 0; i < NX; i++) b.xs[(slot * dm.N + k) * NX + i] = r[i];
    for (int i = 0; i < NU; i++) b.us[(slot * dm.N + k) * NU + i] = r[NX + i];
}

// debugging aid (PDDP_POISON_LDS, run_phase): fill the whole LDS of the compute unit this block lands on with NaNs
static __global__ __launch_bounds__(256) void k_poison_lds(int words) {      // (static: one copy per plant translation unit)
    extern __shared__ unsigned poison_lds[];
    for (int i = threadIdx.x; i < words; i += 256) poison_lds[i] = 0x7fc00000u;
    __syncthreads();
    if (poison_lds[(threadIdx.x * 97) % words] != 0x7fc00000u) __builtin_trap();
}

// k_bp_cl: 16 lanes = (problem, block of knots), lane = column of [A B] / H (bp_cl.hpp); grid ceil(B M / 4), block 64
template <typename P, typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? 2 : 1))) void k_bp_cl(Buffers<T> b, Dims dm, CostWeights<T> cw, int batch, int diag_h) {
    __shared__ BpClLds<T> s[4];
    const int grp = threadIdx.x >> 4, c = threadIdx.x & 15, inst = blockIdx.x * 4 + grp;
    if (inst >= batch * dm.M) return;
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU;
    const int blk = inst % dm.M, pb = inst / dm.M, N = dm.N;
    const SolverState<T>& st = b.state[pb];
    if (st.done) return;
    BpArgs<T> a;                                                         // as bp_body (bodies.hpp)
    a.AB = b.AB + (size_t)pb * N * NX * NM;
    a.Pm = (st.pw ? b.Pp : b.P) + (size_t)pb * N * NX * NX;   a.pv = (st.pw ? b.pp : b.p) + (size_t)pb * N * NX;
    a.Pp = (st.pw ? b.P : b.Pp) + (size_t)pb * N * NX * NX;   a.pp = (st.pw ? b.p : b.pp) + (size_t)pb * N * NX;
    a.H = b.H + (size_t)pb * N * NM * NM;    a.g = b.g + (size_t)pb * N * NM;
    a.KT = b.KT + (size_t)pb * N * NX * NU;  a.du = b.du + (size_t)pb * N * NU;
    a.dcur = b.dcur + (size_t)pb * N * NX;
    a.ApBK = b.ApBK + (size_t)pb * N * NX * NX;  a.Bdu = b.Bdu + (size_t)pb * N * NX;
    a.xcur = b.xb + ((size_t)pb * 2 + st.cur) * N * NX;
    a.xprev2 = b.xb + ((size_t)pb * 2 + st.cur2) * N * NX;
    a.dJexp = b.dJexp + (size_t)pb * 2 * dm.M;
    a.err = b.err + (size_t)pb * dm.M;
    a.rho = st.rho;
    if (bp_cl_block<P, T>(s[grp], c, dm, blk, a, diag_h != 0, P::weight(cw, c, 0, N))) { if (c == 0) a.err[blk] = 1; }
}

// k_bp_mq: one wavefront = (problem, block of knots) of a 12-state / 4-control plant on the matrix cores (bp_mq.hpp); grid B M, block 64.  Replaces backPassKern<<<M, ...>>>
// (bpHelpers.cuh:339-420) like k_bp_cl, which stays for handles whose cost Hessian is not the plant's own diagonal.
#ifndef PDDP_MQ_WAVES
#define PDDP_MQ_WAVES 6      // resident waves per SIMD the float kernel is compiled for (measured: tools/quad_bp_ab.py, profiles/r05_quad_mfma.md)
#endif
#ifndef PDDP_MQ_FUSE_WAVES
#define PDDP_MQ_FUSE_WAVES 6 // ... and the instantiations that compose the sweep maps (four more live registers: six waves without the operand prefetch, PDDP_MQ_FUSE_PREFETCH in bp_mq.hpp)
#endif
template <typename P, typename T, bool DIAGH, bool FUSE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? (FUSE ? PDDP_MQ_FUSE_WAVES : PDDP_MQ_WAVES) : 3, sizeof(T) == 4 ? (FUSE ? PDDP_MQ_FUSE_WAVES : PDDP_MQ_WAVES) : 3))) void k_bp_mq(Buffers<T> b, Dims dm, CostWeights<T> cw, int batch) {
    __shared__ __attribute__((aligned(16))) T lds[kMqLds];
    const int inst = blockIdx.x;
    if (inst >= batch * dm.M) return;
    if (dm.M > 1) mq_bp_block<P, T, true, DIAGH, FUSE>(lds, b, dm, cw, inst / dm.M, inst % dm.M);
    else mq_bp_block<P, T, false, DIAGH>(lds, b, dm, cw, inst / dm.M, inst % dm.M);
}

// k_sweep_maps_cf: grid ceil(B / 4), block 64 -- 16 lanes per problem, lane l < NX owns entry l.  The forward sweep from the per-segment maps k_bp_mq<.., FUSE> composed:
// the sweep x_{k+1} = xcur_{k+1} + (A - B K)_k (x_k - xcur_k) - alpha (B du)_k + [boundary] d_k (forward_sweep, fp.hpp; forwardSweepKern fpHelpers.cuh:19-63) is
// e = t - alpha s with s <- Phi s + gamma (gamma = the map's affine column), t <- Phi t + d_boundary over the segments; at every boundary the start state of every
// candidate goes to its RECORD of the boundary knot, where k_fp_cf reads it (what k_sweep_cf did knot by knot).  The closed-form twin of k_sweep_maps (pddp_mx.hip).
template <typename P, typename T>
__global__ __launch_bounds__(64) void k_sweep_maps_cf(Buffers<T> b, Dims dm, int batch) {
    constexpr int NX = P::NX, REC = P::NX + P::NU;
    static_assert(NX < 16, "one 16-lane row per problem, the map a 16 x 16 tile");
    const int lane = threadIdx.x & 63, l = lane & 15, li = l < NX ? l : NX - 1;
    const int pb_raw = blockIdx.x * 4 + (lane >> 4), pb = pb_raw < batch ? pb_raw : batch - 1;
    const bool live = pb_raw < batch && fp_active<T>(b, dm, pb);
    const int N = dm.N, NBk = dm.NB;
    const T* xcur = b.xb + ((size_t)pb * 2 + b.state[pb].cur) * N * NX;
    const T* dcur = b.dcur + (size_t)pb * N * NX;
    T es = T(0), et = T(0);
    for (int sgm = 0; sgm < dm.M - 1; sgm++) {
        const T* o = b.segmap + ((size_t)pb * dm.M + sgm) * 256;                   // Psi(l, c) at [c * 16 + l]
        const int k = (sgm + 1) * NBk - 1;
        T ns = o[NX * 16 + li], nt = dcur[(size_t)k * NX + li];
#pragma unroll
        for (int cc = 0; cc < NX; cc++) {
            const T ph = o[cc * 16 + li];
            ns += ph * __shfl(es, cc, 16); nt += ph * __shfl(et, cc, 16);
        }
        es = ns; et = nt;
        if (live && l < NX) {
            const T base = xcur[(size_t)(k + 1) * NX + l];
            for (int a = 0; a < dm.A; a++) b.xw[(((size_t)pb * N + k + 1) * dm.A + a) * REC + l] = base + (et - b.alpha[a] * es);
        }
    }
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

// Reference views on demand (pddp_refresh_reference_views): grid (N, B), block 64.  The reference's backward pass leaves A - B K and B du of every knot in d_ApBK / d_Bdu
// (computeFSVars, bpHelpers.cuh:281-312: M > 1) and its next-iteration setup copies the accepted trajectory into EVERY step size's slot (memcpyCurrAKern x 3,
// nisInitHelpers.cuh:24-32,270-272).  The production sweeps do neither (the sweep operands are composed into segment maps inside the backward pass, candidates always start
// from the current trajectory), so the arrays are rebuilt here from what the last sweep left: reference-layout [A B], KT, du -- the operation order of bp_block (bp.hpp) --
// and the current trajectory.  what bit 0: sweep operands, bit 1: step-size slots.
template <typename P, typename T>
__global__ __launch_bounds__(64) void k_reference_views(Buffers<T> b, Dims dm, int what) {
    constexpr int NX = P::NX, NU = P::NU, NM = NX + NU;
    const int k = blockIdx.x, pb = blockIdx.y, N = dm.N;
    const Wave w = this_wave();
    const size_t knot = (size_t)pb * N + k;
    if ((what & 1) && dm.M > 1 && k < N - 1) {
        const T* AB = b.AB + knot * (NX * NM); const T* KT = b.KT + knot * (NX * NU); const T* du = b.du + knot * NU;
        T* F = b.ApBK + knot * (NX * NX); T* Bd = b.Bdu + knot * NX;
        PDDP_FOR(e, NX * NX) {
            const int ky = e / NX, kx = e % NX;
            T val = 0;
            for (int j = 0; j < NU; j++) val += AB[NX * NX + kx + NX * j] * KT[ky + NX * j];
            F[e] = AB[e] - val;
        }
        PDDP_FOR(kx, NX) { T val = 0; for (int j = 0; j < NU; j++) val += AB[NX * NX + kx + NX * j] * du[j]; Bd[kx] = val; }
    }
    if (what & 2) {
        const T* xc = b.xb + (((size_t)pb * 2 + b.state[pb].cur) * N + k) * NX;
        for (int a = 0; a < dm.A; a++) {
            const size_t slot = ((size_t)pb * dm.A + a) * N + k;
            PDDP_FOR(i, NX) { b.xs[slot * NX + i] = xc[i]; b.ds[slot * NX + i] = b.dcur[knot * NX + i]; }
            PDDP_FOR(i, NU) b.us[slot * NU + i] = b.ucur[knot * NU + i];
        }
    }
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
