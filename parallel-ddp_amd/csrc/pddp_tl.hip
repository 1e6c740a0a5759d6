// Thread-lane kernels of the KUKA arm (fp_tl.hpp, plant_arm_tl.hpp): one thread per rollout / per knot.  gfx950 only.
// Built with -ffp-contract=fast semantics inside the tl headers and -fno-slp-vectorize (Makefile).
#include <hip/hip_runtime.h>

#include "bodies.hpp"
#include "tl_launch.hpp"

namespace pddp {

// k_fp_tl: grid ceil(B*M*A / 256), block 256.  Thread i rolls out segment (i / A) % M of candidate i % A of problem i / (M*A): the 8 candidates
// of a (problem, segment) are adjacent lanes and share every gain / reference address.  The linear sweep (k_sweep_lg) runs before it, unchanged.
// Replaces forwardSimKern<<<(M,A),(8,7)>>> + costKern<<<A,N>>> + defectKern<<<A,N>>> (fpHelpers.cuh:366,383,388).
template <typename T, int V>
__global__ __launch_bounds__(256, 2) void k_fp_tl(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, T grav, int batch) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    const int inst = blockIdx.x * 256 + threadIdx.x, per_pb = dm.M * dm.A;
    if (inst >= batch * per_pb) return;
    const int pb = inst / per_pb, rem = inst - pb * per_pb, seg = rem / dm.A, a_idx = rem - seg * dm.A;
    if (!fp_active<T>(b, dm, pb)) return;
    arm_tl_rollout_segment<T>(md, grav, b, dm, cw, dt, pb, a_idx, seg);
}

// k_nis_tl: grid ceil(B*N / 256), block 256.  Thread = knot (global knot index g = pb*N + k).  float: a wave's 64 knots x 147 Jacobian entries are
// staged in LDS (entry-major, padded to 65 so that both the per-thread writes and the transposed reads are conflict-free) and [A B] of the 64 knots
// -- one contiguous 64 x 294 float run -- is written column by column, 64 consecutive columns per pass: every cache line is written once, completely.
// double: direct stores.  Replaces integratorGradientKern + costGradientHessianKern + memcpyCurrAKern x3 (nisInitHelpers.cuh:247-279).
constexpr int kNisTlStage = 147 * 65;
template <typename T, int V>
__global__ __launch_bounds__(256, 1) void k_nis_tl(Buffers<T> b, Dims dm, CostWeights<T> cw, T dt, T grav, int mode, int batch) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    constexpr int NX = 14, NM = 21;
    const int g = blockIdx.x * 256 + threadIdx.x, total = batch * dm.N;
    const int pb = g / dm.N, k = g - pb * dm.N;
    if constexpr (sizeof(T) == 4) {
        __shared__ T stage_all[4 * kNisTlStage];
        T* stage = stage_all + (threadIdx.x >> 6) * kNisTlStage;
        const int lane = threadIdx.x & 63;
        bool valid = false;
        if (g < total) valid = arm_tl_nis_knot<T>(md, grav, b, dm, cw, mode, k, pb, [&](int col, int row, T val) { stage[(col * 7 + row) * 65 + lane] = val; });
        const unsigned long long mask = __ballot(valid);
        wsync();
        if (!mask) return;
        const int g0 = g - lane;                                    // first knot of this wave
        T* AB0 = b.AB + (size_t)g0 * (NX * NM);
        for (int it = 0; it < NM; it++) {                           // 21 passes x 64 lanes = 64 knots x 21 columns, consecutive in memory
            const int pi = it * 64 + lane, kk = pi / NM, col = pi - kk * NM;
            if (!((mask >> kk) & 1ull)) continue;
            T out[NX];
#pragma unroll
            for (int r = 0; r < 7; r++) {
                out[r] = tl_AB_const<T>(r, col, dt);
                out[7 + r] = T(col == 7 + r ? 1 : 0) + dt * stage[(col * 7 + r) * 65 + kk];
            }
            tl_store14(AB0 + (size_t)pi * NX, out);
        }
    } else {
        if (g >= total) return;
        T* AB = b.AB + (size_t)g * (NX * NM);
        const bool valid = arm_tl_nis_knot<T>(md, grav, b, dm, cw, mode, k, pb, [&](int col, int row, T val) { AB[col * NX + 7 + row] = T(col == 7 + row ? 1 : 0) + dt * val; });
        if (valid) for (int col = 0; col < NM; col++) for (int r = 0; r < 7; r++) AB[col * NX + r] = tl_AB_const<T>(r, col, dt);
    }
}

// forward dynamics / gradient of `count` (x, u) samples, one thread each (tests, micro-benchmarks)
template <typename T, int V>
__global__ __launch_bounds__(256, 1) void k_plant_eval_tl(T grav, int count, const T* x, const T* u, T* out, int grad) {
    constexpr ArmTlModel<T> md = arm_tl_builtin<T>(V);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    T xi[14], ui[7], qdd[7];
    for (int e = 0; e < 14; e++) xi[e] = x[(size_t)i * 14 + e];
    for (int e = 0; e < 7; e++) ui[e] = u[(size_t)i * 7 + e];
    ArmTlState<T> st;
    arm_tl_dynamics<T>(md, grav, st, qdd, xi, xi + 7, ui);
    if (!grad) { for (int e = 0; e < 7; e++) out[(size_t)i * 7 + e] = qdd[e]; return; }
    T* o = out + (size_t)i * 147;
    arm_tl_gradient<T>(md, grav, st, xi + 7, qdd, [o](int col, int row, T val) { o[7 * col + row] = val; });
}

template <typename T>
void launch_fp_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int batch) {
    const unsigned inst = (unsigned)batch * dm.M * dm.A;
    if (variant == 0) hipLaunchKernelGGL((k_fp_tl<T, 0>), dim3((inst + 255) / 256), dim3(256), 0, s, b, dm, cw, dt, grav, batch);
    else hipLaunchKernelGGL((k_fp_tl<T, 1>), dim3((inst + 255) / 256), dim3(256), 0, s, b, dm, cw, dt, grav, batch);
}
template <typename T>
void launch_nis_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int mode, int batch) {
    const unsigned knots = (unsigned)batch * dm.N;
    if (variant == 0) hipLaunchKernelGGL((k_nis_tl<T, 0>), dim3((knots + 255) / 256), dim3(256), 0, s, b, dm, cw, dt, grav, mode, batch);
    else hipLaunchKernelGGL((k_nis_tl<T, 1>), dim3((knots + 255) / 256), dim3(256), 0, s, b, dm, cw, dt, grav, mode, batch);
}
template <typename T>
void launch_plant_eval_tl(hipStream_t s, int variant, T grav, int count, const T* x, const T* u, T* out, int grad) {
    if (variant == 0) hipLaunchKernelGGL((k_plant_eval_tl<T, 0>), dim3((count + 255) / 256), dim3(256), 0, s, grav, count, x, u, out, grad);
    else hipLaunchKernelGGL((k_plant_eval_tl<T, 1>), dim3((count + 255) / 256), dim3(256), 0, s, grav, count, x, u, out, grad);
}
template void launch_fp_tl<float>(hipStream_t, int, const Buffers<float>&, const Dims&, const CostWeights<float>&, float, float, int);
template void launch_fp_tl<double>(hipStream_t, int, const Buffers<double>&, const Dims&, const CostWeights<double>&, double, double, int);
template void launch_nis_tl<float>(hipStream_t, int, const Buffers<float>&, const Dims&, const CostWeights<float>&, float, float, int, int);
template void launch_nis_tl<double>(hipStream_t, int, const Buffers<double>&, const Dims&, const CostWeights<double>&, double, double, int, int);
template void launch_plant_eval_tl<float>(hipStream_t, int, float, int, const float*, const float*, float*, int);
template void launch_plant_eval_tl<double>(hipStream_t, int, double, int, const double*, const double*, double*, int);

}  // namespace pddp
