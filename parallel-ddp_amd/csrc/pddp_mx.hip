// Matrix-core kernels of the KUKA arm (bp_mfma.hpp).  gfx950 only.
#include <hip/hip_runtime.h>

#include "bp_mfma.hpp"
#include "mx_launch.hpp"

// resident waves per SIMD the float kernel is compiled for: six for the compact-[A B] variants without the end-effector block (78-80 registers, no scratch; measured
// against five: 1.31 -> 1.27 ms at 16384 problems, profiles/r04_bp_mfma.md), five (<= 102 registers) for the others.  -DPDDP_MX_WAVES=n forces one value (measurement variants).
#ifdef PDDP_MX_WAVES
#define PDDP_MX_WAVES_OF(CAB, HQQ) PDDP_MX_WAVES
#else
#define PDDP_MX_WAVES_OF(CAB, HQQ) ((CAB) ? 6 : 5)
#endif

namespace pddp {

// k_bp_mfma: grid B*M, block 64 -- one wavefront per (problem, block of knots).  Replaces backPassKern<<<M, (8,7)>>> (bpHelpers.cuh:339-420).
template <bool FS, bool DIAGH, bool CAB, bool FUSE, bool HQQ = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PDDP_MX_WAVES_OF(CAB, HQQ), PDDP_MX_WAVES_OF(CAB, HQQ)))) void k_bp_mfma(Buffers<float> b, Dims dm, int batch, float hq1, float hq2, float hr, float dt, int flags) {
    __shared__ __attribute__((aligned(16))) float lds[kMxLds + kMxDmaFloats];
    const int inst = blockIdx.x;
    if (inst >= batch * dm.M) return;
    arm_mx_bp_block<float, FS, DIAGH, CAB, FUSE, HQQ>(lds, b, dm, inst / dm.M, inst % dm.M, hq1, hq2, hr, dt, flags);
}
// The same tile algebra on v_mfma_f64_16x16x4_f64 for double handles (PDDP_BP=mx; no occupancy target: it exists so that the production algebra can be checked
// against the oracle at a precision where every step-size decision of a 40-iteration solve is reproducible -- tests/test_f64_benched_family.py).
template <bool FS, bool DIAGH, bool CAB, bool FUSE, bool HQQ = false>
__global__ __launch_bounds__(64) void k_bp_mfma_f64(Buffers<double> b, Dims dm, int batch, double hq1, double hq2, double hr, double dt, int flags) {
    __shared__ __attribute__((aligned(16))) double lds[kMxLds];
    const int inst = blockIdx.x;
    if (inst >= batch * dm.M) return;
    arm_mx_bp_block<double, FS, DIAGH, CAB, FUSE, HQQ>(lds, b, dm, inst / dm.M, inst % dm.M, hq1, hq2, hr, dt, flags);
}
template <bool FS, bool DIAGH, bool CAB, bool FUSE, bool HQQ = false>
static void launch_one(hipStream_t s, unsigned n, const Buffers<float>& b, const Dims& dm, int batch, float hq1, float hq2, float hr, float dt, int kp) {
    hipLaunchKernelGGL((k_bp_mfma<FS, DIAGH, CAB, FUSE, HQQ>), dim3(n), dim3(64), 0, s, b, dm, batch, hq1, hq2, hr, dt, kp);
}
template <bool FS, bool DIAGH, bool CAB, bool FUSE, bool HQQ = false>
static void launch_one(hipStream_t s, unsigned n, const Buffers<double>& b, const Dims& dm, int batch, double hq1, double hq2, double hr, double dt, int kp) {
    hipLaunchKernelGGL((k_bp_mfma_f64<FS, DIAGH, CAB, FUSE, HQQ>), dim3(n), dim3(64), 0, s, b, dm, batch, hq1, hq2, hr, dt, kp);
}

template <typename T>
void launch_bp_mfma(hipStream_t s, const Buffers<T>& b, const Dims& dm, int batch, bool diag_h, T hq1, T hq2, T hr, T dt, bool keep_P, bool fuse_sweep) {
    const unsigned n = (unsigned)batch * dm.M;
    const int kp = (keep_P ? kMxKeepP : 0) | ((fuse_sweep && b.segmap) ? kMxFuseSweep : 0);
    const bool cab = b.ABc != nullptr && diag_h;
#define PDDP_MX_LAUNCH(FS, DH, CB, FU) launch_one<FS, DH, CB, FU>(s, n, b, dm, batch, hq1, hq2, hr, dt, kp)
    const bool fu = (kp & kMxFuseSweep) != 0;
    if (cab && b.Hc) {                                                // end-effector handles on the thread-lane / matrix-core path: compact [A B] + the compact position block
#define PDDP_MX_LAUNCH_Q(FS, FU) launch_one<FS, true, true, FU, true>(s, n, b, dm, batch, hq1, hq2, hr, dt, kp)
        if (dm.M > 1) { if (fu) PDDP_MX_LAUNCH_Q(true, true); else PDDP_MX_LAUNCH_Q(true, false); }
        else PDDP_MX_LAUNCH_Q(false, false);
#undef PDDP_MX_LAUNCH_Q
        return;
    }
    if (dm.M > 1) {
        if (cab) { if (fu) PDDP_MX_LAUNCH(true, true, true, true); else PDDP_MX_LAUNCH(true, true, true, false); }
        else if (diag_h) { if (fu) PDDP_MX_LAUNCH(true, true, false, true); else PDDP_MX_LAUNCH(true, true, false, false); }
        else { if (fu) PDDP_MX_LAUNCH(true, false, false, true); else PDDP_MX_LAUNCH(true, false, false, false); }
    } else {
        if (cab) PDDP_MX_LAUNCH(false, true, true, false);
        else if (diag_h) PDDP_MX_LAUNCH(false, true, false, false);
        else PDDP_MX_LAUNCH(false, false, false, false);
    }
#undef PDDP_MX_LAUNCH
}
template void launch_bp_mfma<float>(hipStream_t, const Buffers<float>&, const Dims&, int, bool, float, float, float, float, bool, bool);
template void launch_bp_mfma<double>(hipStream_t, const Buffers<double>&, const Dims&, int, bool, double, double, double, double, bool, bool);

// k_sweep_maps: grid ceil(B / PER), block 64.  The forward sweep from the per-segment maps the backward pass composed (bp_mfma.hpp kMxFuseSweep): e <- Phi_s e + gamma_s over the
// segments for the s-sequence (gamma = the map's column 14) and the t-sequence (gamma = the defect of the segment's boundary knot: it enters at the segment's last
// step), lane l < 14 owns entry l; at every boundary the segment start state of every candidate, x = xcur + (t - alpha s), goes to xs.  Replaces forwardSweepKern x A
// (fpHelpers.cuh:19-63) together with the A - B K / B du traffic between the two passes.
// PER problems per wave (1, or 4 with many problems in flight: 16 lanes each -- a quarter of the one-wave workgroups, whose launch is most of what this kernel costs).
template <typename T, int PER>
__global__ __launch_bounds__(64) void k_sweep_maps(Buffers<T> b, Dims dm, int batch) {
    constexpr int NX = 14, W = 64 / PER;                                         // W: lanes per problem (shuffles stay inside them)
    const int lane = threadIdx.x % W, pb = blockIdx.x * PER + threadIdx.x / W;
    if (pb >= batch) return;
    const SolverState<T>& st = b.state[pb];
    if (st.done) return;
    for (int i = 0; i < dm.M; i++) if (b.err[(size_t)pb * dm.M + i]) return;      // failed backward pass: no forward pass this sweep (fp_active)
    const int N = dm.N, NBk = dm.NB, l = lane < NX ? lane : NX - 1;
    const T* xcur = b.xb + ((size_t)pb * 2 + st.cur) * N * NX; const T* dcur = b.dcur + (size_t)pb * N * NX;
    T es = T(0), et = T(0);
    if (dm.M == 4) {                                                             // the usual M: all three segments' operands requested at once (one memory round trip instead of three;
        constexpr int S = 3;                                                     // the stores between the segments keep the compiler from hoisting the later loads itself); same operations
        T ph[S][NX + 1], dk[S], base[S];
#pragma unroll
        for (int sgm = 0; sgm < S; sgm++) {
            const T* o = b.segmap + ((size_t)pb * 4 + sgm) * 256;
            const int k = (sgm + 1) * NBk - 1;
#pragma unroll
            for (int cc = 0; cc <= NX; cc++) ph[sgm][cc] = o[cc * 16 + l];
            dk[sgm] = dcur[(size_t)k * NX + l]; base[sgm] = xcur[(size_t)(k + 1) * NX + l];
        }
#pragma unroll
        for (int sgm = 0; sgm < S; sgm++) {
            const int k = (sgm + 1) * NBk - 1;
            T ns = ph[sgm][NX], nt = dk[sgm];
#pragma unroll
            for (int cc = 0; cc < NX; cc++) { ns = Mx<T>::fma(ph[sgm][cc], __shfl(es, cc, W), ns); nt = Mx<T>::fma(ph[sgm][cc], __shfl(et, cc, W), nt); }
            es = ns; et = nt;
            if (lane < NX) for (int a = 0; a < dm.A; a++) b.xs[(((size_t)pb * dm.A + a) * N + k + 1) * NX + lane] = base[sgm] + (et - b.alpha[a] * es);
        }
        return;
    }
    for (int sgm = 0; sgm < dm.M - 1; sgm++) {
        const T* o = b.segmap + ((size_t)pb * dm.M + sgm) * 256;                   // Psi'(c, l) at [c * 16 + l]
        const int k = (sgm + 1) * NBk - 1;
        T ns = o[14 * 16 + l], nt = dcur[(size_t)k * NX + l];
#pragma unroll
        for (int cc = 0; cc < NX; cc++) {
            const T ph = o[cc * 16 + l];
            ns = Mx<T>::fma(ph, __shfl(es, cc, W), ns); nt = Mx<T>::fma(ph, __shfl(et, cc, W), nt);
        }
        es = ns; et = nt;
        if (lane < NX) {
            const T base = xcur[(size_t)(k + 1) * NX + lane];
            for (int a = 0; a < dm.A; a++) b.xs[(((size_t)pb * dm.A + a) * N + k + 1) * NX + lane] = base + (et - b.alpha[a] * es);
        }
    }
}
template <typename T>
void launch_sweep_maps(hipStream_t s, const Buffers<T>& b, const Dims& dm, int batch) {
    if (batch >= 2048) hipLaunchKernelGGL((k_sweep_maps<T, 4>), dim3((unsigned)((batch + 3) / 4)), dim3(64), 0, s, b, dm, batch);
    else hipLaunchKernelGGL((k_sweep_maps<T, 1>), dim3((unsigned)batch), dim3(64), 0, s, b, dm, batch);
}
template void launch_sweep_maps<float>(hipStream_t, const Buffers<float>&, const Dims&, int);
template void launch_sweep_maps<double>(hipStream_t, const Buffers<double>&, const Dims&, int);

}  // namespace pddp
