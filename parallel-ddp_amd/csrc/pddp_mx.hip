// Matrix-core kernels of the KUKA arm (bp_mfma.hpp).  gfx950 only.
#include <hip/hip_runtime.h>

#include "bp_mfma.hpp"
#include "mx_launch.hpp"

namespace pddp {

// k_bp_mfma: grid B*M, block 64 -- one wavefront per (problem, block of knots); <= 102 registers so that five waves share a SIMD.  Replaces backPassKern<<<M, (8,7)>>> (bpHelpers.cuh:339-420).
template <bool FS, bool DIAGH, bool CAB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_bp_mfma(Buffers<float> b, Dims dm, int batch, float hq1, float hq2, float hr, float dt, int keepP) {
    __shared__ __attribute__((aligned(16))) float lds[96];
    const int inst = blockIdx.x;
    if (inst >= batch * dm.M) return;
    arm_mx_bp_block<FS, DIAGH, CAB>(lds, b, dm, inst / dm.M, inst % dm.M, hq1, hq2, hr, dt, keepP);
}

void launch_bp_mfma(hipStream_t s, const Buffers<float>& b, const Dims& dm, int batch, bool diag_h, float hq1, float hq2, float hr, float dt, bool keep_P) {
    const unsigned n = (unsigned)batch * dm.M;
    const int kp = keep_P ? 1 : 0;
    const bool cab = b.ABc != nullptr && diag_h;
#define PDDP_MX_LAUNCH(FS, DH, CB) hipLaunchKernelGGL((k_bp_mfma<FS, DH, CB>), dim3(n), dim3(64), 0, s, b, dm, batch, hq1, hq2, hr, dt, kp)
    if (dm.M > 1) {
        if (cab) PDDP_MX_LAUNCH(true, true, true);
        else if (diag_h) PDDP_MX_LAUNCH(true, true, false);
        else PDDP_MX_LAUNCH(true, false, false);
    } else {
        if (cab) PDDP_MX_LAUNCH(false, true, true);
        else if (diag_h) PDDP_MX_LAUNCH(false, true, false);
        else PDDP_MX_LAUNCH(false, false, false);
    }
#undef PDDP_MX_LAUNCH
}

}  // namespace pddp
