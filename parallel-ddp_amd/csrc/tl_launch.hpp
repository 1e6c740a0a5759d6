// Launchers of the thread-lane kernels (pddp_tl.hip): they live in their own translation unit because they want a different code
// generation policy than the rest of the library -- fused multiply-adds and NO SLP vectorisation (packed f32 instructions buy no VALU
// throughput on gfx950 and cost register-pair shuffles in scalar code like this).
#pragma once

#include <hip/hip_runtime.h>

#include "fp_tl.hpp"

namespace pddp {

// variant: 0 / 1 = the built-in robot model (arm_tl_builtin) whose constants are folded into the kernels
// store_candidates: also write every candidate's x, u, d (teacher-forcing hook); the sweep passes 0 and lets launch_win_tl re-roll the winner
template <typename T> void launch_fp_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int batch, int store_candidates);
// the linear sweep of all candidates from two sequences (k_sweep_st, float handles)
void launch_sweep_st(hipStream_t s, const Buffers<float>& b, const Dims& dm, int batch);
template <typename T> void launch_win_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int batch);
// the accepted candidate's stored slot -> current trajectory (used instead of launch_win_tl when the rollouts stored their candidates)
template <typename T> void launch_adopt_tl(hipStream_t s, const Buffers<T>& b, const Dims& dm, int batch);
template <typename T> void launch_nis_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int mode, int batch);
template <typename T> void launch_plant_eval_tl(hipStream_t s, int variant, T grav, int count, const T* x, const T* u, T* out, int grad);

}  // namespace pddp
