// Launchers of the thread-lane kernels (pddp_tl.hip): they live in their own translation unit because they want a different code
// generation policy than the rest of the library -- fused multiply-adds and NO SLP vectorisation (packed f32 instructions buy no VALU
// throughput on gfx950 and cost register-pair shuffles in scalar code like this).
#pragma once

#include <hip/hip_runtime.h>

#include "ab_compact.hpp"
#include "fp_tl.hpp"

namespace pddp {

// variant: 0 / 1 = the built-in robot model (arm_tl_builtin) whose constants are folded into the kernels
// store_candidates: 1 = every candidate's x, u, d (teacher-forcing hook); 0 (the sweep) = states and boundary defects only -- what launch_nis_tl adopts the winner from
template <typename T> void launch_fp_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int batch, int store_candidates);
// the rollouts with every step split over two wavefronts (k_fp_tl2: few problems in flight); stores x, u, d of every candidate
void launch_fp_tl2(hipStream_t s, int variant, const Buffers<float>& b, const Dims& dm, const CostWeights<float>& cw, float dt, float grav, int batch);
// the rollouts as a pipeline over four wavefronts (k_fp_tl4, fp_pipe.hpp: few problems in flight; joint-space and end-effector cost); stores x, u, d of every candidate
// (T = double: the parity instantiation, selected with PDDP_FP=tl4 on a double handle)
template <typename T> void launch_fp_tl4(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int batch, const SolverParams& sp, int ls_mode, bool maps);
// the linear sweep of all candidates from two sequences (k_sweep_st, float handles)
void launch_sweep_st(hipStream_t s, const Buffers<float>& b, const Dims& dm, int batch);
// the same sweep with one workgroup per problem and the chain's operands staged in LDS up front (k_sweep_wg: few problems in flight)
void launch_sweep_wg(hipStream_t s, const Buffers<float>& b, const Dims& dm, int batch);
template <typename T> void launch_nis_tl(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int mode, int batch);
// next-iteration setup with one thread per (knot, joint) (k_nis_tl7: few problems in flight); adopts the winner from the candidate-major xs / us / ds
template <typename T> void launch_nis_tl7(hipStream_t s, int variant, const Buffers<T>& b, const Dims& dm, const CostWeights<T>& cw, T dt, T grav, int mode, int batch);
// compact [A B] (ab_compact.hpp) <-> the reference layout b.AB, for the API view of a handle that keeps the compact array
template <typename T> void launch_abc_convert(hipStream_t s, const Buffers<T>& b, int knots, int N, T dt, int to_compact);
template <typename T> void launch_plant_eval_tl(hipStream_t s, int variant, T grav, int count, const T* x, const T* u, T* out, int grad);

}  // namespace pddp
