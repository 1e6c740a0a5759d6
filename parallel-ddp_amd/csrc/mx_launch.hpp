// Launchers of the matrix-core backward pass (pddp_mx.hip / bp_mfma.hpp): handles of the KUKA arm -- float (the production path) and, on request, double.
#pragma once

#include <hip/hip_runtime.h>

#include "solver_state.hpp"

namespace pddp {

// diag_h: the cost Hessian of every running knot is the joint-space cost's own diag(hq1 x 7, hq2 x 7, hr x 7), as the setup kernel wrote it:
// the kernel takes the three numbers from here and does not read H in its loop (the final knot's block is always read)
// b.ABc non-null (and diag_h): [A B] is read from the compact array (ab_compact.hpp), dt rebuilds its constant rows.  keep_P: write every knot's cost-to-go
// (the default; phase hook, MPC handles); otherwise only the block-boundary slots the next pass reads (pddp_config.boundary_cost_to_go_only).
template <typename T>
void launch_bp_mfma(hipStream_t s, const Buffers<T>& b, const Dims& dm, int batch, bool diag_h, T hq1, T hq2, T hr, T dt, bool keep_P, bool fuse_sweep);
// fuse_sweep (and b.segmap): the pass composes every shooting segment's sweep map instead of writing A - B K / B du; launch_sweep_maps then replaces the sweep kernel
template <typename T>
void launch_sweep_maps(hipStream_t s, const Buffers<T>& b, const Dims& dm, int batch);

}  // namespace pddp
