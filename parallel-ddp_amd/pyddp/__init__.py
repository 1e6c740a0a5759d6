"""pyddp -- Python host-side mirror of the reference's runiLQR_GPU / DDPWrappers interface over libpddp.so.

Thin ctypes plumbing around the C ABI in include/pddp.h.  All compute happens in the HIP kernels; there is no
CPU fallback: importing works anywhere, but creating a Solver without a HIP device (or without the built
library) raises.
"""
from .binding import (PddpConfig, PddpKernelSelection, KERNEL_NAMES, set_kernels, PddpError, Solver, Comm, default_config, library_path, build_id, PLANT_DIMS, algorithmic_bytes, algorithmic_bytes_per_kernel,  # noqa: F401
                      PHASE_BP, PHASE_FP, PHASE_LS, PHASE_NIS, PHASE_INIT_NIS, PHASE_INIT_COST, PHASE_BP_COOP, PHASE_BP_FUSED, PHASE_SWEEP_FUSED, PHASE_ROLLOUT)
