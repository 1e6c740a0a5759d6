"""Lane-group kernels (lanegroup.hpp, plant_arm_lg.hpp, fp_lg.hpp) against the wave-cooperative kernels they replace in the
arm's forward pass: the arithmetic is the same operation by operation, so the results must be BIT-IDENTICAL, in float32 and
float64, on the GPU (DPP / ds_swizzle cross-lane moves) and in the host emulation (8 lanes in lock step)."""
import numpy as np
import pytest

from backends import BACKENDS, make_solver

RNG = np.random.default_rng(77)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_lane_group_dynamics_bit_identical_to_cooperative(backend, dtype):
    s = make_solver(backend, 4, dtype=0 if dtype == np.float32 else 1, N=16, M=1, A=1, wafr_urdf=1)
    count = 203                                   # not a multiple of 8: the last wave has idle groups
    x = np.concatenate([RNG.normal(0, 2, (count, 7)), RNG.normal(0, 5, (count, 7))], axis=1).astype(dtype)
    u = RNG.normal(0, 50, (count, 7)).astype(dtype)
    coop, lg = s.plant_eval(0, x, u), s.plant_eval(4, x, u)
    assert np.isfinite(coop).all()
    assert np.array_equal(coop, lg)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_lane_group_gradient_bit_identical_to_cooperative(backend, dtype):
    s = make_solver(backend, 4, dtype=0 if dtype == np.float32 else 1, N=16, M=1, A=1, wafr_urdf=1)
    count = 77
    x = np.concatenate([RNG.normal(0, 2, (count, 7)), RNG.normal(0, 5, (count, 7))], axis=1).astype(dtype)
    u = RNG.normal(0, 50, (count, 7)).astype(dtype)
    coop, lg = s.plant_eval(1, x, u), s.plant_eval(5, x, u)
    assert np.isfinite(coop).all() and np.abs(coop).max() > 0
    assert np.array_equal(coop, lg)
