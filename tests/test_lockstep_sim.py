"""The lock-step experiment around the solver: the simulated robot simulateForward<T, SUBSTEPS>
(examples/WAFR_MPC_examples.cu:111-139) under the trajectory runner's control law getHardwareControls (MPCHelpers.cuh:819-858), and the tool
point compute_eePos_scratch (dynamics_arm.cuh:1953-1960) -- pddp_simulate / pddp_ee_pos against the oracle's restatement.  The plant runs in
double on both sides, the control law in the plan's precision."""
import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import Oracle, default_cfg

RNG = np.random.default_rng(21)
KW = dict(N=32, M=4, A=8, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, max_iter=10, ee_cost=1, ignore_max_rho_exit=0)


def plan(backend, dtype):
    N = KW["N"]
    s = make_solver(backend, 4, dtype=0 if dtype == np.float32 else 1, **KW)
    x0 = np.zeros((N, 14), dtype); x0[:, 1] = 0.7; x0[:, 3] = -0.8; x0[:, 5] = 0.75
    u0 = np.full((N, 7), 0.01, dtype)
    xg = np.zeros(14, dtype); xg[:3] = [0.45, 0.15, 0.75]
    out = s.solve(x0, u0, xg)
    return s, out["x"][0], out["u"][0], out["KT"][0], xg


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-10), (np.float32, 2e-5)])
def test_simulated_robot_follows_the_plan_like_the_reference(backend, dtype, tol):
    s, x, u, KT, xg = plan(backend, dtype)
    o = Oracle(default_cfg(4, cores=8, spawn_threads=0, **KW), dtype)
    step_us = 0.5 / 31 * 1e6
    for elapsed_knots, substeps, noise in ((0.6, 150, 0.0), (2.3, 150, 0.002), (5.0, 40, 0.01)):
        xa = (x[0] + RNG.normal(0, noise, 14)).astype(dtype)
        ro = o.simulate(x.ravel(), u.ravel(), KT.ravel(), 0.0, elapsed_knots * step_us, substeps, xg[:3], xa)
        rs = s.simulate(x, u, KT, 0.0, elapsed_knots * step_us, substeps, xg[:3], xa)
        assert ro[2] == 0 and rs[2] == 0
        np.testing.assert_allclose(rs[0], ro[0], rtol=0, atol=tol * max(1.0, np.abs(ro[0]).max()))
        assert abs(rs[1] - ro[1]) <= max(tol, 2e-7) * max(1.0, abs(ro[1]))
        assert ro[1] > 0
        # with feedback the robot stays near the plan: the state after k knots is close to the plan's knot
        k = int(elapsed_knots)
        assert np.abs(rs[0][:7] - x[k][:7]).max() < 0.2


@pytest.mark.parametrize("backend", BACKENDS)
def test_time_beyond_the_plan_aborts_without_touching_the_state(backend):
    s, x, u, KT, xg = plan(backend, np.float32)
    o = Oracle(default_cfg(4, cores=8, spawn_threads=0, **KW), np.float32)
    step_us = 0.5 / 31 * 1e6
    xa = x[0].copy()
    ro = o.simulate(x.ravel(), u.ravel(), KT.ravel(), 0.0, 31.5 * step_us, 150, xg[:3], xa)
    rs = s.simulate(x, u, KT, 0.0, 31.5 * step_us, 150, xg[:3], xa)
    assert ro[2] == 1 and rs[2] == 1 and ro[1] == 0.0 and rs[1] == 0.0
    assert np.array_equal(rs[0], xa) and np.array_equal(ro[0], xa)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-6)])
def test_tool_point_of_a_batch_of_states(backend, dtype, tol):
    s = make_solver(backend, 4, dtype=0 if dtype == np.float32 else 1, **KW)
    o = Oracle(default_cfg(4, **KW), dtype)
    X = RNG.normal(0, 1.0, (9, 14)).astype(dtype)
    P = s.ee_pos(X)
    for i in range(9):
        np.testing.assert_allclose(P[i], o.ee_pos(X[i], jac=False)[0], rtol=0, atol=tol * 4)
