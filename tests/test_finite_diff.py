"""USE_FINITE_DIFF (config.cuh:68-71): [A B] of the Euler step by central differences of the plant's `dynamics`, column by column
(finiteDiffInner / integratorGradientKern / integratorGradientThreaded, DDPHelpers/nisInitHelpers.cuh:138-201) instead of the analytic gradient.
The reference's own test of its analytic gradient is exactly this pair (test/testDynGrad.cu: analytical vs finite difference, epsilon 1e-3).
"""
import ctypes as C
import os

import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import Oracle, default_cfg, example_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [pytest.param(4, 32, 4, 4, id="kuka"), pytest.param(2, 32, 4, 8, id="cart-euler"), pytest.param(1, 16, 2, 4, id="pend-euler")]


def problem(plant, N, dtype, seed=3):
    rng = np.random.default_rng(seed)
    x, u, xg = example_inputs(plant, N, dtype)
    n, m = x.size // N, u.size // N
    x = (x.reshape(N, n) + rng.normal(0, 0.02, (N, n))).astype(dtype).ravel()
    u = (u.reshape(N, m) * (1 + rng.normal(0, 0.02, (N, m))) + rng.normal(0, 0.01, (N, m))).astype(dtype).ravel()
    return x, u, xg


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("plant,N,M,A", CASES)
def test_finite_difference_jacobian_equals_the_oracle(backend, dtype, plant, N, M, A):
    """The kernels' finite-difference [A B] of every knot against the oracle's restatement of finiteDiffInner: float64 1e-9 of the block's scale; float32
    with the default epsilon 1e-5 is a quotient of rounding errors in the reference itself -- both sides evaluate the same formula, so what is compared is
    the (qdd+ - qdd-) of two dynamics implementations that agree to ~1e-6: absolute 0.2 on entries of size dt / eps * 1e-6."""
    eps = 1e-5 if dtype == np.float64 else 1e-2
    kw = dict(N=N, M=M, A=A, integrator=1, wafr_urdf=1, total_time=0.5 if plant == 4 else 1.0, use_finite_diff=1, finite_diff_epsilon=eps)
    s = make_solver(backend, plant, dtype=0 if dtype == np.float32 else 1, **kw)
    o = Oracle(default_cfg(plant, **kw), dtype)
    n, m = o.n, o.m
    nm = n + m
    x, u, xg = problem(plant, N, dtype)
    s.load(x, u, xg)
    got = s.get("AB").reshape(N, nm, n)[: N - 1]
    ref = np.stack([o.integrator_gradient(x[k * n:(k + 1) * n], u[k * m:(k + 1) * m]).reshape(nm, n) for k in range(N - 1)])
    scale = np.abs(ref).max()
    assert np.abs(got.astype(np.float64) - ref).max() <= (1e-9 if dtype == np.float64 else 2e-3) * scale
    np.testing.assert_array_equal(got[:, :, : n // 2], ref[:, :, : n // 2])          # position rows: the exact constants of dqddk2dxd


@pytest.mark.parametrize("backend", BACKENDS)
def test_finite_difference_jacobian_approaches_the_analytic_one(backend):
    """testDynGrad.cu's comparison, in float64 with a small step: central differences agree with the analytic [A B] to O(eps^2)."""
    kw = dict(N=32, M=4, A=4, integrator=1, wafr_urdf=1, total_time=0.5)
    x, u, xg = problem(4, 32, np.float64)
    sa = make_solver(backend, 4, dtype=1, **kw)
    sf = make_solver(backend, 4, dtype=1, use_finite_diff=1, finite_diff_epsilon=1e-6, **kw)
    sa.load(x, u, xg); sf.load(x, u, xg)
    a, f = sa.get("AB").reshape(32, 21, 14)[:31], sf.get("AB").reshape(32, 21, 14)[:31]
    assert np.abs(a - f).max() <= 2e-8 * np.abs(a).max()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("plant,N,M,A", CASES[:2])
def test_whole_solve_with_finite_differences_follows_the_oracle(backend, plant, N, M, A):
    kw = dict(N=N, M=M, A=A, integrator=1, wafr_urdf=1, total_time=0.5 if plant == 4 else 1.0, max_iter=6, tol_cost=0.0, use_finite_diff=1, finite_diff_epsilon=1e-5)
    s = make_solver(backend, plant, dtype=1, **kw)
    o = Oracle(default_cfg(plant, cores=1, spawn_threads=0, **kw), np.float64)
    x, u, xg = example_inputs(plant, N, np.float64)
    out = s.solve(x, u, xg)
    ref = o.run_ilqr_gpusem(x, u, xg)
    it = ref["iters"]
    assert out["iters"][0] == it and list(out["alphaOut"][0][: it + 1]) == list(ref["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], ref["Jout"][: it + 1], rtol=1e-7)
    assert (np.asarray(ref["alphaOut"][1: it + 1]) >= 0).any()


@pytest.mark.gpu
def test_finite_differences_need_the_euler_rule():
    import pyddp
    with pytest.raises(pyddp.PddpError):
        make_solver("hip", 2, dtype=1, N=32, M=4, A=4, integrator=3, use_finite_diff=1)


def test_cpu_entry_point_with_finite_differences():
    """runiLQR_CPU with USE_FINITE_DIFF 1 (integratorGradientThreaded's finite-difference definition, nisInitHelpers.cuh:185-201): the product's CPU path
    against the oracle's CPU path, float64."""
    from test_cpu_twin import run_cpu_twin as run_cpu_product      # the ctypes harness of the CPU entry point
    kw = dict(N=32, M=4, A=4, integrator=1, wafr_urdf=1, total_time=0.5, max_iter=5, tol_cost=0.0, use_finite_diff=1, finite_diff_epsilon=1e-5)
    x, u, xg = example_inputs(4, 32, np.float64)
    got = run_cpu_product(4, np.float64, x, u, xg, cores=4, **kw)
    ref = Oracle(default_cfg(4, cores=4, spawn_threads=0, **kw), np.float64).run_ilqr_cpu(x, u, xg)
    it = ref["iters"]
    assert got["iters"] == it and list(got["alphaOut"][: it + 1]) == list(ref["alphaOut"][: it + 1])
    np.testing.assert_allclose(got["Jout"][: it + 1], ref["Jout"][: it + 1], rtol=1e-7)
