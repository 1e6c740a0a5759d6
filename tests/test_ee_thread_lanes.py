"""End-effector cost family on the THREAD-LANE / matrix-core kernels (k_fp_tl<EE>, k_nis_tl<EE>, k_bp_mfma with the compact position block of the Gauss-Newton Hessian):
what BASELINE configs[3] (64 concurrent Kuka MPC rollouts, end-effector cost) and large end-effector batches run from 512 problems up (kernels fp=tl, bp=mx pin the
selection on small handles).  Checked against the oracle, whose end-effector family is pinned by the reference's own statements (tests/test_phase_pins.py):

  * tool point + Jacobian of the thread-lane world chain (plant_arm_tl.hpp) against compute_eePos's restatement, float64 1e-9;
  * setup kernel: H_k, g_k, per-knot cost knot by knot (float64 1e-9 of the largest entry; float32 1e-4);
  * float64 whole solves: identical step-size decisions, J / x / u to 1e-8 -- with / without the initial rollout, M in {4, 1, 2};
  * (GPU) a float32 batch equals single-problem solves bit for bit, and follows the float32 oracle over the leading iterations.
"""
import os

import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import Oracle, default_cfg

RNG = np.random.default_rng(77)
EE = dict(N=32, M=4, A=8, wafr_urdf=1, mpc_mode=1, ee_cost=1, tol_cost=1e-5, total_time=0.5, max_iter=8, ignore_max_rho_exit=0)
TL = dict(fp="tl", bp="mx")


def tl_solver(backend, **kw):
    return make_solver(backend, 4, **kw, kernels=dict(TL))


def start(N, dtype):
    x0 = np.zeros((N, 14), dtype); x0[:, 1] = 0.7; x0[:, 3] = -0.8; x0[:, 5] = 0.75
    u0 = np.full((N, 7), 0.01, dtype)
    xg = np.zeros(14, dtype); xg[:3] = [0.45, 0.15, 0.75]
    return x0, u0, xg


@pytest.mark.parametrize("backend", BACKENDS)
def test_thread_lane_tool_point_and_jacobian(backend):
    s = make_solver(backend, 4, dtype=1, **EE)
    o = Oracle(default_cfg(4, **EE), np.float64)
    x = RNG.normal(0, 1.2, (12, 14)); u = np.zeros((12, 7))
    out = s.plant_eval(9, x, u)
    for k in range(12):
        pos, dpos = o.ee_pos(x[k])
        assert np.abs(out[k][:6] - pos).max() <= 1e-9 and np.abs(out[k][6:].reshape(7, 6) - dpos).max() <= 1e-9 * max(1.0, np.abs(dpos).max()), k


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 1e-4)])
def test_setup_kernel_knot_by_knot(backend, dtype, tol):
    N = 32
    kw = {**EE, "Q_EE2": 0.02, "QF_EE2": 3.0, "Q_xEE": 0.05}
    s = tl_solver(backend, dtype=0 if dtype == np.float32 else 1, **kw)
    if backend == "hip":
        assert "k_nis_tl" in dict(s.time_kernels(1))
    o = Oracle(default_cfg(4, cores=8, spawn_threads=0, **kw), dtype)
    x = RNG.normal(0, 0.8, (N, 14)).astype(dtype); u = RNG.normal(0, 5.0, (N, 7)).astype(dtype)
    goal = np.zeros(14, dtype); goal[:6] = [0.4, -0.1, 0.7, 0.1, -0.2, 0.3]
    s.load(x, u, goal)
    H, g, ck = s.get("H").reshape(N, 21, 21), s.get("g").reshape(N, 21), s.get("costk")
    refs = [o.ee_cost_grad(x[k], u[k], goal[:6], k) for k in range(N)]
    scale_H = max(np.abs(r[0]).max() for r in refs); scale_g = max(np.abs(r[1]).max() for r in refs)
    for k in range(N):
        Ho, go = refs[k]
        assert np.abs(H[k] - Ho).max() <= tol * scale_H, k
        assert np.abs(g[k] - go).max() <= tol * scale_g, k
        co = o.ee_cost(x[k], u[k], goal[:6], k)
        assert abs(ck[k] - co) <= tol * max(abs(co), 1.0), k
    # ... and the same arrays after a sweep (mode 0: the position block travels compact, "H" is the expanded view)
    s.iterate(1); s.sync()
    st = s.get_state()[0]
    xc = s.get("xb").reshape(2, N, 14)[st.cur]; uc = s.get("ucur").reshape(N, 7)
    H2 = s.get("H").reshape(N, 21, 21)
    for k in range(0, N, 5):
        Ho, _ = o.ee_cost_grad(xc[k], uc[k], goal[:6], k)
        assert np.abs(H2[k] - Ho).max() <= max(tol, 1e-6) * max(np.abs(Ho).max(), 1.0), k


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rollout", [0, 1])
@pytest.mark.parametrize("M,A", [(4, 8), (1, 8), (2, 3)])
def test_float64_whole_solve(backend, rollout, M, A):
    kw = {**EE, "M": M, "A": A}
    N = kw["N"]
    s = tl_solver(backend, dtype=1, **kw)
    if backend == "hip":
        names = dict(s.time_kernels(1))
        assert "k_fp_tl" in names and "k_nis_tl" in names and "k_bp_mfma" in names, names
    o = Oracle(default_cfg(4, cores=8, spawn_threads=0, **kw), np.float64)
    x0, u0, xg = start(N, np.float64)
    r = o.run_ilqr_gpusem(x0.ravel(), u0.ravel(), xg, rollout=rollout)
    out = s.solve(x0, u0, xg, forward_rollout=rollout)
    it = r["iters"]
    assert out["iters"][0] == it
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-8)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-8 * np.abs(r["x"]).max())
    np.testing.assert_allclose(out["u"][0].ravel(), r["u"], rtol=0, atol=1e-7 * np.abs(r["u"]).max())


@pytest.mark.gpu
def test_float32_batch_equals_single_problem_solves_and_follows_the_oracle():
    kw = {**EE, "N": 64, "max_iter": 8}
    B, N = 600, 64
    rng = np.random.default_rng(5)
    xs, us, gs = [], [], []
    for b in range(B):
        x0, u0, xg = start(N, np.float32)
        x0[:, :7] += rng.normal(0, 0.01, (1, 7)).astype(np.float32)
        xg = xg.copy(); xg[1] = 0.2 * np.sin(2 * np.pi * b / B)
        xs.append(x0.ravel()); us.append(u0.ravel()); gs.append(xg)
    s = make_solver("hip", 4, dtype=0, batch=B, **kw)                  # 600 problems: the automatic selection is the thread-lane / matrix-core family
    names = dict(s.time_kernels(1))
    assert "k_fp_tl" in names and "k_nis_tl" in names and "k_bp_mfma" in names, names
    out = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    s1 = tl_solver("hip", dtype=0, batch=1, **kw)
    o32 = Oracle(default_cfg(4, cores=1, spawn_threads=0, **kw), np.float32)
    lead = []
    for b in rng.choice(B, 6, replace=False):
        o1 = s1.solve(xs[b], us[b], gs[b])
        for k in ("alphaOut", "Jout", "x", "u"):
            assert np.array_equal(o1[k][0], out[k][b]), (int(b), k)
        r = o32.run_ilqr_gpusem(xs[b], us[b], gs[b])
        n_same = next((i for i in range(9) if out["alphaOut"][b][i] != r["alphaOut"][i]), 9)
        lead.append(n_same)
        np.testing.assert_allclose(out["Jout"][b][0], r["Jout"][0], rtol=1e-5)
    assert np.median(lead) >= 3, lead
