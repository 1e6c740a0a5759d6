"""End-effector cost family of the KUKA arm (plants/cost_arm.cuh:206-389, compute_eePos
plants/dynamics_arm.cuh:1879-1925, in-simulation cost accumulation fpHelpers.cuh:259-265,298-300, costKern<T,MODE> :169-190, the EE branch
of costGradientHessianKern nisInitHelpers.cuh:52-84).

The oracle's restatement of this family is PINNED by the reference's own statements executed at fixture-generation time (tests/test_phase_pins.py: tool point,
in-sim cost, gradient / Gauss-Newton Hessian, a whole EE_COST solve -- bit for bit).  What is checked here:
  * the oracle's restatement against finite differences of itself (kinematics Jacobian, cost gradient) and against an independent
    numpy forward-kinematics of the tool point built from the model tables;
  * the kernels against the oracle: per-knot cost / gradient / Gauss-Newton Hessian (teacher-forced), whole solves in float64 with
    identical step-size decisions (with and without the initial rollout), receding-horizon sequences including the reference's
    initial-cost quirk after a solve that ended on a shortened step.
float32 whole solves: this problem amplifies a one-ulp change of the INPUT to 2e-4 in the first accepted cost and 2e-3 two iterations later
in the oracle itself (measured; the Gauss-Newton Hessian ignores the cost weights, so the steps are long and the line search does the
work), hence the loose solver-level float32 bounds; the float32 arithmetic itself is checked knot by knot."""
import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import Oracle, OracleMpc, default_cfg

RNG = np.random.default_rng(11)
EE = dict(N=32, M=4, A=8, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, max_iter=15, ee_cost=1, ignore_max_rho_exit=0)


def start(N, dtype):
    """loadInitialState mode 1-like pose (utils/exampleUtils.cuh:40-46) held over the horizon, u = 0.01 (loadTraj :49-58)."""
    x0 = np.zeros((N, 14), dtype); x0[:, 1] = 0.7; x0[:, 3] = -0.8; x0[:, 5] = 0.75
    u0 = np.full((N, 7), 0.01, dtype)
    xg = np.zeros(14, dtype); xg[:3] = [0.45, 0.15, 0.75]
    return x0, u0, xg


# ------------------------------------------------------------------------------------------------ the oracle against analysis
def fk_tool_point(q, wafr_urdf=1, z=0.0635):
    """Independent forward kinematics from the committed model tables: T = prod_i F_i Rz(q_i); tool point = T_7 [0, 0, z, 1]."""
    import os
    import re
    text = open(os.path.join(os.path.dirname(__file__), "..", "oracle", "iiwa14_model_data.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    blk = text[text.index("IIWA14_JOINT_FRAME"):]
    blk = blk[blk.index("=") + 1: blk.index("};")]
    vals = [float(v) for v in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", blk)]
    assert len(vals) == 2 * 7 * 16
    F = np.asarray(vals).reshape(2, 7, 4, 4).transpose(0, 1, 3, 2)[wafr_urdf]      # stored column-major
    T = np.eye(4)
    for i in range(7):
        c, s = np.cos(q[i]), np.sin(q[i])
        T = T @ F[i] @ np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    return T @ np.array([0, 0, z, 1.0]), T


def test_oracle_tool_point_matches_independent_forward_kinematics():
    o = Oracle(default_cfg(4, **EE), np.float64)
    for _ in range(8):
        x = RNG.normal(0, 1.2, 14)
        pos, _ = o.ee_pos(x, jac=False)
        p, T = fk_tool_point(x[:7])
        np.testing.assert_allclose(pos[:3], p[:3], atol=1e-12)
        np.testing.assert_allclose(pos[3], np.arctan2(T[2, 1], T[2, 2]), atol=1e-12)
        np.testing.assert_allclose(pos[4], np.arctan2(-T[2, 0], np.hypot(T[2, 1], T[2, 2])), atol=1e-12)
        np.testing.assert_allclose(pos[5], np.arctan2(T[1, 0], T[0, 0]), atol=1e-12)


def test_oracle_jacobian_and_cost_gradient_against_finite_differences():
    o = Oracle(default_cfg(4, Q_EE2=0.02, QF_EE2=3.0, Q_xEE=0.05, **EE), np.float64)
    goal = np.array([0.4, -0.1, 0.7, 0.1, -0.2, 0.3])
    eps = 1e-6
    for k in (3, 31):
        x, u = RNG.normal(0, 1.0, 14), RNG.normal(0, 5.0, 7)
        pos, J = o.ee_pos(x)
        Jfd = np.stack([(o.ee_pos(x + eps * np.eye(14)[j], False)[0] - o.ee_pos(x - eps * np.eye(14)[j], False)[0]) / (2 * eps) for j in range(7)])
        np.testing.assert_allclose(J, Jfd, atol=2e-8)
        H, g = o.ee_cost_grad(x, u, goal, k)
        z = np.concatenate([x, u])
        gfd = np.array([(o.ee_cost(*np.split(z + eps * np.eye(21)[i], [14]), goal, k) - o.ee_cost(*np.split(z - eps * np.eye(21)[i], [14]), goal, k)) / (2 * eps)
                        for i in range(21)])
        np.testing.assert_allclose(g, gfd, rtol=1e-6, atol=1e-6)
        # the Hessian is the UNWEIGHTED Gauss-Newton term J'J plus the diagonal weights (cost_arm.cuh:366 has the weights commented out)
        fin = k == 31
        diag = np.concatenate([np.full(7, 0.0 if fin else 0.05), np.full(7, 1000.0 if fin else 0.1), np.full(7, 0.0 if fin else 1e-4)])
        Hexp = np.diag(diag); Hexp[:7, :7] += J @ J.T
        np.testing.assert_allclose(H, Hexp, atol=1e-12)


# ------------------------------------------------------------------------------------------------ kernels against the oracle
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 3e-5)])
def test_setup_kernel_cost_gradient_hessian_knot_by_knot(backend, dtype, tol):
    """initAlgGPU's costGradientHessianKern on a random trajectory: H_k, g_k and the per-knot cost d_JT[k], then costKern<T,1>."""
    N = 32
    kw = {**EE, "Q_EE2": 0.02, "QF_EE2": 3.0, "Q_xEE": 0.05}
    s = make_solver(backend, 4, dtype=0 if dtype == np.float32 else 1, **kw)
    o = Oracle(default_cfg(4, cores=8, spawn_threads=0, **kw), dtype)
    x = RNG.normal(0, 0.8, (N, 14)).astype(dtype); u = RNG.normal(0, 5.0, (N, 7)).astype(dtype)
    goal = np.zeros(14, dtype); goal[:6] = [0.4, -0.1, 0.7, 0.1, -0.2, 0.3]
    s.load(x, u, goal)
    H, g, ck = s.get("H").reshape(N, 21, 21), s.get("g").reshape(N, 21), s.get("costk")
    scale_H = scale_g = 0.0
    refs = [o.ee_cost_grad(x[k], u[k], goal[:6], k) for k in range(N)]
    scale_H = max(np.abs(r[0]).max() for r in refs); scale_g = max(np.abs(r[1]).max() for r in refs)
    for k in range(N):
        Ho, go = refs[k]
        assert np.abs(H[k] - Ho).max() <= tol * scale_H, k
        assert np.abs(g[k] - go).max() <= tol * scale_g, k
        co = o.ee_cost(x[k], u[k], goal[:6], k)
        assert abs(ck[k] - co) <= tol * max(abs(co), 1.0), k
    st = s.get_state()[0]
    ref_total = o.run_ilqr_gpusem(x.ravel(), u.ravel(), goal)["Jout"][0]
    assert abs(st.prevJ - 2 * kw["tol_cost"] - ref_total) <= max(tol, 1e-6) * abs(ref_total)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rollout", [0, 1])
@pytest.mark.parametrize("M,A", [(4, 8), (1, 8), (2, 3)])
def test_float64_whole_solve(backend, rollout, M, A):
    kw = {**EE, "M": M, "A": A}
    N = kw["N"]
    s = make_solver(backend, 4, dtype=1, **kw)
    o = Oracle(default_cfg(4, cores=8, spawn_threads=0, **kw), np.float64)
    x0, u0, xg = start(N, np.float64)
    r = o.run_ilqr_gpusem(x0.ravel(), u0.ravel(), xg, rollout=rollout)
    out = s.solve(x0, u0, xg, forward_rollout=rollout)
    it = r["iters"]
    assert out["iters"][0] == it
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    if not (rollout and M == 1) and A == 8:        # (the other scenarios may be rejected throughout: the rho schedule and exit are what they exercise)
        assert sum(a >= 0 for a in r["alphaOut"][1: it + 1]) >= 1, "the scenario must contain accepted steps"
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-8)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-8 * np.abs(r["x"]).max())
    np.testing.assert_allclose(out["u"][0].ravel(), r["u"], rtol=0, atol=1e-7 * np.abs(r["u"]).max())


@pytest.mark.parametrize("backend", BACKENDS)
def test_float32_whole_solve_leading_iterations(backend):
    N = EE["N"]
    s = make_solver(backend, 4, dtype=0, **EE)
    o = Oracle(default_cfg(4, cores=8, spawn_threads=0, **EE), np.float32)
    x0, u0, xg = start(N, np.float32)
    r = o.run_ilqr_gpusem(x0.ravel(), u0.ravel(), xg)
    out = s.solve(x0, u0, xg)
    assert list(out["alphaOut"][0][:4]) == list(r["alphaOut"][:4])
    np.testing.assert_allclose(out["Jout"][0][0], r["Jout"][0], rtol=2e-6)
    np.testing.assert_allclose(out["Jout"][0][1], r["Jout"][1], rtol=5e-4)        # see the module docstring: input-ulp sensitivity is 2e-4 here
    np.testing.assert_allclose(out["Jout"][0][:6], r["Jout"][:6], rtol=2e-2)
    it = int(out["iters"][0])
    assert out["Jout"][0][it] < 0.95 * out["Jout"][0][0]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cost_shift", [0, 1])
def test_receding_horizon_with_the_end_effector_cost(backend, cost_shift):
    """runiLQR_MPC_GPU with EE_COST (the configuration of examples/WAFR_MPC_examples.cu): warm-started solves against the oracle's
    persistent state, including the reference's initial cost read from d_JT[alphaIndex] (nisInitHelpers.cuh:392): after a solve that
    ended with step-size index a > 0 the next solve starts from prevJ = cost of knot a, fails every line search and falls back."""
    kw = {**EE, "max_iter": 8, "ee_cost_shift": cost_shift}
    N = kw["N"]
    s = make_solver(backend, 4, dtype=1, **kw)
    o = OracleMpc(default_cfg(4, cores=8, spawn_threads=0, **kw), np.float64)
    x0, u0, xg = start(N, np.float64)
    s.load(x0, u0, xg)
    o.set_traj(x0.ravel(), u0.ravel())
    xact = x0[0].copy()
    plan = [(0, 1, 8), (1, 0, 4), (2, 0, 4), (1, 0, 3), (1, 0, 4), (2, 0, 4)]
    quirk_seen = 0
    prev_alpha_end = 0
    for step, (shift, clear, mi) in enumerate(plan):
        r = o.mpc_solve(xact, xg, shift, clear_vars=clear, full_rollout=1, max_iter=mi)
        g = s.mpc_solve(xact, xg, shift, clear_vars=clear, full_rollout=1, max_iter=mi)
        it = r["iters"]
        assert g["iters"][0] == it and g["success"][0] == r["success"], (step, g["iters"], it, g["success"], r["success"])
        assert list(g["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1]), step
        np.testing.assert_allclose(g["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-6, err_msg=str(step))
        if prev_alpha_end > 0:
            quirk_seen += 1
            assert r["success"] == 0 and all(a == -1 for a in r["alphaOut"][1: it + 1]), "prevJ = one knot's cost: nothing can improve on it"
        if r["success"]:
            np.testing.assert_allclose(g["x"][0].ravel(), r["x"], rtol=0, atol=1e-6 * max(np.abs(r["x"]).max(), 1.0))
            np.testing.assert_allclose(g["u"][0].ravel()[: (N - 1) * 7], r["u"][: (N - 1) * 7], rtol=0, atol=1e-6 * max(np.abs(r["u"]).max(), 1.0))
        else:
            break          # the fall-back's last knots are stale reference buffers (tests/test_mpc_parity.py)
        acc = [a for a in r["alphaOut"][1: it + 1] if a >= 0]
        prev_alpha_end = 0 if r["alphaOut"][it] == -1 else r["alphaOut"][it]
        nxt = plan[step + 1][0] if step + 1 < len(plan) else 0
        xact = g["x"][0][nxt] + RNG.normal(0, 0.0002, 14)
    assert step >= 1


@pytest.mark.parametrize("backend", BACKENDS)
def test_initial_cost_fix_flag_is_opt_in_and_only_changes_the_quirk(backend):
    """pddp_config.ee_initial_cost_fix = 1: the warm-started solve after a solve that ended on a shortened step starts from the whole
    trajectory's cost and makes progress; with the default 0 it reproduces the reference (previous test)."""
    kw = {**EE, "max_iter": 8}
    N = kw["N"]
    x0, u0, xg = start(N, np.float64)
    res = {}
    for fix in (0, 1):
        s = make_solver(backend, 4, dtype=1, ee_initial_cost_fix=fix, **kw)
        s.load(x0, u0, xg)
        first = s.mpc_solve(x0[0], xg, 0, clear_vars=1, max_iter=8)
        assert first["alphaOut"][0][first["iters"][0]] > 0          # ends on a shortened step: the quirk's precondition
        res[fix] = (first, s.mpc_solve(first["x"][0][1], xg, 1, max_iter=4))
    assert np.array_equal(res[0][0]["Jout"], res[1][0]["Jout"])        # the first solve (alphaIndex = 0 at its start) is untouched
    q, f = res[0][1], res[1][1]
    assert q["success"][0] == 0 and q["Jout"][0][0] < 1.0              # one knot's cost
    assert f["success"][0] == 1 and f["Jout"][0][0] > 1.0 and f["Jout"][0][f["iters"][0]] < f["Jout"][0][0]


def _replay(kw, plan, xacts, xg, x0, u0):
    o = OracleMpc(default_cfg(4, cores=8, spawn_threads=0, **kw), np.float64)
    o.set_traj(x0.ravel(), u0.ravel())
    r = None
    for (shift, clear, mi), xa in zip(plan, xacts):
        r = o.mpc_solve(xa, xg, shift, clear_vars=clear, full_rollout=1, max_iter=mi)
    return r


@pytest.mark.parametrize("backend", BACKENDS)
def test_receding_horizon_chain_of_successful_end_effector_solves(backend):
    """The healthy case of the same loop: every solve of the chain ends with alphaIndex = 0 (last iteration rejected, or a full step), so the
    next one starts from the true initial cost.  The per-solve iteration caps that make it so are found on the oracle first."""
    kw = {**EE, "max_iter": 12}
    N = kw["N"]
    x0, u0, xg = start(N, np.float64)
    xg[:3] = [0.3, 0.1, 0.95]
    rng = np.random.default_rng(5)
    plan, xacts, xa = [], [], x0[0].copy()
    for step, shift in enumerate([0, 1, 2, 1]):
        pick = None
        for mi in range(2, 13):
            r = _replay(kw, plan + [(shift, 1 if step == 0 else 0, mi)], xacts + [xa], xg, x0, u0)
            if r["success"] and r["alphaOut"][r["iters"]] <= 0:
                pick = (mi, r); break
        if pick is None:
            break
        plan.append((shift, 1 if step == 0 else 0, pick[0])); xacts.append(xa)
        nxt = [0, 1, 2, 1, 0][step + 1]
        xa = pick[1]["x"].reshape(N, 14)[nxt] + rng.normal(0, 0.0002, 14)
    assert len(plan) >= 3, plan
    s = make_solver(backend, 4, dtype=1, **kw)
    o = OracleMpc(default_cfg(4, cores=8, spawn_threads=0, **kw), np.float64)
    s.load(x0, u0, xg); o.set_traj(x0.ravel(), u0.ravel())
    for step, ((shift, clear, mi), xa) in enumerate(zip(plan, xacts)):
        r = o.mpc_solve(xa, xg, shift, clear_vars=clear, full_rollout=1, max_iter=mi)
        g = s.mpc_solve(xa, xg, shift, clear_vars=clear, full_rollout=1, max_iter=mi)
        it = r["iters"]
        assert g["iters"][0] == it and g["success"][0] == r["success"] == 1, step
        assert list(g["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1]), step
        np.testing.assert_allclose(g["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-6, err_msg=str(step))
        np.testing.assert_allclose(g["x"][0].ravel(), r["x"], rtol=0, atol=1e-6 * max(np.abs(r["x"]).max(), 1.0))
        if step > 0:
            assert r["Jout"][0] > 1.0, "the warm-started solve must start from the whole trajectory's cost, not from one knot's"


@pytest.mark.gpu
def test_end_effector_cost_at_the_mpc_example_shape_batch():
    """BASELINE configs[3] stage 4b: 64 independent rollouts x 8 alphas, N=64, M=4, EE cost, MPC mode, float32; every problem of the batch
    equals the same problem solved alone (bitwise), and the costs decrease."""
    kw = dict(N=64, M=4, A=8, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, max_iter=10, ee_cost=1, ignore_max_rho_exit=0)
    B = 64
    rng = np.random.default_rng(2024)      # own stream: the inputs must not depend on which tests ran before
    x0 = np.zeros((B, 64, 14), np.float32); x0[:, :, 1] = 0.7; x0[:, :, 3] = -0.8; x0[:, :, 5] = 0.75
    x0 += rng.normal(0, 0.01, (B, 1, 14)).astype(np.float32)
    u0 = np.full((B, 64, 7), 0.01, np.float32)
    t = np.linspace(0, 2 * np.pi, B, endpoint=False)
    xg = np.zeros((B, 14), np.float32); xg[:, 0] = 0.5 + 0.1 * np.cos(t); xg[:, 1] = 0.2 * np.sin(t); xg[:, 2] = 0.6 + 0.1 * np.sin(2 * t)
    sb = make_solver("hip", 4, batch=B, **kw)
    out = sb.solve(x0, u0, xg)
    for b in (0, 17, 63):
        s1 = make_solver("hip", 4, batch=1, **kw)
        o1 = s1.solve(x0[b], u0[b], xg[b])
        assert np.array_equal(o1["Jout"][0], out["Jout"][b]) and np.array_equal(o1["x"][0], out["x"][b])
    it = out["iters"]
    # (a rejected iteration records prevJ, which carries the 2 TOL_COST epsilon of initAlgGPU until the first acceptance, nisInitHelpers.cuh:393)
    assert all(out["Jout"][b][it[b]] <= out["Jout"][b][0] + 3 * kw["tol_cost"] for b in range(B))
    improved = sum(out["Jout"][b][it[b]] < 0.98 * out["Jout"][b][0] for b in range(B))
    assert improved >= B // 4, (improved, [float(out["Jout"][b][0]) for b in range(4)], [float(out["Jout"][b][it[b]]) for b in range(4)])


@pytest.mark.gpu
def test_baseline_config3_at_size_float64_every_rollout_follows_the_oracle():
    """BASELINE configs[3] at its stated size (stage 4b): 64 concurrent rollouts x 8 alphas, N=64, M=4, T=0.5 s,
    MPC_MODE (gravity 0), end-effector cost, TOL_COST 1e-5 -- as ONE float64 handle of 64 problems (goal r at phase r/64 of a figure, start
    pose perturbed per rollout).  Every rollout must follow the oracle's GPU-semantics driver: identical step-size indices (rejections
    included), J / x / u to 1e-8.  (Sharded 8 per GPU the same problems run independently: tests/test_shard.py covers the partition.)"""
    kw = dict(N=64, M=4, A=8, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, max_iter=8, ee_cost=1, ignore_max_rho_exit=0)
    B = 64
    rng = np.random.default_rng(64)
    x0 = np.zeros((B, 64, 14)); x0[:, :, 1] = 0.7; x0[:, :, 3] = -0.8; x0[:, :, 5] = 0.75
    x0[:, :, :7] += rng.normal(0, 0.01, (B, 1, 7))
    u0 = np.full((B, 64, 7), 0.01)
    t = np.linspace(0, 2 * np.pi, B, endpoint=False)
    xg = np.zeros((B, 14)); xg[:, 0] = 0.5 + 0.1 * np.cos(t); xg[:, 1] = 0.2 * np.sin(t); xg[:, 2] = 0.6 + 0.1 * np.sin(2 * t)
    s = make_solver("hip", 4, dtype=1, batch=B, **kw)
    out = s.solve(x0, u0, xg)
    o = Oracle(default_cfg(4, cores=1, spawn_threads=0, **kw), np.float64)
    for r in range(B):
        ref = o.run_ilqr_gpusem(x0[r].ravel(), u0[r].ravel(), xg[r])
        it = ref["iters"]
        assert out["iters"][r] == it, r
        assert list(out["alphaOut"][r][: it + 1]) == list(ref["alphaOut"][: it + 1]), r
        np.testing.assert_allclose(out["Jout"][r][: it + 1], ref["Jout"][: it + 1], rtol=1e-8)
        np.testing.assert_allclose(out["x"][r].ravel(), ref["x"], rtol=0, atol=1e-8 * max(np.abs(ref["x"]).max(), 1))
        np.testing.assert_allclose(out["u"][r].ravel(), ref["u"], rtol=0, atol=1e-7 * max(np.abs(ref["u"]).max(), 1))
