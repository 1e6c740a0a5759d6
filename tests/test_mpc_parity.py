"""MPC wrapper: pddp_mpc_solve against the oracle's restatement of loadVarsGPU_MPC / runiLQR_MPC_GPU /
storeVarsGPU_MPC (DDPHelpers/MPCHelpers.cuh:602-655, 864-1016, 755-774; joint-space cost).  A receding-horizon sequence: every solve
shifts the previous solution, rolls it out from a perturbed "measured" state and iterates with a small max_iter.  float64: identical
step-size indices, iteration counts and success flags; J, x, u, K to 1e-6 (the open-loop rollouts of the unstable arm amplify the last
bits of a float64 by four to five orders of magnitude over a horizon; costs reach 1e7 in the failed solves of this sequence)."""
import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import OracleMpc, default_cfg, example_inputs

def _rng(*key):
    """A generator per test (and parameter set): the inputs of a test must not depend on which tests ran before it in the process (pytest -k / -n / -m gpu select
    different subsets; until round 6 one module-level generator made every test's data a function of the run's selection and order)."""
    return np.random.default_rng([42, *[int(k) for k in key]])


def close(a, b, tol=1e-6):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1.0)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("plant,kw,full", [
    (4, dict(N=32, M=4, A=8, wafr_urdf=1, mpc_mode=1, total_time=0.5, tol_cost=1e-5, max_iter=8), 1),     # the MPC example's defaults: FULL_ROLLOUT 1
    (4, dict(N=32, M=4, A=8, wafr_urdf=1, mpc_mode=1, total_time=0.5, tol_cost=1e-5, max_iter=8), 0),     # segment rollout + tail with feedback
    (2, dict(N=32, M=2, A=8, integrator=3, total_time=2.0, tol_cost=1e-5, max_iter=8), 1),
])
def test_receding_horizon_sequence(backend, plant, kw, full):
    RNG = _rng(1, plant, full)
    dtype = np.float64
    N = kw["N"]
    s = make_solver(backend, plant, dtype=1, **kw)
    o = OracleMpc(default_cfg(plant, cores=8, spawn_threads=0, **kw), dtype)
    n, m = o.n, o.m
    x0, u0, xg = example_inputs(plant, N, dtype, noise=RNG.normal(0, 0.001, (N, n)))
    if plant == 4:
        u0 = np.full(N * m, 0.01)                      # MPC_MODE: no gravity to compensate (utils/exampleUtils.cuh:49-58)
        xg = x0[:n].copy(); xg[:7] += 0.1; xg[7:] = 0  # a nearby goal, as in tracking: the open-loop rollouts of the warm start stay tame
    s.load(x0, u0, xg)                                 # seeds the handle's "previous solution"
    o.set_traj(x0, u0)
    xact = x0[:n].copy()
    plan = [(0, 1, 6), (2, 0, 4), (1, 0, 4), (3, 0, 3), (2, 0, 3), (1, 0, 4), (4, 0, 2)]      # (shift, clear_vars, max_iter); see the fall-back note below
    compared = 0
    for step, (shift, clear, mi) in enumerate(plan):
        r = o.mpc_solve(xact, xg, shift, clear_vars=clear, full_rollout=full, max_iter=mi)
        g = s.mpc_solve(xact, xg, shift, clear_vars=clear, full_rollout=full, max_iter=mi)
        it = r["iters"]
        assert g["iters"][0] == it and g["success"][0] == r["success"], (step, g["iters"], it, g["success"], r["success"])
        assert list(g["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1]), step
        assert close(g["Jout"][0][: it + 1], r["Jout"][: it + 1]), step
        # A failed solve falls back to the reference's d_xp / d_up.  Those are only (re)written by a shift > 0 without clear_vars, and even
        # then shiftAndCopy never writes their knot DIM_N - 1: otherwise they hold leftovers of earlier iterations.  The library stores the
        # shifted CURRENT solution instead; compare what is well defined in both.
        kx = N - 1 if not r["success"] else N
        ku = N - 2 if not r["success"] else N
        if r["success"] or (shift > 0 and not clear):
            assert close(g["x"][0][:kx], r["x"].reshape(N, n)[:kx]), step
            assert close(g["u"][0][:ku], r["u"].reshape(N, m)[:ku]), step
        assert close(g["KT"][0], r["KT"].reshape(N, m, n), 1e-5), step
        if not r["success"]:
            break        # from here on the two sides start from different (stale vs. clean) last knots of the fall-back: see above
        compared += 1
        # next measured state: where the plan says the system should be after `shift_next` knots, plus a disturbance
        nxt = plan[step + 1][0] if step + 1 < len(plan) else 0
        xact = g["x"][0][nxt] + RNG.normal(0, 0.0005, n)
    assert compared >= 3, "the sequence must contain several successful warm-started solves"


@pytest.mark.parametrize("backend", BACKENDS)
def test_mpc_batch_rollouts_are_independent(backend):
    """64-rollout shape of BASELINE configs[3], scaled down: a batch of MPC problems with different measured states and shifts gives,
    problem by problem, exactly what separate single-problem handles give."""
    RNG = _rng(2)
    kw = dict(N=32, M=4, A=8, wafr_urdf=1, mpc_mode=1, total_time=0.5, tol_cost=1e-5, max_iter=5)
    B = 3
    xs, us, gs = [], [], []
    for b in range(B):
        x0, u0, xg = example_inputs(4, 32, np.float32, noise=RNG.normal(0, 0.001, (32, 14)))
        xs.append(x0); us.append(np.full(32 * 7, 0.01, np.float32)); gs.append(xg + np.float32(0.05 * b))
    sb = make_solver(backend, 4, batch=B, **kw)
    sb.load(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    xact = np.stack([x[:14] for x in xs]) + RNG.normal(0, 0.002, (B, 14)).astype(np.float32)
    shifts = np.asarray([0, 2, 1], np.int32)
    first = sb.mpc_solve(xact, np.stack(gs), 0, clear_vars=1)
    second = sb.mpc_solve(first["x"][np.arange(B), shifts], np.stack(gs), shifts)
    for b in range(B):
        s1 = make_solver(backend, 4, batch=1, **kw)
        s1.load(xs[b], us[b], gs[b])
        f1 = s1.mpc_solve(xact[b], gs[b], 0, clear_vars=1)
        f2 = s1.mpc_solve(f1["x"][0][shifts[b]], gs[b], int(shifts[b]))
        assert np.array_equal(f1["Jout"][0], first["Jout"][b]) and np.array_equal(f2["Jout"][0], second["Jout"][b])
        assert np.array_equal(f2["x"][0], second["x"][b]) and f2["success"][0] == second["success"][b]


@pytest.mark.parametrize("backend", BACKENDS)
def test_mpc_batch_with_a_per_call_iteration_limit_keeps_the_trace_rows(backend):
    """A per-call max_iter below config.max_iter must not move the rows of Jout / alphaOut: their stride is fixed at allocation
    (config.max_iter + 2).  Batch of 3 against single-problem handles, trace rows and iteration counts compared."""
    RNG = _rng(3)
    kw = dict(N=32, M=4, A=8, wafr_urdf=1, mpc_mode=1, total_time=0.5, tol_cost=1e-5, max_iter=8)
    B = 3
    xs, us, gs = [], [], []
    for b in range(B):
        x0, u0, xg = example_inputs(4, 32, np.float32, noise=RNG.normal(0, 0.001, (32, 14)))
        xs.append(x0); us.append(np.full(32 * 7, 0.01, np.float32)); gs.append(xg + np.float32(0.05 * b))
    sb = make_solver(backend, 4, batch=B, **kw)
    sb.load(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    xact = np.stack([x[:14] for x in xs]) + RNG.normal(0, 0.002, (B, 14)).astype(np.float32)
    got = sb.mpc_solve(xact, np.stack(gs), 0, clear_vars=1, max_iter=3)
    for b in range(B):
        s1 = make_solver(backend, 4, batch=1, **kw)
        s1.load(xs[b], us[b], gs[b])
        one = s1.mpc_solve(xact[b], gs[b], 0, clear_vars=1, max_iter=3)
        it = int(one["iters"][0])
        assert it == int(got["iters"][b]) and it <= 3
        assert np.array_equal(one["Jout"][0][:it + 1], got["Jout"][b][:it + 1]) and np.array_equal(one["alphaOut"][0][:it + 1], got["alphaOut"][b][:it + 1])
        assert np.array_equal(one["x"][0], got["x"][b])
        s1.close()
    sb.close()


@pytest.mark.gpu
def test_plain_solve_then_polled_mpc_solves_on_one_handle():
    """runiLQR_GPU followed by the MPC loop on the same handle (the canonical flow): the status polls of the plain solve allocate the pinned state
    buffer BEFORE the first pddp_mpc_solve grows its staging area, and an MPC call with max_iter > poll_every polls again.  Polled and un-polled
    calls must give the same bits."""
    RNG = _rng(4)
    kw = dict(N=32, M=4, A=8, wafr_urdf=1, total_time=0.5, tol_cost=0.0, max_iter=8)
    x0, u0, xg = example_inputs(4, 32, np.float32, noise=RNG.normal(0, 0.001, (32, 14)))
    res = []
    for poll in (2, 16):
        s = make_solver("hip", 4, **kw)
        first = s.solve(x0, u0, xg)                                   # pddp_status runs here
        xa = first["x"][0][1].copy()
        seq = [s.mpc_solve(xa, xg, 1, clear_vars=0, max_iter=6, poll_every=poll)]
        seq.append(s.mpc_solve(seq[0]["x"][0][1], xg, 1, clear_vars=0, max_iter=5, poll_every=poll))
        done, iters = s.status()                                      # and once more after the staging area exists
        assert done.all()
        res.append(seq)
        s.close()
    for a, b in zip(*res):
        assert np.isfinite(a["Jout"][0][: int(a["iters"][0]) + 1]).all()
        assert int(a["iters"][0]) == int(b["iters"][0])
        assert np.array_equal(a["Jout"], b["Jout"]) and np.array_equal(a["x"], b["x"]) and np.array_equal(a["alphaOut"], b["alphaOut"])


@pytest.mark.gpu
def test_warm_started_mpc_after_a_plain_solve_shifts_the_whole_cost_to_go():
    """A handle WITHOUT mpc_mode runs a plain solve on the matrix-core backward pass, then a warm-started MPC call (shift > 0, clear_vars = 0): the shift moves
    interior cost-to-go slots into the block boundaries (MPCHelpers.cuh:602-655), so the plain solve must have written them.  float64, against the lane-group
    family (which the receding-horizon tests above hold against the oracle): identical step-size indices, J / x to 1e-6."""
    RNG = _rng(5)
    import os
    kw = dict(N=32, M=4, A=8, wafr_urdf=1, total_time=0.5, tol_cost=0.0, max_iter=6)
    x0, u0, xg = example_inputs(4, 32, np.float64, noise=RNG.normal(0, 0.001, (32, 14)))
    outs = []
    for env in (dict(bp="mx", fp="tl"), {}):
        s = make_solver("hip", 4, dtype=1, **kw, kernels=dict(env))
        first = s.solve(x0, u0, xg)
        xa = first["x"][0][2] + 1e-3
        outs.append((first, s.mpc_solve(xa, xg, 2, clear_vars=0, max_iter=4)))
        s.close()
    (f_mx, m_mx), (f_lg, m_lg) = outs
    assert list(f_mx["alphaOut"][0]) == list(f_lg["alphaOut"][0])
    it = int(m_lg["iters"][0])
    assert int(m_mx["iters"][0]) == it and list(m_mx["alphaOut"][0][: it + 1]) == list(m_lg["alphaOut"][0][: it + 1])
    assert close(m_mx["Jout"][0][: it + 1], m_lg["Jout"][0][: it + 1]) and close(m_mx["x"][0], m_lg["x"][0])
