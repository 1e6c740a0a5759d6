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
    coop, lg, lg_packed = s.plant_eval(0, x, u), s.plant_eval(4, x, u), s.plant_eval(6, x, u)
    assert np.isfinite(coop).all()
    assert np.array_equal(coop, lg)
    assert np.array_equal(coop, lg_packed)          # the forward pass's variant: 6x6 products two rows per packed instruction


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


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("N,M", [(32, 4), (16, 1), (64, 8)])
def test_lane_group_backward_pass_bit_identical_to_cooperative(backend, dtype, N, M):
    """k_bp_lg (PHASE_BP for the arm) against the wave-cooperative k_bp (PHASE_BP_COOP) from identical inputs: boundary
    cost-to-go, defects, shifted trajectory, regulariser, arbitrary (non-diagonal) cost Hessians."""
    import os
    import pyddp
    from oracle_binding import example_inputs
    n, m, nm = 14, 7, 21
    # small batches default to the cooperative kernel: pin the lane-group one for PHASE_BP
    s = make_solver(backend, 4, dtype=0 if dtype == np.float32 else 1, N=N, M=M, A=2, wafr_urdf=1, total_time=0.5, batch=3, kernels=dict(bp="lg"))
    rng = np.random.default_rng(5)
    xs, us, gs = [], [], []
    for b in range(3):
        x, u, xg = example_inputs(4, N, dtype, noise=rng.normal(0, 0.01, (N, n)))
        xs.append(x); us.append(u); gs.append(xg)
    s.load(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    NB = N // M
    H = s.get("H").reshape(3, N, nm, nm).copy()
    H += (rng.normal(0, 1e-3, H.shape)).astype(dtype)                  # break the diagonal structure
    Pp, pp, d = np.zeros((3, N, n, n), dtype), np.zeros((3, N, n), dtype), np.zeros((3, N, n), dtype)
    for pb in range(3):
        for b in range(M - 1):
            k = NB * (b + 1) - 1
            Q = rng.normal(0, 1, (n, n)); Pp[pb, k] = (Q @ Q.T / n + np.eye(n)) * 10; pp[pb, k] = rng.normal(0, 1, n); d[pb, k] = rng.normal(0, 0.01, n)
    xb = s.get("xb").reshape(3, 2, N, n).copy()
    xb[:, 1] = xb[:, 0] + rng.normal(0, 0.005, (3, N, n)).astype(dtype)
    st = s.get_state()
    for pb in range(3):
        st[pb].rho = 3.5 + pb; st[pb].cur = 0; st[pb].cur2 = 1
    outs = {}
    for phase in (pyddp.PHASE_BP_COOP, pyddp.PHASE_BP):
        s.set("H", H); s.set("Pp", Pp); s.set("pp", pp); s.set("dcur", d); s.set("xb", xb); s.set_state(st)
        for name in ("P", "p", "KT", "du", "ApBK", "Bdu", "dJexp"):
            s.set(name, np.zeros_like(s.get(name)))
        s.set("err", np.ones(3 * M, np.int32))
        s.run_phase(phase)
        outs[phase] = {name: s.get(name).copy() for name in ("P", "p", "KT", "du", "ApBK", "Bdu", "dJexp", "err")}
    ref, lg = outs[pyddp.PHASE_BP_COOP], outs[pyddp.PHASE_BP]
    assert np.abs(ref["KT"]).max() > 0 and np.isfinite(ref["P"]).all()
    for name in ref:
        assert np.array_equal(ref[name], lg[name]), name


@pytest.mark.parametrize("backend", BACKENDS)
def test_whole_solve_identical_with_either_backward_pass(backend):
    """The same solves with the lane-group and with the cooperative backward pass: identical cost traces, bit for bit."""
    import os
    from oracle_binding import example_inputs
    kw = dict(N=64, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=10, batch=4)
    xs, us, gs = [], [], []
    for b in range(4):
        x, u, xg = example_inputs(4, 64, np.float32, noise=RNG.normal(0, 0.002 * (b + 1), (64, 14)))
        xs.append(x); us.append(u); gs.append(xg)
    outs = {}
    for mode in ("lg", "coop", "wide"):
        s = make_solver(backend, 4, kernels=dict(bp=mode), **kw)
        outs[mode] = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    assert np.array_equal(outs["lg"]["Jout"], outs["coop"]["Jout"]) and np.array_equal(outs["lg"]["alphaOut"], outs["coop"]["alphaOut"])
    assert np.array_equal(outs["lg"]["x"], outs["coop"]["x"]) and np.array_equal(outs["lg"]["KT"], outs["coop"]["KT"])
    assert np.array_equal(outs["wide"]["Jout"], outs["coop"]["Jout"]) and np.array_equal(outs["wide"]["KT"], outs["coop"]["KT"])
    assert outs["lg"]["Jout"][0][10] < outs["lg"]["Jout"][0][0]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rollout,N,M,A", [(0, 32, 4, 8), (1, 32, 4, 8), (0, 64, 8, 16)])
def test_end_effector_cost_identical_on_lane_groups_and_cooperative_kernels(backend, rollout, N, M, A):
    """The end-effector cost family (ee_cost_lg.hpp vs ee_cost.hpp): tool point, Jacobian, Gauss-Newton Hessian, in-rollout cost accumulation --
    whole float32 solves on the lane-group kernels and on the wave-cooperative ones (kernels fp=coop) give the same bits."""
    import os
    kw = dict(N=N, M=M, A=A, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, max_iter=8 if N == 32 else 4, ee_cost=1, ignore_max_rho_exit=0, batch=3,
              Q_EE2=0.02, QF_EE2=3.0, Q_xEE=0.05)
    B = 3
    x0 = np.zeros((B, N, 14), np.float32); x0[:, :, 1] = 0.7; x0[:, :, 3] = -0.8; x0[:, :, 5] = 0.75
    x0 += RNG.normal(0, 0.02, (B, 1, 14)).astype(np.float32)
    u0 = np.full((B, N, 7), 0.01, np.float32)
    xg = np.zeros((B, 14), np.float32); xg[:, :6] = [0.45, 0.15, 0.75, 0.1, -0.2, 0.3]; xg[:, 1] += np.float32(0.05) * np.arange(B, dtype=np.float32)
    outs, arrs = {}, {}
    for mode in ("lg", "coop"):
        # fp "lg": keep the lane-group kernels (few problems in flight default to the thread-lane pipeline, fp_pipe.hpp);
        # sweep "alpha": the per-candidate linear sweep (k_sweep_lg), the one whose operation order equals the cooperative kernel's
        s = make_solver(backend, 4, kernels=dict(fp=mode, sweep="alpha"), **kw)
        s.load(x0, u0, xg, forward_rollout=rollout)
        arrs[mode] = {k: s.get(k).copy() for k in ("H", "g", "AB", "costk")}
        outs[mode] = s.solve(x0, u0, xg, forward_rollout=rollout)
    for k in ("H", "g", "AB"):
        assert np.array_equal(arrs["lg"][k], arrs["coop"][k]), k
    if not rollout:
        assert np.array_equal(arrs["lg"]["costk"], arrs["coop"]["costk"])
    for k in ("Jout", "alphaOut", "x", "u", "KT"):
        assert np.array_equal(outs["lg"][k], outs["coop"][k]), k
    assert any(outs["lg"]["alphaOut"][b][1: outs["lg"]["iters"][b] + 1].max() >= 0 for b in range(B))
