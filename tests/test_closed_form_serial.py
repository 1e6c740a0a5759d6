"""Thread-serial kernels of the closed-form plants (k_bp_ts / k_fp_ts / k_nis_ts: one THREAD per block of knots / candidate / knot, the same bodies as the
wave-cooperative kernels with a one-lane wave) -- what BASELINE configs[1] (cart-pole N=128, A=8) and configs[4] (quadrotor N=256, RK3, A=16) run once a batch
fills the device.

  * bit for bit the cooperative kernels' results (every output element is computed by one lane / one thread with the same operations in the same order);
  * float64 whole solves follow the oracle's GPU-semantics driver decision for decision -- the oracle whose phases at these plants' sizes (1x1 and 4x4 adjugate
    inversion, the three integration rules) are pinned by the reference's own statements (tests/test_phase_pins.py).
"""
import os

import numpy as np
import pytest

from backends import make_solver
from oracle_binding import Oracle, default_cfg, example_inputs

pytestmark = pytest.mark.gpu
CASES = [pytest.param(2, dict(N=128, M=4, A=8, integrator=3, total_time=4.0, max_iter=10), id="cartpole-config1"),
         pytest.param(3, dict(N=256, M=4, A=16, integrator=3, total_time=4.0, max_iter=5), id="quadrotor-config4"),
         pytest.param(1, dict(N=64, M=4, A=4, integrator=1, total_time=4.0, max_iter=8), id="pendulum"),
         pytest.param(2, dict(N=64, M=1, A=8, integrator=2, total_time=2.0, max_iter=8), id="cartpole-midpoint-single-shooting")]


@pytest.mark.parametrize("plant,kw", CASES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_thread_serial_equals_cooperative_bit_for_bit(plant, kw, dtype):
    B = 5
    rng = np.random.default_rng(17)
    n = {1: 2, 2: 4, 3: 12}[plant]
    xs, us, gs = [], [], []
    for b in range(B):
        x0, u0, xg = example_inputs(plant, kw["N"], dtype, noise=rng.normal(0, 0.001 * (b + 1), (kw["N"], n)))
        xs.append(x0); us.append(u0); gs.append(xg)
    outs = {}
    for mode in ("ts", "coop"):
        s = make_solver("hip", plant, dtype=0 if dtype == np.float32 else 1, batch=B, tol_cost=0.0, kernels=dict(cf=mode), **kw)
        names = dict(s.time_kernels(1))
        assert ("k_fp_ts" in names) == (mode == "ts") and ("k_bp_ts" in names) == (mode == "ts") and ("k_nis_ts" in names) == (mode == "ts"), names   # kernels.cf forces every phase
        outs[mode] = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
        outs[mode]["P"] = s.get_cost_to_go()[0]
        s.close()
    a, c = outs["ts"], outs["coop"]
    assert (a["iters"] == c["iters"]).all() and a["iters"].min() >= 1
    for k in ("alphaOut", "Jout", "x", "u", "KT", "P"):
        assert np.array_equal(a[k], c[k], equal_nan=True), k


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("B", [5, 64])
def test_quadrotor_full_device_kernels_equal_cooperative_bit_for_bit(dtype, B):
    """BASELINE configs[4] with the device full runs k_fp_cf (a wavefront = 4 problems x 16 step sizes, each knot's operands fetched once into an LDS stage a step ahead)
    and k_nis_kb (lane = knot for the three stage gradients, then lane = column of [A B]), from 2048 problems also k_bp_cl (16 lanes per block of knots, lane = column; forced
    here): per output element the operations of the cooperative kernels in their order.
    5 problems: the last wavefront is partly empty; 64: the automatic selection."""
    plant, kw = 3, dict(N=256, M=4, A=16, integrator=3, total_time=4.0, max_iter=5)
    rng = np.random.default_rng(19)
    xs, us, gs = [], [], []
    for b in range(B):
        x0, u0, xg = example_inputs(plant, kw["N"], dtype, noise=rng.normal(0, 0.001 * (b % 7 + 1), (kw["N"], 12)))
        xs.append(x0); us.append(u0); gs.append(xg)
    outs = {}
    for mode, env in (("new", dict(cf_fp="cf", cf_nis="kb16", cf_bp="cl") if B < 64 else dict(cf_bp="cl")), ("coop", dict(cf="coop"))):
        s = make_solver("hip", plant, dtype=0 if dtype == np.float32 else 1, batch=B, tol_cost=0.0, kernels=env, **kw)
        names = dict(s.time_kernels(1))
        assert ("k_fp_cf" in names) == (mode == "new") and ("k_nis_kb" in names) == (mode == "new") and ("k_bp_cl" in names) == (mode == "new"), names
        outs[mode] = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
        outs[mode]["P"] = s.get_cost_to_go()[0]
        s.close()
    a, c = outs["new"], outs["coop"]
    assert (a["iters"] == c["iters"]).all() and a["iters"].min() >= 1
    for k in ("alphaOut", "Jout", "x", "u", "KT", "P"):
        assert np.array_equal(a[k], c[k], equal_nan=True), k


@pytest.mark.parametrize("plant,kw", CASES)
def test_thread_serial_float64_solves_follow_the_oracle(plant, kw):
    n = {1: 2, 2: 4, 3: 12}[plant]
    x0, u0, xg = example_inputs(plant, kw["N"], np.float64, noise=np.random.default_rng(23).normal(0, 0.001, (kw["N"], n)))
    r = Oracle(default_cfg(plant, cores=8, spawn_threads=0, tol_cost=0.0, **kw), np.float64).run_ilqr_gpusem(x0, u0, xg)
    s = make_solver("hip", plant, dtype=1, tol_cost=0.0, kernels=dict(cf="ts"), **kw)
    out = s.solve(x0, u0, xg)
    it = r["iters"]
    assert out["iters"][0] == it and list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-7)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-7 * max(np.abs(r["x"]).max(), 1.0))
    s.close()


def test_large_batch_selection_is_thread_serial_and_equals_single_problem_solves():
    """The automatic selection: 512 cart-pole problems run the thread-serial kernels, a single problem the cooperative ones -- same bits."""
    kw = dict(N=128, M=4, A=8, integrator=3, total_time=4.0, max_iter=6, tol_cost=0.0)
    B = 512
    rng = np.random.default_rng(31)
    xs, us = [], []
    for b in range(B):
        x0, u0, xg = example_inputs(2, 128, np.float32, noise=rng.normal(0, 0.001, (128, 4)))
        xs.append(x0); us.append(u0)
    s = make_solver("hip", 2, dtype=0, batch=B, **kw)
    assert "k_fp_cf" in dict(s.time_kernels(1)) and "k_nis_ts" in dict(s.time_kernels(1))      # thread per rollout (operands staged per wavefront: 8 step sizes), thread per knot
    out = s.solve(np.concatenate(xs), np.concatenate(us), np.tile(xg, B))
    s1 = make_solver("hip", 2, dtype=0, batch=1, **kw)
    assert "k_fp" in dict(s1.time_kernels(1))
    for b in rng.choice(B, 6, replace=False):
        o1 = s1.solve(xs[b], us[b], xg)
        for k in ("alphaOut", "Jout", "x", "u"):
            assert np.array_equal(o1[k][0], out[k][b]), (int(b), k)


@pytest.mark.gpu
def test_candidate_arrays_are_views_of_the_records_on_staged_handles():
    """Closed-form handles whose production rollouts keep knot-major records (k_fp_cf) adopt from those records: pddp_get_array("xs" / "us") after production sweeps must
    show the LAST rollouts (not the arrays' stale contents), and pddp_set_array of either must reach the records."""
    plant, kw = 3, dict(N=64, M=4, A=16, integrator=3, total_time=2.0, max_iter=6, tol_cost=0.0)
    B = 4
    rng = np.random.default_rng(23)
    probs = [example_inputs(plant, kw["N"], np.float32, noise=rng.normal(0, 0.002, (kw["N"], 12))) for _ in range(B)]
    x0, u0, xg = (np.concatenate([p[i] for p in probs]) for i in range(3))
    outs = {}
    for mode, sel in (("records", dict(cf_fp="cf", cf="ts")), ("plain", dict(cf="ts", cf_fp="ts"))):
        s = make_solver("hip", plant, dtype=0, batch=B, use_graph=0, kernels=sel, **kw)
        names = [n for n, _ in s.time_kernels(1)]
        assert ("k_fp_cf" in names) == (mode == "records"), names
        s.load(x0, u0, xg); s.iterate(3); s.sync()
        outs[mode] = (s.get("xs").copy(), s.get("us").copy())
        if mode == "records":
            xs = outs[mode][0].copy(); xs[:12] += 1.0
            s.set("xs", xs)
            assert np.array_equal(s.get("xs"), xs, equal_nan=True)      # (candidates that run away hold NaNs)
            xw = s.get("xw").reshape(B, kw["N"], 16, 16)
            assert np.array_equal(xw[0, 0, 0, :12], xs[:12])                 # knot 0, candidate 0 of problem 0 reached its record
        s.close()
    N, n, m, A = kw["N"], 12, 4, 16
    xr, xp = (outs[k][0].reshape(B, A, N, n) for k in ("records", "plain"))
    ur, up = (outs[k][1].reshape(B, A, N, m) for k in ("records", "plain"))
    assert np.array_equal(xr, xp, equal_nan=True) and np.array_equal(ur[:, :, : N - 1], up[:, :, : N - 1], equal_nan=True)


@pytest.mark.gpu
def test_candidate_views_follow_the_records_across_graph_replays():
    """with use_graph = 1 only the FIRST pddp_iterate passes through the launch functions (capture); later ones replay the hipGraph.  The flag that says
    "the records are newer than xs / us" must be raised by pddp_iterate itself: iterate, get, iterate, get -- the second get has to show the second rollouts, on the graph
    handle exactly what a handle without a graph shows.  And after pddp_refresh_reference_views put the winner into every slot, a get must return THAT."""
    plant, kw = 3, dict(N=64, M=4, A=16, integrator=3, total_time=2.0, max_iter=8, tol_cost=0.0)
    B = 4
    rng = np.random.default_rng(29)
    probs = [example_inputs(plant, kw["N"], np.float32, noise=rng.normal(0, 0.002, (kw["N"], 12))) for _ in range(B)]
    x0, u0, xg = (np.concatenate([p[i] for p in probs]) for i in range(3))
    seen = {}
    for graph in (1, 0):
        s = make_solver("hip", plant, dtype=0, batch=B, use_graph=graph, kernels=dict(cf_fp="cf", cf="ts"), **kw)
        assert "k_fp_cf" in [n for n, _ in s.time_kernels(1)]
        s.load(x0, u0, xg)
        got = []
        for _ in range(3):
            s.iterate(1); s.sync()
            got.append((s.get("xs").copy(), s.get("us").copy()))
        seen[graph] = got
        if graph:
            s.refresh_reference_views()
            xs = s.get("xs").reshape(B, 16, kw["N"], 12)
            assert all(np.array_equal(xs[:, a], xs[:, 0], equal_nan=True) for a in range(16)), "every slot holds the accepted trajectory after the refresh"
        s.close()
    for it in range(3):
        for j, name in enumerate(("xs", "us")):
            assert np.array_equal(seen[1][it][j], seen[0][it][j], equal_nan=True), (it, name)
    assert not np.array_equal(seen[1][0][0], seen[1][1][0], equal_nan=True) and not np.array_equal(seen[1][1][0], seen[1][2][0], equal_nan=True), "the sweeps must move the candidates"
