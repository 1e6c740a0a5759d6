"""Solver-level parity (SURVEY.md G3/G4 shape): the whole runiLQR_GPU loop of the kernels against the oracle's
GPU-semantics driver on the stored example inputs (examples/WAFR_iLQR_examples.cu:69-121).

float64: alpha indices identical and J / x / u / KT to 1e-8 over the whole solve -- this proves that bookkeeping, block
boundaries, defects, line search, accept/reject and the rho schedule are the reference's.
float32: iLQR amplifies one-ulp differences (a changed alpha choice changes everything after it, SURVEY.md section 7
"hard parts" 2), so the test checks alphaOut equality FIRST over the leading iterations and then J to 2e-3 there.
"""
import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import Oracle, default_cfg, example_inputs

RNG = np.random.default_rng(7)


def run_pair(backend, plant, dtype, noise_std=0.0, batch=1, **kw):
    s = make_solver(backend, plant, dtype=0 if dtype == np.float32 else 1, batch=batch, **kw)
    o = Oracle(default_cfg(plant, cores=8, spawn_threads=0, **kw), dtype)
    N = kw["N"]
    xs, us, gs, refs = [], [], [], []
    for b in range(batch):
        noise = RNG.normal(0, noise_std, (N, o.n)) if noise_std else None
        x0, u0, xg = example_inputs(plant, N, dtype, noise=noise)
        xs.append(x0); us.append(u0); gs.append(xg)
        refs.append(o.run_ilqr_gpusem(x0, u0, xg))
    out = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    return out, refs, s


KUKA = dict(N=64, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=12)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("M", [4, 1])
def test_kuka_float64_whole_solve(backend, M):
    out, refs, _ = run_pair(backend, 4, np.float64, **{**KUKA, "M": M})
    r = refs[0]
    it = r["iters"]
    assert out["iters"][0] == it and out["done"][0] == 2
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-8)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-8 * np.abs(r["x"]).max())
    np.testing.assert_allclose(out["u"][0].ravel(), r["u"], rtol=0, atol=1e-8 * np.abs(r["u"]).max())
    np.testing.assert_allclose(out["KT"][0].ravel(), r["KT"], rtol=0, atol=1e-7 * np.abs(r["KT"]).max())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ee_type", [0, 2])
def test_kuka_ee_type_variants_of_the_default_urdf(backend, ee_type):
    """EE_TYPE 0 (no end effector) and 2 (flange + peg), plants/dynamics_arm.cuh:50-65: with the default URDF link 7's inertia is the base values x INERTIA_MODIFIER
    (1 / 5) and its mass 1.2 + WEIGHT_MODIFIER (0 / 0.5) (initI, :338-347) -- pddp_config.ee_type.  The handle's robot tables then equal neither built-in model, so the
    kernels that take the robot as literals step aside; whole float64 solves follow the oracle (which carries the same switch) decision for decision, and differ from
    EE_TYPE 1's."""
    kw = dict(N=32, M=4, A=8, wafr_urdf=0, tol_cost=0.0, total_time=0.5, max_iter=8)
    x0, u0, xg = example_inputs(4, 32, np.float64, noise=np.random.default_rng(5).normal(0, 0.002, (32, 14)), wafr_urdf=0)
    ref = Oracle(default_cfg(4, cores=1, spawn_threads=0, ee_type=ee_type, **kw), np.float64).run_ilqr_gpusem(x0, u0, xg)
    ref1 = Oracle(default_cfg(4, cores=1, spawn_threads=0, ee_type=1, **kw), np.float64).run_ilqr_gpusem(x0, u0, xg)
    s = make_solver(backend, 4, dtype=1, ee_type=ee_type, **kw)
    out = s.solve(x0, u0, xg)
    it = ref["iters"]
    assert list(out["alphaOut"][0][: it + 1]) == list(ref["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], ref["Jout"][: it + 1], rtol=1e-8)
    np.testing.assert_allclose(out["x"][0].ravel(), ref["x"], rtol=0, atol=1e-8 * np.abs(ref["x"]).max())
    assert abs(ref["Jout"][it] - ref1["Jout"][it]) > 1e-6 * abs(ref1["Jout"][it])      # (the switch does change the robot)
    s.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_kuka_float32_headline_config(backend):
    """BASELINE config 3: Kuka, N=128, A=8, M=4, Euler, float."""
    out, refs, _ = run_pair(backend, 4, np.float32, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=20)
    r = refs[0]
    lead = 8
    assert list(out["alphaOut"][0][:lead]) == list(r["alphaOut"][:lead])
    np.testing.assert_allclose(out["Jout"][0][:lead], r["Jout"][:lead], rtol=2e-3)
    np.testing.assert_allclose(out["Jout"][0][:3], r["Jout"][:3], rtol=1e-5)      # before any amplification
    # convergence quality is the same even after the traces part ways
    assert out["Jout"][0][20] < 0.5 * out["Jout"][0][0] and abs(out["Jout"][0][20] - r["Jout"][20]) < 0.25 * r["Jout"][20]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("plant,kw", [
    (2, dict(N=128, M=4, A=8, integrator=3, total_time=4.0, max_iter=10)),          # BASELINE config 2 (cart-pole)
    (2, dict(N=64, M=1, A=8, integrator=1, total_time=2.0, max_iter=10)),
    (1, dict(N=64, M=1, A=1, integrator=1, total_time=4.0, max_iter=10)),           # BASELINE config 1 (pendulum, 1 alpha)
    (1, dict(N=64, M=4, A=1, integrator=1, total_time=4.0, max_iter=10)),
    (3, dict(N=64, M=4, A=16, integrator=3, total_time=2.0, max_iter=6)),           # BASELINE config 5 shape (quadrotor, RK3)
])
def test_other_plants_float64(backend, plant, kw):
    out, refs, _ = run_pair(backend, plant, np.float64, noise_std=0.001, tol_cost=0.0, **kw)
    r = refs[0]
    it = r["iters"]
    assert out["iters"][0] == it
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-7)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-7 * max(np.abs(r["x"]).max(), 1))


@pytest.mark.parametrize("backend", BACKENDS)
def test_quadrotor_fp32_vs_fp64_sweep(backend):
    """BASELINE config 5: same stored inputs in float and double; report where they part (asserted loosely)."""
    kw = dict(N=64, M=4, A=16, integrator=3, total_time=2.0, max_iter=8, tol_cost=0.0)
    RNG2 = np.random.default_rng(11)
    noise = RNG2.normal(0, 0.001, (64, 12))
    res = {}
    for dt in (np.float32, np.float64):
        s = make_solver(backend, 3, dtype=0 if dt == np.float32 else 1, **kw)
        x0, u0, xg = example_inputs(3, 64, dt, noise=noise)
        res[dt] = s.solve(x0, u0, xg)
    a32, a64 = res[np.float32]["alphaOut"][0], res[np.float64]["alphaOut"][0]
    first_diff = next((i for i in range(9) if a32[i] != a64[i]), 9)
    assert first_diff >= 2
    np.testing.assert_allclose(res[np.float32]["Jout"][0][:first_diff], res[np.float64]["Jout"][0][:first_diff], rtol=5e-3)


@pytest.mark.parametrize("backend", BACKENDS)
def test_batch_equals_independent_solves_and_exits(backend):
    """Batch axis: B problems in one handle give bit-identical results to B single solves; tolerance exit per problem."""
    kw = dict(N=32, M=4, A=4, wafr_urdf=1, tol_cost=1e-3, total_time=0.5, max_iter=25)
    s3 = make_solver(backend, 4, batch=3, **kw)
    xs, us, gs = [], [], []
    for b in range(3):
        x0, u0, xg = example_inputs(4, 32, np.float32, noise=RNG.normal(0, 0.01 * (b + 1), (32, 14)))
        xs.append(x0); us.append(u0); gs.append(xg)
    out3 = s3.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    assert out3["done"].all()
    for b in range(3):
        s1 = make_solver(backend, 4, batch=1, **kw)
        o1 = s1.solve(xs[b], us[b], gs[b])
        assert o1["iters"][0] == out3["iters"][b] and o1["done"][0] == out3["done"][b]
        assert np.array_equal(o1["Jout"][0], out3["Jout"][b]) and np.array_equal(o1["x"][0], out3["x"][b])
    assert len(set(out3["iters"])) >= 1


@pytest.mark.parametrize("backend", BACKENDS)
def test_full_size_properties(backend):
    """Size-independent properties at BASELINE's full size (Kuka N=128, A=8, M=4): accepted iterations decrease J,
    rejected ones keep it, exits are in-band, defects live only on segment boundaries, x[0] is never moved."""
    kw = dict(N=128, M=4, A=8, wafr_urdf=1, tol_cost=1e-4, total_time=0.5, max_iter=100)
    s = make_solver(backend, 4, **kw)
    x0, u0, xg = example_inputs(4, 128, np.float32, noise=RNG.normal(0, 0.001, (128, 14)))
    out = s.solve(x0, u0, xg)
    it, J, a = out["iters"][0], out["Jout"][0], out["alphaOut"][0]
    assert out["done"][0] in (1, 2) and 1 <= it <= 100
    for i in range(1, it + 1):
        if a[i] >= 0:
            assert J[i] <= J[i - 1] * (1 + 1e-6) and 0 <= a[i] < 8
        else:
            assert a[i] == -1 and J[i] == J[i - 1]
    assert J[it] < 0.2 * J[0]
    assert np.array_equal(out["x"][0][0], x0.reshape(128, 14)[0])
    ds = s.get("ds").reshape(8, 128, 14)
    nonb = [k for k in range(128) if not (((k + 1) % 32 == 0) and k < 127)]
    assert not ds[:, nonb].any()
    assert (out["dmax"] >= 0).all()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("plant,kw", [
    (4, dict(N=64, M=4, A=8, wafr_urdf=1, total_time=0.5, max_iter=8)),
    (2, dict(N=64, M=2, A=8, integrator=3, total_time=2.0, max_iter=8)),
])
def test_forward_rollout_flag(backend, plant, kw):
    """forwardRolloutFlag = 1 (nisInitHelpers.cuh:642-648): the loaded trajectory is first rolled out segment by segment;
    alphaOut[0] = 0, and the whole solve follows the oracle's GPU-semantics driver with rollout = 1."""
    dtype = np.float64
    s = make_solver(backend, plant, dtype=1, tol_cost=0.0, **kw)
    o = Oracle(default_cfg(plant, cores=8, spawn_threads=0, tol_cost=0.0, **kw), dtype)
    x0, u0, xg = example_inputs(plant, kw["N"], dtype, noise=RNG.normal(0, 0.001, (kw["N"], o.n)))
    r = o.run_ilqr_gpusem(x0, u0, xg, rollout=1)
    out = s.solve(x0, u0, xg, forward_rollout=1)
    it = r["iters"]
    assert out["alphaOut"][0][0] == 0 and r["alphaOut"][0] == 0
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-8)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-8 * max(np.abs(r["x"]).max(), 1))


@pytest.mark.parametrize("backend", BACKENDS)
def test_warm_start_arrays(backend):
    """clearVarsFlag = 0 (nisInitHelpers.cuh:621-628): KT0, P0, p0, d0 handed back by the caller seed a warm start that is reproducible
    and differs from a cold start (the boundary cost-to-go enters the first backward pass); NULL arrays keep the device values."""
    kw = dict(N=64, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=6)
    x0, u0, xg = example_inputs(4, 64, np.float32, noise=RNG.normal(0, 0.001, (64, 14)))
    s = make_solver(backend, 4, **kw)
    first = s.solve(x0, u0, xg)
    P_last, p_last, _, _ = s.get_cost_to_go()
    warm = dict(KT0=s.get("KT"), P0=P_last, p0=p_last, d0=s.get("dcur"))
    x1, u1 = first["x"][0].ravel(), first["u"][0].ravel()
    kept = s.solve(x1, u1, xg, clear_vars=0)                       # device values kept: Pp is the cost-to-go of the iteration BEFORE the exit
    given = [make_solver(backend, 4, **kw).solve(x1, u1, xg, clear_vars=0, **warm) for _ in range(2)]   # P0 seeds P and Pp alike (:621-624)
    cold = make_solver(backend, 4, **kw).solve(x1, u1, xg, clear_vars=1)
    assert np.array_equal(given[0]["Jout"][0], given[1]["Jout"][0]) and np.array_equal(given[0]["x"][0], given[1]["x"][0])
    assert not np.array_equal(given[0]["Jout"][0][1:], cold["Jout"][0][1:])
    assert not np.array_equal(kept["Jout"][0][1:], cold["Jout"][0][1:])


@pytest.mark.parametrize("backend", BACKENDS)
def test_baseline_config4a_mpc_rollout_batch(backend):
    """BASELINE configs[3], stage 4a: the WAFR_MPC_examples problem shape -- Kuka, N=64, T=0.5 s, M=4, A=8,
    MPC_MODE gravity 0, TOL_COST 1e-5 -- as a batch of independent rollouts with different goals and start states, joint-space
    cost.  Every rollout must follow the oracle's GPU-semantics driver (float64: identical alpha sequence, J to 1e-8)."""
    kw = dict(N=64, M=4, A=8, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, max_iter=8, ignore_max_rho_exit=0)
    B = 5
    s = make_solver(backend, 4, dtype=1, batch=B, **kw)
    o = Oracle(default_cfg(4, cores=8, spawn_threads=0, **kw), np.float64)
    xs, us, gs, refs = [], [], [], []
    for r in range(B):
        x0, u0, xg = example_inputs(4, 64, np.float64, noise=RNG.normal(0, 0.01, (64, 14)))
        x0 = x0.reshape(64, 14); x0[:, :7] += 0.05 * r; x0 = x0.ravel()          # rollout r starts elsewhere ...
        u0 = np.full(64 * 7, 0.01)                                                 # ... with the MPC example's u = 0.01 (no gravity to hold)
        xg = xg.copy(); xg[:7] += 0.1 * np.sin(2 * np.pi * r / B)                  # ... and tracks another point of the goal curve
        xs.append(x0); us.append(u0); gs.append(xg)
        refs.append(o.run_ilqr_gpusem(x0, u0, xg))
    out = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    for r in range(B):
        it = refs[r]["iters"]
        assert out["iters"][r] == it
        assert list(out["alphaOut"][r][: it + 1]) == list(refs[r]["alphaOut"][: it + 1])
        np.testing.assert_allclose(out["Jout"][r][: it + 1], refs[r]["Jout"][: it + 1], rtol=1e-8)
        np.testing.assert_allclose(out["x"][r].ravel(), refs[r]["x"], rtol=0, atol=1e-8 * max(np.abs(refs[r]["x"]).max(), 1))


@pytest.mark.gpu
def test_baseline_config5_quadrotor_full_size_fp32_vs_fp64():
    """BASELINE configs[4]: quadrotor, N=256, RK3, 16 alphas, M=4, T=4 s.  float64 follows the oracle; float32 and float64 from the
    same stored inputs: report where they part, require the leading iterations to agree."""
    kw = dict(N=256, M=4, A=16, integrator=3, total_time=4.0, max_iter=6, tol_cost=0.0)
    noise = np.random.default_rng(11).normal(0, 0.001, (256, 12))
    res = {}
    for dt in (np.float32, np.float64):
        s = make_solver("hip", 3, dtype=0 if dt == np.float32 else 1, **kw)
        x0, u0, xg = example_inputs(3, 256, dt, noise=noise)
        res[dt] = s.solve(x0, u0, xg)
    o = Oracle(default_cfg(3, cores=8, spawn_threads=0, **kw), np.float64)
    r = o.run_ilqr_gpusem(*example_inputs(3, 256, np.float64, noise=noise))
    it = r["iters"]
    assert list(res[np.float64]["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(res[np.float64]["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-7)
    a32, a64 = res[np.float32]["alphaOut"][0], res[np.float64]["alphaOut"][0]
    first_diff = next((i for i in range(7) if a32[i] != a64[i]), 7)
    print("quadrotor N=256 RK3 A=16: float32 and float64 alpha sequences agree for", first_diff, "iterations;",
          "J rel dev", np.abs(res[np.float32]["Jout"][0][:first_diff] / res[np.float64]["Jout"][0][:first_diff] - 1).max())
    assert first_diff >= 2
    np.testing.assert_allclose(res[np.float32]["Jout"][0][:first_diff], res[np.float64]["Jout"][0][:first_diff], rtol=5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [4, 1])
def test_kuka_float64_headline_size_whole_solve(M):
    """BASELINE configs[2] at full size (N=128, A=8), float64, 40 iterations: the kernels follow the oracle's GPU-semantics driver
    decision for decision (identical step-size indices, rejections included) and J / x / u / K to 1e-7."""
    kw = dict(N=128, M=M, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=40)
    out, refs, _ = run_pair("hip", 4, np.float64, noise_std=0.001, **kw)
    r = refs[0]
    it = r["iters"]
    assert out["iters"][0] == it == 40
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-7)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-7 * np.abs(r["x"]).max())
    np.testing.assert_allclose(out["u"][0].ravel(), r["u"], rtol=0, atol=1e-7 * np.abs(r["u"]).max())
    np.testing.assert_allclose(out["KT"][0].ravel(), r["KT"], rtol=0, atol=1e-6 * np.abs(r["KT"]).max())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("N,M,A", [(16, 1, 1), (16, 8, 1), (32, 2, 3), (64, 16, 5), (32, 1, 13)])
def test_kuka_ragged_shapes_float64(backend, N, M, A):
    """Edge shapes of the arm path: a single candidate, a single segment, segments of 2 knots (N/M = 2, the minimum), candidate counts that do not fill
    the 8 lane groups of a wave, A*M not a multiple of 8 -- the block / segment / defect-boundary bookkeeping must stay the reference's."""
    kw = dict(N=N, M=M, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5 * N / 128, max_iter=6)
    out, refs, _ = run_pair(backend, 4, np.float64, **kw)
    r = refs[0]
    it = r["iters"]
    assert out["iters"][0] == it
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-8)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-8 * np.abs(r["x"]).max())
    np.testing.assert_allclose(out["KT"][0].ravel(), r["KT"], rtol=0, atol=1e-7 * max(np.abs(r["KT"]).max(), 1.0))


@pytest.mark.gpu
def test_kuka_long_horizon_needs_more_than_the_default_dynamic_lds():
    """N = 512, A = 16 in float64: the forward pass's cost table takes 16 * (512 + 4) * 8 = 66 KB of dynamic LDS (above the 64 KB default limit)."""
    kw = dict(N=512, M=4, A=16, wafr_urdf=1, tol_cost=0.0, total_time=2.0, max_iter=3)
    out, refs, _ = run_pair("hip", 4, np.float64, **kw)
    r = refs[0]
    it = r["iters"]
    assert out["iters"][0] == it and list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-7)


@pytest.mark.gpu
def test_kuka_maximum_candidate_grid_float64():
    """The largest candidate grid one workgroup takes (A * M = 128: 16 alphas x 8 segments) on a long horizon (N = 256), float64, against the oracle."""
    kw = dict(N=256, M=8, A=16, wafr_urdf=1, tol_cost=0.0, total_time=1.0, max_iter=5)
    out, refs, _ = run_pair("hip", 4, np.float64, **kw)
    r = refs[0]
    it = r["iters"]
    assert out["iters"][0] == it and list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-7)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-7 * np.abs(r["x"]).max())
