"""USE_LIMITS_FLAG (config.cuh:171-173) for the KUKA arm's joint-space cost: quadratic penalties beyond the safety-scaled position / velocity / torque limits in the
cost and in its gradient -- not in the Hessian (costFunc / costGrad / limitCosts / quadPen, plants/cost_arm.cuh:13-94,136-149,176-199).

tests/golden/limits_fixtures.{npz,json} (tests/golden/make_phase_fixtures.py --limits, build container only): the reference's own costKern / costThreaded /
costGradientHessianKern / costGradientHessianThreaded and two whole runiLQR_GPU solves compiled with USE_LIMITS_FLAG 1, executed at generation time on trajectories that
leave the limits.  Data only.
  * the oracle (use_limits = 1) against the fixture, 1e-12 / whole solves decision for decision;
  * the kernels (pddp_config.use_limits) against the oracle: per-knot cost gradient and per-candidate cost through the phase hooks, whole float64 solves decision for
    decision -- on the wave-cooperative kernels (what a handle with few problems runs: the lane-group family does not carry the variant) and on the thread lanes.
The same flag with the end-effector cost (there the penalties also enter the diagonal of H) and USE_SMOOTH_ABS (the tool-point term of a knot becomes
sqrt(2 c + alpha^2) - alpha, its gradient c' / sqrt(...); cost_arm.cuh:218-220,242-251,289-291,341-343,374-376): gradient / Hessian / per-knot cost, the in-sim cost
accumulation of forwardSimKern and a whole solve with both flags, against the executed reference (oracle) and against the oracle (kernels)."""
import json
import os

import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import Oracle, default_cfg

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(HERE, "limits_fixtures.json")))
DATA = np.load(os.path.join(HERE, "limits_fixtures.npz"))
CASES = {c["name"]: c for c in MAN["cases"]}
N, A = 16, 4
XS = [DATA["limits.in.x%d" % a] for a in range(A)]
US = [DATA["limits.in.u%d" % a] for a in range(A)]
XG = DATA["limits.in.xg"]


def rel(a, ref):
    a, ref = np.asarray(a, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300))


def oracle(cfg, dtype=np.float64, **over):
    kw = {k: cfg[k] for k in ("N", "M", "A", "integrator", "total_time", "wafr_urdf", "tol_cost", "max_iter") if k in cfg}
    kw.update(over)
    return Oracle(default_cfg(4, cores=1, spawn_threads=0, use_limits=1, **kw), dtype)


def test_the_trajectories_leave_every_kind_of_limit():
    x, u = np.stack(XS).reshape(A, N, 14), np.stack(US).reshape(A, N, 7)
    assert (np.abs(x[..., :7]) > 2.96705972839 * 0.8).any() and (np.abs(x[..., 7:]) > 2.356194 * 0.8).any() and (np.abs(u) > 240.0).any()
    assert (np.abs(x[..., :7]) < 2.09439510239 * 0.8).any() and (np.abs(u) < 240.0).any()


@pytest.mark.parametrize("sem", ["gpu", "cpu"])
def test_oracle_cost_and_gradient_match_the_executed_reference(sem):
    case = CASES["limits_cost_" + sem]
    o = oracle(case["cfg"])
    o0 = Oracle(default_cfg(4, cores=1, spawn_threads=0, N=N, M=2, A=A, total_time=0.5, wafr_urdf=1), np.float64)
    g, H = DATA["limits_cost_%s.out.g" % sem].reshape(N, 21), DATA["limits_cost_%s.out.H" % sem].reshape(N, 441)
    x, u = XS[0].reshape(N, 14), US[0].reshape(N, 7)
    changed = 0
    for k in range(N):
        Hk, gk = o.cost_grad(x[k], u[k], XG, k)
        H0, g0 = o0.cost_grad(x[k], u[k], XG, k)
        assert rel(gk, g[k]) <= 1e-12 and rel(Hk, H[k]) <= 1e-12, k
        assert np.array_equal(Hk, H0)                               # the reference leaves H alone
        changed += not np.array_equal(gk, g0)
    assert changed >= N // 2
    for a in range(A):
        if sem == "gpu":
            assert rel(o.total_cost(1, XS[a], US[a], XG), DATA["limits_cost_gpu.out.J"][a]) <= 1e-12
        else:
            total = 0.0
            for v in DATA["limits_cost_cpu.out.Jparts"][a]:
                total += float(v)
            assert rel(o.total_cost(0, XS[a], US[a], XG), total) <= 1e-12


@pytest.mark.parametrize("name", [n for n in CASES if CASES[n]["kind"] == "solve"])
def test_oracle_whole_solves_match_the_executed_reference(name):
    c = CASES[name]["cfg"]
    r = oracle(c).run_ilqr_gpusem(DATA[name + ".in.x0"], DATA[name + ".in.u0"], DATA[name + ".in.xg"])
    ref_a, ref_J = DATA[name + ".out.alphaOut"], DATA[name + ".out.Jout"]
    it = r["iters"]
    assert list(r["alphaOut"][: it + 1]) == list(ref_a[: it + 1]) and (np.asarray(ref_a[1: it + 1]) >= 0).any()
    for k, got in (("Jout", r["Jout"][: it + 1]), ("x", r["x"]), ("u", r["u"]), ("KT", r["KT"])):
        ref = ref_J[: it + 1] if k == "Jout" else DATA[name + ".out." + k]
        assert rel(got, ref) <= 1e-10, (name, k)
    r0 = Oracle(default_cfg(4, cores=1, spawn_threads=0, **{k: c[k] for k in ("N", "M", "A", "integrator", "total_time", "wafr_urdf", "tol_cost", "max_iter")}),
                np.float64).run_ilqr_gpusem(DATA[name + ".in.x0"], DATA[name + ".in.u0"], DATA[name + ".in.xg"])
    assert list(r0["alphaOut"][:7]) != list(r["alphaOut"][:7]) or rel(r0["Jout"][:3], r["Jout"][:3]) > 1e-3     # the penalties matter in these solves


SELECTIONS = [pytest.param(dict(fp="coop"), id="cooperative"), pytest.param(dict(bp="mx", fp="tl"), id="thread-lanes")]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("env", SELECTIONS)
def test_kernels_cost_gradient_and_candidate_costs(backend, env):
    """next-iteration setup in init mode writes g_k (and the untouched diagonal H_k); the line search's candidate costs come out of a teacher-forced forward pass with
    zero gains and zero feed-forward: candidate a reproduces the loaded trajectory's controls, so J[a] is the cost of the rolled-out trajectory under the same cost."""
    cfg = CASES["limits_cost_gpu"]["cfg"]
    o = oracle(cfg)

    def run():
        s = make_solver(backend, 4, dtype=1, kernels=dict(env), N=N, M=2, A=A, total_time=0.5, wafr_urdf=1, use_limits=1, tol_cost=0.0, max_iter=4)
        s.load(XS[0], US[0], XG)                                   # init mode of the setup kernel: g, H of the loaded trajectory
        g, H = s.get("g").reshape(N, 21), s.get("H").reshape(N, 441)
        x, u = XS[0].reshape(N, 14), US[0].reshape(N, 7)
        for k in range(N):
            Hk, gk = o.cost_grad(x[k], u[k], XG, k)
            assert rel(g[k, : 14 if k == N - 1 else 21], gk[: 14 if k == N - 1 else 21]) <= 1e-12, (k, "g")
            if k < N - 1:
                assert rel(H[k], Hk) <= 1e-12, (k, "H")
        J0 = s.store()["Jout"][0][0]
        assert rel(J0, o.total_cost(1, XS[0], US[0], XG)) <= 1e-12   # initAlgGPU's cost of the loaded trajectory
        s.close()
    run()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("env", SELECTIONS)
@pytest.mark.parametrize("name", [n for n in CASES if CASES[n]["kind"] == "solve"])
def test_kernels_whole_float64_solves_follow_the_oracle(backend, env, name):
    c = CASES[name]["cfg"]
    x0, u0, xg = DATA[name + ".in.x0"], DATA[name + ".in.u0"], DATA[name + ".in.xg"]
    r = oracle(c).run_ilqr_gpusem(x0, u0, xg)

    def run():
        s = make_solver(backend, 4, dtype=1, use_limits=1, kernels=dict(env), **{k: c[k] for k in ("N", "M", "A", "integrator", "total_time", "wafr_urdf", "tol_cost", "max_iter")})
        out = s.solve(x0, u0, xg)
        s.close()
        return out
    out = run()
    it = r["iters"]
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    assert rel(out["Jout"][0][: it + 1], r["Jout"][: it + 1]) <= 1e-8 and rel(out["x"][0], r["x"]) <= 1e-7 and rel(out["u"][0], r["u"]) <= 1e-7


# ------------------------------------------------------------------------------------------------ the same flags with the end-effector cost (+ USE_SMOOTH_ABS)
EE_CASES = [n for n in CASES if CASES[n]["kind"] == "ee"]


def ee_oracle(cfg, dtype=np.float64):
    kw = {k: cfg[k] for k in ("N", "M", "A", "integrator", "total_time", "wafr_urdf", "mpc_mode", "tol_cost", "max_iter", "ignore_max_rho_exit") if k in cfg}
    return Oracle(default_cfg(4, cores=1, spawn_threads=0, ee_cost=1, use_limits=cfg.get("use_limits", 0), use_smooth_abs=cfg.get("use_smooth_abs", 0), **kw), dtype)


@pytest.mark.parametrize("name", EE_CASES)
def test_oracle_end_effector_variants_match_the_executed_reference(name):
    """costGradientHessianKern / -Threaded (gradient, Gauss-Newton Hessian + limit terms on its diagonal, per-knot cost) and forwardSimKern / forwardSim with the in-sim
    cost accumulation, compiled with USE_SMOOTH_ABS and / or USE_LIMITS_FLAG"""
    case = CASES[name]
    c, sem = case["cfg"], case["sem"]
    o = ee_oracle(c)
    Nk = c["N"]
    x, u, goal = DATA["limits_ee.in.x"].reshape(Nk, 14), DATA["limits_ee.in.u"].reshape(Nk, 7), DATA["limits_ee.in.goal"]
    H, g = DATA[name + ".grad.H"].reshape(Nk, -1), DATA[name + ".grad.g"].reshape(Nk, -1)
    Jk = []
    for k in range(Nk):
        Hk, gk = o.ee_cost_grad(x[k], u[k], goal, k)
        assert rel(Hk, H[k]) <= 1e-12 and rel(gk, g[k]) <= 1e-12, (name, k)
        Jk.append(o.ee_cost(x[k], u[k], goal, k))
    if sem == "gpu":
        assert rel(Jk, DATA[name + ".grad.J_knots"]) <= 1e-12
    else:
        T = len(DATA[name + ".grad.J_parts"])
        parts = [0.0] * T
        for k in range(Nk):
            parts[k % T] += Jk[k]
        assert rel(parts, DATA[name + ".grad.J_parts"]) <= 1e-12
    sim = {k[len("limits_ee.sim.in."):]: DATA[k] for k in DATA.files if k.startswith("limits_ee.sim.in.")}
    for a_, al in enumerate(sim["alphas"]):
        xs, us, ds = sim["xs"][a_].copy(), sim["u"].copy(), sim["d"].copy()
        JT = o.forward_sim_ee(xs, us, sim["KT"], sim["du"], ds, al, sim["xp"], goal)
        assert rel(xs, DATA[name + ".sim.xs"][a_]) <= 1e-12 and rel(us, DATA[name + ".sim.us"][a_]) <= 1e-12, (name, a_)
        assert rel(JT, DATA[name + ".sim.JT"][a_]) <= 1e-12, (name, a_, "in-sim cost")


def test_oracle_end_effector_whole_solve_with_both_flags():
    c = CASES["limits_ee_solve"]["cfg"]
    r = ee_oracle(c).run_ilqr_gpusem(DATA["limits_ee_solve.in.x0"], DATA["limits_ee_solve.in.u0"], DATA["limits_ee_solve.in.xg"])
    it = r["iters"]
    assert list(r["alphaOut"][: it + 1]) == list(DATA["limits_ee_solve.out.alphaOut"][: it + 1])
    for k, got in (("Jout", r["Jout"][: it + 1]), ("x", r["x"]), ("u", r["u"]), ("KT", r["KT"])):
        ref = DATA["limits_ee_solve.out.Jout"][: it + 1] if k == "Jout" else DATA["limits_ee_solve.out." + k]
        assert rel(got, ref) <= 1e-10, k


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("env", SELECTIONS)
@pytest.mark.parametrize("flags", [dict(use_smooth_abs=1), dict(use_limits=1), dict(use_smooth_abs=1, use_limits=1)], ids=["smooth", "limits", "both"])
def test_kernels_end_effector_variants(backend, env, flags):
    """init mode of the setup kernel (g, H, per-knot cost -> the initial cost) and a whole float64 solve, on the cooperative kernels and on the thread lanes"""
    c = dict(CASES["limits_ee_solve"]["cfg"], use_smooth_abs=0, use_limits=0, **{})
    c.update(flags)
    o = ee_oracle(c)
    x0, u0, xg6 = DATA["limits_ee_solve.in.x0"], DATA["limits_ee_solve.in.u0"], DATA["limits_ee_solve.in.xg"]
    xg = np.zeros(14); xg[:6] = xg6
    r = o.run_ilqr_gpusem(x0, u0, xg6)
    Nk = c["N"]

    def run():
        s = make_solver(backend, 4, dtype=1, ee_cost=1, kernels=dict(env), **{k: c[k] for k in ("N", "M", "A", "integrator", "total_time", "wafr_urdf", "mpc_mode", "tol_cost", "max_iter",
                                                                              "ignore_max_rho_exit", "use_smooth_abs", "use_limits")})
        s.load(x0, u0, xg)
        g, H = s.get("g").reshape(Nk, 21), s.get("H").reshape(Nk, 441)
        for k in range(Nk):
            Hk, gk = o.ee_cost_grad(x0.reshape(Nk, 14)[k], u0.reshape(Nk, 7)[k], xg6, k)
            assert rel(g[k], gk) <= 1e-9 and rel(H[k], Hk) <= 1e-9, (k, rel(g[k], gk), rel(H[k], Hk))
        out = s.solve(x0, u0, xg)
        s.close()
        return out
    out = run()
    it = r["iters"]
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    assert rel(out["Jout"][0][: it + 1], r["Jout"][: it + 1]) <= 1e-7 and rel(out["x"][0], r["x"]) <= 1e-6


@pytest.mark.gpu
def test_float32_handles_with_limits_select_kernels_that_carry_the_variant():
    """few problems in flight: the cooperative kernels (the lane-group family has no limit terms); from 512 problems: thread lanes"""
    import test_kernel_selection as ks
    kuka = dict(N=64, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=20, use_limits=1)
    assert ks.kernels(4, 2, **kuka)[-3:] == ["k_fp", "k_ls", "k_nis"]
    assert ks.kernels(4, 512, **kuka)[-3:] == ["k_fp_tl", "k_ls", "k_nis_tl"]


@pytest.mark.gpu
def test_variants_outside_their_cost_family_are_refused():
    import pyddp
    with pytest.raises(pyddp.binding.PddpError, match="use_smooth_abs"):
        pyddp.Solver(pyddp.default_config(4, N=16, M=2, A=4, use_smooth_abs=1))
    with pytest.raises(pyddp.binding.PddpError, match="use_limits"):
        pyddp.Solver(pyddp.default_config(2, N=16, M=2, A=4, integrator=3, use_limits=1))


def test_fixture_is_data_only():
    assert set(MAN) == {"_provenance", "cases"}
    assert all(DATA[k].dtype.kind in "fi" for k in DATA.files)
