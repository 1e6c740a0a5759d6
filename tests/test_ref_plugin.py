"""A plant file + cost file in the REFERENCE'S OWN plug-in form compiled in as plant 5 (SURVEY.md section 8b "plug-in functions a plant must provide";
parallel-ddp_amd/csrc/ref_plugin.hpp, `make user PLANT_FILE=... COST_FILE=...`).

The pair under test is builder-written in exactly that form -- parallel-ddp_amd/examples/plants/{dynamics,cost}_twolink.cuh: `dynamics`, `dynamicsGradient`, `initI`, `initT`
with the reference's argument lists (plants/dynamics_arm.cuh:2097,2167), `costFunc` / `costGrad` with the five run-time weights (plants/cost_arm.cuh:130,158) and a
NON-DIAGONAL Hessian (joint coupling, torque coupling, velocity-torque cross blocks).  The oracle restates the reference's solver; the plug-in is an input to it, so its
plant 5 calls the same two files compiled for the host by tests/plugin/plugin_shim.cpp (the reference's one-thread loop helpers, written independently of the adapter).
  * the plug-in itself: analytic gradient against central differences, costGrad against differences of costFunc;
  * the adapter + every kernel body on the host (test tool), the CPU entry points, and on the GPU both kernel selections: float64 whole solves equal the oracle
    decision for decision;
  * what the adapter refuses (N other than the compile-time NUM_TIME_STEPS);
  * in the build container only: the reference's own plants/dynamics_cart.cuh + cost_cart.cuh fed through the same build reproduce the closed-form pins and the
    built-in cart-pole bit for bit (no reference file travels: the test skips where /root/reference is absent)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import pyddp
from oracle_binding import Oracle, default_cfg, register_plugin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parallel-ddp_amd")
PLANT = os.path.join(PKG, "examples", "plants", "dynamics_twolink.cuh")
COST = os.path.join(PKG, "examples", "plants", "cost_twolink.cuh")
N = 64
MAKE = ["make", "-C", PKG, "-s", f"PLANT_FILE={PLANT}", f"COST_FILE={COST}", "NUM_POS=2", "CONTROL_SIZE=2", f"NUM_TIME_STEPS={N}", "USER_TAG=twolink"]
HOSTSIM = os.path.join(ROOT, "tests", "hostsim", "libpddp_hostsim_twolink.so")
CPULIB = os.path.join(PKG, "lib", "libpddp_cpu_twolink.so")
HIPLIB = os.path.join(PKG, "lib", "libpddp_twolink.so")


def build_host():
    subprocess.check_call(MAKE + ["../tests/hostsim/libpddp_hostsim_twolink.so", "lib/libpddp_cpu_twolink.so"])


def build_shim(plant=PLANT, cost=COST, npos=2, m=2, n_steps=N, tag="twolink"):
    out = os.path.join(ROOT, "tests", "plugin", f"libplugin_{tag}.so")
    src = os.path.join(ROOT, "tests", "plugin", "plugin_shim.cpp")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in (src, plant, cost)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", f'-DPLUGIN_PLANT_FILE="{plant}"', f'-DPLUGIN_COST_FILE="{cost}"',
                               f"-DPLUGIN_NUM_POS={npos}", f"-DPLUGIN_CONTROL_SIZE={m}", f"-DPLUGIN_NUM_TIME_STEPS={n_steps}", "-o", out, src])
    return out


KW = dict(N=N, M=4, A=8, integrator=3, total_time=2.0, tol_cost=0.0, max_iter=12, rho_init=10.0, Q1=0.5, Q2=0.01, R=0.001, QF1=500.0, QF2=50.0)


def inputs(dtype=np.float64, batch=1, seed=3):
    rng = np.random.default_rng(seed)
    x0 = np.zeros((batch, N, 4), dtype); x0[:, :, 0] = -np.pi / 2                    # hanging down, at rest + a little velocity noise
    x0[:, :, 2:] = rng.normal(0, 0.001, (batch, N, 2))
    u0 = np.full((batch, N, 2), 0.01, dtype)
    xg = np.tile(np.asarray([np.pi / 2 - 0.3, 0.6, 0.0, 0.0], dtype), (batch, 1))    # up, slightly bent
    return (x0[0], u0[0], xg[0]) if batch == 1 else (x0, u0, xg)


def oracle(dtype=np.float64, **kw):
    register_plugin(build_shim())
    return Oracle(default_cfg(5, **{"cores": 1, "spawn_threads": 0, **KW, **kw}), dtype)


def solve(path, dtype=np.float64, batch=1, **kw):
    cfg = pyddp.default_config(5, _lib_path=path, dtype=0 if dtype == np.float32 else 1, batch=batch, **{**KW, **kw})
    s = pyddp.Solver(cfg, _lib_path=path)
    return s, s.solve(*inputs(dtype, batch))


def test_the_plugin_gradient_is_the_derivative_of_its_dynamics_and_costgrad_of_costfunc():
    o = oracle()
    rng = np.random.default_rng(11)
    h = 1e-6
    for _ in range(8):
        x, u, xg = rng.normal(0, 1.5, 4), rng.normal(0, 4.0, 2), rng.normal(0, 1.0, 4)
        dq, qdd = o.dynamics_gradient(x, u)
        assert np.array_equal(qdd, o.dynamics(x, u))                                 # the qdd of dynamicsGradient IS dynamics() (what the adapter checks at creation)
        z = np.concatenate([x, u])
        for col in range(6):
            e = np.zeros(6); e[col] = h
            fd = (o.dynamics((z + e)[:4], (z + e)[4:]) - o.dynamics((z - e)[:4], (z - e)[4:])) / (2 * h)
            np.testing.assert_allclose(dq[2 * col: 2 * col + 2], fd, rtol=2e-6, atol=1e-7)
        for k in (5, N - 1):
            H, g = o.cost_grad(x, u, xg, k)
            H = H.reshape(6, 6)
            np.testing.assert_array_equal(H, H.T)
            nz = 4 if k == N - 1 else 6
            gfd = np.zeros(6); Hfd = np.zeros((6, 6))
            for i in range(nz):
                e = np.zeros(6); e[i] = 1e-4
                gfd[i] = (o.cost_func((z + e)[:4], (z + e)[4:], xg, k) - o.cost_func((z - e)[:4], (z - e)[4:], xg, k)) / 2e-4
                Hp, gp = o.cost_grad((z + e)[:4], (z + e)[4:], xg, k); Hm, gm = o.cost_grad((z - e)[:4], (z - e)[4:], xg, k)
                Hfd[:, i] = (gp - gm) / 2e-4
            np.testing.assert_allclose(g[:nz], gfd[:nz], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(H[:nz, :nz], Hfd[:nz, :nz], rtol=1e-6, atol=1e-8)
            if k < N - 1: assert H[0, 1] != 0 and H[4, 5] != 0 and H[2, 4] != 0     # joint, torque and cross coupling: not a diagonal cost
    # a solve on this cost makes progress and the oracle accepts most iterations (the parity tests below are about a real solve)
    r = o.run_ilqr_gpusem(*inputs())
    assert sum(a >= 0 for a in r["alphaOut"][1: r["iters"] + 1]) >= 6 and r["Jout"][r["iters"]] < 0.2 * r["Jout"][0]


def check_against_oracle(out, ref, b=0, rtol=1e-8):
    it = ref["iters"]
    assert out["iters"][b] == it and list(out["alphaOut"][b][: it + 1]) == list(ref["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][b][: it + 1], ref["Jout"][: it + 1], rtol=rtol)
    np.testing.assert_allclose(out["x"][b].ravel(), ref["x"].ravel(), rtol=rtol, atol=1e-9)
    np.testing.assert_allclose(out["u"][b].ravel(), ref["u"].ravel(), rtol=rtol, atol=1e-9)
    np.testing.assert_allclose(out["KT"][b].ravel(), ref["KT"].ravel(), rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("integrator,M", [(3, 4), (1, 4), (2, 1)])
def test_host_emulation_of_the_kernels_with_the_plugin_equals_the_oracle(integrator, M):
    build_host()
    ref = oracle(integrator=integrator, M=M).run_ilqr_gpusem(*inputs())
    s, out = solve(HOSTSIM, integrator=integrator, M=M)
    check_against_oracle(out, ref)
    # the H the setup kernel stored is the user's: the coupling entries sit where costGrad put them
    H = s.get("H").reshape(N, 6, 6)
    assert np.all(H[: N - 1, 0, 1] == -0.2) and np.all(H[: N - 1, 4, 5] == -0.0005) and np.allclose(H[: N - 1, 2, 4], 0.001 * 0.5) and np.all(H[N - 1, 0, 1] == -100.0)


def test_cpu_entry_points_with_the_plugin_equal_the_oracles_cpu_path():
    build_host()
    from test_cpu_twin import CpuBuffers
    lib = C.CDLL(CPULIB)
    cfg = pyddp.default_config(1, dtype=1, **KW); cfg.plant = 5
    n, m, M, A, mi = 4, 2, KW["M"], KW["A"], KW["max_iter"]
    nm = n + m
    sizes = dict(x=n * N, xp=n * N, xp2=n * N, u=m * N, up=n * N, P=n * n * N, p=n * N, Pp=n * n * N, pp=n * N, AB=n * nm * N, H=nm * nm * N, g=nm * N, KT=n * m * N,
                 du=m * N, d=n * N, dp=n * N, ApBK=n * n * N, Bdu=n * N, alpha=A, JT=8, dJexp=8)                     # the sizes of allocateMemory_CPU (nisInitHelpers.cuh:886-925)
    arrs = {k: np.zeros(v) for k, v in sizes.items()}
    arrs["alpha"][:] = [cfg.alpha_base ** i for i in range(A)]
    err = np.zeros(8, np.int32)
    buf = CpuBuffers(**{k: v.ctypes.data for k, v in arrs.items()}, err=err.ctypes.data)
    x0, u0, xg = inputs()
    x0, u0 = x0.copy(), u0.copy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    Jout, aout, tt, iters = np.zeros(mi + 2), np.zeros(mi + 2, np.int32), [np.zeros(mi + 2) for _ in range(6)], C.c_int(0)
    assert lib.pddp_cpu_run_ilqr(C.byref(cfg), C.byref(buf), p(x0), p(u0), None, None, None, None, p(xg), p(Jout), p(aout), 0, 1, 1, *[p(t) for t in tt], 1, C.byref(iters)) == 0
    ref = oracle().run_ilqr_cpu(*inputs())
    it = ref["iters"]
    assert iters.value == it and list(aout[: it + 1]) == list(ref["alphaOut"][: it + 1])
    np.testing.assert_allclose(Jout[: it + 1], ref["Jout"][: it + 1], rtol=1e-8)
    np.testing.assert_allclose(x0.ravel(), ref["x"].ravel(), rtol=1e-8, atol=1e-9)


def test_a_handle_with_another_horizon_than_the_cost_files_num_time_steps_is_refused():
    build_host()
    with pytest.raises(pyddp.PddpError, match="NUM_TIME_STEPS = 64"):
        pyddp.Solver(pyddp.default_config(5, _lib_path=HOSTSIM, dtype=1, **{**KW, "N": 32}), _lib_path=HOSTSIM)


# ---- the reference's own closed-form plug-in through the same build (build container only: nothing of /root/reference travels)
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "plants", "dynamics_cart.cuh")), reason="needs the reference tree (build container only)")
def test_the_references_cart_pole_files_through_the_adapter_reproduce_the_pins_and_the_builtin_plant():
    plant, cost = os.path.join(REF, "plants", "dynamics_cart.cuh"), os.path.join(REF, "plants", "cost_cart.cuh")
    lib = os.path.join(ROOT, "tests", "hostsim", "libpddp_hostsim_refcart.so")
    subprocess.check_call(["make", "-C", PKG, "-s", f"PLANT_FILE={plant}", f"COST_FILE={cost}", "NUM_POS=2", "CONTROL_SIZE=1", "NUM_TIME_STEPS=128", "USER_TAG=refcart",
                           "../tests/hostsim/libpddp_hostsim_refcart.so"])
    kw = dict(N=128, M=4, A=8, integrator=3, total_time=4.0, tol_cost=0.0, max_iter=10, rho_init=10.0, max_defect=0.75)
    s5 = pyddp.Solver(pyddp.default_config(5, _lib_path=lib, dtype=1, **kw), _lib_path=lib)
    # plant level against the fixtures generated from the reference's executed statements (tests/golden/closed_form_plants.json)
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "closed_form_plants.json")))["cart"]["cases"]
    x, u = np.asarray([c["x"] for c in cases]), np.asarray([c["u"] for c in cases])
    qdd, dqdd = np.asarray([c["qdd"] for c in cases]), np.asarray([c["dqdd"] for c in cases])
    np.testing.assert_allclose(s5.plant_eval(0, x, u).reshape(qdd.shape), qdd, rtol=1e-12, atol=1e-12 * np.abs(qdd).max())
    np.testing.assert_allclose(s5.plant_eval(1, x, u).reshape(dqdd.shape), dqdd, rtol=1e-12, atol=1e-12 * np.abs(dqdd).max())
    # solver level: the same problem on plant 5 (the reference's files) and on the built-in cart-pole (plant 2) -- identical decisions, J, x, K
    rng = np.random.default_rng(5)
    x0 = np.zeros((128, 4)); x0[:, 2:] = rng.normal(0, 0.001, (128, 2)); u0 = np.full((128, 1), 0.01); xg = np.asarray([0.0, 3.1416, 0.0, 0.0])
    a = s5.solve(x0, u0, xg)
    s2 = pyddp.Solver(pyddp.default_config(2, _lib_path=lib, dtype=1, **kw), _lib_path=lib)
    b = s2.solve(x0, u0, xg)
    it = b["iters"][0]
    assert a["iters"][0] == it and np.array_equal(a["alphaOut"][0][: it + 1], b["alphaOut"][0][: it + 1]) and sum(v >= 0 for v in b["alphaOut"][0][1: it + 1]) >= 5
    np.testing.assert_allclose(a["Jout"][0][: it + 1], b["Jout"][0][: it + 1], rtol=1e-12)
    np.testing.assert_allclose(a["x"][0], b["x"][0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(a["KT"][0], b["KT"][0], rtol=1e-8, atol=1e-10)


# ---- the GPU: both kernel selections the library makes for a closed-form plant
@pytest.mark.gpu
def test_gpu_kernels_with_the_plugin_equal_the_oracle_float64():
    subprocess.check_call(MAKE + ["user"])
    ref = oracle().run_ilqr_gpusem(*inputs())
    s, out = solve(HIPLIB)                                             # one problem: a wave per unit, the plug-in on lane 0
    check_against_oracle(out, ref)
    B = 96                                                             # 96 x 4 (problem, segment) units: the thread-serial kernels (64 plug-in evaluations per wave)
    x0, u0, xg = inputs(batch=B, seed=9)
    cfg = pyddp.default_config(5, _lib_path=HIPLIB, dtype=1, batch=B, **KW)
    sb = pyddp.Solver(cfg, _lib_path=HIPLIB)
    outb = sb.solve(x0, u0, xg)
    o = oracle()
    for b in (0, 17, 95):
        check_against_oracle(outb, o.run_ilqr_gpusem(x0[b], u0[b], xg[b]), b)
    H = sb.get("H").reshape(B, N, 6, 6)
    assert np.all(H[:, : N - 1, 0, 1] == -0.2) and np.all(H[:, N - 1, 0, 1] == -100.0)


@pytest.mark.gpu
def test_gpu_kernels_with_the_plugin_float32_follow_the_oracle_over_the_leading_iterations():
    subprocess.check_call(MAKE + ["user"])
    x0, u0, xg = inputs(np.float32)
    r32, r64 = oracle(np.float32).run_ilqr_gpusem(x0, u0, xg), oracle().run_ilqr_gpusem(*inputs())
    _, out = solve(HIPLIB, np.float32)
    lead = next((i for i in range(r64["iters"] + 1) if not (out["alphaOut"][0][i] == r32["alphaOut"][i] == r64["alphaOut"][i])), r64["iters"] + 1)
    assert lead >= 5, (out["alphaOut"][0], r32["alphaOut"], r64["alphaOut"])
    for i in range(lead):
        ek, eo = abs(float(out["Jout"][0][i]) - r64["Jout"][i]) / r64["Jout"][i], abs(float(r32["Jout"][i]) - r64["Jout"][i]) / r64["Jout"][i]
        assert ek <= max(1e-4, 5 * eo), (i, ek, eo)
