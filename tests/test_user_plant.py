"""The build-time plant plug-in (SURVEY.md section 8b "config.cuh plant/cost plug-in surface"; parallel-ddp_amd/Makefile `user`, csrc/plants.hpp):
examples/plants/damped_pendulum.hpp compiled in as plant 5.  No oracle knows this plant, so it is pinned three ways:
  * with zero damping it IS the built-in pendulum: whole solves through plant 5 of the user build equal plant 1 bit for bit;
  * its analytic gradient agrees with central finite differences of its own dynamics (through pddp_plant_eval);
  * the damped plant's solve decreases the cost and reaches the goal.
CPU: the host emulation and the CPU entry points built with the same header; GPU: lib/libpddp_user.so."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import pyddp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parallel-ddp_amd")
POLICY = os.path.join(PKG, "examples", "plants", "damped_pendulum.hpp")


def build_user(extra=""):
    """make user with the example policy; extra: additional -D for the policy's own parameters (a second build directory would be cleaner, but the
    zero-damping variant is only needed by this test, so it is built on the side with the same commands)."""
    subprocess.check_call(["make", "-C", PKG, "-s", "user", f"PLANT_POLICY={POLICY}"])
    return {"hostsim": os.path.join(ROOT, "tests", "hostsim", "libpddp_hostsim_user.so"), "hip": os.path.join(PKG, "lib", "libpddp_user.so"),
            "cpu": os.path.join(PKG, "lib", "libpddp_cpu_user.so")}


def build_hostsim_variant(defs, out):
    # (the host emulation is one translation unit per plant since round 6: tests/hostsim/hostsim_*.cpp; the arm's double half is hostsim_arm.cpp compiled a second time)
    hs = os.path.join(ROOT, "tests", "hostsim")
    flags = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unknown-pragmas", f'-DPDDP_USER_PLANT_HEADER="{POLICY}"', "-I" + os.path.join(PKG, "csrc")] + defs
    objdir = out + ".objs"
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for unit, extra in (("hostsim", []), ("hostsim_pend", []), ("hostsim_cart", []), ("hostsim_quad", []), ("hostsim_arm", []), ("hostsim_arm", ["-DPDDP_HOSTSIM_ARM_HALF=1"]), ("hostsim_user", [])):
        obj = os.path.join(objdir, unit + ("64" if extra else "") + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen(["g++"] + flags + extra + ["-c", "-o", obj, os.path.join(hs, unit + ".cpp")]))
    assert all(p_.wait() == 0 for p_ in procs)
    subprocess.check_call(["g++", "-shared", "-o", out] + objs)
    return out


def inputs(N, dtype):
    rng = np.random.default_rng(3)
    x0 = np.zeros((N, 2), dtype); x0[:, 1] = rng.normal(0, 0.001, N)
    u0 = np.full((N, 1), 0.01, dtype)
    return x0, u0, np.asarray([3.1416, 0.0], dtype)


KW = dict(N=64, M=4, A=8, integrator=3, total_time=4.0, tol_cost=0.0, max_iter=15)


def solve(path, plant, dtype=np.float64, **kw):
    cfg = pyddp.default_config(plant, _lib_path=path, dtype=0 if dtype == np.float32 else 1, **{**KW, **kw})
    s = pyddp.Solver(cfg, _lib_path=path)
    x0, u0, xg = inputs(cfg.N, dtype)
    return s, s.solve(x0, u0, xg)


def test_user_plant_with_zero_damping_is_the_builtin_pendulum_bit_for_bit():
    libs = build_user()
    zero = build_hostsim_variant(["-DUSER_PENDULUM_DAMPING=0.0"], os.path.join(ROOT, "tests", "hostsim", "libpddp_hostsim_user0.so"))
    for dtype in (np.float64, np.float32):
        _, a = solve(zero, 5, dtype)
        _, b = solve(zero, 1, dtype)
        assert a["iters"][0] == b["iters"][0] and np.array_equal(a["alphaOut"][0], b["alphaOut"][0])
        assert np.array_equal(a["Jout"][0], b["Jout"][0]) and np.array_equal(a["x"][0], b["x"][0]) and np.array_equal(a["KT"][0], b["KT"][0])
    assert os.path.exists(libs["hip"]) and os.path.exists(libs["cpu"])


def check_gradient_and_solve(path):
    s, out = solve(path, 5)
    it = out["iters"][0]
    J = out["Jout"][0]
    assert J[it] < 0.05 * J[0] and abs(out["x"][0][-1][0] - 3.1416) < 0.05           # swings up
    acc = [a for a in out["alphaOut"][0][1: it + 1] if a >= 0]
    assert len(acc) >= 5
    rng = np.random.default_rng(8)
    x, u = rng.normal(0, 2.0, (16, 2)), rng.normal(0, 5.0, (16, 1))
    g = s.plant_eval(1, x, u).reshape(16, 3)                                          # dqdd[col]: d/dq, d/dqd, d/du
    h = 1e-6
    for col, (dx, du) in enumerate((([h, 0], [0]), ([0, h], [0]), ([0, 0], [h]))):
        fd = (s.plant_eval(0, x + dx, u + du) - s.plant_eval(0, x - np.asarray(dx), u - np.asarray(du))) / (2 * h)
        np.testing.assert_allclose(g[:, col], fd[:, 0], rtol=1e-6, atol=1e-7)
    assert np.allclose(g[:, 1], -0.2) and np.allclose(g[:, 2], 1.0)                   # -b / (m l^2), 1 / (m l^2) of the example's parameters
    return out


def test_user_plant_host_emulation_gradient_and_solve():
    check_gradient_and_solve(build_user()["hostsim"])


def test_user_plant_through_the_cpu_entry_point():
    """runiLQR_CPU's library built with the same policy header: plant 5 solves (CPU semantics), and the standard library refuses plant 5."""
    libs = build_user()
    from test_cpu_twin import CpuBuffers
    lib = C.CDLL(libs["cpu"])
    std = C.CDLL(os.path.join(PKG, "lib", "libpddp_cpu.so"))
    cfg = pyddp.default_config(1, dtype=1, **KW); cfg.plant = 5
    N, n, m, M, A, mi = 64, 2, 1, 4, 8, 15
    nm = n + m
    sizes = dict(x=n * N, xp=n * N, xp2=n * N, u=m * N, up=n * N, P=n * n * N, p=n * N, Pp=n * n * N, pp=n * N, AB=n * nm * N, H=nm * nm * N, g=nm * N, KT=n * m * N,
                 du=m * N, d=n * N, dp=n * N, ApBK=n * n * N, Bdu=n * N, alpha=A, JT=8, dJexp=8)
    arrs = {k: np.zeros(v) for k, v in sizes.items()}
    arrs["alpha"][:] = [cfg.alpha_base ** i for i in range(A)]
    err = np.zeros(8, np.int32)
    buf = CpuBuffers(**{k: v.ctypes.data for k, v in arrs.items()}, err=err.ctypes.data)
    x0, u0, xg = inputs(N, np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    Jout, aout, tt, iters = np.zeros(mi + 2), np.zeros(mi + 2, np.int32), [np.zeros(mi + 2) for _ in range(6)], C.c_int(0)
    args = (C.byref(cfg), C.byref(buf), p(x0), p(u0), None, None, None, None, p(xg), p(Jout), p(aout), 0, 1, 1, *[p(t) for t in tt], 8, C.byref(iters))
    assert lib.pddp_cpu_run_ilqr(*args) == 0
    assert Jout[iters.value] < 0.2 * Jout[0]
    assert std.pddp_cpu_run_ilqr(*args) != 0                                          # a library without the policy does not know plant 5


@pytest.mark.gpu
def test_user_plant_on_the_gpu_matches_its_host_emulation():
    libs = build_user()
    ref = check_gradient_and_solve(libs["hostsim"])
    out = check_gradient_and_solve(libs["hip"])
    it = ref["iters"][0]
    assert out["iters"][0] == it and list(out["alphaOut"][0][: it + 1]) == list(ref["alphaOut"][0][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], ref["Jout"][0][: it + 1], rtol=1e-9)
    with pytest.raises(pyddp.PddpError, match="plant must be"):
        pyddp.default_config(5)                                                       # the standard library has no plant 5
