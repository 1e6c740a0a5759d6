"""FLOAT32 pins of the oracle (VERDICT r3 "missing" 3a): tests/golden/phase_fixtures_f32.{npz,json} hold the reference's OWN statements executed in float32 at
fixture-generation time -- refc2py's float32 mode: T is a float32 value type with C's usual arithmetic conversions (float op float in float32 with one rounding per
operation, float op double in double, conversion on assignment / argument passing / return / cast), no contraction, libm's float functions for sin / cos / atan2 -- that is,
the reference compiled with algType = float (config.cuh:74) and strict IEEE arithmetic; device branches under the SIMT emulation with the reference's launch geometry.

At 1e-12 in float64 (tests/test_phase_pins.py) the ORDER of a sum is invisible: any order passes.  In float32 it is not -- the pairwise tree of costKern / reduceSum
(cudaUtils.h:187-207), computeExpRed's partial sums (bpHelpers.cuh:326-332), the inner products of backprop / computeCTG / computeKTdu, the RK3 stage chains -- so the bar
here is BIT-FOR-BIT equality of the float32 oracle (oracle/liboracle.so, `_f32` entry points, built -ffp-contract=off) with every stored array, integers included.
This is what makes oracle32 a pinned member of the float32 bar of tests/test_fp32_bar.py (which compares kernels with it) rather than a plausible float32 evaluation."""
import json
import os

import numpy as np
import pytest

from oracle_binding import Oracle, default_cfg

HERE = os.path.dirname(os.path.abspath(__file__))
MAN = json.load(open(os.path.join(HERE, "golden", "phase_fixtures_f32.json")))
DATA = dict(np.load(os.path.join(HERE, "golden", "phase_fixtures_f32.npz")))
MAN6 = json.load(open(os.path.join(HERE, "golden", "phase_fixtures_f32_r06.json")))            # round 6: float32 sweeps from the solver's invariant
DATA.update(np.load(os.path.join(HERE, "golden", "phase_fixtures_f32_r06.npz")))
MAN["cases"] = MAN["cases"] + MAN6["cases"]
CASES = {c["name"]: c for c in MAN["cases"]}
F32 = np.float32


def inp(case, key):
    a = np.array(DATA["%s/in/%s" % (case["name"], key)])
    return a.astype(F32) if a.dtype.kind == "f" else a


def out(case, key):
    return np.array(DATA["%s/out/%s" % (case["name"], key)])


def oracle_for(case, **kw):
    c = case["cfg"]
    w = {k.strip("_"): v for k, v in case.get("weights", {}).items()}
    for k in ("wafr_urdf", "mpc_mode", "ee_cost", "tol_cost", "max_iter"):
        if k in c:
            kw.setdefault(k, c[k])
    return Oracle(default_cfg(c["plant"], N=c["N"], M=c["M"], A=c["A"], integrator=c["integrator"], total_time=c["total_time"], cores=8, spawn_threads=0, **w, **kw), F32)


def same(got, ref, what):
    got, ref = np.asarray(got, F32).ravel(), np.asarray(ref, F32).ravel()
    assert got.shape == ref.shape, what
    if not np.array_equal(got, ref):
        bad = np.flatnonzero(got != ref)
        raise AssertionError((what, "%d of %d entries differ" % (bad.size, got.size), [(int(i), float(got[i]), float(ref[i])) for i in bad[:5]]))


def names(kind):
    return [c["name"] for c in MAN["cases"] if c["kind"] == kind]


def test_fixture_is_float32_data_only():
    assert all(DATA[k].dtype.kind in "fi" for k in DATA)
    assert all(DATA[k].dtype == np.float32 for k in DATA if "/out/" in k and DATA[k].dtype.kind == "f")
    for c in MAN["cases"]:
        assert set(c) <= {"name", "kind", "cfg", "sem", "inputs", "outputs", "rho", "weights", "flags"}


@pytest.mark.parametrize("name", names("backward_pass"))
def test_backward_pass_float32_bit_for_bit(name):
    case = CASES[name]
    o = oracle_for(case)
    n, m, N, M = o.n, o.m, case["cfg"]["N"], case["cfg"]["M"]
    a = {k: inp(case, k) for k in ("AB", "P", "p", "Pp", "pp", "H", "g", "d", "x", "xp")}
    KT, du, ApBK, Bdu = np.zeros(N * n * m, F32), np.zeros(N * m, F32), np.zeros(N * n * n, F32), np.zeros(N * n, F32)
    fail, dJexp, err = o.backward_pass(1 if case["sem"] == "gpu" else 0, a["AB"], a["P"], a["p"], a["Pp"], a["pp"], a["H"], a["g"], KT, du, a["d"], ApBK, Bdu, a["x"], a["xp"], F32(case["rho"]))
    assert list(err) == list(out(case, "err")) and not fail
    for k, got in (("KT", KT), ("du", du), ("dJexp", dJexp)):
        same(got, out(case, k), (name, k))
    same(a["P"][: (N - 1) * n * n], out(case, "P")[: (N - 1) * n * n], (name, "P")); same(a["p"][: (N - 1) * n], out(case, "p")[: (N - 1) * n], (name, "p"))
    if M > 1:
        same(ApBK[: (N - 1) * n * n], out(case, "ApBK")[: (N - 1) * n * n], (name, "ApBK")); same(Bdu[: (N - 1) * n], out(case, "Bdu")[: (N - 1) * n], (name, "Bdu"))
    if case["sem"] == "cpu":
        same(a["H"], out(case, "H"), (name, "H in place")); same(a["g"], out(case, "g"), (name, "g in place"))      # the host path accumulates H, g in place (bpHelpers.cuh:90-91)


@pytest.mark.parametrize("name", names("forward_sweep"))
def test_forward_sweep_float32_bit_for_bit(name):
    case = CASES[name]
    o = oracle_for(case)
    for a_, al in enumerate(inp(case, "alphas")):
        x = inp(case, "x")
        o.forward_sweep(x, inp(case, "ApBK"), inp(case, "Bdu"), inp(case, "d"), inp(case, "xp"), F32(al))
        same(x, out(case, "xs")[a_], (name, a_))


@pytest.mark.parametrize("name", names("forward_sim"))
def test_forward_sim_float32_bit_for_bit(name):
    case = CASES[name]
    o = oracle_for(case)
    xs = inp(case, "xs")
    for a_, al in enumerate(inp(case, "alphas")):
        x, u, d = xs[a_].copy(), inp(case, "u"), inp(case, "d")
        o.forward_sim(x, u, inp(case, "KT"), inp(case, "du"), d, F32(al), inp(case, "xp"))
        same(x, out(case, "xs")[a_], (name, "x", a_)); same(u, out(case, "us")[a_], (name, "u", a_)); same(d, out(case, "ds")[a_], (name, "d", a_))


@pytest.mark.parametrize("name", names("integrator_gradient"))
def test_integrator_gradients_float32_bit_for_bit(name):
    case = CASES[name]
    o = oracle_for(case)
    n, m, N = o.n, o.m, case["cfg"]["N"]
    x, u = inp(case, "x").reshape(N, n), inp(case, "u").reshape(N, m)
    ref = out(case, "AB").reshape(N, -1)
    for k in range(N - 1):
        same(o.integrator_gradient(x[k], u[k]), ref[k], (name, k))


@pytest.mark.parametrize("name", names("total_cost"))
def test_total_cost_float32_bit_for_bit(name):
    """costKern: the pairwise tree over the knots (sem gpu); costThreaded: COST_THREADS strided partial sums (sem cpu)"""
    case = CASES[name]
    o = oracle_for(case)
    xs, us, xg = inp(case, "xs"), inp(case, "us"), inp(case, "xg")
    for a_ in range(len(xs)):
        if case["sem"] == "gpu":
            same([o.total_cost(1, xs[a_], us[a_], xg)], [out(case, "J")[a_]], (name, a_))
        else:
            total = F32(0)
            for v in out(case, "Jparts")[a_]:
                total = F32(total + F32(v))
            same([o.total_cost(0, xs[a_], us[a_], xg)], [total], (name, a_))


@pytest.mark.parametrize("name", names("max_defect"))
def test_max_defect_float32_bit_for_bit(name):
    case = CASES[name]
    o = oracle_for(case)
    ds = inp(case, "ds")
    for a_ in range(len(ds)):
        same([o.max_defect(1, ds[a_])], [out(case, "dmax")[a_]], (name, a_))


@pytest.mark.parametrize("name", names("cost_gradient_hessian"))
def test_cost_gradient_hessian_float32_bit_for_bit(name):
    case = CASES[name]
    o = oracle_for(case)
    n, m, N = o.n, o.m, case["cfg"]["N"]
    nm = n + m
    x, u, xg = inp(case, "x").reshape(N, n), inp(case, "u").reshape(N, m), inp(case, "xg")
    H, g = out(case, "H").reshape(N, nm, nm), out(case, "g").reshape(N, nm)
    for k in range(N):
        Hk, gk = o.cost_grad(x[k], u[k], xg, k)
        Hk = Hk.reshape(nm, nm)
        same(gk, g[k], (name, "g", k))
        if k < N - 1:
            same(Hk, H[k], (name, "H", k))
        else:
            same(Hk[:n, :n], H[k][:n, :n], (name, "H final", k))


@pytest.mark.parametrize("name", names("solve"))
def test_whole_solve_of_runiLQR_GPU_float32_bit_for_bit(name):
    """runiLQR_GPU end to end in float32 (host driver + every kernel under the SIMT emulation): step-size indices, J, x, u, K -- the same bits"""
    case = CASES[name]
    o = oracle_for(case)
    r = o.run_ilqr_gpusem(inp(case, "x0"), inp(case, "u0"), inp(case, "xg"))
    it = r["iters"]
    ref_a = out(case, "alphaOut")
    assert list(r["alphaOut"][: it + 1]) == list(ref_a[: it + 1]), (list(r["alphaOut"][: it + 1]), list(ref_a))
    assert sum(a >= 0 for a in ref_a[1: it + 1]) >= 3
    same(r["Jout"][: it + 1], out(case, "Jout")[: it + 1], (name, "J")); same(r["x"], out(case, "x"), (name, "x")); same(r["u"], out(case, "u"), (name, "u")); same(r["KT"], out(case, "KT"), (name, "KT"))
