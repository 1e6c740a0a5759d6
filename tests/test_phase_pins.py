"""Pins of the SOLVER-level oracle: tests/golden/phase_fixtures.{npz,json} hold the outputs of the reference's OWN statements
-- linearXfrmOrLoad, backprop, invHuu / invHuu_dim4 / computeKTdu_dim1 + invertMatrix, computeKTdu, computeCTG, computeFSVars, computeExpRed,
forwardSweepInner, computeControlKT, forwardSimInner, _integrator / _integratorGradient (Euler, midpoint, RK3), costFunc / costGrad (arm, joint space),
costKern / costThreaded, defectKern / defectComp, reduceSum / reduceMax, matMult / matVMult / dotProd and the host line search of forwardSimGPU --
executed in float64 on stored inputs at generation time (tests/golden/make_phase_fixtures.py + refc2py.py: a mechanical statement-by-statement translation;
the `__CUDA_ARCH__` branches under a SIMT emulation with the reference's launch geometry for sem = "gpu", the host branches through the per-thread drivers
for sem = "cpu").  The oracle's float64 instantiation has to reproduce them:

    |oracle64 - fixture| <= 1e-12 x max|fixture|        per output array;  integers (err flags, step-size index, ignore_defect) identical.

Beyond the phases, the fixture holds WHOLE SOLVES of the reference's runiLQR_GPU (host driver + every kernel it launches, emulated end to end), so the loop's
bookkeeping -- initial cost and its epsilon, rho schedule, accept / reject with restore, Pp <- P, winner broadcast, exits -- is pinned as well; and the arm's own
dynamics / dynamicsGradient and the end-effector cost family (compute_eePos, costFunc / costGrad with EE_COST).

This is what makes the oracle a PINNED checker for the backward pass, the forward sweep, the rollouts, the three integration rules with their quirks, the
cost / defect reductions and the line search -- in both the kernel semantics the HIP path is compared with and the host semantics of runiLQR_CPU.
"""
import json
import os

import numpy as np
import pytest

from oracle_binding import Oracle, default_cfg

HERE = os.path.dirname(os.path.abspath(__file__))
MAN = json.load(open(os.path.join(HERE, "golden", "phase_fixtures.json")))
DATA = dict(np.load(os.path.join(HERE, "golden", "phase_fixtures.npz")))
# round 4 (make_phase_fixtures.py --round4): sweeps / rollouts that start from the solver's invariant, and a whole runiLQR_GPU solve at the HEADLINE size N = 128, M = 4, A = 8
_MAN4 = json.load(open(os.path.join(HERE, "golden", "phase_fixtures_r04.json")))
DATA.update(np.load(os.path.join(HERE, "golden", "phase_fixtures_r04.npz")))
MAN["cases"] = MAN["cases"] + _MAN4["cases"]
CASES = {c["name"]: c for c in MAN["cases"]}
TOL = 1e-12


def inp(case, key):
    return np.array(DATA["%s/in/%s" % (case.get("inputs_of", case["name"]), key)], np.float64)


def out(case, key):
    return np.array(DATA["%s/out/%s" % (case["name"], key)])


def oracle_for(case, **kw):
    c = case["cfg"]
    extra = dict(case.get("weights", {}))
    w = {k.strip("_"): v for k, v in extra.items()}
    # cores = 8 -> COST_THREADS = 4, BP_THREADS = FSIM_THREADS = min(M, 8): the thread partition the fixture's host drivers were called with
    for k in ("wafr_urdf", "mpc_mode", "ee_cost"):
        if k in c:
            kw.setdefault(k, c[k])
    return Oracle(default_cfg(c["plant"], N=c["N"], M=c["M"], A=c["A"], integrator=c["integrator"], total_time=c["total_time"], cores=8, spawn_threads=0, **w, **kw), np.float64)


def close(got, ref, what, scale=None):
    ref = np.asarray(ref, np.float64).ravel(); got = np.asarray(got, np.float64).ravel()
    assert got.shape == ref.shape, what
    s = scale if scale is not None else max(np.abs(ref).max(), 1e-300)
    e = np.abs(got - ref).max() / s
    assert e <= TOL, (what, e)


def names(kind):
    return [c["name"] for c in MAN["cases"] if c["kind"] == kind]


@pytest.mark.parametrize("name", names("backward_pass"))
def test_backward_pass(name):
    case = CASES[name]
    o = oracle_for(case)
    n, m, N, M = o.n, o.m, case["cfg"]["N"], case["cfg"]["M"]
    a = {k: inp(case, k) for k in ("AB", "P", "p", "Pp", "pp", "H", "g", "d", "x", "xp")}
    KT, du, ApBK, Bdu = np.zeros(N * n * m), np.zeros(N * m), np.zeros(N * n * n), np.zeros(N * n)
    fail, dJexp, err = o.backward_pass(1 if case["sem"] == "gpu" else 0, a["AB"], a["P"], a["p"], a["Pp"], a["pp"], a["H"], a["g"], KT, du, a["d"], ApBK, Bdu, a["x"], a["xp"], case["rho"])
    assert list(err) == list(out(case, "err")) and fail == int(out(case, "err").any())
    if "P" not in case["outputs"]:
        return                                                   # a failing inversion: the flags are the contract, what the aborted blocks left behind is not
    got = dict(P=a["P"], p=a["p"], KT=KT, du=du, ApBK=ApBK, Bdu=Bdu, dJexp=dJexp, H=a["H"], g=a["g"], Pp=a["Pp"], pp=a["pp"])
    for k in ("KT", "du", "dJexp", "H", "g", "Pp", "pp"):
        close(got[k], out(case, k), (name, k))
    # P, p: slot j holds the cost-to-go at knot j + 1; the last slot (N - 1) is never written by either side: compare the written ones
    close(got["P"][: (N - 1) * n * n], out(case, "P")[: (N - 1) * n * n], (name, "P"))
    close(got["p"][: (N - 1) * n], out(case, "p")[: (N - 1) * n], (name, "p"))
    if M > 1:
        close(got["ApBK"][: (N - 1) * n * n], out(case, "ApBK")[: (N - 1) * n * n], (name, "ApBK"))
        close(got["Bdu"][: (N - 1) * n], out(case, "Bdu")[: (N - 1) * n], (name, "Bdu"))


@pytest.mark.parametrize("name", names("forward_sweep"))
def test_forward_sweep(name):
    case = CASES[name]
    o = oracle_for(case)
    alphas = inp(case, "alphas")
    for a_, al in enumerate(alphas):
        x = inp(case, "x")
        o.forward_sweep(x, inp(case, "ApBK"), inp(case, "Bdu"), inp(case, "d"), inp(case, "xp"), al)
        close(x, out(case, "xs")[a_], (name, a_))


@pytest.mark.parametrize("name", names("forward_sim"))
def test_forward_sim(name):
    case = CASES[name]
    o = oracle_for(case)
    alphas, xs = inp(case, "alphas"), inp(case, "xs")
    for a_, al in enumerate(alphas):
        x, u, d = xs[a_].copy(), inp(case, "u"), inp(case, "d")
        o.forward_sim(x, u, inp(case, "KT"), inp(case, "du"), d, al, inp(case, "xp"))
        close(x, out(case, "xs")[a_], (name, "x", a_)); close(u, out(case, "us")[a_], (name, "u", a_))
        close(d, out(case, "ds")[a_], (name, "d", a_), scale=max(np.abs(out(case, "xs")[a_]).max(), 1.0))


@pytest.mark.parametrize("name", names("integrator"))
def test_integrator_rules(name):
    case = CASES[name]
    o = oracle_for(case)
    x, u = inp(case, "x").reshape(-1, o.n), inp(case, "u").reshape(-1, o.m)
    for k in range(len(x)):
        close(o.integrator(x[k], u[k]), out(case, "xn")[k], (name, k))


@pytest.mark.parametrize("name", names("integrator_gradient"))
def test_integrator_gradients(name):
    case = CASES[name]
    o = oracle_for(case)
    n, m, N = o.n, o.m, case["cfg"]["N"]
    x, u = inp(case, "x").reshape(N, n), inp(case, "u").reshape(N, m)
    ref = out(case, "AB").reshape(N, -1)
    for k in range(N - 1):
        close(o.integrator_gradient(x[k], u[k]), ref[k], (name, k))
    assert not ref[N - 1].any()                                  # the last knot has no [A B] (grid N - 1)


@pytest.mark.parametrize("name", names("total_cost"))
def test_total_cost(name):
    case = CASES[name]
    o = oracle_for(case)
    xs, us, xg = inp(case, "xs"), inp(case, "us"), inp(case, "xg")
    for a_ in range(len(xs)):
        if case["sem"] == "gpu":
            close(o.total_cost(1, xs[a_], us[a_], xg), out(case, "J")[a_], (name, a_))
        else:                                                    # COST_THREADS strided partial sums, added in thread order (fpHelpers.cuh:454)
            total = 0.0
            for v in out(case, "Jparts")[a_]:
                total += float(v)
            close(o.total_cost(0, xs[a_], us[a_], xg), total, (name, a_))


@pytest.mark.parametrize("name", names("max_defect"))
def test_max_defect(name):
    case = CASES[name]
    o = oracle_for(case)
    ds = inp(case, "ds")
    for a_ in range(len(ds)):
        got = o.max_defect(1 if case["sem"] == "gpu" else 0, ds[a_])
        assert abs(got - out(case, "dmax")[a_]) <= TOL * max(1.0, abs(out(case, "dmax")[a_])), (name, a_)
    if case["sem"] == "cpu":
        assert not out(case, "dmax").any()                       # defectComp never updates its maximum (fpHelpers.cuh:123): the reference's CPU path sees 0


@pytest.mark.parametrize("name", names("cost_gradient_hessian"))
def test_cost_gradient_hessian(name):
    case = CASES[name]
    o = oracle_for(case)
    n, m, N = o.n, o.m, case["cfg"]["N"]
    nm = n + m
    x, u, xg = inp(case, "x").reshape(N, n), inp(case, "u").reshape(N, m), inp(case, "xg")
    H, g = out(case, "H").reshape(N, -1), out(case, "g").reshape(N, -1)
    for k in range(N):
        Hk, gk = o.cost_grad(x[k], u[k], xg, k)
        close(Hk, H[k], (name, "H", k)); close(gk, g[k], (name, "g", k))
    # ... and through the oracle's next-iteration setup (what the solver loop calls): same H, g for every knot
    if case["sem"] == "gpu":
        _, H2, g2 = oracle_for(case, wafr_urdf=1).next_iteration_setup(x.ravel().copy(), u.ravel().copy(), xg)
        close(H2, H, (name, "H via setup")); close(g2, g, (name, "g via setup"))


@pytest.mark.parametrize("name", names("arm_plant"))
def test_arm_plant_functions(name):
    """dynamics<T> / dynamicsGradient<T> of plants/dynamics_arm.cuh (:2097-2289, with everything they call: load_Tb, compute_T_TA_J, compute_Iw_Icrbs_twist, compute_JdotV,
    compute_M_Tau, invertMatrix, compute_dT_dTA_dJ, compute_dM, compute_dtwist, compute_dJdotV, compute_dWb, compute_dTau, finish_dqdd ...) on the tables of initI / initT"""
    case = CASES[name]
    o = oracle_for(case)
    x, u = inp(case, "x"), inp(case, "u")
    for k in range(len(x)):
        close(o.dynamics(x[k], u[k]), out(case, "qdd")[k], (name, "qdd", k))
        dq, qdd = o.dynamics_gradient(x[k], u[k])
        close(dq, out(case, "dqdd")[k], (name, "dqdd", k)); close(qdd, out(case, "qdd")[k], (name, "qdd of the gradient call", k))


def test_arm_model_tables_are_initI_initT():
    """the robot constants the product and the oracle carry (iiwa14_model_data.h) against the tables initI / initT fill (plants/dynamics_arm.cuh:73-427), both URDF variants"""
    import re
    txt = open(os.path.join(os.path.dirname(HERE), "oracle", "iiwa14_model_data.h")).read()
    def table(name):
        body = re.sub(r"/\*.*?\*/", "", txt[txt.index(name):], flags=re.S)
        body = body[body.index("="):body.index("};")]
        return np.asarray([float(v) for v in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)])
    I_all, F_all = table("IIWA14_SPATIAL_INERTIA").reshape(2, 7, 36), table("IIWA14_JOINT_FRAME").reshape(2, 7, 16)
    for nm, v in (("armplant_w1_g1", 1), ("armplant_w0_g1", 0)):
        case = CASES[nm]
        close(I_all[v].ravel(), out(case, "model_I"), (nm, "I"))
        # initT fills a 36-float slot per link whose first 16 floats hold the 4x4 joint frame at q = 0; the six q-dependent rotation entries (0, 1, 2, 4, 5, 6) are
        # written by updateT at run time, the constant ones must be the table's
        Tb = out(case, "model_T").reshape(7, 36)[:, :16]
        const = [i for i in range(16) if i not in (0, 1, 2, 4, 5, 6)]
        close(F_all[v][:, const], Tb[:, const], (nm, "T constants"))


@pytest.mark.parametrize("name", names("ee_pos"))
def test_tool_point_kinematics(name):
    case = CASES[name]
    o = oracle_for(case)
    for k, x in enumerate(inp(case, "x")):
        close(o.ee_pos(x, jac=False)[0], out(case, "eePos")[k], (name, k))


@pytest.mark.parametrize("name", names("ee_forward_sim"))
def test_end_effector_forward_sim_with_in_sim_cost(name):
    """forwardSimKern / forwardSim with EE_COST: control law, dynamics with the tool point, Euler step, defects, and the cost accumulated on the way in seven per-joint sums"""
    case = CASES[name]
    o = oracle_for(case)
    alphas, xs, goal = inp(case, "alphas"), inp(case, "xs"), inp(case, "goal")
    for a_, al in enumerate(alphas):
        x, u, d = xs[a_].copy(), inp(case, "u"), inp(case, "d")
        JT = o.forward_sim_ee(x, u, inp(case, "KT"), inp(case, "du"), d, al, inp(case, "xp"), goal)
        close(x, out(case, "xs")[a_], (name, "x", a_)); close(u, out(case, "us")[a_], (name, "u", a_))
        close(d, out(case, "ds")[a_], (name, "d", a_), scale=max(np.abs(out(case, "xs")[a_]).max(), 1.0))
        close(JT, out(case, "JT")[a_], (name, "JT", a_))


@pytest.mark.parametrize("name", names("ee_cost_gradient_hessian"))
def test_end_effector_cost_gradient_hessian(name):
    """costGradientHessianKern / -Threaded, end-effector branch: compute_eePos with its Jacobian, costGrad (gradient through the Jacobian, unweighted Gauss-Newton Hessian), the
    knot's cost; device: costKern<T,1> tree sum over the knots"""
    case = CASES[name]
    o = oracle_for(case)
    N = case["cfg"]["N"]
    x, u, goal = inp(case, "x").reshape(N, 14), inp(case, "u").reshape(N, 7), inp(case, "goal")
    H, g = out(case, "H").reshape(N, -1), out(case, "g").reshape(N, -1)
    Jk = []
    for k in range(N):
        Hk, gk = o.ee_cost_grad(x[k], u[k], goal, k)
        close(Hk, H[k], (name, "H", k)); close(gk, g[k], (name, "g", k))
        Jk.append(o.ee_cost(x[k], u[k], goal, k))
    if case["sem"] == "gpu":
        close(Jk, out(case, "J_knots"), (name, "J per knot"))
        v = list(Jk)                                             # costKern<T,1>: pairwise tree over blockDim.x = N entries (reduceSum)
        step = N // 2
        while step >= 1:
            for i in range(step):
                v[i] = v[i] + v[i + step]
            step //= 2
        close([v[0]], out(case, "J_total"), (name, "J total"))
    else:                                                        # host: COST_THREADS strided partial sums (costGradientHessianThreaded adds into JT[tid])
        T = len(out(case, "J_parts"))
        parts = [0.0] * T
        for k in range(N):
            parts[k % T] += Jk[k]
        close(parts, out(case, "J_parts"), (name, "J parts"))


def test_line_search_of_forwardSimGPU():
    ls = MAN["line_search"]
    c = ls["cfg"]
    o = Oracle(default_cfg(c["plant"], N=c["N"], M=c["M"], A=c["A"], exp_red_min=ls["constants"]["EXP_RED_MIN"], exp_red_max=ls["constants"]["EXP_RED_MAX"],
                           max_defect=ls["constants"]["MAX_DEFECT_SIZE"]), np.float64)
    accepted = 0
    for t in ls["cases"]:
        ai, ign, dJ, z = o.line_search_gpu(t["J"], t["dmax"], t["dJexp"], t["prevJ"], t["ignore_defect"], t["alphaIndex"])
        e = t["expect"]
        assert (ai, ign) == (e["alphaIndex"], e["ignore_defect"]), t
        assert abs(dJ - e["dJ"]) <= TOL * max(1.0, abs(e["dJ"])) and abs(z - e["z"]) <= TOL * max(1.0, abs(e["z"])), t
        accepted += e["dJ"] >= 0
    assert 5 <= accepted <= len(ls["cases"]) - 5


@pytest.mark.parametrize("name", names("solve"))
def test_whole_solves_of_the_reference_runiLQR_GPU(name):
    """runiLQR_GPU itself (DDPWrappers.cuh:10-138) executed at generation time -- host driver, loadVarsGPU, initAlgGPU, backwardPassGPU, forwardSimGPU + line search,
    acceptRejectTrajGPU, nextIterationSetupGPU, storeVarsGPU as they stand, every kernel they launch under the SIMT emulation -- against the oracle's GPU-semantics driver:
    identical step-size indices (rejections, the initial -1 / 0, the exit iteration) and J, x, u, K to 1e-10 (measured: bit-identical)."""
    case = CASES[name]
    c = case["cfg"]
    kw = {k: c[k] for k in ("wafr_urdf", "mpc_mode", "ee_cost", "ignore_max_rho_exit") if k in c}
    o = Oracle(default_cfg(4, N=c["N"], M=c["M"], A=c["A"], integrator=c["integrator"], total_time=c["total_time"], tol_cost=c["tol_cost"], max_iter=c["max_iter"],
                           cores=1, spawn_threads=0, **kw), np.float64)
    fl = case.get("flags", {})
    r = o.run_ilqr_gpusem(inp(case, "x0"), inp(case, "u0"), inp(case, "xg"), rollout=fl.get("rollout", 0), ignore_first_defect=fl.get("ifd", 1))
    ref_a, ref_J = out(case, "alphaOut"), out(case, "Jout")
    it = r["iters"]
    assert list(r["alphaOut"][: it + 1]) == list(ref_a[: it + 1]), (name, list(r["alphaOut"][: it + 1]), list(ref_a))
    assert not ref_J[it + 1:].any()                               # the reference wrote exactly iter + 1 entries
    for k, got in (("Jout", r["Jout"][: it + 1]), ("x", r["x"]), ("u", r["u"]), ("KT", r["KT"])):
        ref = ref_J[: it + 1] if k == "Jout" else out(case, k)
        e = np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-300)
        assert e <= 1e-10, (name, k, e)


def test_fixture_provenance_is_data_only():
    """the fixture holds numbers and case descriptions, no reference text"""
    assert set(MAN) == {"_provenance", "cases", "line_search"}
    for c in MAN["cases"]:
        assert set(c) <= {"name", "kind", "cfg", "sem", "inputs", "outputs", "rho", "weights", "inputs_of", "flags"}
    assert all(DATA[k].dtype.kind in "fi" for k in DATA)
