"""The kernels against the reference's EXECUTED statements with NO oracle in the chain ("missing" 4).

tests/golden/phase_fixtures.npz holds inputs and outputs of the reference's own backPassKern, forwardSweepKern, forwardSimKern (with its integrators and plant
plug-ins), integratorGradientKern, costGradientHessianKern, costKern / defectKern and the host line search of forwardSimGPU, executed in float64 at fixture-generation
time (tests/golden/make_phase_fixtures.py + refc2py.py; tests/test_phase_pins.py pins the ORACLE to them).  Here the same stored inputs go straight into float64
handles through the teacher-forcing hooks of the C ABI (pddp_set_array / pddp_set_state / pddp_run_phase) and the kernels' outputs are held against the stored
outputs at 1e-9 (integers identical): bpHelpers.cuh:339-420, fpHelpers.cuh:57-63, 279-301, 134-152, 96-111, nisInitHelpers.cuh:46-93, 205-221.
Backends: the kernel bodies on the host (test tool, CPU suite) and the HIP kernels on the GPU -- for the arm every float64 selection the library carries
(lane groups, wave-cooperative, matrix cores, thread lanes)."""
import json
import os

import numpy as np
import pytest

import pyddp
from backends import BACKENDS, make_solver

HERE = os.path.dirname(os.path.abspath(__file__))
MAN = json.load(open(os.path.join(HERE, "golden", "phase_fixtures.json")))
MAN4 = json.load(open(os.path.join(HERE, "golden", "phase_fixtures_r04.json")))        # round 4: sweeps from the solver's invariant, the headline-size whole solve
DATA = dict(np.load(os.path.join(HERE, "golden", "phase_fixtures.npz")))
DATA.update(np.load(os.path.join(HERE, "golden", "phase_fixtures_r04.npz")))
MAN["cases"] = MAN["cases"] + MAN4["cases"]
CASES = {c["name"]: c for c in MAN["cases"]}
TOL = 1e-9
DIMS = {1: (1, 2, 1), 2: (2, 4, 1), 3: (6, 12, 4), 4: (7, 14, 7)}


def names(kind, sem="gpu"):
    return [c["name"] for c in MAN["cases"] if c["kind"] == kind and c.get("sem") == sem]


def inp(case, key):
    return np.array(DATA["%s/in/%s" % (case.get("inputs_of", case["name"]), key)], np.float64)


def out(case, key):
    return np.array(DATA["%s/out/%s" % (case["name"], key)])


def close(got, ref, what, scale=None, tol=TOL):
    ref = np.asarray(ref, np.float64).ravel(); got = np.asarray(got, np.float64).ravel()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    s = scale if scale is not None else max(np.abs(ref).max(), 1e-300)
    e = np.abs(got - ref).max() / s
    assert e <= tol, (what, e)


# the float64 kernel selections of the arm (pddp_config.kernels)
ARM_SELECTIONS = [pytest.param({}, id="default"), pytest.param(dict(bp="coop", fp="coop"), id="coop"), pytest.param(dict(bp="mx", fp="tl"), id="mx-tl")]


def handle(backend, case, env=None, **kw):
    c = case["cfg"]
    w = {k.strip("_"): v for k, v in case.get("weights", {}).items()}
    for k in ("wafr_urdf", "mpc_mode", "ee_cost"):
        if k in c: kw.setdefault(k, c[k])
    s = make_solver(backend, c["plant"], dtype=1, N=c["N"], M=c["M"], A=c["A"], integrator=c["integrator"], total_time=c["total_time"], **w, **kw, kernels=dict(env or {}))
    return s


def prime(s, case, x=None, u=None, xg=None):
    """a handle needs a loaded problem before its arrays can be overwritten (state, step sizes, pointer tables)"""
    npos, n, m = DIMS[case["cfg"]["plant"]]
    N = case["cfg"]["N"]
    s.load(np.zeros(N * n) if x is None else x, np.zeros(N * m) if u is None else u, np.zeros(n) if xg is None else xg)
    return n, m, N


def selections_for(case):
    if case["cfg"]["plant"] == 3:       # the quadrotor's size: also the matrix-core backward pass (k_bp_mq, bp_mq.hpp; the library's choice with the device full)
        return [pytest.param({}, id="default"), pytest.param(dict(cf_bp="mq"), id="matrix-core")]
    return ARM_SELECTIONS if case["cfg"]["plant"] == 4 else [pytest.param({}, id="default")]


def bp_params():
    for name in names("backward_pass"):
        for sel in selections_for(CASES[name]):
            yield pytest.param(name, sel.values[0], id=name + "-" + sel.id)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,env", list(bp_params()))
def test_backward_pass_kernels_on_the_references_inputs(backend, name, env):
    case = CASES[name]
    if (env.get("bp") == "mx" or env.get("cf_bp") == "mq") and backend != "hip": pytest.skip("the matrix-core backward pass exists on the GPU only")
    s = handle(backend, case, env)
    n, m, N = prime(s, case)
    M = case["cfg"]["M"]
    for k in ("AB", "H", "g", "P", "p", "Pp", "pp"):
        s.set(k, inp(case, k))
    s.set("dcur", inp(case, "d"))
    s.set("xb", np.stack([inp(case, "x").reshape(N, n), inp(case, "xp").reshape(N, n)]))     # half 0: the current trajectory, half 1: the one Pp / pp were computed at (d_xp2)
    st = s.get_state()
    st[0].rho = case["rho"]; st[0].cur = 0; st[0].cur2 = 1; st[0].pw = 0
    s.set_state(st)
    s.run_phase(pyddp.PHASE_BP)
    assert list(s.get("err")[:M]) == list(out(case, "err"))
    if "P" not in case["outputs"]:
        return                                                    # a failing inversion: the flags are the contract
    for k in ("KT", "du", "dJexp"):
        close(s.get(k)[: out(case, k).size], out(case, k), (name, k))
    close(s.get("P")[: (N - 1) * n * n], out(case, "P")[: (N - 1) * n * n], (name, "P"))
    close(s.get("p")[: (N - 1) * n], out(case, "p")[: (N - 1) * n], (name, "p"))
    if M > 1:
        close(s.get("ApBK")[: (N - 1) * n * n], out(case, "ApBK")[: (N - 1) * n * n], (name, "ApBK"))
        close(s.get("Bdu")[: (N - 1) * n], out(case, "Bdu")[: (N - 1) * n], (name, "Bdu"))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", [n_ for n_ in names("forward_sweep") if n_.startswith("sweepinv")])
def test_forward_sweep_kernels_on_the_references_inputs(backend, name):
    """forwardSweepKern's observable effect is every candidate's SEGMENT START states (the rollouts that follow overwrite the rest of x).  The cases are the ones whose
    candidate slots hold the current trajectory when the sweep starts -- the invariant of the solver loop under which the reference's in-place
    `xkp1 += ...` (fpHelpers.cuh:43) and the library's sweeps from the current trajectory are the same computation; the general in-place form is pinned on the oracle
    (test_phase_pins.py)."""
    case = CASES[name]
    envs = [{}] if case["cfg"]["plant"] != 4 else [{}, dict(fp="coop")]
    for env in envs:
        s = handle(backend, case, env)
        n, m, N = prime(s, case)
        M, A = case["cfg"]["M"], case["cfg"]["A"]
        alphas = inp(case, "alphas")
        assert len(alphas) == A
        s.set("alpha", alphas); s.set("ApBK", inp(case, "ApBK")); s.set("Bdu", inp(case, "Bdu")); s.set("dcur", inp(case, "d"))
        s.set("xb", np.stack([inp(case, "xp").reshape(N, n), inp(case, "xp").reshape(N, n)]))
        s.set("xs", np.tile(inp(case, "x").reshape(1, N, n), (A, 1, 1)))
        s.run_phase(pyddp.PHASE_FP)
        xs = s.get("xs").reshape(A, N, n)
        ref = out(case, "xs").reshape(A, N, n)
        starts = [b * (N // M) for b in range(1, M)]
        close(xs[:, starts], ref[:, starts], (name, env))


def sim_params():
    for name in names("forward_sim"):
        for sel in selections_for(CASES[name]):
            if sel.values[0].get("bp") == "mx": sel = pytest.param(dict(fp="tl"), id="tl")
            yield pytest.param(name, sel.values[0], id=name + "-" + sel.id)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,env", list(sim_params()))
def test_rollout_kernels_on_the_references_inputs(backend, name, env):
    """forwardSimKern / forwardSimInner with computeControlKT, the three integrators (midpoint's start-velocity quirk, RK3) and the plant plug-ins, from the stored segment
    start states of every candidate (PDDP_PHASE_ROLLOUT: no sweep in front)."""
    case = CASES[name]
    s = handle(backend, case, env)
    n, m, N = prime(s, case)
    M, A = case["cfg"]["M"], case["cfg"]["A"]
    alphas = inp(case, "alphas")
    s.set("alpha", alphas); s.set("KT", inp(case, "KT")); s.set("du", inp(case, "du")); s.set("ucur", inp(case, "u"))
    s.set("xb", np.stack([inp(case, "xp").reshape(N, n), inp(case, "xp").reshape(N, n)]))
    s.set("dcur", inp(case, "d")); s.set("ds", np.tile(inp(case, "d").reshape(1, N, n), (A, 1, 1)))
    s.set("xs", inp(case, "xs")); s.set("us", np.tile(inp(case, "u").reshape(1, N, m), (A, 1, 1)))
    s.run_phase(pyddp.PHASE_ROLLOUT)
    xs, us, ds = s.get("xs").reshape(A, N, n), s.get("us").reshape(A, N, m), s.get("ds").reshape(A, N, n)
    rx, ru, rd = out(case, "xs").reshape(A, N, n), out(case, "us").reshape(A, N, m), out(case, "ds").reshape(A, N, n)
    NB = N // M
    # The round-3 cases draw x_0 and xp_0 independently; the reference's segment 0 then starts from the candidate's x_0 with a control law that sees x_0 - xp_0 != 0, a state
    # the solver never produces (knot 0 of every candidate IS the current trajectory's; the library's segment 0 starts from it).  Those cases are held on the segments
    # b >= 1, the round-4 cases ("siminv_*": x_0 = xp_0) on every knot.
    first = 0 if name.startswith("siminv") else NB
    if first >= N:
        pytest.skip("single shooting with an independent x_0: covered by the siminv_* cases")
    bnd = [k for k in range(first, N - 1) if (k + 1) % NB == 0]
    for a in range(A):
        close(xs[a][first:], rx[a][first:], (name, "x", a)); close(us[a][first: N - 1], ru[a][first: N - 1], (name, "u", a))
        if bnd: close(ds[a][bnd], rd[a][bnd], (name, "d", a), scale=max(np.abs(rx[a]).max(), 1.0))


def nis_params(kind):
    for name in names(kind):
        for sel in selections_for(CASES[name]):
            if sel.values[0].get("bp") == "mx": sel = pytest.param(dict(fp="tl"), id="tl")
            yield pytest.param(name, sel.values[0], id=name + "-" + sel.id)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,env", list(nis_params("integrator_gradient")))
def test_integrator_gradient_kernels_on_the_references_inputs(backend, name, env):
    """integratorGradientKern -> _integratorGradient (Euler, midpoint, RK3 with its stage-state quirk) with every plant's dynamicsGradient: [A B] of a loaded trajectory."""
    case = CASES[name]
    s = handle(backend, case, env)
    npos, n, m = DIMS[case["cfg"]["plant"]]
    N = case["cfg"]["N"]
    s.load(inp(case, "x"), inp(case, "u"), np.zeros(n))              # loadVarsGPU + initAlgGPU: the derivatives of the loaded trajectory
    ref = out(case, "AB").reshape(N, -1)
    close(s.get("AB").reshape(N, -1)[: N - 1], ref[: N - 1], name)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,env", list(nis_params("cost_gradient_hessian")))
def test_cost_gradient_hessian_kernels_on_the_references_inputs(backend, name, env):
    case = CASES[name]
    s = handle(backend, case, env)
    npos, n, m = DIMS[case["cfg"]["plant"]]
    N, nm = case["cfg"]["N"], n + m
    s.load(inp(case, "x"), inp(case, "u"), inp(case, "xg"))
    H, rH = s.get("H").reshape(N, nm, nm), out(case, "H").reshape(N, nm, nm)
    close(H[: N - 1], rH[: N - 1], (name, "H")); close(H[N - 1, :n, :n], rH[N - 1, :n, :n], (name, "H final"))     # the final knot's costGrad writes the state block only (cost_arm.cuh:159-174)
    close(s.get("g"), out(case, "g"), (name, "g"))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", names("total_cost"))
def test_cost_and_defect_reductions_on_the_references_inputs(backend, name):
    """costKern (pairwise tree over the knots) for every candidate: the initial-cost kernel evaluates a loaded trajectory."""
    case = CASES[name]
    npos, n, m = DIMS[case["cfg"]["plant"]]
    N = case["cfg"]["N"]
    xs, us, xg = inp(case, "xs").reshape(-1, N, n), inp(case, "us").reshape(-1, N, m), inp(case, "xg")
    s = handle(backend, case)
    for a in range(len(xs)):
        s.load(xs[a], us[a], xg)
        J0 = s.get("Jout")[0]
        close(J0, out(case, "J")[a], (name, a))


@pytest.mark.parametrize("backend", BACKENDS)
def test_line_search_kernel_on_the_references_tables(backend):
    """the host loop of forwardSimGPU (fpHelpers.cuh:395-408) as the reference executes it, 40 stored cost tables: step-size index, ignore_defect, z.  The partial sums
    of the expected reduction are handed over as block 0's pair (the kernel adds the blocks' pairs in order: 0 + 0 + ... stays exact)."""
    ls = MAN["line_search"]
    c = ls["cfg"]
    s = make_solver(backend, c["plant"], dtype=1, N=c["N"], M=c["M"], A=c["A"], integrator=c["integrator"], total_time=c["total_time"], exp_red_min=ls["constants"]["EXP_RED_MIN"],
                    exp_red_max=ls["constants"]["EXP_RED_MAX"], max_defect=ls["constants"]["MAX_DEFECT_SIZE"], wafr_urdf=1, tol_cost=0.0)
    npos, n, m = DIMS[c["plant"]]
    accepted = 0
    for t in ls["cases"]:
        s.load(np.zeros(c["N"] * n), np.zeros(c["N"] * m), np.zeros(n))
        s.set("alpha", np.asarray(t["alpha"])); s.set("J", np.asarray(t["J"])); s.set("dmax", np.asarray(t["dmax"]))
        dj = np.zeros(2 * c["M"]); dj[0], dj[1] = t["dJexp"]
        s.set("dJexp", dj)
        st = s.get_state()
        st[0].prevJ = t["prevJ"]; st[0].ignore_defect = t["ignore_defect"]; st[0].alphaIndex = t["alphaIndex"]
        s.set_state(st)
        s.run_phase(pyddp.PHASE_LS)
        st = s.get_state()
        e = t["expect"]
        if e["dJ"] < 0:
            assert st[0].accepted == 0, t
        else:
            accepted += 1
            assert st[0].accepted == 1 and st[0].alphaIndex == e["alphaIndex"] and st[0].ignore_defect == e["ignore_defect"], (t, st[0].alphaIndex)
            assert abs(st[0].z - e["z"]) <= 1e-12 * max(1, abs(e["z"]))
    assert 5 <= accepted <= len(ls["cases"]) - 5


def solve_params():
    for c in MAN["cases"]:
        if c["kind"] != "solve": continue
        for sel in ARM_SELECTIONS:
            env = sel.values[0]
            if c["cfg"].get("ee_cost") and env.get("fp") == "coop": continue
            yield pytest.param(c["name"], env, id=c["name"] + "-" + sel.id)
        yield pytest.param(c["name"], dict(bp="mx", fp="tl4"), id=c["name"] + "-one-problem")


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,env", list(solve_params()))
def test_whole_solves_on_the_references_inputs(backend, name, env):
    """runiLQR_GPU executed end to end at generation time (DDPWrappers.cuh:10-138 with every kernel it launches) against pddp_solve on the same inputs, float64: identical
    step-size indices (the initial -1 / 0, rejections, the exit iteration), J / x / u to 1e-7, K to 1e-6 -- up to the headline size N = 128, M = 4, A = 8
    (solve_arm_N128_M4_A8: BASELINE configs[2]); every float64 kernel selection of the arm, including the parity instantiations of the benched families."""
    case = CASES[name]
    c = case["cfg"]
    if backend != "hip" and env.get("bp") == "mx": pytest.skip("matrix-core / pipeline kernels: GPU only")
    kw = {k: c[k] for k in ("wafr_urdf", "mpc_mode", "ee_cost", "ignore_max_rho_exit", "tol_cost", "max_iter") if k in c}
    s = handle(backend, dict(cfg=c), env, **{k: v for k, v in kw.items() if k not in ("wafr_urdf", "mpc_mode", "ee_cost")})
    fl = case.get("flags", {})
    xg = inp(case, "xg")
    if xg.size < 14: xg = np.concatenate([xg, np.zeros(14 - xg.size)])
    res = s.solve(inp(case, "x0"), inp(case, "u0"), xg, forward_rollout=fl.get("rollout", 0), ignore_first_defect=fl.get("ifd", 1))
    ref_a, ref_J = out(case, "alphaOut"), out(case, "Jout")
    it = int(res["iters"][0])
    assert list(res["alphaOut"][0][: it + 1]) == list(ref_a[: it + 1]), (list(res["alphaOut"][0][: it + 1]), list(ref_a))
    # (a whole solve carries rounding-order differences from iteration to iteration: 1e-7 on what it returns, like tests/test_f64_benched_family.py; the phases above hold 1e-9)
    close(res["Jout"][0][: it + 1], ref_J[: it + 1], (name, "J"), tol=1e-7)
    close(res["x"][0], out(case, "x"), (name, "x"), tol=1e-7); close(res["u"][0].ravel()[: (c["N"] - 1) * 7], out(case, "u")[: (c["N"] - 1) * 7], (name, "u"), tol=1e-7)
    close(res["KT"][0].ravel()[: (c["N"] - 1) * 98], out(case, "KT")[: (c["N"] - 1) * 98], (name, "KT"), tol=1e-6)
