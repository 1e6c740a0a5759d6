"""FLOAT32 kernels against the reference's statements EXECUTED IN FLOAT32, with no oracle in the chain (VERDICT r5 "missing" 3 / task 1b).

tests/golden/phase_fixtures_f32.npz holds inputs and outputs of the reference's own backPassKern, forwardSweepKern, forwardSimKern, integratorGradientKern,
costGradientHessianKern, costKern / defectKern and one whole runiLQR_GPU solve executed in float32 at fixture-generation time (refc2py's float32 mode: strict IEEE single
precision, one rounding per operation, the reference's launch geometry under the SIMT emulation -- so the pairwise tree of reduceSum, cudaUtils.h:187-207, computeExpRed's
partial sums, bpHelpers.cuh:326-332, and every inner product's ORDER are in the numbers).  tests/test_phase_pins_f32.py pins the ORACLE to them on the CPU; here the same
stored inputs go straight into float32 handles through the C ABI's teacher-forcing hooks (pddp_set_array / pddp_set_state / pddp_run_phase) on the kernel families that are
built to execute the reference's IEEE operations one for one (-ffp-contract=off: the wave-cooperative kernels and the arm's lane-group kernels):

  * everything that is additions and multiplications -- backward pass (gains, feed-forward, cost-to-go, sweep operands, expected reduction, error flags), linear forward
    sweep, cost gradient / Hessian, the cost tree, the defect maximum -- BIT FOR BIT;
  * what passes through sin / cos (rollouts, integrator gradients, the whole solve): the device's single-precision sine and cosine are not glibc's sinf / cosf (the
    fixture's), so those are held to a few units in the last place of the trajectory's scale instead, with identical step-size decisions for the solve.

Backends: the kernel bodies on the host (test tool, CPU suite) and the HIP kernels on the GPU (-m gpu)."""
import json
import os

import numpy as np
import pytest

import pyddp
from backends import BACKENDS, make_solver

HERE = os.path.dirname(os.path.abspath(__file__))
MAN = json.load(open(os.path.join(HERE, "golden", "phase_fixtures_f32.json")))
DATA = dict(np.load(os.path.join(HERE, "golden", "phase_fixtures_f32.npz")))
MAN6 = json.load(open(os.path.join(HERE, "golden", "phase_fixtures_f32_r06.json")))            # round 6: float32 sweeps that start from the solver's invariant (candidate slots = current trajectory)
DATA.update(np.load(os.path.join(HERE, "golden", "phase_fixtures_f32_r06.npz")))
MAN["cases"] = MAN["cases"] + MAN6["cases"]
CASES = {c["name"]: c for c in MAN["cases"]}
F32 = np.float32
DIMS = {1: (1, 2, 1), 2: (2, 4, 1), 3: (6, 12, 4), 4: (7, 14, 7)}
TRIG_TOL = 4e-6          # max |a - ref| / max |ref| for phases of the closed-form plants, whose operands pass through the device's sinf / cosf (about 30 units in the last place of the largest entry)
ARM_TOL = 1e-4           # the arm's dynamics / dynamicsGradient are other ALGORITHMS than the reference's (body-frame recursions, plant_arm*.hpp): north_star's float32 tolerance


def tol_of(case):
    return ARM_TOL if case["cfg"]["plant"] == 4 else TRIG_TOL

# the bit-exact float32 families: wave-cooperative (every plant) and, for the arm, 8-lane groups
ARM_FAMILIES = [pytest.param(dict(bp="coop", fp="coop"), id="cooperative"), pytest.param(dict(bp="lg", fp="lg", sweep="alpha"), id="lane-groups")]
CF_FAMILIES = [pytest.param(dict(cf="coop"), id="cooperative")]


def names(kind, sem="gpu"):
    return [c["name"] for c in MAN["cases"] if c["kind"] == kind and c.get("sem") == sem]


def families(case):
    return ARM_FAMILIES if case["cfg"]["plant"] == 4 else CF_FAMILIES


def params(kind):
    for name in names(kind):
        for fam in families(CASES[name]):
            yield pytest.param(name, fam.values[0], id=name + "-" + fam.id)


def inp(case, key):
    a = np.array(DATA["%s/in/%s" % (case["name"], key)])
    return a.astype(F32) if a.dtype.kind == "f" else a


def out(case, key):
    return np.array(DATA["%s/out/%s" % (case["name"], key)])


def same(got, ref, what):
    got, ref = np.asarray(got).ravel(), np.asarray(ref).ravel()
    assert got.dtype == F32 and got.shape == ref.shape, (what, got.dtype, got.shape, ref.shape)
    ref = ref.astype(F32)
    if not np.array_equal(got, ref, equal_nan=True):
        bad = np.flatnonzero(got != ref)
        raise AssertionError((what, "%d of %d entries differ" % (bad.size, got.size), [(int(i), float(got[i]), float(ref[i])) for i in bad[:5]]))


def near(got, ref, what, tol=TRIG_TOL, scale=None):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    assert got.shape == ref.shape, what
    e = np.abs(got - ref).max() / (scale if scale is not None else max(np.abs(ref).max(), 1e-30))
    assert e <= tol, (what, e)
    return e


def handle(backend, case, sel, **kw):
    c = case["cfg"]
    w = {k.strip("_"): v for k, v in (case.get("weights") or {}).items()}
    for k in ("wafr_urdf", "mpc_mode", "ee_cost"):
        if k in c:
            kw.setdefault(k, c[k])
    return make_solver(backend, c["plant"], dtype=0, N=c["N"], M=c["M"], A=c["A"], integrator=c["integrator"], total_time=c["total_time"], kernels=dict(sel), **w, **kw)


def prime(s, case):
    npos, n, m = DIMS[case["cfg"]["plant"]]
    N = case["cfg"]["N"]
    s.load(np.zeros(N * n, F32), np.zeros(N * m, F32), np.zeros(n, F32))
    return n, m, N


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,sel", list(params("backward_pass")))
def test_backward_pass_float32_kernels_bit_for_bit(backend, name, sel):
    """backPassKern (bpHelpers.cuh:339-420) in float32: arm 7 x 7 (invertMatrix), cart-pole 1 x 1 (computeKTdu_dim1), quadrotor 4 x 4 (invHuu_dim4)."""
    case = CASES[name]
    s = handle(backend, case, sel)
    n, m, N = prime(s, case)
    M = case["cfg"]["M"]
    for k in ("AB", "H", "g", "P", "p", "Pp", "pp"):
        s.set(k, inp(case, k))
    s.set("dcur", inp(case, "d"))
    s.set("xb", np.stack([inp(case, "x").reshape(N, n), inp(case, "xp").reshape(N, n)]))
    st = s.get_state()
    st[0].rho = case["rho"]; st[0].cur = 0; st[0].cur2 = 1; st[0].pw = 0
    s.set_state(st)
    s.run_phase(pyddp.PHASE_BP)
    assert list(s.get("err")[:M]) == list(out(case, "err"))
    for k in ("KT", "du", "dJexp"):
        same(s.get(k)[: out(case, k).size], out(case, k), (name, k))          # dJexp: every block's pair of partial sums, in computeExpRed's order
    same(s.get("P")[: (N - 1) * n * n], out(case, "P")[: (N - 1) * n * n], (name, "P"))
    same(s.get("p")[: (N - 1) * n], out(case, "p")[: (N - 1) * n], (name, "p"))
    if M > 1:
        same(s.get("ApBK")[: (N - 1) * n * n], out(case, "ApBK")[: (N - 1) * n * n], (name, "ApBK"))
        same(s.get("Bdu")[: (N - 1) * n], out(case, "Bdu")[: (N - 1) * n], (name, "Bdu"))
    s.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,sel", list(params("forward_sweep")))
def test_forward_sweep_float32_kernels_bit_for_bit(backend, name, sel):
    """forwardSweepKern (fpHelpers.cuh:19-63): every candidate's segment start states.  The stored case sweeps a candidate x that differs from the current trajectory xp in place;
    the library's sweep starts from the current trajectory (the solver loop's invariant), so the cases held here are the round-6 ones whose candidate slots hold the current
    trajectory (sweepinv32_*: the same computation then); the in-place form of the other two is pinned on the oracle (tests/test_phase_pins_f32.py)."""
    case = CASES[name]
    x, xp = inp(case, "x"), inp(case, "xp")
    N, n = case["cfg"]["N"], DIMS[case["cfg"]["plant"]][1]
    if not np.array_equal(x, xp):
        pytest.skip("candidate slots differ from the current trajectory: the in-place form, pinned on the oracle (tests/test_phase_pins_f32.py)")
    s = handle(backend, case, sel)
    prime(s, case)
    M, A = case["cfg"]["M"], case["cfg"]["A"]
    s.set("alpha", inp(case, "alphas")); s.set("ApBK", inp(case, "ApBK")); s.set("Bdu", inp(case, "Bdu")); s.set("dcur", inp(case, "d"))
    s.set("xb", np.stack([xp.reshape(N, n), xp.reshape(N, n)]))
    s.set("xs", np.tile(x.reshape(1, N, n), (A, 1, 1)))
    s.run_phase(pyddp.PHASE_FP)
    xs, ref = s.get("xs").reshape(A, N, n), out(case, "xs").reshape(A, N, n)
    starts = [b * (N // M) for b in range(1, M)]
    same(xs[:, starts], ref[:, starts], name)
    s.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,sel", list(params("total_cost")))
def test_cost_tree_float32_kernels_bit_for_bit(backend, name, sel):
    """costKern + reduceSum (fpHelpers.cuh:134-152, cudaUtils.h:187-207): the pairwise tree over the knots, every candidate."""
    case = CASES[name]
    npos, n, m = DIMS[case["cfg"]["plant"]]
    N = case["cfg"]["N"]
    xs, us, xg = inp(case, "xs").reshape(-1, N, n), inp(case, "us").reshape(-1, N, m), inp(case, "xg")
    s = handle(backend, case, sel)
    for a in range(len(xs)):
        s.load(xs[a], us[a], xg)
        same(s.get("Jout")[:1], out(case, "J")[a: a + 1], (name, a))
    s.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,sel", list(params("cost_gradient_hessian")))
def test_cost_gradient_hessian_float32_kernels_bit_for_bit(backend, name, sel):
    case = CASES[name]
    s = handle(backend, case, sel)
    npos, n, m = DIMS[case["cfg"]["plant"]]
    N, nm = case["cfg"]["N"], n + m
    s.load(inp(case, "x"), inp(case, "u"), inp(case, "xg"))
    H, rH = s.get("H").reshape(N, nm, nm), out(case, "H").reshape(N, nm, nm)
    same(H[: N - 1], rH[: N - 1], (name, "H")); same(H[N - 1, :n, :n], rH[N - 1, :n, :n], (name, "H final"))
    same(s.get("g"), out(case, "g"), (name, "g"))
    s.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,sel", list(params("integrator_gradient")))
def test_integrator_gradient_float32_kernels(backend, name, sel):
    """integratorGradientKern (nisInitHelpers.cuh:205-221) with every plant's dynamicsGradient: sin / cos inside -- the device's, not glibc's."""
    case = CASES[name]
    s = handle(backend, case, sel)
    npos, n, m = DIMS[case["cfg"]["plant"]]
    N = case["cfg"]["N"]
    s.load(inp(case, "x"), inp(case, "u"), np.zeros(n, F32))
    ref = out(case, "AB").reshape(N, -1)
    e = near(s.get("AB").reshape(N, -1)[: N - 1], ref[: N - 1], name, tol=tol_of(case))
    print(f"{name}: max |AB - reference float32| / max |AB| = {e:.2e}")
    s.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,sel", list(params("solve")))
def test_whole_float32_solve_follows_the_references_float32_solve(backend, name, sel):
    """runiLQR_GPU executed end to end in float32 (DDPWrappers.cuh:10-138) against pddp_solve on the bit-exact families.  What a whole float32 solve can be held to WITHOUT an
    oracle in the chain: the integers -- every step-size index, rejections included, over all eight iterations -- and the initial cost bit for bit (costKern's tree on the loaded
    trajectory).  The costs after that are NOT a parity bar here: the arm's dynamicsGradient is another algorithm than the reference's (1e-5 relative in [A B], test above) and
    the float32 Riccati recursion amplifies that to 3e-4 in J after one iteration and 2e-2 after five on this N = 16 problem -- the size of the distance between the
    reference's own float32 and float64 evaluations; the per-phase tests above and tests/test_fp32_bar.py (per iteration, teacher-forced) are the bars.  J[1] is held to 1e-3."""
    case = CASES[name]
    c = case["cfg"]
    s = handle(backend, case, sel, **{k: c[k] for k in ("tol_cost", "max_iter") if k in c})
    xg = inp(case, "xg")
    if xg.size < 14:
        xg = np.concatenate([xg, np.zeros(14 - xg.size, F32)])
    fl = case.get("flags") or {}
    res = s.solve(inp(case, "x0"), inp(case, "u0"), xg, forward_rollout=fl.get("rollout", 0), ignore_first_defect=fl.get("ifd", 1))
    it = int(res["iters"][0])
    ref_a, ref_J = out(case, "alphaOut"), out(case, "Jout")
    assert it == c["max_iter"] and list(res["alphaOut"][0][: it + 1]) == list(ref_a[: it + 1]), (list(res["alphaOut"][0][: it + 1]), list(ref_a))
    assert (np.asarray(ref_a[1: it + 1]) >= 0).sum() >= 3, "the case must accept iterations"
    same(res["Jout"][0][:1], ref_J[:1], (name, "initial cost"))
    near(res["Jout"][0][1:2], ref_J[1:2], (name, "J[1]"), tol=1e-3)
    rel = np.abs(res["Jout"][0][: it + 1].astype(np.float64) - ref_J[: it + 1]) / ref_J[: it + 1]
    print(f"{name}: |J - reference float32 J| / J per iteration: " + " ".join(f"{v:.1e}" for v in rel))
    s.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,sel", list(params("forward_sim")))
def test_rollout_float32_kernels(backend, name, sel):
    """forwardSimKern / forwardSimInner with computeControlKT and the integrators (fpHelpers.cuh:202-301), from every candidate's stored segment start states
    (PDDP_PHASE_ROLLOUT: no sweep in front); held on the segments b >= 1 like the float64 cases that draw x_0 and xp_0 independently (tests/test_fixtures_direct.py)."""
    case = CASES[name]
    s = handle(backend, case, sel)
    n, m, N = prime(s, case)
    M, A = case["cfg"]["M"], case["cfg"]["A"]
    s.set("alpha", inp(case, "alphas")); s.set("KT", inp(case, "KT")); s.set("du", inp(case, "du")); s.set("ucur", inp(case, "u"))
    s.set("xb", np.stack([inp(case, "xp").reshape(N, n), inp(case, "xp").reshape(N, n)]))
    s.set("dcur", inp(case, "d")); s.set("ds", np.tile(inp(case, "d").reshape(1, N, n), (A, 1, 1)))
    s.set("xs", inp(case, "xs")); s.set("us", np.tile(inp(case, "u").reshape(1, N, m), (A, 1, 1)))
    s.run_phase(pyddp.PHASE_ROLLOUT)
    xs, us, ds = s.get("xs").reshape(A, N, n), s.get("us").reshape(A, N, m), s.get("ds").reshape(A, N, n)
    rx, ru, rd = out(case, "xs").reshape(A, N, n), out(case, "us").reshape(A, N, m), out(case, "ds").reshape(A, N, n)
    NB = N // M
    first = NB
    bnd = [k for k in range(first, N - 1) if (k + 1) % NB == 0]
    worst = 0.0
    for a in range(A):
        worst = max(worst, near(xs[a][first:], rx[a][first:], (name, "x", a), tol=tol_of(case)), near(us[a][first: N - 1], ru[a][first: N - 1], (name, "u", a), tol=tol_of(case)))
        if bnd:
            near(ds[a][bnd], rd[a][bnd], (name, "d", a), tol=tol_of(case), scale=max(np.abs(rx[a]).max(), 1.0))
    print(f"{name}: worst relative distance of x / u to the reference's float32 rollouts {worst:.2e}")
    s.close()
