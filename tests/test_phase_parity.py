"""Teacher-forced phase parity: every phase of a DDP sweep is fed the same inputs in the kernels and in
the oracle's GPU-semantics restatement and its outputs are compared.  Integer outputs (knot / defect bookkeeping, err
flags, alpha index) must be identical; float64 to 1e-9; float32 norm-wise to 1e-4 (north_star) -- the backward pass
is compared after ONE pass from identical inputs, so no iteration-to-iteration amplification enters.
"""
import numpy as np
import pytest

import pyddp
from backends import BACKENDS, make_solver
from oracle_binding import Oracle, default_cfg, example_inputs

CASES = [  # plant, N, M, A, integrator
    pytest.param(4, 32, 4, 4, 1, id="kuka-N32-M4"),
    pytest.param(4, 32, 1, 6, 1, id="kuka-N32-M1"),
    pytest.param(2, 32, 4, 8, 3, id="cart-N32-M4-rk3"),
    pytest.param(3, 32, 2, 4, 3, id="quad-N32-M2-rk3"),
    pytest.param(1, 16, 4, 1, 1, id="pend-N16-M4-euler"),
]


def nrel(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64).ravel() - ref.ravel()).max() / max(np.abs(ref).max(), 1e-30))


def setup(backend, plant, N, M, A, integ, dtype, seed=1):
    rng = np.random.default_rng(seed)
    kw = dict(N=N, M=M, A=A, integrator=integ, wafr_urdf=1, total_time=0.5 if plant == 4 else 1.0)
    s = make_solver(backend, plant, dtype=0 if dtype == np.float32 else 1, **kw)
    o = Oracle(default_cfg(plant, **kw), dtype)
    n, m = o.n, o.m
    x, u, xg = example_inputs(plant, N, dtype)
    x = (x.reshape(N, n) + rng.normal(0, 0.01, (N, n))).astype(dtype).ravel()      # a trajectory whose knots all differ
    u = (u.reshape(N, m) * (1 + rng.normal(0, 0.01, (N, m)))).astype(dtype).ravel()
    return s, o, rng, x, u, xg


def kw_of(s):
    """the oracle configuration of a solver handle's problem (setup() builds both from the same keywords)"""
    c = s.cfg
    return dict(N=c.N, M=c.M, A=c.A, integrator=c.integrator, wafr_urdf=c.wafr_urdf, total_time=c.total_time)


def tol_for(dtype):
    return 1e-9 if dtype == np.float64 else 1e-4


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("plant,N,M,A,integ", CASES)
def test_sweep_phases_teacher_forced(backend, dtype, plant, N, M, A, integ):
    s, o, rng, x, u, xg = setup(backend, plant, N, M, A, integ, dtype)
    n, m, nm = o.n, o.m, o.n + o.m
    tol = tol_for(dtype)
    z = lambda *shape: np.zeros(shape, dtype)

    # ---- load + init: derivatives and initial cost (initAlgGPU)
    s.load(x, u, xg)
    AB, H, g = o.next_iteration_setup(x, u, xg)
    assert nrel(s.get("AB")[: (N - 1) * n * nm], AB[: (N - 1) * n * nm]) <= tol
    Hk, Ho = s.get("H").reshape(N, nm, nm), H.reshape(N, nm, nm)
    assert nrel(Hk[: N - 1], Ho[: N - 1]) <= tol and nrel(Hk[N - 1, :n, :n], Ho[N - 1, :n, :n]) <= tol
    assert nrel(s.get("g"), g) <= tol
    st = s.get_state()
    J0 = o.total_cost(1, x, u, xg)
    assert abs(st[0].prevJ - (J0 + 2 * s.cfg.tol_cost)) <= tol * abs(J0)
    assert st[0].iter == 1 and st[0].done == 0 and s.get("alphaOut")[0] == -1

    # ---- backward pass from identical inputs (boundary cost-to-go, defects, shifted trajectory, rho)
    NB = N // M
    Pp, pp, d = z(N, n, n), z(N, n), z(N, n)
    xprev = (x.reshape(N, n) + rng.normal(0, 0.005, (N, n))).astype(dtype)
    for b in range(M - 1):
        k = NB * (b + 1) - 1
        Q = rng.normal(0, 1, (n, n)); Pp[k] = (Q @ Q.T / n + np.eye(n)) * 10; pp[k] = rng.normal(0, 1, n)
        d[k] = rng.normal(0, 0.01, n)
    rho = 3.5
    Hin = s.get("H").copy()          # use the kernel-side derivatives for both (final knot's u-block is defined there)
    ABin, gin = s.get("AB").copy(), s.get("g").copy()
    s.set("Pp", Pp); s.set("pp", pp); s.set("dcur", d)
    xb = np.stack([x.reshape(N, n), xprev])      # half 0 = current trajectory, half 1 = trajectory of Pp/pp
    s.set("xb", xb)
    st[0].rho = rho; st[0].cur = 0; st[0].cur2 = 1
    s.set_state(st)
    s.run_phase(pyddp.PHASE_BP)
    P, p, KT, du, ApBK, Bdu = z(N, n, n), z(N, n), z(N, m, n), z(N, m), z(N, n, n), z(N, n)
    fail, dJexp, err = o.backward_pass(1, ABin, P, p, Pp.copy(), pp.copy(), Hin.copy(), gin.copy(), KT, du, d, ApBK, Bdu,
                                       np.ascontiguousarray(x), np.ascontiguousarray(xprev), rho)
    assert list(s.get("err")) == list(err) and fail == 0
    if plant == 4 and dtype == np.float32:
        # The arm's float32 backward pass may run on the matrix cores (bp_mfma.hpp), which sum in another order than oracle32: it is held against
        # the float32 noise floor of the reference algorithm measured against oracle64 on the same inputs (test_fp32_bar.bp_noise_floor).
        from test_fp32_bar import bp_noise_floor, bp_quantities, bar
        q = dict(AB=ABin, Pp=Pp, pp=pp, H=Hin, g=gin, d=d, x=x, xp2=xprev)
        o64 = Oracle(default_cfg(plant, **kw_of(s)), np.float64)
        q64 = {k_: np.ascontiguousarray(v, np.float64).ravel() for k_, v in q.items()}
        z64 = lambda *sh: np.zeros(sh, np.float64)
        r = dict(P=z64(N * n * n), p=z64(N * n), KT=z64(N * n * m), du=z64(N * m), ApBK=z64(N * n * n), Bdu=z64(N * n))
        _, r["dJexp"], _ = o64.backward_pass(1, q64["AB"], r["P"], r["p"], q64["Pp"].copy(), q64["pp"].copy(), q64["H"].copy(), q64["g"].copy(), r["KT"], r["du"],
                                             q64["d"], r["ApBK"], r["Bdu"], q64["x"], q64["xp2"], rho)
        o32f = Oracle(default_cfg(plant, **kw_of(s)), np.float32, variant="fma")
        floor, _, _ = bp_noise_floor(o, o32f, q, rho, r, 7, n, N, M)
        kern = {name: s.get(name) for name in ("KT", "du", "P", "p", "dJexp", "ApBK", "Bdu")}
        for name, v, ref in bp_quantities(kern, r, n, N, M):
            assert bar(nrel(v, ref), floor[name]), (name, nrel(v, ref), floor[name])
    else:
        for name, ref in (("KT", KT), ("du", du), ("P", P), ("p", p), ("dJexp", dJexp)):
            assert nrel(s.get(name), ref) <= tol, name
        if M > 1:
            assert nrel(s.get("ApBK")[: (N - 1) * n * n], ApBK.ravel()[: (N - 1) * n * n]) <= tol
            assert nrel(s.get("Bdu")[: (N - 1) * n], Bdu.ravel()[: (N - 1) * n]) <= tol

    # ---- forward pass: sweep + rollout + cost + defect for every alpha, from the oracle's gains
    s.set("KT", KT); s.set("du", du); s.set("ApBK", ApBK); s.set("Bdu", Bdu)
    s.set("ucur", u)
    s.run_phase(pyddp.PHASE_FP)
    xs, us, ds = s.get("xs").reshape(A, N, n), s.get("us").reshape(A, N, m), s.get("ds").reshape(A, N, n)
    Jk, dk = s.get("J"), s.get("dmax")
    alphas = s.get("alpha")
    bnd = [k for k in range(N) if ((k + 1) % NB == 0) and k < N - 1]
    Jo, do, n_div, tols = [], [], 0, []
    for a in range(A):
        xa, ua, da = x.copy(), u.copy(), d.copy().ravel()
        if M > 1:
            o.forward_sweep(xa, ApBK, Bdu, d, np.ascontiguousarray(x), alphas[a])
        o.forward_sim(xa, ua, KT, du, da, alphas[a], np.ascontiguousarray(x))
        if not (np.isfinite(xa).all() and np.abs(xa).max() < 1e3):     # a diverging candidate: both must diverge
            assert not (np.isfinite(xs[a]).all() and np.abs(xs[a]).max() < 1e3)
            Jo.append(np.inf); do.append(np.inf); n_div += 1; tols.append(np.inf)
            continue
        # candidates whose rollout runs away (|qd| > 4 rad/s) amplify one-ulp differences of the dynamics: looser bar there
        tol_a = tol if (dtype == np.float64 or np.abs(xa.reshape(N, n)[:, o.npos:]).max() < (4.0 if plant == 4 else 1e3)) else 100 * tol
        tols.append(tol_a)
        assert nrel(xs[a], xa) <= tol_a and nrel(us[a], ua) <= tol_a
        if bnd:
            assert nrel(ds[a][bnd], da.reshape(N, n)[bnd]) <= max(tol, 1e-3 if dtype == np.float32 else 0), "defects are differences of nearby states"
        Jo.append(o.total_cost(1, xa, ua, xg)); do.append(o.max_defect(1, da))
    assert n_div < A, "test inputs must leave at least one stable candidate"
    ok = np.isfinite(Jo)
    for a in range(A):
        if ok[a]:
            assert abs(Jk[a] - Jo[a]) <= tols[a] * abs(Jo[a]), (a, Jk[a], Jo[a])
    assert min(tols) == tol, "test inputs must leave at least one well-behaved candidate"
    if M > 1:
        np.testing.assert_allclose(dk[ok], np.asarray(do)[ok], rtol=max(tol, 1e-3 if dtype == np.float32 else 0), atol=1e-6)
    Jo = [j if np.isfinite(j) else 1e30 for j in Jo]; do = [v if np.isfinite(v) else 1e30 for v in do]

    # ---- line search + accept/reject decision from identical tables
    st = s.get_state()
    prevJ = float(np.max([j for j in Jo if j < 1e29])) * 1.0001
    st[0].prevJ = prevJ; st[0].ignore_defect = 1; st[0].alphaIndex = 0
    s.set_state(st)
    s.set("J", np.asarray(Jo, dtype)); s.set("dmax", np.asarray(do, dtype)); s.set("dJexp", dJexp)
    dsum = dJexp.copy()
    for i in range(1, M):
        dsum[0] += dJexp[2 * i]; dsum[1] += dJexp[2 * i + 1]
    ai, ign, dJ, zz = o.line_search_gpu(Jo, do, dsum, np.asarray(prevJ, dtype), 1, 0)
    s.run_phase(pyddp.PHASE_LS)
    st = s.get_state()
    if dJ < 0:
        assert st[0].accepted == 0 and s.get("alphaOut")[1] == -1
    else:
        assert st[0].accepted == 1 and st[0].alphaIndex == ai and s.get("alphaOut")[1] == ai and st[0].ignore_defect == ign
        assert abs(st[0].z - zz) <= 1e-5 * max(1, abs(zz))

    # ---- next-iteration setup: winner -> current trajectory, new derivatives there
    s.run_phase(pyddp.PHASE_NIS)
    if st[0].accepted == 1:
        w = st[0].alphaIndex
        cur = st[0].cur
        assert np.array_equal(s.get("xb").reshape(2, N, n)[cur], xs[w]) and np.array_equal(s.get("ucur").reshape(N, m), us[w])
        AB2, H2, g2 = o.next_iteration_setup(np.ascontiguousarray(xs[w]).ravel(), np.ascontiguousarray(us[w]).ravel(), xg)
        assert nrel(s.get("AB")[: (N - 1) * n * nm], AB2[: (N - 1) * n * nm]) <= max(tol, 2e-4 if dtype == np.float32 else 0)
        assert nrel(s.get("g"), g2) <= tol
    # Pp <- P, pp <- p of nextIterationSetupGPU (:266-267) is a flip of the double buffer: the next backward pass reads what this one wrote
    assert s.get_state()[0].pw == 1


@pytest.mark.parametrize("backend", BACKENDS)
def test_backward_pass_reports_non_positive_huu(backend):
    """m = 1 (cart-pole): a non-positive Huu must raise the err flag, bump rho and leave the iteration count alone
    (computeKTdu_dim1, bpHelpers.cuh:101; backwardPassGPU retry, :497-511)."""
    s, o, rng, x, u, xg = setup(backend, 2, 16, 2, 2, 1, np.float32)
    s.load(x, u, xg)
    H = s.get("H").reshape(16, 5, 5)
    H[:, 4, 4] = -1e6              # Huu_cost hugely negative at every knot
    s.set("H", H)
    rho0 = s.get_state()[0].rho
    s.iterate(1)
    st = s.get_state()
    assert s.get("err").any() and st[0].accepted == -1 and st[0].iter == 1 and st[0].rho > rho0 and st[0].bp_retries == 1
