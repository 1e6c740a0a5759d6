"""The kernel family bench.py runs -- matrix-core backward pass with the sweep maps composed in it (k_bp_mfma), k_sweep_maps, thread-lane rollouts
(k_fp_tl) and setup (k_nis_tl) with the compact [A B] and the knot-major candidate states -- instantiated in FLOAT64 and held against the oracle at
tight tolerance (VERDICT r2 "missing #1").

In float32 those kernels can only be held against a bar as wide as float32's own error on this problem (tests/test_fp32_bar.py: 1e-2 .. 1e-1 in the
gains from iteration ~10 on), which could not catch a small structural error.  The same source -- one template over the element type: the tile algebra of
bp_mfma.hpp on v_mfma_f64_16x16x4_f64 instead of v_mfma_f32_16x16x4_f32 (Mx<T>), the thread-lane bodies of fp_tl.hpp / plant_arm_tl.hpp, the staging and
the compact [A B] of k_nis_tl -- compiled for double (kernels bp=mx, fp=tl on a dtype-1 handle) has to follow the oracle's GPU-semantics driver decision
for decision over 40 iterations at the headline size, and reproduce every phase's outputs to 1e-9 under teacher forcing, the fused sweep's segment start
states against oracle.forward_sweep included.  Reference path: bpHelpers.cuh:339-420, fpHelpers.cuh:19-63, 225-301, nisInitHelpers.cuh:205-279.
"""
import os

import numpy as np
import pytest

import pyddp
from backends import make_solver
from gpusem_steps import gpusem_iterations
from oracle_binding import Oracle, default_cfg, example_inputs

pytestmark = pytest.mark.gpu
FAMILY = dict(bp="mx", fp="tl")
# The two float selections the library makes for the arm, each in its float64 (parity) instantiation:
#   large batch (what bench.py's headline runs): k_bp_mfma + k_sweep_maps + k_fp_tl + k_nis_tl (compact [A B], knot-major candidate records)
#   one problem (what the latency / MPC figures run; VERDICT r3 "missing" 2): k_bp_mfma + k_fp_tl4 (the four-wave rollout pipeline, fp_pipe.hpp; it begins with the sweep over the maps) + k_nis_tl7
#   (thread = (knot, joint)).  Reference path of the latter two: fpHelpers.cuh:225-301, nisInitHelpers.cuh:205-279.
FAMILIES = {"large-batch": (FAMILY, ("k_bp_mfma", "k_fp_tl", "k_nis_tl")), "one-problem": (dict(bp="mx", fp="tl4"), ("k_bp_mfma", "k_fp_tl4", "k_nis_tl7"))}


def nrel(a, ref):
    ref = np.asarray(ref, np.float64).ravel()
    a = np.asarray(a, np.float64).ravel()
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300))


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("M", [4, 1])
def test_kuka_float64_headline_size_whole_solve_on_the_benched_family(M, family):
    """BASELINE configs[2] at full size (N=128, A=8), float64, 40 iterations on the matrix-core / thread-lane family: identical step-size indices
    (rejections included) and J / x / u to 1e-7, K to 1e-6 against the oracle's GPU-semantics driver."""
    kw = dict(N=128, M=M, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=40)
    noise = np.random.default_rng(7).normal(0, 0.001, (128, 14))
    x0, u0, xg = example_inputs(4, 128, np.float64, noise=noise)
    r = Oracle(default_cfg(4, cores=8, spawn_threads=0, **kw), np.float64).run_ilqr_gpusem(x0, u0, xg)
    s = make_solver("hip", 4, dtype=1, kernels=FAMILIES[family][0], **kw)
    out = s.solve(x0, u0, xg)
    names = dict(s.time_kernels(1))
    assert all(k in names for k in FAMILIES[family][1]) and ("k_sweep_maps" in names) == (M > 1 and family == "large-batch"), names
    it = r["iters"]
    assert out["iters"][0] == it == 40
    assert list(out["alphaOut"][0][: it + 1]) == list(r["alphaOut"][: it + 1])
    np.testing.assert_allclose(out["Jout"][0][: it + 1], r["Jout"][: it + 1], rtol=1e-7)
    np.testing.assert_allclose(out["x"][0].ravel(), r["x"], rtol=0, atol=1e-7 * np.abs(r["x"]).max())
    np.testing.assert_allclose(out["u"][0].ravel(), r["u"], rtol=0, atol=1e-7 * np.abs(r["u"]).max())
    np.testing.assert_allclose(out["KT"][0].ravel(), r["KT"], rtol=0, atol=1e-6 * np.abs(r["KT"]).max())
    s.close()


def test_lean_cost_to_go_option_changes_nothing_the_solver_returns():
    """pddp_config.boundary_cost_to_go_only: the interior cost-to-go slots are not an input of any later phase of runiLQR_GPU -- same bits in every output
    with and without them; without the option (the default) the arrays P, p hold every knot's cost-to-go like the reference's d_P, d_p, and a warm-started
    MPC call is refused on a handle that iterated with it."""
    kw = dict(N=64, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=12)
    x0, u0, xg = example_inputs(4, 64, np.float32, noise=np.random.default_rng(3).normal(0, 0.001, (64, 14)))
    outs = []
    for lean in (0, 1):
        s = make_solver("hip", 4, dtype=0, boundary_cost_to_go_only=lean, kernels=FAMILY, **kw)
        outs.append((s.solve(x0, u0, xg), s.get_cost_to_go()[0].reshape(64, 14, 14)))
        if lean:
            with pytest.raises(pyddp.PddpError):
                s.mpc_solve(x0.reshape(64, 14)[1], xg, 1, clear_vars=0, max_iter=2)
            s.mpc_solve(x0.reshape(64, 14)[1], xg, 1, clear_vars=1, max_iter=2)       # a cold start is fine, and from then on every slot is kept
            s.mpc_solve(x0.reshape(64, 14)[1], xg, 1, clear_vars=0, max_iter=2)
        s.close()
    (a, Pa), (b, Pb) = outs
    for k in ("Jout", "x", "u", "KT", "alphaOut"):
        assert np.array_equal(a[k], b[k]), k
    interior = [k for k in range(63) if (k + 1) % 16 != 0]
    assert np.abs(Pa[interior]).max(axis=(1, 2)).min() > 0          # default: every knot's cost-to-go is in the array
    assert not Pb[interior].any()                                    # lean: only the block-boundary slots were ever written
    assert np.array_equal(Pa[[15, 31, 47]], Pb[[15, 31, 47]])


KUKA = dict(N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5)


@pytest.mark.parametrize("kw,iterations", [pytest.param(KUKA, 24, id="headline-M4"), pytest.param({**KUKA, "M": 1}, 10, id="single-shooting"),
                                            pytest.param({**KUKA, "N": 64, "M": 2, "A": 16}, 10, id="N64-M2-A16")])
@pytest.mark.parametrize("family", list(FAMILIES))
def test_every_phase_of_the_benched_family_in_float64_teacher_forced(kw, iterations, family):
    """Every iteration of a real float64 solve of the oracle, every phase given the oracle's inputs of that phase in ONE handle (slot b = iteration b):
    setup (AB, g), backward pass (K, du, P, p, expected reduction; A - B K / B du through the separate-sweep hook), the FUSED production sweep (segment
    maps composed inside the backward pass + k_sweep_maps: every candidate's segment start states against oracle.forward_sweep), rollouts of every
    candidate (x, u, J, boundary defects), line search (integers).  Bar: 1e-9 of the quantity's size."""
    TOL = 1e-9
    o = Oracle(default_cfg(4, cores=1, spawn_threads=0, **kw), np.float64)
    n, m, N, M, A = o.n, o.m, kw["N"], kw["M"], kw["A"]
    nm, NB = n + m, N // M
    x0, u0, xg = example_inputs(4, N, np.float64, noise=np.random.default_rng(21).normal(0, 0.001, (N, n)))
    with np.errstate(all="ignore"):
        recs = list(gpusem_iterations(o, x0, u0, xg, iterations))
    B = len(recs)
    s = make_solver("hip", 4, dtype=1, batch=B, kernels=FAMILIES[family][0], **kw)
    s.load(np.tile(x0, B), np.tile(u0, B), np.tile(xg, B))
    stack = lambda key: np.stack([np.asarray(r[key]).ravel() for r in recs])
    worst = {}

    def check(name, got, ref, scale=None):
        ref = np.asarray(ref, np.float64).ravel(); got = np.asarray(got, np.float64).ravel()
        e = float(np.abs(got - ref).max() / (scale if scale is not None else max(np.abs(ref).max(), 1e-300)))
        worst[name] = max(worst.get(name, 0.0), e)

    def set_states(fn):
        st = s.get_state()
        for b_ in range(B):
            fn(st[b_], recs[b_])
        s.set_state(st)

    bnd = [k for k in range(N) if ((k + 1) % NB == 0) and k < N - 1]

    def st_common(st, rec):
        st.cur = 0; st.cur2 = 1; st.pw = 0; st.rho = rec.rho; st.drho = rec.drho; st.done = 0; st.accepted = 0; st.iter = rec.iter
    set_states(st_common)
    s.set("xb", np.concatenate([stack("x").reshape(B, 1, N * n), stack("xp2").reshape(B, 1, N * n)], axis=1))
    s.set("ucur", stack("u")); s.set("dcur", stack("d"))
    # ---- setup (k_nis_tl: thread = knot, Jacobian staged through LDS, compact [A B]; read back through the reference-layout view)
    s.run_phase(pyddp.PHASE_INIT_NIS)
    ABk, gk = s.get("AB").reshape(B, -1), s.get("g").reshape(B, -1)
    nAB = (N - 1) * n * nm
    for i, rec in enumerate(recs):
        check("nis.AB", ABk[i][:nAB], rec.AB[:nAB]); check("nis.g", gk[i], rec.g)
    # ---- backward pass from the oracle's inputs: once through the separate-sweep hook (writes A - B K, B du), once fused (composes the segment maps)
    for name in ("AB", "g", "Pp", "pp"):
        s.set(name, stack(name))
    for fused in (False, True) if M > 1 else (False,):
        s.run_phase(pyddp.PHASE_BP_FUSED if fused else pyddp.PHASE_BP)
        out = {name: s.get(name).reshape(B, -1) for name in ("KT", "du", "P", "p", "dJexp", "ApBK", "Bdu")}
        err = s.get("err").reshape(B, M)
        tag = "bp_fused." if fused else "bp."
        for i, rec in enumerate(recs):
            assert list(err[i]) == list(rec.err)
            for name in ("KT", "du", "P", "p"):
                cnt = {"KT": (N - 1) * n * m, "du": (N - 1) * m, "P": (N - 1) * n * n, "p": (N - 1) * n}[name]
                check(tag + name, out[name][i][:cnt], np.asarray(rec[name]).ravel()[:cnt])
            check(tag + "dJexp", [out["dJexp"][i][0::2].sum(), out["dJexp"][i][1::2].sum()], [rec.dJexp[0::2].sum(), rec.dJexp[1::2].sum()])
            if M > 1 and not fused:
                check("bp.ApBK", out["ApBK"][i][: (N - 1) * n * n], rec.ApBK[: (N - 1) * n * n]); check("bp.Bdu", out["Bdu"][i][: (N - 1) * n], rec.Bdu[: (N - 1) * n])
    # ---- the fused forward sweep: k_sweep_maps from the maps of the fused backward pass above -> segment start states of every candidate
    if M > 1:
        s.set("xs", np.zeros(B * A * N * n))
        s.run_phase(pyddp.PHASE_SWEEP_FUSED)
        xs = s.get("xs").reshape(B, A, N, n)
        starts = [k + 1 for k in bnd]
        for i, rec in enumerate(recs):
            for a in range(A):
                xa = rec.x.copy()
                with np.errstate(all="ignore"):
                    o.forward_sweep(xa, rec.ApBK, rec.Bdu, rec.d, rec.x, rec.alphas[a])
                ref = xa.reshape(N, n)[starts]
                if np.isfinite(ref).all():
                    check("sweep_fused.x_start", xs[i][a][starts], ref)
    # ---- rollouts of every candidate from the oracle's gains (the separate sweep kernel supplies the start states here: its operands are teacher-forced)
    for name in ("KT", "du", "ApBK", "Bdu"):
        s.set(name, stack(name))
    s.run_phase(pyddp.PHASE_FP)
    xs, us, ds = s.get("xs").reshape(B, A, N, n), s.get("us").reshape(B, A, N, m), s.get("ds").reshape(B, A, N, n)
    Jk = s.get("J").reshape(B, A)
    in_play = 0
    for i, rec in enumerate(recs):
        for a in range(A):
            ref_x = rec.xs[a]
            if not (np.isfinite(ref_x).all() and rec.J[a] <= 1.5 * rec.prevJ):          # a runaway candidate only has to be rejected by everybody
                assert not (Jk[i][a] <= rec.prevJ)
                continue
            in_play += 1
            check("fp.x", xs[i][a], ref_x); check("fp.u", us[i][a], rec.us[a]); check("fp.J", Jk[i][a], rec.J[a])
            if bnd:
                check("fp.d", ds[i][a][bnd], rec.ds[a].reshape(N, n)[bnd], scale=np.abs(ref_x).max())
    assert in_play >= B
    # ---- line search + accept / reject from the oracle's cost tables: integers
    def st_ls(st, rec):
        st.prevJ = rec.prevJ; st.ignore_defect = rec.ignore_defect; st.alphaIndex = 0
    set_states(st_ls)
    s.set("J", stack("J")); s.set("dmax", stack("dmax")); s.set("dJexp", stack("dJexp"))
    s.run_phase(pyddp.PHASE_LS)
    st = s.get_state()
    for b_, rec in enumerate(recs):
        if rec.accepted:
            assert st[b_].accepted == 1 and st[b_].alphaIndex == rec.ls_alpha and st[b_].ignore_defect == rec.ls_ignore_defect
        else:
            assert st[b_].accepted == 0
        assert abs(st[b_].rho - rec.rho_next) <= 1e-12 * rec.rho_next
    s.close()
    print("float64 teacher-forced, worst error per quantity:", {k: f"{v:.2e}" for k, v in sorted(worst.items())})
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, bad


def test_mx_tile_layout_of_both_element_types_agree_on_full_hessian_inputs():
    """The full-Hessian instantiation (what the end-effector cost runs) in float64 against the oracle on a dense cost Hessian handed in through the phase hook."""
    kw = dict(N=32, M=2, A=4, wafr_urdf=1, tol_cost=0.0, total_time=0.125)
    o = Oracle(default_cfg(4, cores=1, spawn_threads=0, **kw), np.float64)
    n, m, N, M = o.n, o.m, 32, 2
    nm = n + m
    rng = np.random.default_rng(5)
    x0, u0, xg = example_inputs(4, N, np.float64, noise=rng.normal(0, 0.01, (N, n)))
    with np.errstate(all="ignore"):
        rec = next(iter(gpusem_iterations(o, x0, u0, xg, 2)))
    # a dense symmetric positive definite cost Hessian per knot and a nonzero boundary cost-to-go
    H = np.zeros((N, nm, nm))
    for k in range(N):
        Q = rng.normal(0, 0.3, (nm, nm)); H[k] = Q @ Q.T + np.diag(np.abs(rng.normal(1, 0.2, nm)))
    Pp = np.zeros((N, n, n)); pp = rng.normal(0, 0.1, (N, n))
    for k in range(N):
        Q = rng.normal(0, 0.2, (n, n)); Pp[k] = Q @ Q.T + np.eye(n)
    d = np.zeros((N, n)); d[N // M - 1] = rng.normal(0, 0.01, n)
    z = lambda *sh: np.zeros(sh)
    P, p, KT, du, ApBK, Bdu = z(N * n * n), z(N * n), z(N * n * m), z(N * m), z(N * n * n), z(N * n)
    fail, dJexp, err = o.backward_pass(1, rec.AB, P, p, Pp.ravel().copy(), pp.ravel().copy(), H.ravel().copy(), rec.g.copy(), KT, du, d.ravel(), ApBK, Bdu, rec.x, rec.xp2, 3.0)
    s = make_solver("hip", 4, dtype=1, kernels=dict(bp="mx", fp="lg"), **kw)
    s.load(x0, u0, xg)
    st = s.get_state(); st[0].cur = 0; st[0].cur2 = 1; st[0].pw = 0; st[0].rho = 3.0; st[0].done = 0; s.set_state(st)
    s.set("xb", np.concatenate([rec.x, rec.xp2])); s.set("dcur", d); s.set("H", H); s.set("AB", rec.AB); s.set("g", rec.g); s.set("Pp", Pp); s.set("pp", pp)
    s.run_phase(pyddp.PHASE_BP)
    for name, ref, cnt in (("KT", KT, (N - 1) * n * m), ("du", du, (N - 1) * m), ("P", P, (N - 1) * n * n), ("p", p, (N - 1) * n), ("ApBK", ApBK, (N - 1) * n * n), ("Bdu", Bdu, (N - 1) * n)):
        assert nrel(s.get(name)[:cnt], ref[:cnt]) <= 1e-10, name
    got = s.get("dJexp")
    np.testing.assert_allclose([got[0::2].sum(), got[1::2].sum()], [dJexp[0::2].sum(), dJexp[1::2].sum()], rtol=1e-10)
    s.close()
