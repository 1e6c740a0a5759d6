"""Solver-level cross-checks of the oracle against the numbers recorded in SURVEY.md section 8(c) (tests/golden/survey_kat.json).

STATUS: NOT a pin of the reference.  Those numbers came from a build of the reference's host code against stand-in CUDA headers, and carry two
artefacts of that build (integer min/max in the rho schedule, sin/cos bound to the double libm functions); the tests below therefore run the oracle
in an EMULATION of that build (`survey_int_minmax`, `survey_double_trig`), a mode no parity test uses.  They guard the restatement of the driver
logic (line search, accept/reject, rho schedule, multiple-shooting bookkeeping) against regressions -- nothing more.  The pins that count come from
data the reference itself holds: tests/test_urdf_pins.py (iiwa14.urdf, the example's gravity-balancing torques, printDyn / testDynGrad states),
tests/test_closed_form_pins.py (the reference's own pendulum / cart-pole / quadrotor formulas), tests/test_fig8_pins.py (the recorded figure-eight
run) and tests/test_wire_format.py (lcm-gen hashes).  CPU-only.
"""
import json
import os

import numpy as np
import pytest

from oracle_binding import Oracle, default_cfg, example_inputs

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_kat.json")))


def f32(v):
    return np.float32(v)


def kuka_cfg(M, **kw):
    k = KAT["kuka_joint_cost"]
    return default_cfg(4, N=k["N"], M=M, A=k["A"], wafr_urdf=1, tol_cost=0.0, total_time=k["total_time"],
                       alpha_base=k["alpha_base"], cores=k["cores"], spawn_threads=0, max_iter=30, **kw)


@pytest.mark.parametrize("M", [4, 1])
def test_runiLQR_CPU_J_trace_bit_exact(M):
    """G3: runiLQR_CPU J trace and alpha indices, bit-exact (printed with 6 decimals = unique float32)."""
    ref = KAT["G3_runiLQR_CPU"]["M%d" % M]
    o = Oracle(kuka_cfg(M, survey_int_minmax=1, survey_double_trig=1))
    r = o.run_ilqr_cpu(*example_inputs(4, 128))
    n = len(ref["J"])
    assert list(r["alphaOut"][:n]) == ref["alpha"]
    assert [f32(v) for v in r["Jout"][:n]] == [f32(v) for v in ref["J"]]
    if "J_iter30" in ref:
        assert f32(r["Jout"][30]) == f32(ref["J_iter30"])
    if "stalls_at" in ref:
        assert all(a == -1 for a in r["alphaOut"][n:31]) and f32(r["Jout"][30]) == f32(ref["stalls_at"])


@pytest.mark.parametrize("M", [4, 1])
def test_gpu_semantics_matches_G4_cross_check(M):
    """G4: GPU-semantics driver vs the survey's modified-CPU2 cross-check: identical alphas, J to 1e-6 rel."""
    ref = KAT["G4_near_gpu_semantics_runiLQR_CPU2_modified"]["M%d" % M]
    o = Oracle(kuka_cfg(M, survey_int_minmax=1, survey_double_trig=1))
    r = o.run_ilqr_gpusem(*example_inputs(4, 128))
    n = len(ref["J"])
    assert list(r["alphaOut"][:n]) == ref["alpha"]
    np.testing.assert_allclose(r["Jout"][:n], ref["J"], rtol=1e-6)
    np.testing.assert_allclose(r["Jout"][30], ref["J_iter30"], rtol=1e-6)


def test_plant_probes():
    p = KAT["plant_probes"]
    for plant, key in ((1, "pendulum_qdd"), (2, "cartpole_qdd"), (3, "quadrotor_qdd")):
        o = Oracle(default_cfg(plant))
        x = np.array([0.1 * (i + 1) for i in range(o.n)], np.float32)
        u = np.array([0.5 + i for i in range(o.m)], np.float32)
        np.testing.assert_allclose(o.dynamics(x, u), p[key], rtol=6e-7)  # printed with %.7g


def test_backprop_regularisation_block_structure():
    """Appendix E probe: rho lands in the x-row/u-col block of H, not in the u-row/x-col block K reads."""
    n, m, N = 14, 7, 8
    nm = n + m
    o = Oracle(default_cfg(4, N=N, M=1, A=1))
    AB = np.zeros((N, nm, n), np.float32)
    for k in range(N):
        AB[k, :n, :] = np.eye(n)
        for j in range(m):
            AB[k, n + j, j] = 1
    H = np.zeros((N, nm, nm), np.float32)
    H[N - 1, :n, :n] = 2 * np.eye(n)
    H[N - 1, 0, 1] = H[N - 1, 1, 0] = 0.25
    g = np.zeros((N, nm), np.float32)
    z = lambda *s: np.zeros(s, np.float32)
    P, p, Pp, pp = z(N, n, n), z(N, n), z(N, n, n), z(N, n)
    KT, du, d, ApBK, Bdu, x = z(N, m, n), z(N, m), z(N, n), z(N, n, n), z(N, n), z(N, n)
    o.backward_pass(0, AB, P, p, Pp, pp, H, g, KT, du, d, ApBK, Bdu, x, x.copy(), 5.0)
    Hk = H[N - 2]   # [col][row], host semantics accumulate in place
    b = KAT["backprop_probe"]
    assert Hk[0, 0] == b["H_x0_x0"] and Hk[n, n] == b["H_u0_u0"]
    assert Hk[n, 0] == b["H_rowx0_colu0"] and Hk[0, n] == b["H_rowu0_colx0"]
