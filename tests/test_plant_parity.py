"""Plant-level parity: dynamics, dynamicsGradient, _integrator, _integratorGradient of the kernels vs
the oracle, on the reference's own test distribution (test/testDynGrad.cu:12-19: q~N(0,2), qd~N(0,5), u~N(0,50)),
on near-trajectory states, and on the axis-aligned poses of test/printDyn.cu:39-49.

Tolerances: results are compared norm-wise (max abs error / max abs reference value).  float64 must agree to 1e-9.
float32: the bar is 1e-4 (BASELINE.json north_star) unless the reference algorithm's own float32 noise is larger:
the float32 oracle is only accurate to ~1-2e-4 against its own float64 instantiation for dqdd (mass-matrix
conditioning amplifies the rounding of its dense tensor chain), so the bar is max(1e-4, 2 x that measured self-error).
The kernels' error against float64 is smaller than the oracle's ("gradient accuracy").
"""
import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import Oracle, default_cfg

RNG = np.random.default_rng(20260929)
PLANTS = {1: "pend", 2: "cart", 3: "quad", 4: "arm"}


def states(plant, kind, count, dtype):
    npos, n, m = {1: (1, 2, 1), 2: (2, 4, 1), 3: (6, 12, 4), 4: (7, 14, 7)}[plant]
    if kind == "extreme":
        x = np.concatenate([RNG.normal(0, 2, (count, npos)), RNG.normal(0, 5, (count, npos))], axis=1)
        u = RNG.normal(0, 50, (count, m))
    elif kind == "traj":
        x = np.concatenate([RNG.normal(0, 1.0, (count, npos)), RNG.normal(0, 0.5, (count, npos))], axis=1)
        u = RNG.normal(0, 20, (count, m))
    else:  # axis-aligned poses (test/printDyn.cu:39-49): one joint at +-pi/2, zero velocity
        x = np.zeros((2 * npos + 1, n)); u = np.zeros((2 * npos + 1, m))
        for i in range(npos):
            x[2 * i + 1, i] = np.pi / 2; x[2 * i + 2, i] = -np.pi / 2
    if plant == 3:
        x[:, 4] = np.clip(x[:, 4], -1.2, 1.2)   # keep cos(pitch) away from 0 (1/cos in the quadrotor model)
    return x.astype(dtype), u.astype(dtype)


def nrel(a, ref):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(ref, np.float64)).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("plant,integ", [(4, 1), (2, 1), (2, 2), (2, 3), (3, 3), (3, 1), (1, 1), (1, 3)])
def test_plant_functions_match_oracle(backend, dtype, plant, integ):
    s = make_solver(backend, plant, dtype=0 if dtype == np.float32 else 1, integrator=integ, N=16, M=1, A=1, wafr_urdf=1)
    cfg = default_cfg(plant, integrator=integ, N=16, M=1, A=1, wafr_urdf=1)
    o, o64 = Oracle(cfg, dtype), Oracle(cfg, np.float64)
    for kind, count in (("traj", 40), ("extreme", 40), ("axis", 0)):
        x, u = states(plant, kind, count, dtype)
        got = [s.plant_eval(w, x, u) for w in range(4)]
        worst = [0.0] * 4
        self_err = [0.0] * 4
        for i in range(x.shape[0]):
            dq, qdd = o.dynamics_gradient(x[i], u[i])
            ref = [qdd, dq, o.integrator(x[i], u[i]), o.integrator_gradient(x[i], u[i])]
            dq64, qdd64 = o64.dynamics_gradient(x[i], u[i])
            ref64 = [qdd64, dq64, o64.integrator(x[i], u[i]), o64.integrator_gradient(x[i], u[i])]
            for w in range(4):
                worst[w] = max(worst[w], nrel(got[w][i], ref[w]))
                self_err[w] = max(self_err[w], nrel(ref[w], ref64[w]))
        for w, name in enumerate(("dynamics", "dynamicsGradient", "integrator", "integratorGradient")):
            tol = 1e-9 if dtype == np.float64 else max(1e-4, 2 * self_err[w])
            assert worst[w] <= tol, f"{PLANTS[plant]} {name} [{kind}] rel err {worst[w]:.3g} > {tol:.3g} (oracle f32 self-error {self_err[w]:.3g})"


@pytest.mark.parametrize("plant", [1, 2, 3, 4])
def test_oracle_gradient_is_the_derivative_of_its_dynamics(plant):
    """The reference's own check (test/testDynGrad.cu): analytic gradient vs central differences, here in float64
    with a 1e-6 bar instead of the reference's 10 %.  Quadrotor: only the translational rows are checked -- the
    reference's hand-derived rotational Jacobian is inconsistent with its own dynamics (e.g. dynamics_quad.cuh:66 uses
    0.6125*diffU4*sin(x3) where the Jacobian :133,137 uses 6.125*sin(x3)); the oracle and the kernels restate both
    as they are, and parity with the oracle is what test_plant_functions_match_oracle asserts."""
    cfg = default_cfg(plant, wafr_urdf=1)
    o = Oracle(cfg, np.float64)
    x, u = states(plant, "traj", 5, np.float64)
    eps = 1e-6
    worst = 0.0
    for i in range(x.shape[0]):
        dq, _ = o.dynamics_gradient(x[i], u[i])
        dq = dq.reshape(o.n + o.m, o.npos).T
        fd = np.zeros_like(dq)
        for j in range(o.n + o.m):
            xp, xm, up, um = x[i].copy(), x[i].copy(), u[i].copy(), u[i].copy()
            if j < o.n:
                xp[j] += eps; xm[j] -= eps
            else:
                up[j - o.n] += eps; um[j - o.n] -= eps
            fd[:, j] = (o.dynamics(xp, up) - o.dynamics(xm, um)) / (2 * eps)
        rows = slice(0, 3) if plant == 3 else slice(None)
        worst = max(worst, nrel(dq[rows], fd[rows]))
    assert worst < 1e-6
