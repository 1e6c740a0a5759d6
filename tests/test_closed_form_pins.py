"""Pins of the pendulum / cart-pole / quadrotor plug-ins: tests/golden/closed_form_plants.json holds the reference's OWN dynamics / dynamicsGradient
statements (plants/dynamics_{pend,cart,quad}.cuh) executed in float64 on stored inputs by the committed tests/golden/make_closed_form_plants.py, the
diagonal cost weights of plants/cost_{pend,cart,quad}.cuh, and the hover thrust the example holds (examples/WAFR_iLQR_examples.cu:90).

  oracle float64  == fixture to 1e-12 (same formulas, same operation order up to the compiler)
  kernels float64 == fixture to 1e-12; float32 within  max(1e-4, 1.5 x err(oracle32, fixture))  relative to the largest output entry
"""
import json
import os

import numpy as np
import pytest

from backends import BACKENDS, make_solver
from oracle_binding import Oracle, default_cfg

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "closed_form_plants.json")))
PLANTS = {1: "pend", 2: "cart", 3: "quad"}


def rel(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("plant", [1, 2, 3])
def test_oracle_f64_equals_the_reference_formulas(plant):
    o = Oracle(default_cfg(plant), np.float64)
    for c in GOLD[PLANTS[plant]]["cases"]:
        assert rel(o.dynamics(c["x"], c["u"]), c["qdd"]) < 1e-12
        dq, qdd = o.dynamics_gradient(c["x"], c["u"])
        assert rel(dq, c["dqdd"]) < 1e-12 and rel(qdd, c["qdd"]) < 1e-12


def test_quadrotor_hover_thrust_held_by_the_example():
    c = GOLD["quad"]["cases"][-1]
    assert c["u"] == [1.22625] * 4
    assert np.abs(Oracle(default_cfg(3), np.float64).dynamics(c["x"], c["u"])).max() < 1e-12


@pytest.mark.parametrize("plant,N", [(1, 64), (2, 128), (3, 256)])
def test_oracle_cost_weights_are_the_reference_macros(plant, N):
    w = GOLD[PLANTS[plant]]["cost_weights"][str(N)]
    o = Oracle(default_cfg(plant, N=N), np.float64)
    n, m = o.n, o.m
    x, u, xg = np.arange(1, n + 1) * 0.25, np.arange(1, m + 1) * 0.5, np.arange(n) * 0.1
    for k, key in ((3, "running"), (N - 1, "final")):
        H, g = o.cost_grad(x, u, xg, k)
        np.testing.assert_allclose(np.diag(H.reshape(n + m, n + m)), w[key], rtol=1e-15)
        assert np.count_nonzero(H) == np.count_nonzero(w[key])
        np.testing.assert_allclose(g, np.asarray(w[key]) * np.concatenate([x - xg, u]), rtol=1e-14)
        cost = 0.5 * np.sum(np.asarray(w[key]) * np.concatenate([x - xg, u]) ** 2)
        assert o.cost_func(x, u, xg, k) == pytest.approx(cost, rel=1e-13)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("plant", [1, 2, 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kernels_match_the_reference_formulas(backend, plant, dtype):
    cases = GOLD[PLANTS[plant]]["cases"]
    s = make_solver(backend, plant, dtype=0 if dtype == np.float32 else 1, N=8, M=1, A=1)
    o32 = Oracle(default_cfg(plant), np.float32)
    x = np.array([c["x"] for c in cases], dtype); u = np.array([c["u"] for c in cases], dtype)
    qdd, dq = s.plant_eval(0, x, u), s.plant_eval(1, x, u)
    for i, c in enumerate(cases):
        for got, ref, ora in ((qdd[i], c["qdd"], lambda: o32.dynamics(x[i], u[i])), (dq[i], c["dqdd"], lambda: o32.dynamics_gradient(x[i], u[i])[0])):
            e = rel(got, ref)
            if dtype == np.float64:
                assert e < 1e-12, (plant, i, e)
            else:
                assert e <= max(1e-4, 1.5 * rel(ora(), ref)), (plant, i, e)
    s.close()
