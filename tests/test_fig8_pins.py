"""Pins from DATA the reference's test directory holds: the recorded lock-step figure-eight run in test/WAFR_fig8.py
(tests/golden/fig8_trace.json, extracted by make_fig8_trace.py) and the 200-point goal table of loadFig8Goal (tests/golden/fig8_goals.csv).

  * every recorded goal lies on the piecewise-linear figure through the table (pins the table and the interpolation form of loadFig8Goal);
  * the recorded error norm is |tool point - goal| (pins evNorm, utils/exampleUtils.cuh:84-93);
  * the first recorded tool point is the zero pose's link-7 origin, 1.261 m up: the oracle's kinematic chain with a zero tool offset gives exactly that
    (that recorded run predates the flange offset EE_ON_LINK_Z = 0.0635 of dynamics_arm.cuh:57-58: with it the zero pose gives 1.3245);
  * the recorded run's last running average is the published 0.087763;
  * hostapi's loadFig8Goal (C++) reproduces numpy's interpolation over the same table, including the wrap-around count."""
import json
import os
import subprocess

import numpy as np

from oracle_binding import Oracle, default_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TRACE = json.load(open(os.path.join(GOLD, "fig8_trace.json")))
GOALS = np.loadtxt(os.path.join(GOLD, "fig8_goals.csv"), delimiter=",", comments="#")


def dist_to_polyline(p, pts):
    a, b = pts[:-1], pts[1:]
    ab = b - a
    t = np.clip(((p - a) * ab).sum(1) / np.maximum((ab * ab).sum(1), 1e-30), 0, 1)
    return np.linalg.norm(a + t[:, None] * ab - p, axis=1).min()


def test_recorded_goals_lie_on_the_figure_and_errors_are_tool_point_distances():
    assert GOALS.shape == (200, 3)
    closed = np.vstack([GOALS, GOALS[:1]])                      # ru wraps to entry 0 (loadFig8Goal's `% numGoals`)
    for ee, goal, enorm, avg, vnorm in TRACE["records"]:
        assert dist_to_polyline(np.asarray(goal), closed) < 2e-6          # the trace is printed with 6 decimals
        assert abs(np.linalg.norm(np.asarray(ee) - np.asarray(goal)) - enorm) < 3e-6
    assert TRACE["records"][0][1] == [round(v, 6) for v in GOALS[0]]
    assert abs(TRACE["records"][-1][3] - TRACE["published_average_tracking_error"]) < 1e-6 and TRACE["published_average_tracking_error"] == 0.087763


def test_kinematic_chain_against_the_first_recorded_tool_point():
    o0 = Oracle(default_cfg(4, N=64, wafr_urdf=1, mpc_mode=1, ee_cost=1, ee_on_link_z=0.0), np.float64)
    o1 = Oracle(default_cfg(4, N=64, wafr_urdf=1, mpc_mode=1, ee_cost=1), np.float64)
    z0 = o0.ee_pos(np.zeros(14), False)[0]
    np.testing.assert_allclose(z0[:3], [0, 0, 1.261], atol=1e-9)
    np.testing.assert_allclose(o1.ee_pos(np.zeros(14), False)[0][2], 1.261 + 0.0635, atol=1e-9)
    first = np.asarray(TRACE["records"][0][0])                 # after one short control cycle from the zero pose: millimetres away
    assert np.linalg.norm(first - z0[:3]) < 2.5e-3


def test_hostapi_goal_generator_matches_numpy_interpolation():
    exe = os.path.join(ROOT, "tests", "wire", "wire_driver")
    pkg = os.path.join(ROOT, "parallel-ddp_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-Wall", os.path.join(ROOT, "tests", "wire", "wire_driver.cpp"), "-L" + os.path.join(pkg, "lib"), "-lpddp",
                           "-Wl,-rpath," + os.path.join(pkg, "lib"), "-o", exe])
    out = subprocess.run([exe, os.path.join(GOLD, "fig8_goals.csv")], capture_output=True, text=True, check=True).stdout
    rows = [l.split() for l in out.splitlines() if l.startswith("goal")]
    assert len(rows) == 6
    total, n = 10.0e6, 200
    for _, t, rep, gx, gy, gz in rows:
        t = float(t)
        num = t / (total / (n - 1))
        fr = np.float32(num - np.floor(num)); rd, ru = int(np.floor(num)) % n, int(np.ceil(num)) % n
        want = (np.float32(1) - fr) * GOALS[rd].astype(np.float32) + fr * GOALS[ru].astype(np.float32)
        np.testing.assert_allclose([float(gx), float(gy), float(gz)], want, rtol=2e-7)
        assert int(rep) == int(np.floor(num)) // n
    assert [int(r[2]) for r in rows] == [0, 0, 0, 0, 0, 1]       # one whole figure is over when floor(goalNum) reaches 200, a little after totalTime
