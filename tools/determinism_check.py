#!/usr/bin/env python3
"""Two checks of a handle's phase kernels on fixed inputs:
  * repeat every phase and compare the outputs bit for bit between repetitions (races);
  * run every phase again with each compute unit's LDS filled with NaNs beforehand (PDDP_POISON_LDS, solver_impl.hpp run_phase) and compare with the clean
    run: a kernel that reads LDS it has not written itself shows up as NaNs or changed values (what is in LDS when a kernel starts belongs to whichever
    kernel ran on that compute unit before -- results that depend on it differ from process to process).
usage: tools/determinism_check.py [reps] [batch] [ee|joint|cart|quad]   (env PDDP_BP / PDDP_FP / PDDP_CF_* select the kernel family; COLD=1: no warm-up sweeps)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import pyddp
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tests'))
import sys as _s, os as _o; _s.path.insert(0, _o.path.dirname(_o.path.abspath(__file__))); import _sel; _sel.install()      # PDDP_BP / PDDP_FP / ... on this tool's command line -> pddp_config.kernels (tools/_sel.py; the library reads no environment)
from backends import make_solver
from oracle_binding import example_inputs
from test_fp32_bar import EE_KW, KUKA, ee_start
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
what = sys.argv[3] if len(sys.argv) > 3 else "ee"
F32 = np.float32
rng = np.random.default_rng(5)
if what == "ee":
    plant, kw = 4, EE_KW
    x0, u0, xg = ee_start(kw["N"], np.float64, False)
elif what == "joint":
    plant, kw = 4, KUKA
    x0, u0, xg = example_inputs(4, kw["N"], np.float64)
elif what == "cart":
    plant, kw = 2, dict(N=128, M=4, A=8, integrator=3, total_time=4.0, tol_cost=0.0, max_iter=12)
    x0, u0, xg = example_inputs(2, kw["N"], np.float64)
else:
    plant, kw = 3, dict(N=256, M=4, A=16, integrator=3, total_time=4.0, tol_cost=0.0, max_iter=12)
    x0, u0, xg = example_inputs(3, kw["N"], np.float64)
s = make_solver("hip", plant, dtype=0, batch=B, **kw)
xs = np.concatenate([(x0 + rng.normal(0, 1e-3, x0.shape)).astype(F32) for _ in range(B)])
s.load(xs, np.tile(u0.astype(F32), B), np.tile(xg.astype(F32), B))
if not os.environ.get("COLD"):
    s.iterate(3); s.sync()
    print(what, "B", B, "kernels", [k for k, _ in s.time_kernels(1)])
names = {pyddp.PHASE_INIT_NIS: ("AB", "g", "H", "costk"), pyddp.PHASE_BP: ("KT", "du", "P", "p", "dJexp", "ApBK", "Bdu"), pyddp.PHASE_FP: ("xs", "us", "ds", "J", "dmax")}
if plant == 4 and kw["M"] > 1 and not os.environ.get("PDDP_SWEEP") and os.environ.get("PDDP_BP", "mx") == "mx":       # the production pair of the matrix-core family: maps composed in the backward pass
    names[pyddp.PHASE_BP_FUSED] = ("KT", "du", "P", "p", "dJexp"); names[pyddp.PHASE_SWEEP_FUSED] = ("xs",)
st0 = s.get_state()


def differs(a, b):
    a, b = a.ravel(), b.ravel()
    d = np.flatnonzero((a != b) & ~(np.isnan(a) & np.isnan(b)))
    return None if d.size == 0 else (int(d.size), int(d[0]), int(np.isnan(b[d]).sum()))


for ph, outs in names.items():
    ref = None; bad = {}
    for r in range(reps):
        s.set_state(st0)
        s.run_phase(ph); s.sync()
        cur = {k: s.get(k).copy() for k in outs}
        if ref is None:
            ref = cur
        else:
            for k in outs:
                d = differs(ref[k], cur[k])
                if d:
                    bad.setdefault(k, []).append((r,) + d)
    print("phase", ph, "differences between repetitions:", {k: v[:4] for k, v in bad.items()} if bad else "none")
    os.environ["PDDP_POISON_LDS"] = "1"
    s.set_state(st0)
    s.run_phase(ph); s.sync()
    os.environ.pop("PDDP_POISON_LDS")
    pois = {k: differs(ref[k], s.get(k)) for k in outs}
    print("phase", ph, "with poisoned LDS (entries that differ, first index, of which NaN):", {k: v for k, v in pois.items() if v} or "same bits")
    s.set_state(st0); s.run_phase(ph); s.sync()          # leave clean outputs behind for the next phase
s.close()
