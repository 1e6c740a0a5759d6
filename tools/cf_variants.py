#!/usr/bin/env python3
"""Per-kernel times of the closed-form plants' kernels, thread-serial vs wave-cooperative, per phase (PDDP_CF_BP / _FP / _NIS): picks the library's defaults."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import pyddp
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tests'))
import sys as _s, os as _o; _s.path.insert(0, _o.path.dirname(_o.path.abspath(__file__))); import _sel; _sel.install()      # PDDP_BP / PDDP_FP / ... on this tool's command line -> pddp_config.kernels (tools/_sel.py; the library reads no environment)
from bench import closed_form_inputs
rng = np.random.default_rng(1)
for name, plant, B, kw in (("cart N128 A8 M4 rk3", 2, 16384, dict(N=128, M=4, A=8, integrator=3, total_time=4.0)),
                           ("cart N128 A8 M4 rk3", 2, 1024, dict(N=128, M=4, A=8, integrator=3, total_time=4.0)),
                           ("quad N256 A16 M4 rk3", 3, 4096, dict(N=256, M=4, A=16, integrator=3, total_time=4.0)),
                           ("quad N256 A16 M4 rk3", 3, 256, dict(N=256, M=4, A=16, integrator=3, total_time=4.0)),
                           ("pend N64 A8 M4 euler", 1, 16384, dict(N=64, M=4, A=8, integrator=1, total_time=4.0))):
    for mode in ("ts", "coop"):
        os.environ["PDDP_CF"] = mode
        s = pyddp.Solver(pyddp.default_config(plant, batch=B, max_iter=100, tol_cost=0.0, use_graph=1, **kw))
        x0, u0, xg = closed_form_inputs(plant, kw["N"], rng, B)
        s.load(x0, u0, xg); s.iterate(3); s.sync()
        print(name, "B", B, mode, {k: round(v, 4) for k, v in s.time_kernels(5)}, flush=True)
        s.close()
