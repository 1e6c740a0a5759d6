#!/usr/bin/env python3
"""Quadrotor (12 states): 16-lane group kernels (k_nis_gl, k_bp_gl) against the cooperative / thread-serial ones, per phase.  usage: tools/cf_gl.py [batch]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import pyddp
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tests'))
import sys as _s, os as _o; _s.path.insert(0, _o.path.dirname(_o.path.abspath(__file__))); import _sel; _sel.install()      # PDDP_BP / PDDP_FP / ... on this tool's command line -> pddp_config.kernels (tools/_sel.py; the library reads no environment)
from bench import closed_form_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kw = dict(N=256, M=4, A=16, integrator=3, total_time=4.0)
ref = None
for dtype in (0, 1):
    for env in ({}, {"PDDP_CF_NIS": "coop", "PDDP_CF_BP": "coop"}, {"PDDP_CF_NIS": "gl", "PDDP_CF_BP": "gl"}, {"PDDP_CF_NIS": "gl8"}):
        for k in ("PDDP_CF_NIS", "PDDP_CF_BP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        rng = np.random.default_rng(1)
        s = pyddp.Solver(pyddp.default_config(3, batch=B, max_iter=100, tol_cost=0.0, use_graph=1, dtype=dtype, **kw))
        x0, u0, xg = closed_form_inputs(3, kw["N"], rng, B)
        s.load(x0, u0, xg); s.iterate(3); s.sync()
        t = {k: round(v, 4) for k, v in s.time_kernels(5)}
        out = s.store()
        J = out["Jout"][0][:4].tolist()
        print("quad B", B, "f64" if dtype else "f32", env or "default", t, "sum", round(sum(t.values()), 3), "J", J, flush=True)
        s.close()
