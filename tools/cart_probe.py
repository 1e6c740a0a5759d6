#!/usr/bin/env python3
"""Cart-pole (BASELINE configs[1]) sweeps alone, per kernel.  usage: tools/cart_probe.py [batch] [dtype 0|1] [sweeps]   (PDDP_CF_FP=cf|ts selects the rollout kernel)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import pyddp
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tests'))
import sys as _s, os as _o; _s.path.insert(0, _o.path.dirname(_o.path.abspath(__file__))); import _sel; _sel.install()      # PDDP_BP / PDDP_FP / ... on this tool's command line -> pddp_config.kernels (tools/_sel.py; the library reads no environment)
from bench import closed_form_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dtype = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kw = dict(N=128, M=4, A=8, integrator=3, total_time=4.0)
rng = np.random.default_rng(1)
s = pyddp.Solver(pyddp.default_config(2, batch=B, max_iter=100, tol_cost=0.0, use_graph=1, dtype=dtype, **kw))
x0, u0, xg = closed_form_inputs(2, kw["N"], rng, B)
s.load(x0, u0, xg); s.iterate(3); s.sync()
t = {k: round(v, 4) for k, v in s.time_kernels(sweeps)}
print("cart B", B, "f64" if dtype else "f32", t, "sum", round(sum(t.values()), 3), "J", s.store()["Jout"][0][:4].tolist(), flush=True)
s.close()
