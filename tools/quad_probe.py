#!/usr/bin/env python3
"""Quadrotor (BASELINE configs[4]) sweeps alone, for counter passes and A/B runs.  usage: tools/quad_probe.py [batch] [dtype 0|1] [sweeps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import pyddp
from bench import closed_form_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dtype = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kw = dict(N=256, M=4, A=16, integrator=3, total_time=4.0)
rng = np.random.default_rng(1)
LIB = os.environ.get("PDDP_LIB")          # alternative build of libpddp (A/B measurements of build variants)
s = pyddp.Solver(pyddp.default_config(3, batch=B, max_iter=100, tol_cost=0.0, use_graph=1, dtype=dtype, _lib_path=LIB, **kw), _lib_path=LIB)
x0, u0, xg = closed_form_inputs(3, kw["N"], rng, B)
s.load(x0, u0, xg); s.iterate(3); s.sync()
t = {k: round(v, 4) for k, v in s.time_kernels(sweeps)}
print("quad B", B, "f64" if dtype else "f32", t, "sum", round(sum(t.values()), 3), "J", s.store()["Jout"][0][:4].tolist(), flush=True)
s.close()
