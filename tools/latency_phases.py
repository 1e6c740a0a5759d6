#!/usr/bin/env python3
"""Per-phase kernel times of ONE problem in flight (the latency case): HIP-event times per phase and the replayed-graph time per sweep."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tests'))
import sys as _s, os as _o; _s.path.insert(0, _o.path.dirname(_o.path.abspath(__file__))); import _sel; _sel.install()      # PDDP_BP / PDDP_FP / ... on this tool's command line -> pddp_config.kernels (tools/_sel.py; the library reads no environment)
rng = np.random.default_rng(1)
for N, A, bp in ((128, 8, None), (128, 8, "coop"), (64, 16, None)):
    if bp: os.environ["PDDP_BP"] = bp
    else: os.environ.pop("PDDP_BP", None)
    lib = os.environ.get("PDDP_LIB")            # a build variant (parallel-ddp_amd/lib/libpddp_<tag>.so)
    cfg = pyddp.default_config(4, N=N, M=4, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=1, max_iter=100, use_graph=1, _lib_path=lib)
    s = pyddp.Solver(cfg, _lib_path=lib)
    x0, u0, xg = bench.example_inputs(N, rng, 1)
    s.load(x0, u0, xg); s.set_benchmark_mode(1); s.iterate(5); s.sync()
    tot, ph = s.time_sweeps(50, phases=True)
    plain, _ = s.time_sweeps(50, phases=False)
    print(N, A, bp, "per-phase us", [round(v / 50 * 1e3, 1) for v in ph], "sum", round(tot / 50 * 1e3, 1), "graph replay us/sweep", round(plain / 50 * 1e3, 1))
    s.close()
