#!/usr/bin/env python3
"""Per-kernel average times (HIP events, pddp_time_kernels) of the bench sweep: usage tools/kernel_times.py [batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
LIB = os.environ.get("PDDP_LIB")          # alternative build of libpddp (A/B measurements of build variants)
for B in [int(v) for v in sys.argv[1:]] or [4096]:
    cfg = pyddp.default_config(4, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=B, max_iter=200, use_graph=1, _lib_path=LIB)
    s = pyddp.Solver(cfg, _lib_path=LIB)
    x0, u0, xg = bench.example_inputs(128, np.random.default_rng(1), B)
    s.load(x0, u0, xg); s.iterate(5); s.sync()
    k = s.time_kernels(30)
    s.load(x0, u0, xg); s.iterate(5); s.sync()      # the graph replays over the SAME iterations (6..35, bench.py's window): later ones reject more and more steps, whose setup kernel is idle
    plain, _ = s.time_sweeps(30, phases=False)
    print(B, " ".join(f"{n}={ms * 1e3:.1f}us" for n, ms in k), f"sum={sum(ms for _, ms in k) * 1e3:.1f}us graph={plain / 30 * 1e3:.1f}us/sweep -> {B * 30 / (plain * 1e-3):.0f} it/s", flush=True)
    s.close()
