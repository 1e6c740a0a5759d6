#!/usr/bin/env python3
"""Milliseconds per sweep over successive windows of DDP iterations of the bench configuration (does the sweep time depend on how far the solves have progressed?):
usage tools/sweep_windows.py [batch] [boundary_only 0|1] [windows] [sweeps per window]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
bo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cfg = pyddp.default_config(4, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=B, max_iter=nw * K + 10, use_graph=1, boundary_cost_to_go_only=bo)
s = pyddp.Solver(cfg)
x0, u0, xg = bench.example_inputs(128, np.random.default_rng(1234), B)
s.load(x0, u0, xg); s.iterate(5); s.sync()
out = []
for w in range(nw):
    t0 = time.perf_counter(); s.iterate(K); s.sync(); wall = (time.perf_counter() - t0) / K * 1e3
    out.append(f"{wall:.3f}")
print(B, "boundary_only", bo, "ms/sweep per window of", K, ":", " ".join(out), flush=True)
s.close()
