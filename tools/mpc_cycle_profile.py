#!/usr/bin/env python3
"""Control cycles of the published MPC shape (Kuka N=64, A=16, M=4, 4 iterations per cycle, shifted by one knot): wall clock per cycle and, under
rocprofv3 --kernel-trace --stats, the kernels of a cycle.  usage: tools/mpc_cycle_profile.py [ee_cost 0|1] [cycles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
ee = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 200
full = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # full_rollout (0: the warm-start rollout covers the first shooting segment only)
N = 64
rng = np.random.default_rng(77)
kw = dict(wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, ignore_max_rho_exit=0, use_graph=1)
s = pyddp.Solver(pyddp.default_config(4, N=N, M=4, A=16, batch=1, max_iter=100, ee_cost=ee, **kw))
x0, u0, xg = bench.ee_inputs(N, rng, 1)
if not ee:
    xg[0, :7] = [0.5, 0.6, -0.3, -0.9, 0.2, 0.7, 0.1]
s.load(x0, u0, xg)
first = s.mpc_solve(x0[0, 0], xg, 0, clear_vars=1, max_iter=100)
xa = first["x"][0][1]
ms = []
for c in range(cycles):
    t0 = time.perf_counter()
    r = s.mpc_solve(xa + rng.normal(0, 0.0005, 14).astype(np.float32), xg, 1, max_iter=4, full_rollout=full)
    ms.append((time.perf_counter() - t0) * 1e3)
    xa = r["x"][0][1]
print(f"ee_cost={ee}: median control cycle {np.median(ms[5:]):.3f} ms over {cycles} cycles of 4 iterations (min {np.min(ms[5:]):.3f})")
