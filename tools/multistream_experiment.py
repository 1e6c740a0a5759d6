#!/usr/bin/env python3
"""Experiment: the same 4096 problems as ONE handle (one stream) or as several handles (one stream each) iterated concurrently -- does the GPU overlap the
latency-bound backward pass of one sub-batch with the issue-bound rollouts of another?  Prints problems x sweeps / s for each split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench

TOTAL, K, W = 4096, 40, 5
for parts in (1, 2, 4, 8):
    B = TOTAL // parts
    rng = np.random.default_rng(1234)
    hs = []
    for p in range(parts):
        cfg = pyddp.default_config(4, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=B, max_iter=100, use_graph=1)
        s = pyddp.Solver(cfg)
        x0, u0, xg = bench.example_inputs(128, rng, B)
        s.load(x0, u0, xg)
        hs.append(s)
    for s in hs:
        s.iterate(W)
    for s in hs:
        s.sync()
    t0 = time.perf_counter()
    for i in range(K):
        for s in hs:
            s.iterate(1)
    for s in hs:
        s.sync()
    t = time.perf_counter() - t0
    print(f"{parts} handle(s) x {B} problems: {TOTAL * K / t:,.0f} problems x sweeps / s  ({1e3 * t / K:.3f} ms per sweep of all {TOTAL})")
    for s in hs:
        s.close()
