#!/usr/bin/env python3
"""Whole float32 solves of the smoke case (Kuka N=128 A=8 M=4, 10 iterations) over four noise seeds: per iteration the relative error of J against the float64 oracle for
three kernel selections (one-problem, large-batch, lane groups) next to the float32 oracle's own -- what __graft_entry__.smoke()'s tolerance is read against."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, pyddp
from oracle_binding import Oracle, default_cfg, example_inputs
kw = dict(N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=10)
for seed in (5, 6, 7, 8):
    noise = np.random.default_rng(seed).normal(0, 0.001, (128, 14))
    x64, u64, g64 = example_inputs(4, 128, np.float64, noise=noise)
    r64 = Oracle(default_cfg(4, cores=1, spawn_threads=0, **kw), np.float64).run_ilqr_gpusem(x64, u64, g64)
    x32, u32, g32 = (a.astype(np.float32) for a in (x64, u64, g64))
    r32 = Oracle(default_cfg(4, cores=1, spawn_threads=0, **kw), np.float32).run_ilqr_gpusem(x32, u32, g32)
    for label, sel in (("one", None), ("large", dict(bp="mx", fp="tl")), ("lg", dict(bp="lg", fp="lg"))):
        s = pyddp.Solver(pyddp.default_config(4, dtype=0, kernels=sel, **kw))
        out = s.solve(x32, u32, g32)
        lead = next((i for i in range(11) if not (out["alphaOut"][0][i] == r32["alphaOut"][i] == r64["alphaOut"][i])), 11)
        rows = [(i, "%.1e" % (abs(float(out["Jout"][0][i]) - r64["Jout"][i]) / r64["Jout"][i]), "%.1e" % (abs(float(r32["Jout"][i]) - r64["Jout"][i]) / r64["Jout"][i])) for i in range(lead)]
        print(seed, label, "lead", lead, rows)
        s.close()
