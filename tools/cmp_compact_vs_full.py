#!/usr/bin/env python3
"""One-off comparisons of builds / layouts of the float large-batch kernel selection (matrix-core backward pass + thread lanes) on ragged shapes:
   tools/cmp_compact_vs_full.py builds <tagA> <tagB>   -- the same solves on parallel-ddp_amd/lib/libpddp<tag>.so, outputs compared bit for bit
   tools/cmp_compact_vs_full.py layouts [<tag>]        -- compact [A B] (LDS prefetch) against PDDP_AB=full on one build, largest differences after 1, 2, 6 iterations"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, pyddp
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tests'))
import sys as _s, os as _o; _s.path.insert(0, _o.path.dirname(_o.path.abspath(__file__))); import _sel; _sel.install()      # PDDP_BP / PDDP_FP / ... on this tool's command line -> pddp_config.kernels (tools/_sel.py; the library reads no environment)
from oracle_binding import example_inputs
SHAPES = [(32, 4), (32, 1), (64, 4), (128, 1), (128, 8), (256, 4), (128, 4)]
def run(lib, env, batch, N, M, iters):
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    try:
        cfg = pyddp.default_config(4, _lib_path=lib, dtype=0, batch=batch, use_graph=0, N=N, M=M, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=iters)
        s = pyddp.Solver(cfg, _lib_path=lib)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    rng = np.random.default_rng(11); xs, us, gs = [], [], []
    for _ in range(batch):
        x0, u0, xg = example_inputs(4, N, np.float32, noise=rng.normal(0, 0.002, (N, 14))); xs.append(x0); us.append(u0); gs.append(xg)
    out = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    res = {k: np.array(out[k]) for k in ("x", "u", "KT", "Jout", "alphaOut")}
    s.close(); return res
lib = lambda tag: os.path.join(ROOT, "parallel-ddp_amd", "lib", f"libpddp{tag}.so")
ENV = {"PDDP_BP": "mx", "PDDP_FP": "tl"}
if sys.argv[1] == "builds":
    for N, M in SHAPES:
        a, b = run(lib(sys.argv[2]), ENV, 5, N, M, 6), run(lib(sys.argv[3]), ENV, 5, N, M, 6)
        print(f"N={N} M={M}: compact path of '{sys.argv[2]}' vs '{sys.argv[3]}':", {k: bool(np.array_equal(a[k], b[k])) for k in a}, "alpha", a["alphaOut"][0][:7], flush=True)
else:
    tag = sys.argv[2] if len(sys.argv) > 2 else ""
    for iters in (1, 2, 6):
        c, f = run(lib(tag), ENV, 5, 64, 4, iters), run(lib(tag), dict(ENV, PDDP_AB="full"), 5, 64, 4, iters)
        print(tag or "current", "iters", iters, {k: float(np.max(np.abs(c[k].astype(np.float64) - f[k]))) for k in c}, flush=True)
