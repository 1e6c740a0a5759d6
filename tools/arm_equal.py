#!/usr/bin/env python3
"""Two builds of the library (parallel-ddp_amd/lib/libpddp_<tag>.so; "product" = lib/libpddp.so) on the arm's matrix-core backward pass: are whole solves the same BITS?
usage (through gpurun): tools/arm_equal.py <tagA> <tagB>  -- float and double handles, M = 4 / 2 / 1, the large-batch and the one-problem selections, joint-space and
end-effector cost, 70 problems (a ragged last wavefront) and one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
bad = 0
cases = [(0, 4, dict(bp="mx", fp="tl"), 70, 0), (0, 1, dict(bp="mx", fp="tl"), 70, 0), (0, 2, dict(bp="mx", fp="tl"), 5, 0), (0, 4, {}, 1, 0), (1, 4, dict(bp="mx", fp="tl"), 5, 0),
         (0, 4, dict(bp="mx", fp="tl"), 64, 1), (0, 4, {}, 1, 1), (0, 4, dict(bp="mx", fp="lg"), 3, 0)]
for dtype, M, sel, B, ee in cases:
    outs, names = [], []
    for tag in sys.argv[1:3]:
        lib = None if tag == "product" else os.path.join(ROOT, "parallel-ddp_amd", "lib", f"libpddp_{tag}.so")
        N = 64 if ee else 128
        kw = dict(N=N, M=M, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=12, batch=B, dtype=dtype, use_graph=1, kernels=sel, _lib_path=lib)
        if ee:
            kw.update(ee_cost=1, mpc_mode=1, ignore_max_rho_exit=0)
        s = pyddp.Solver(pyddp.default_config(4, **kw), _lib_path=lib)
        x0, u0, xg = bench.ee_inputs(N, np.random.default_rng(7), B) if ee else bench.example_inputs(N, np.random.default_rng(7), B)
        if dtype:
            x0, u0, xg = (np.asarray(v, np.float64) for v in (x0, u0, xg))
        s.load(x0, u0, xg); s.iterate(12); s.sync()
        o = s.store(); o["P"] = s.get_cost_to_go()[0]
        outs.append(o); names.append([n for n, _ in s.time_kernels(1) if n][0])
        s.close()
    same = all(np.array_equal(outs[0][k], outs[1][k], equal_nan=True) for k in ("x", "u", "KT", "Jout", "alphaOut", "P"))
    bad += not same
    print("f64" if dtype else "f32", "M", M, "B", B, "ee" if ee else "joint", sel, names, "identical" if same else "DIFFERENT", "steps taken:", int((outs[0]["alphaOut"][:, 1:12] >= 0).sum()), flush=True)
sys.exit(1 if bad else 0)
