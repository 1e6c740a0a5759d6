#!/usr/bin/env python3
"""Two builds of the library (parallel-ddp_amd/lib/libpddp_<tag>.so) on the quadrotor with the matrix-core backward pass: are the solves the same BITS?
usage (through gpurun): tools/quad_mq_equal.py <tagA> <tagB>      -- float and double handles, M = 4 and M = 1, the plant's own cost Hessian and one read from H_k."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
bad = 0
for dtype in (0, 1):
    for M in (4, 1):
        for override_h in (0, 1):
            outs = []
            for tag in sys.argv[1:3]:
                lib = os.path.join(ROOT, "parallel-ddp_amd", "lib", f"libpddp_{tag}.so")
                B = 64
                s = pyddp.Solver(pyddp.default_config(3, batch=B, N=64, M=M, A=8, integrator=3, total_time=4.0, max_iter=12, tol_cost=0.0, dtype=dtype, use_graph=1,
                                                      kernels=dict(cf_bp="mq"), _lib_path=lib), _lib_path=lib)
                x0, u0, xg = bench.closed_form_inputs(3, 64, np.random.default_rng(5), B)
                if dtype:
                    x0, u0, xg = (np.asarray(v, np.float64) for v in (x0, u0, xg))
                s.load(x0, u0, xg)
                if override_h:                                   # the instantiation that reads H_k: hand the handle its own Hessian back
                    s.set("H", s.get("H"))
                s.iterate(16); s.sync()
                o = s.store()
                outs.append(o)
                names = [n for n, _ in s.time_kernels(1)]
                s.close()
            same = all(np.array_equal(outs[0][k], outs[1][k], equal_nan=True) for k in ("x", "u", "KT", "Jout", "alphaOut"))
            bad += not same
            print("f64" if dtype else "f32", "M", M, "H read" if override_h else "diag H", names[0], "identical" if same else "DIFFERENT",
                  "steps taken:", int((outs[0]["alphaOut"][:, 1:8] >= 0).sum()), flush=True)
sys.exit(1 if bad else 0)
