#!/usr/bin/env python3
"""BASELINE configs[4] (quadrotor N=256, A=16, M=4, RK3) with the device full: per-kernel times of the sweep with the backward pass on the matrix cores (k_bp_mq, the
library's choice) and on the lane-per-column kernel it replaced (k_bp_cl), alternating on ONE box.  usage (through gpurun): tools/quad_bp_ab.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tags = sys.argv[2:]                                  # build variants (parallel-ddp_amd/lib/libpddp_<tag>.so), float only: tools/quad_bp_ab.py 2 mqw5 mqw6 ...
for r in range(rounds):
    for dtype, B in ((0, 16384),) if tags else ((0, 16384), (1, 8192)):
        for sel in (tags or ("mq", "cl")):
            kw = dict(N=256, M=4, A=16, integrator=3, total_time=4.0)
            lib = os.path.join(ROOT, "parallel-ddp_amd", "lib", f"libpddp_{sel}.so") if tags else None
            s = pyddp.Solver(pyddp.default_config(3, batch=B, max_iter=100, tol_cost=0.0, dtype=dtype, use_graph=1, kernels=dict(cf_bp="mq" if tags else sel), _lib_path=lib, **kw), _lib_path=lib)
            x0, u0, xg = bench.closed_form_inputs(3, 256, np.random.default_rng(99), B)
            s.load(x0, u0, xg); s.iterate(3); s.sync()
            ms, _ = s.time_sweeps(10, phases=False)
            s.load(x0, u0, xg); s.iterate(3); s.sync()
            k = s.time_kernels(10)
            print(("f64" if dtype else "f32"), B, sel, " ".join(f"{n}={v:.3f}" for n, v in k if n), f"sweep={ms / 10:.3f} ms", flush=True)
            s.close()
