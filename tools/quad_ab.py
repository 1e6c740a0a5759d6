#!/usr/bin/env python3
"""BASELINE configs[4] (quadrotor N=256, A=16, M=4, RK3) with the device full: per-kernel times of the sweep on build variants (parallel-ddp_amd/lib/libpddp_<tag>.so;
tag "product" = lib/libpddp.so), alternating on ONE box.  usage (through gpurun): tools/quad_ab.py <rounds> <dtypes: f32|f64|both> <tag[:field=value,...]> ...
e.g. tools/quad_ab.py 2 f32 base new new:cf_nis=kb20"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
rounds, which = int(sys.argv[1]), sys.argv[2]
specs = sys.argv[3:]
for r in range(rounds):
    for dtype, B in [(0, 16384)] * (which in ("f32", "both")) + [(1, 8192)] * (which in ("f64", "both")):
        for spec in specs:
            tag, _, sel = spec.partition(":")
            kernels = dict(kv.split("=") for kv in sel.split(",")) if sel else {}
            kw = dict(N=256, M=4, A=16, integrator=3, total_time=4.0)
            lib = None if tag == "product" else os.path.join(ROOT, "parallel-ddp_amd", "lib", f"libpddp_{tag}.so")
            s = pyddp.Solver(pyddp.default_config(3, batch=B, max_iter=100, tol_cost=0.0, dtype=dtype, use_graph=1, kernels=kernels, _lib_path=lib, **kw), _lib_path=lib)
            x0, u0, xg = bench.closed_form_inputs(3, 256, np.random.default_rng(99), B)
            s.load(x0, u0, xg); s.iterate(3); s.sync()
            ms, _ = s.time_sweeps(10, phases=False)
            s.load(x0, u0, xg); s.iterate(3); s.sync()
            k = s.time_kernels(10)
            print(("f64" if dtype else "f32"), B, spec, " ".join(f"{n}={v:.3f}" for n, v in k if n), f"sweep={ms / 10:.3f} ms", flush=True)
            s.close()
