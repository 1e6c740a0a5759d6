#!/usr/bin/env python3
"""Per-phase kernel times (HIP events, ms per sweep) over the batch size for every kernel selection of the arm's sweep:
backward pass wide / coop / lg, forward pass + setup lg / tl.  Kuka N=128 A=8 M=4 float32 (BASELINE configs[2]).  Output -> profiles/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tests'))
import sys as _s, os as _o; _s.path.insert(0, _o.path.dirname(_o.path.abspath(__file__))); import _sel; _sel.install()      # PDDP_BP / PDDP_FP / ... on this tool's command line -> pddp_config.kernels (tools/_sel.py; the library reads no environment)
batches = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 8, 64, 256, 1024, 2048, 4096, 8192, 16384]
sel = sys.argv[2].split(",") if len(sys.argv) > 2 else ["auto", "wide+lg", "coop+lg", "lg+lg", "lg+tl", "wide+tl"]
rng = np.random.default_rng(1)
print("batch selection  bp_ms fp_ms ls_ms nis_ms  sum_ms  graph_ms_per_sweep  iterations_per_s")
for B in batches:
    x0, u0, xg = bench.example_inputs(128, rng, B)
    for name in sel:
        if name == "auto":
            os.environ.pop("PDDP_BP", None); os.environ.pop("PDDP_FP", None)
        else:
            bp, fp = name.split("+"); os.environ["PDDP_BP"] = bp; os.environ["PDDP_FP"] = fp
        if name.startswith("wide") and B * 4 > 8192: continue
        if name.startswith("coop") and B > 8192: continue
        lib = os.environ.get("PDDP_LIB")
        cfg = pyddp.default_config(4, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=B, max_iter=200, use_graph=1, _lib_path=lib)
        s = pyddp.Solver(cfg, _lib_path=lib)
        s.load(x0, u0, xg); s.iterate(5); s.sync()
        K = 30 if B <= 4096 else 12
        tot, ph = s.time_sweeps(K, phases=True)
        plain, _ = s.time_sweeps(K, phases=False)
        print(f"{B:6d} {name:8s} " + " ".join(f"{v / K:8.4f}" for v in ph) + f" {tot / K:8.4f} {plain / K:8.4f} {B * K / (plain * 1e-3):12.0f}", flush=True)
        s.close()
