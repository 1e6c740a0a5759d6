#!/usr/bin/env python3
"""k_bp_mfma launch time (HIP events) of build variants of libpddp on ONE box: usage tools/bp_exp_times.py <lib tag> ... (parallel-ddp_amd/lib/libpddp_<tag>.so)
Both cost-to-go modes (bench.py's boundary-only handle and the library default); benchmark mode keeps every problem iterating whatever a measurement variant computes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
B = int(os.environ.get("PDDP_BATCH", "16384"))
ROUNDS = int(os.environ.get("PDDP_ROUNDS", "2"))
for r in range(ROUNDS):
    for tag in sys.argv[1:]:
        lib = os.path.join(ROOT, "parallel-ddp_amd", "lib", f"libpddp_{tag}.so")
        out = []
        for bnd in (1, 0):
            cfg = pyddp.default_config(4, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=B, max_iter=200, use_graph=1, boundary_cost_to_go_only=bnd, _lib_path=lib)
            s = pyddp.Solver(cfg, _lib_path=lib)
            x0, u0, xg = bench.example_inputs(128, np.random.default_rng(1), B)
            s.load(x0, u0, xg); s.set_benchmark_mode(1); s.iterate(5); s.sync()
            k = dict(s.time_kernels(20))
            bp = [v for n, v in k.items() if n.startswith("k_bp")][0]
            out.append(f"{'boundary' if bnd else 'all-slots'} " + " ".join(f"{n.split('<')[0][2:]}={v * 1e3:.0f}" for n, v in k.items()) + f" sum={sum(k.values()) * 1e3:.1f}us")
            s.close()
        print(f"{tag:6s}", " | ".join(out), flush=True)
