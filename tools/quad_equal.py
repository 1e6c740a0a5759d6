#!/usr/bin/env python3
"""Two builds of the library (parallel-ddp_amd/lib/libpddp_<tag>.so; "product" = lib/libpddp.so) on the quadrotor's full-device kernels (k_fp_cf / k_sweep_cf, k_nis_kb, and
the bit-exact backward pass k_bp_cl): are whole solves the same BITS?  usage (through gpurun): tools/quad_equal.py <tagA[:field=value,...]> <tagB[:field=value,...]>
-- float and double handles, M = 4 and M = 1, 16 and 8 step sizes, a ragged batch (70 problems: the last wavefront of every kernel is partly empty)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
bad = 0
for dtype in (0, 1):
    for M in (4, 1):
        for A in (16, 8):
            outs, names = [], []
            for spec in sys.argv[1:3]:
                tag, _, sel = spec.partition(":")
                kernels = dict(cf_bp="cl", cf_fp="cf", cf_nis="kb16")
                kernels.update(dict(kv.split("=") for kv in sel.split(",")) if sel else {})
                lib = None if tag == "product" else os.path.join(ROOT, "parallel-ddp_amd", "lib", f"libpddp_{tag}.so")
                B = 70
                s = pyddp.Solver(pyddp.default_config(3, batch=B, N=64, M=M, A=A, integrator=3, total_time=4.0, max_iter=12, tol_cost=0.0, dtype=dtype, use_graph=1, kernels=kernels, _lib_path=lib), _lib_path=lib)
                x0, u0, xg = bench.closed_form_inputs(3, 64, np.random.default_rng(5), B)
                if dtype:
                    x0, u0, xg = (np.asarray(v, np.float64) for v in (x0, u0, xg))
                s.load(x0, u0, xg)
                s.iterate(12); s.sync()
                o = s.store()
                o["AB"] = s.get("AB"); o["g"] = s.get("g"); o["P"] = s.get_cost_to_go()[0]
                outs.append(o)
                names.append([n for n, _ in s.time_kernels(1) if n])
                s.close()
            same = all(np.array_equal(outs[0][k], outs[1][k], equal_nan=True) for k in ("x", "u", "KT", "Jout", "alphaOut", "AB", "g", "P"))
            bad += not same
            print("f64" if dtype else "f32", "M", M, "A", A, names[0], "|", names[1], "identical" if same else "DIFFERENT", "steps taken:", int((outs[0]["alphaOut"][:, 1:12] >= 0).sum()), flush=True)
sys.exit(1 if bad else 0)
