#!/usr/bin/env python3
"""The fused forward-sweep maps of the quadrotor's matrix-core backward pass (bp_mq.hpp FUSE + k_sweep_maps_cf) against the per-knot form (A - B K | B du + k_sweep_cf /
the phase hook's rollouts) ON THE SAME HANDLE, teacher-forced through the phase hooks, and whole solves of a fused handle against one with kernels.sweep = st.
usage (through gpurun): tools/quad_fused_check.py <lib tag | product> [problems]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, ROOT)
import numpy as np, pyddp, bench
tag = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = None if tag == "product" else os.path.join(ROOT, "parallel-ddp_amd", "lib", f"libpddp_{tag}.so")
PIN = dict(cf_bp="mq", cf_fp="cf", cf_nis="kb16", ls="many")
N, M, A, n, m = 256, 4, 16, 12, 4
NB = N // M
bnd = [k for k in range(N) if (k + 1) % NB == 0 and k < N - 1]
for dtype in (1, 0):
    T = np.float64 if dtype else np.float32
    mk = lambda **kern: pyddp.Solver(pyddp.default_config(3, batch=B, max_iter=12, tol_cost=0.0, dtype=dtype, use_graph=1, kernels=dict(PIN, **kern), N=N, M=M, A=A, integrator=3, total_time=4.0, _lib_path=lib), _lib_path=lib)
    x0, u0, xg = bench.closed_form_inputs(3, N, np.random.default_rng(7), B)
    s = mk()
    print(("f64" if dtype else "f32"), "kernels", [k for k, _ in s.time_kernels(1) if k])
    s.load(x0, u0, xg); s.iterate(3); s.sync()
    s.run_phase(pyddp.PHASE_BP)
    ref = {k: s.get(k).copy() for k in ("KT", "du", "P", "p", "dJexp", "ApBK", "Bdu")}
    s.run_phase(pyddp.PHASE_FP)
    xs = s.get("xs").reshape(B, A, N, n).copy()
    s.run_phase(pyddp.PHASE_BP_FUSED)
    same = {k: bool(np.array_equal(s.get(k), ref[k], equal_nan=True)) for k in ("KT", "du", "P", "p", "dJexp")}
    s.set("xw", np.zeros(B * N * A * (n + m), T))
    s.run_phase(pyddp.PHASE_SWEEP_FUSED)
    xw = s.get("xw").reshape(B, N, A, n + m)
    worst = 0.0
    for k in bnd:
        a_ = xw[:, k + 1, :, :n].astype(np.float64); b_ = xs[:, :, k + 1, :].astype(np.float64)
        ok = np.isfinite(b_).all(axis=2)
        worst = max(worst, float((np.abs(a_ - b_)[ok]).max() / np.abs(b_[ok]).max()))
    ApBK = s.get("ApBK"); views_ok = bool(np.allclose(ApBK, ref["ApBK"], rtol=1e-5 if not dtype else 1e-12, atol=1e-6 if not dtype else 1e-12))
    print("  gains / cost-to-go of the fused instantiation equal the plain one's bits:", same, "| start states of all candidates at the boundaries, fused vs per-knot sweep, max rel:", f"{worst:.2e}", "| A - B K through pddp_get_array after the fused pass:", views_ok)
    s.close()
    # whole solves: fused handle vs per-knot handle
    outs = []
    for kern in ({}, {"sweep": "st"}):
        s = mk(**kern)
        names = [k for k, _ in s.time_kernels(1) if k]
        outs.append((names, s.solve(x0, u0, xg))); s.close()
    (na, oa), (nb, ob) = outs
    al_a, al_b = oa["alphaOut"], ob["alphaOut"]
    first = [int(np.argmax(al_a[b] != al_b[b])) if (al_a[b] != al_b[b]).any() else -1 for b in range(B)]
    Ja, Jb = oa["Jout"].astype(np.float64), ob["Jout"].astype(np.float64)
    it = int(min(oa["iters"].min(), ob["iters"].min()))
    rel = np.abs(Ja[:, : it + 1] - Jb[:, : it + 1]) / np.abs(Jb[:, : it + 1])
    print("  whole solves", na[1], "vs", nb[1], ": problems whose step-size decisions all agree:", sum(f < 0 for f in first), "of", B, "| first differing iteration (others):", sorted(set(f for f in first if f >= 0)),
          "| max rel J difference by iteration:", " ".join(f"{v:.1e}" for v in rel.max(axis=0)))
