#!/usr/bin/env python3
"""The float32 backward pass of the quadrotor at N = 256 against the oracle32 ensemble, record by record: the error of every ensemble member next to the errors of the
matrix-core kernel (k_bp_mq), the lane-per-column kernel (k_bp_cl) and the cooperative kernel on the same float32 inputs.  usage (through gpurun): tools/quad_bar_probe.py [seeds] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "parallel-ddp_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np, pyddp
import test_fp32_bar as bar
from backends import make_solver
from gpusem_steps import gpusem_iterations
from oracle_binding import Oracle, default_cfg, example_inputs
F32 = np.float32
seeds, iterations = (int(sys.argv[1]) if len(sys.argv) > 1 else 3), (int(sys.argv[2]) if len(sys.argv) > 2 else 12)
kw = dict(N=256, M=4, A=16, integrator=3, total_time=4.0, tol_cost=0.0, max_iter=iterations)
o64 = Oracle(default_cfg(3, cores=1, spawn_threads=0, **kw), np.float64); o32 = Oracle(default_cfg(3, cores=1, spawn_threads=0, **kw), np.float32)
o32f = Oracle(default_cfg(3, cores=1, spawn_threads=0, **kw), np.float32, variant="fma")
n, m, N, M = 12, 4, 256, 4
recs = []
for sd in range(seeds):
    x0, u0, xg = example_inputs(3, N, np.float64, noise=np.random.default_rng(41 + sd).normal(0, 0.001, (N, n)))
    with np.errstate(all="ignore"):
        recs += list(gpusem_iterations(o64, x0, u0, xg, iterations))
R = len(recs)
r32 = [{k: (v.astype(F32) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in rec.items()} for rec in recs]
stack = lambda key: np.stack([r[key].ravel() for r in r32])
outs = {}
for name, sel in (("mq", dict(cf_bp="mq")), ("cl", dict(cf_bp="cl")), ("coop", dict(cf_bp="coop"))):
    s = make_solver("hip", 3, dtype=0, batch=R, kernels=sel, **kw)
    s.load(np.tile(x0.astype(F32), R), np.tile(u0.astype(F32), R), np.tile(xg.astype(F32), R))
    st = s.get_state()
    for b_ in range(R):
        rec = recs[b_]; st[b_].cur = 0; st[b_].cur2 = 1; st[b_].pw = 0; st[b_].rho = rec.rho; st[b_].drho = rec.drho; st[b_].done = 0; st[b_].accepted = 0; st[b_].iter = rec.iter
    s.set_state(st)
    s.set("xb", np.concatenate([stack("x").reshape(R, 1, N * n), stack("xp2").reshape(R, 1, N * n)], axis=1)); s.set("ucur", stack("u")); s.set("dcur", stack("d"))
    for k in ("AB", "g", "Pp", "pp"):
        s.set(k, stack(k))
    s.run_phase(pyddp.PHASE_BP)
    outs[name] = {k: s.get(k).reshape(R, -1) for k in ("KT", "du", "P", "p", "dJexp", "ApBK", "Bdu")}
    print(name, [nm for nm, _ in s.time_kernels(1) if nm][:1], flush=True)
    s.close()
ratios = {k: [] for k in outs}
for i, rec in enumerate(recs):
    q = r32[i]
    members = [bar.oracle_bp(o32, q, rec.rho), bar.oracle_bp(o32f, q, rec.rho)]
    members += [bar.oracle_bp(o32, q, rec.rho, np.random.default_rng(1000 * (j + 1) + i)) for j in range(3)]
    members += [bar.oracle_bp(o32f, q, rec.rho, np.random.default_rng(1000 * (j + 4) + i)) for j in range(3)]
    errs = [{nm: bar.nrel(v, r) for nm, v, r in bar.bp_quantities(mem, rec, n, N, M)} for mem in members]
    for nm in errs[0]:
        floor = max(e[nm] for e in errs)
        line = f"rec {i:2d} it {rec.iter:2d} rho {rec.rho:8.3g} {nm:6s} members " + " ".join(f"{e[nm]:.1e}" for e in errs)
        for kname, o in outs.items():
            ek = dict((a, bar.nrel(v, r)) for a, v, r in bar.bp_quantities({k: o[k][i] for k in o}, rec, n, N, M))[nm]
            ratio = ek / max(floor, 1e-4 / 1.5)
            ratios[kname].append(ratio)
            line += f" | {kname} {ek:.1e} ({ratio:.2f})"
        if any(ratios[k][-1] > 1.0 for k in outs):
            print(line, flush=True)
for kname, r in ratios.items():
    r = np.asarray(r)
    print(f"{kname}: n {len(r)} median {np.median(r):.2f} p90 {np.percentile(r, 90):.2f} p99 {np.percentile(r, 99):.2f} max {r.max():.2f} frac<=1.5 {np.mean(r <= 1.5):.4f}")
