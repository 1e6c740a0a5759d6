"""TEST TOOL: prints how far the HIP kernels are from the oracle (float32 and float64 instantiations), per level.

    python tests/parity_report.py [path/to/libpddp*.so ...]      (on the GPU box; default: the product library)

For float32 it reports three distances, all norm-wise (max |a-b| / max |b|):
    kernel(f32) vs oracle(f32)   -- the parity number,
    oracle(f32) vs oracle(f64)   -- the reference algorithm's own float32 rounding noise on these inputs,
    kernel(f32) vs oracle(f64)   -- the kernel's distance from the exact answer.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "parallel-ddp_amd")):
    sys.path.insert(0, p)

import pyddp  # noqa: E402
from oracle_binding import Oracle, default_cfg, example_inputs  # noqa: E402


def nrel(a, ref):
    ref = np.asarray(ref, np.float64).ravel()
    return float(np.abs(np.asarray(a, np.float64).ravel() - ref).max() / max(np.abs(ref).max(), 1e-30))


def plant_level(path):
    rng = np.random.default_rng(5)
    cfgk = dict(N=16, M=1, A=1, wafr_urdf=1)
    s = pyddp.Solver(pyddp.default_config(4, _lib_path=path, dtype=0, **cfgk), _lib_path=path)
    o32, o64 = Oracle(default_cfg(4, **cfgk), np.float32), Oracle(default_cfg(4, **cfgk), np.float64)
    x = np.concatenate([rng.normal(0, 1.0, (64, 7)), rng.normal(0, 0.5, (64, 7))], axis=1).astype(np.float32)
    u = rng.normal(0, 20, (64, 7)).astype(np.float32)
    got = [s.plant_eval(w, x, u) for w in range(4)]
    names = ("dynamics", "dynamicsGradient", "integrator", "integratorGradient")
    for w, name in enumerate(names):
        e = [0.0, 0.0, 0.0]
        for i in range(64):
            r32 = [o32.dynamics_gradient(x[i], u[i])[1], o32.dynamics_gradient(x[i], u[i])[0], o32.integrator(x[i], u[i]), o32.integrator_gradient(x[i], u[i])][w]
            r64 = [o64.dynamics_gradient(x[i], u[i])[1], o64.dynamics_gradient(x[i], u[i])[0], o64.integrator(x[i], u[i]), o64.integrator_gradient(x[i], u[i])][w]
            e[0] = max(e[0], nrel(got[w][i], r32)); e[1] = max(e[1], nrel(r32, r64)); e[2] = max(e[2], nrel(got[w][i], r64))
        print(f"  plant {name:20s} kernel-vs-oracle32 {e[0]:.2e}   oracle32-vs-oracle64 {e[1]:.2e}   kernel-vs-oracle64 {e[2]:.2e}")
    s.close()


def backward_pass_level(path, N=32, M=4):
    """One backward pass from identical (float32) inputs taken from a real solve's first iteration."""
    kw = dict(N=N, M=M, A=4, wafr_urdf=1, total_time=0.5)
    n, m = 14, 7
    s = pyddp.Solver(pyddp.default_config(4, _lib_path=path, dtype=0, **kw), _lib_path=path)
    x, u, xg = example_inputs(4, N, np.float32)
    rng = np.random.default_rng(1)
    x = (x.reshape(N, n) + rng.normal(0, 0.01, (N, n))).astype(np.float32).ravel()
    s.load(x, u, xg)
    AB, H, g = s.get("AB"), s.get("H"), s.get("g")
    s.run_phase(pyddp.PHASE_BP)
    outs = {}
    for dt in (np.float32, np.float64):
        o = Oracle(default_cfg(4, **kw), dt)
        z = lambda *sh: np.zeros(sh, dt)
        P, p, Pp, pp, KT, du, d, ApBK, Bdu = z(N, n, n), z(N, n), z(N, n, n), z(N, n), z(N, m, n), z(N, m), z(N, n), z(N, n, n), z(N, n)
        o.backward_pass(1, AB.astype(dt), P, p, Pp, pp, H.astype(dt), g.astype(dt), KT, du, d, ApBK, Bdu, x.astype(dt), x.astype(dt), 12.5)
        outs[dt] = dict(KT=KT, du=du, P=P, p=p)
    for name in ("KT", "du", "P", "p"):
        k = s.get(name)
        print(f"  bp    {name:20s} kernel-vs-oracle32 {nrel(k, outs[np.float32][name]):.2e}   oracle32-vs-oracle64 "
              f"{nrel(outs[np.float32][name], outs[np.float64][name]):.2e}   kernel-vs-oracle64 {nrel(k, outs[np.float64][name]):.2e}")
    s.close()


def solver_level(path):
    kw = dict(N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=30)
    s = pyddp.Solver(pyddp.default_config(4, _lib_path=path, dtype=0, **kw), _lib_path=path)
    x0, u0, xg = example_inputs(4, 128, np.float32)
    out = s.solve(x0, u0, xg)
    r = Oracle(default_cfg(4, cores=1, spawn_threads=0, **kw), np.float32).run_ilqr_gpusem(x0, u0, xg)
    r64 = Oracle(default_cfg(4, cores=1, spawn_threads=0, **kw), np.float64).run_ilqr_gpusem(x0, u0, xg)
    a, ar, a64 = list(out["alphaOut"][0][:31]), list(r["alphaOut"][:31]), list(r64["alphaOut"][:31])
    same = next((i for i in range(31) if a[i] != ar[i]), 31)
    same64 = next((i for i in range(31) if ar[i] != a64[i]), 31)
    print(f"  solve Kuka N=128 M=4 A=8 f32: alpha sequence identical to oracle32 for {same} iterations "
          f"(oracle32 vs oracle64: {same64});  J rel err over those: {nrel(out['Jout'][0][:same], r['Jout'][:same]):.2e};"
          f"  J[30] kernel {out['Jout'][0][30]:.3f} oracle32 {r['Jout'][30]:.3f} oracle64 {r64['Jout'][30]:.3f}")
    print("        kernel  ", a)
    print("        oracle32", ar)
    print("        oracle64", a64)
    s.close()


def ee_cost_level(path, N=64):
    """End-effector cost family (oracle pinned: tests/test_phase_pins.py): per-knot H_k, g_k, cost of the setup kernel on a random trajectory, and the first costs of a solve."""
    rng = np.random.default_rng(9)
    kw = dict(N=N, M=4, A=8, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, max_iter=10, ee_cost=1, ignore_max_rho_exit=0, Q_EE2=0.02, QF_EE2=3.0, Q_xEE=0.05)
    s = pyddp.Solver(pyddp.default_config(4, _lib_path=path, dtype=0, **kw), _lib_path=path)
    o32, o64 = Oracle(default_cfg(4, **kw), np.float32), Oracle(default_cfg(4, **kw), np.float64)
    x = rng.normal(0, 0.8, (N, 14)).astype(np.float32); u = rng.normal(0, 5.0, (N, 7)).astype(np.float32)
    goal = np.zeros(14, np.float32); goal[:6] = [0.4, -0.1, 0.7, 0.1, -0.2, 0.3]
    s.load(x, u, goal)
    H, g, ck = s.get("H").reshape(N, 21, 21), s.get("g").reshape(N, 21), s.get("costk")
    r32 = [o32.ee_cost_grad(x[k], u[k], goal[:6], k) for k in range(N)]; r64 = [o64.ee_cost_grad(x[k], u[k], goal[:6], k) for k in range(N)]
    c32 = [o32.ee_cost(x[k], u[k], goal[:6], k) for k in range(N)]; c64 = [o64.ee_cost(x[k], u[k], goal[:6], k) for k in range(N)]
    for name, got, a, b in (("H", H, np.stack([r[0] for r in r32]), np.stack([r[0] for r in r64])), ("g", g, np.stack([r[1] for r in r32]), np.stack([r[1] for r in r64])),
                            ("cost", ck, np.asarray(c32), np.asarray(c64))):
        print(f"  ee    {name:20s} kernel-vs-oracle32 {nrel(got, a):.2e}   oracle32-vs-oracle64 {nrel(a, b):.2e}   kernel-vs-oracle64 {nrel(got, b):.2e}")
    x0 = np.zeros((N, 14), np.float32); x0[:, 1] = 0.7; x0[:, 3] = -0.8; x0[:, 5] = 0.75
    u0 = np.full((N, 7), 0.01, np.float32); xg = np.zeros(14, np.float32); xg[:3] = [0.45, 0.15, 0.75]
    out = s.solve(x0, u0, xg)
    a32 = o32.run_ilqr_gpusem(x0.ravel(), u0.ravel(), xg); a64 = o64.run_ilqr_gpusem(x0.astype(np.float64).ravel(), u0.astype(np.float64).ravel(), xg.astype(np.float64))
    same = 0
    while same < 10 and out["alphaOut"][0][same] == a32["alphaOut"][same]:
        same += 1
    same64 = 0
    while same64 < 10 and a32["alphaOut"][same64] == a64["alphaOut"][same64]:
        same64 += 1
    print(f"  ee    solve N={N} f32: alpha sequence identical to oracle32 for {same} iterations (oracle32 vs oracle64: {same64});  "
          f"J[1] kernel {out['Jout'][0][1]:.5f} oracle32 {a32['Jout'][1]:.5f} oracle64 {a64['Jout'][1]:.5f}")


def quadrotor_tolerance_sweep(path, iters=25):
    """BASELINE configs[4]: quadrotor, N=256, RK3, 16 alphas, M=4, T=4 s -- float32 and float64 from the same stored inputs:
    per iteration the relative deviation of J and the max deviations of x and K^T at the end, and the first iteration where the step-size index differs."""
    kw = dict(N=256, M=4, A=16, integrator=3, total_time=4.0, max_iter=iters, tol_cost=0.0)
    noise = np.random.default_rng(11).normal(0, 0.001, (256, 12))
    res = {}
    for dt in (np.float32, np.float64):
        s = pyddp.Solver(pyddp.default_config(3, _lib_path=path, dtype=0 if dt == np.float32 else 1, **kw), _lib_path=path)
        x0, u0, xg = example_inputs(3, 256, dt, noise=noise)
        res[dt] = s.solve(x0, u0, xg)
    a32, a64 = res[np.float32]["alphaOut"][0], res[np.float64]["alphaOut"][0]
    J32, J64 = res[np.float32]["Jout"][0], res[np.float64]["Jout"][0]
    first = next((i for i in range(iters + 1) if a32[i] != a64[i]), iters + 1)
    dev = [abs(float(J32[i]) / float(J64[i]) - 1) for i in range(iters + 1)]
    print(f"  quad  N=256 RK3 A=16 M=4: float32 vs float64 step-size indices agree for {first} of {iters + 1} entries; J[0] {J64[0]:.4f} -> J[{iters}] f64 {J64[iters]:.4f} f32 {J32[iters]:.4f}")
    print("        |J32/J64 - 1| per iteration: " + " ".join(f"{d:.1e}" for d in dev))
    print(f"        final x: {nrel(res[np.float32]['x'][0], res[np.float64]['x'][0]):.2e}   final K^T: {nrel(res[np.float32]['KT'][0], res[np.float64]['KT'][0]):.2e}")


if __name__ == "__main__":
    paths = sys.argv[1:] or [pyddp.library_path()]
    for path in paths:
        print(os.path.basename(path))
        plant_level(path); backward_pass_level(path); backward_pass_level(path, N=128, M=4); solver_level(path); ee_cost_level(path); quadrotor_tolerance_sweep(path)
