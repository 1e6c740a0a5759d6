import sys, os, json, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"parallel-ddp_amd"))
import pyddp, bench
lib = sys.argv[1] if len(sys.argv) > 1 else None
def run(name, B, N, dtype, ee, A=8):
    kw = dict(wafr_urdf=1, tol_cost=0.0, total_time=0.5, use_graph=1)
    cfg = pyddp.default_config(4, N=N, M=4, A=A, batch=B, max_iter=100, dtype=dtype, ee_cost=ee, mpc_mode=ee, _lib_path=lib, **kw)
    s = pyddp.Solver(cfg, _lib_path=lib)
    if ee: x0,u0,xg = bench.ee_inputs(N, np.random.default_rng(1), B)
    else: x0,u0,xg = bench.example_inputs(N, np.random.default_rng(1), B)
    if dtype: x0,u0,xg = x0.astype(np.float64),u0.astype(np.float64),xg.astype(np.float64)
    s.load(x0,u0,xg); s.set_benchmark_mode(1); s.iterate(5); s.sync()
    k = s.time_kernels(20)
    print(name, " ".join(f"{n}={ms*1e3:.1f}us" for n,ms in k if n), flush=True)
    s.close()
run("ee f32 B=64 N=64", 64, 64, 0, 1)
run("ee f32 B=4096 N=64", 4096, 64, 0, 1)
run("joint f64 B=1 N=128", 1, 128, 1, 0)
run("joint f64 B=64 N=128", 64, 128, 1, 0)
