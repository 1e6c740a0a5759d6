import faulthandler, sys, os
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd"))
import numpy as np
import pyddp
def P(*a): print(*a, flush=True)
mode = sys.argv[1]
if "torch" in mode:
    import torch; torch.cuda.set_device(0); P("torch ok", torch.cuda.is_available())
for graph in (0, 1):
    cfg = pyddp.default_config(4, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=64, max_iter=100, use_graph=graph)
    s = pyddp.Solver(cfg); P("created graph=", graph)
    sys.path.insert(0, ROOT)
    import bench
    x0, u0, xg = bench.example_inputs(128, np.random.default_rng(0), 64)
    s.load(x0, u0, xg); P("loaded")
    s.iterate(3); s.sync(); P("iterated")
    out = s.store(); P("stored", out["Jout"][0][:4], out["alphaOut"][0][:4])
    P("time", s.time_sweeps(5, phases=True))
    if "torch" in mode:
        t = torch.as_tensor(s.device_array("Jout"), device="cuda"); P("as_tensor", t[:4])
    s.close()
P("done")
