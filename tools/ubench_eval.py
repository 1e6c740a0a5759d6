"""GPU micro-benchmark of the plant evaluation kernels (run under rocprofv3 --kernel-trace --stats).
   PDDP_EVAL_GRID=1 -> one wave evaluates everything serially (latency per evaluation);  default grid -> throughput."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parallel-ddp_amd"))
import pyddp
what = int(sys.argv[1]); count = int(sys.argv[2])
s = pyddp.Solver(pyddp.default_config(4, N=16, M=1, A=1, wafr_urdf=1))
rng = np.random.default_rng(0)
x = np.concatenate([rng.normal(0, 1, (count, 7)), rng.normal(0, 0.5, (count, 7))], axis=1); u = rng.normal(0, 20, (count, 7))
for _ in range(3):
    s.plant_eval(what, x, u)
