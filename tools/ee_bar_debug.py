#!/usr/bin/env python3
"""Runs the end-effector float32 bar flow of tests/test_fp32_bar.py once and prints every failing comparison (debugging aid).  env PDDP_POISON_LDS=1: with poisoned LDS."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import sys as _s, os as _o; _s.path.insert(0, _o.path.dirname(_o.path.abspath(__file__))); import _sel; _sel.install()      # PDDP_BP / PDDP_FP / ... on this tool's command line -> pddp_config.kernels (tools/_sel.py; the library reads no environment)
import test_fp32_bar as t
env = dict(bp="mx", fp="tl") if len(sys.argv) > 1 and sys.argv[1] == "large" else {}
rows, fails, ints_ok, names = t.run_bar_ee("hip", dict(t.EE_KW), env, 10, True)
print("kernels", list(names), "ints_ok", ints_ok, "rows", len(rows), "fails", len(fails))
c = collections.Counter((it, ph.split("[")[0], nm, "nan" if np.isnan(ek) else "num") for it, ph, nm, ek, eo, ok in fails)
for k, v in sorted(c.items()): print("  ", k, v)
r = t._run_bar.bp_ratio
print("bp ratio: share <= 1.5:", float(np.mean(r <= 1.5)), "max", float(r.max()))
w = t.summarize(rows)
print({f"{k[0]}.{k[1]}": f"{v[0]:.1e}|{v[1]:.1e}@{v[2]}" for k, v in sorted(w.items()) if k[0] == "bp"})
