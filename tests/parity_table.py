"""Prints the float32 bar table of tests/test_fp32_bar.py (worst kernel error and the oracle32 error at that point, per phase and quantity)
for a backend:  python tests/parity_table.py hip|hostsim   (the GPU run's output is committed under profiles/)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "parallel-ddp_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import test_fp32_bar as t  # noqa: E402

backend = sys.argv[1] if len(sys.argv) > 1 else "hostsim"
np.seterr(all="ignore")
cases = (("Kuka N=128 A=8 M=4 float32, matrix-core backward pass (diagonal-H path) + thread-lane forward pass / setup", 4, t.KUKA, dict(bp="mx", fp="tl"), 40, False),
         ("Kuka N=128 A=8 M=4 float32, matrix-core backward pass (full-H path) + lane-group forward pass / setup", 4, t.KUKA, dict(bp="mx", fp="lg"), 40, True),
         ("Kuka N=128 A=8 M=4 float32, lane-group backward pass + thread-lane forward pass / setup", 4, t.KUKA, dict(bp="lg", fp="tl"), 40, False),
         ("cart-pole N=128 A=8 M=4 RK3 float32", 2, t.CART, {}, 12, False))
for name, plant, kw, env, its, full_h in cases:
    ens = env.get("bp") == "mx"
    rows, fails, ints = t.run_bar(backend, plant, kw, env, 5, its, full_h=full_h, ensemble=ens)
    if ens:
        r = t._run_bar.bp_ratio
        print("   backward pass against the float32 noise-floor ensemble (8 float32 evaluations of the reference algorithm: strict, FMA-contracted, each on 3 draws of one-ulp-jittered inputs): err(kernel)/max(ensemble): median %.2f, 90th pct %.2f, max %.2f; inside 1.5x: %.1f %%" % (np.median(r), np.percentile(r, 90), r.max(), 100 * np.mean(r <= 1.5)))
    print(f"{name}: {its} iterations teacher-forced from oracle64, {len(rows)} comparisons, integers identical: {ints}, outside the bar: {len(fails)}")
    print("   phase quantity   worst err(kernel32,oracle64)   err(oracle32,oracle64) there   iteration")
    for k, v in sorted(t.summarize(rows).items()):
        print("   %-5s %-8s %.2e   %.2e   %d %s" % (k[0], k[1], v[0], v[1], v[2], v[3]))
    for f in fails[:10]:
        print("   OUTSIDE", f)
