#!/usr/bin/env python3
"""Generates tests/golden/fig8_trace.json from DATA the reference's test directory holds: the recorded lock-step run in test/WAFR_fig8.py (the list `a` of
[tool point xyz, goal xyz, eNorm, running average error, vNorm] per control cycle, printed by fig8Simulate's debugMode 1, examples/WAFR_MPC_examples.cu:167)
and the published summary in its comment (:5-6).  Every 8th record plus the first and last three are kept.  Run in the build container (needs /root/reference)."""
import json
import os
import re

SRC = "/root/reference/test/WAFR_fig8.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fig8_trace.json")
text = open(SRC).read()
a = json.loads(re.search(r"^a = (\[.*\])\s*$", text, re.M).group(1))
avg = float(re.search(r"Average tracking error: \[([0-9.]+)\]", text).group(1))
keep = sorted(set(list(range(0, len(a), 8)) + [0, 1, 2, len(a) - 3, len(a) - 2, len(a) - 1]))
json.dump({"source": "test/WAFR_fig8.py", "records_total": len(a), "published_average_tracking_error": avg, "index": keep, "records": [a[i] for i in keep]},
          open(OUT, "w"))
print("wrote", OUT, len(keep), "records")
