#!/usr/bin/env python3
"""Runs only the `widening` rows of bench.py (MPC wrapper, end-effector cost family) and prints them as JSON --
the command that profiles/r01g_ee_cost_* was recorded from (rocprofv3 --kernel-trace --stats -- python tools/bench_widening.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench.widening_rows(0)))
