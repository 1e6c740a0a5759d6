#!/usr/bin/env python3
"""profiles/<dir>/counters.txt (tools/pmc_rows.sh: separate rocprofv3 --pmc passes over `python bench.py --rows`) -> profiles/rows_traffic.json, the per-kernel HBM traffic
and instruction counts bench.py attaches to the rows beside the headline.  usage: tools/make_rows_traffic.py profiles/r04_rows"""
import json, os, re, subprocess, sys
src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "parallel-ddp_amd"))
import pyddp  # noqa: E402
bid_file = os.path.join(src, "build_id.json")          # written on the GPU box by tools/pmc_rows.sh: the tree / library the counters were taken on
build = json.load(open(bid_file)) if os.path.exists(bid_file) else pyddp.build_id()
try:
    build["git_head"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
except Exception:      # noqa: BLE001
    build["git_head"] = None
out = {"build": build, "source": f"{src}/counters.txt (tools/pmc_rows.sh: kernel trace + FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* passes of `python bench.py --rows`; HBM bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB)", "kernels": {}}
for ln in open(os.path.join(src, "counters.txt")):
    if ln.startswith("#") or " grid=" not in ln:
        continue
    name, rest = ln.split(" grid=", 1)
    kv = dict(re.findall(r"(\w+)=([0-9.e+]+)", "grid=" + rest))
    short = name.split("<")[0]
    plant = "quad" if "Quad" in name else "cart" if "Cart" in name else "pend" if "Pend" in name else "arm"
    dtype = "f64" if "double" in name else "f32"
    rec = {"name": name.strip(), "grid": int(float(kv["grid"])), "launches": int(float(kv["launches"]))}
    if "hbm_read_MB" in kv:
        rec["hbm_read_bytes"] = float(kv["hbm_read_MB"]) * 1e6; rec["hbm_write_bytes"] = float(kv["hbm_write_MB"]) * 1e6
    for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if c in kv:
            rec[c] = float(kv[c])
    out["kernels"].setdefault(f"{short}|{plant}|{dtype}", []).append(rec)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(src.rstrip("/"))), "rows_traffic.json"), "w"), indent=1)
print({k: [r["grid"] for r in v] for k, v in out["kernels"].items()})
