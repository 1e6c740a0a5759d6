#!/usr/bin/env python3
"""profiles/roofline_traffic.json and profiles/<name>/summary.md from the counter passes of tools/pmc_pass.sh (gpurun_out/pmc_<tag>).
usage: tools/make_traffic_json.py <tag> <profiles subdir> <batch> [handle options: "library defaults" | "lean"]
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the factor 2 on FETCH_SIZE is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md "HBM": rocprofv3 tallies 128-byte read requests as 64 bytes)."""
import collections, csv, glob, json, os, statistics, subprocess, sys
tag, sub, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
handle_options = sys.argv[4] if len(sys.argv) > 4 else "library defaults"     # "lean" for passes taken with bench.py --lean-ctg
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "pmc_" + tag)
d = collections.defaultdict(list)
for f in sorted(glob.glob(src + "/*/run_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("pddp::", "").split("<")[0].split("(")[0]
        if k.startswith("k_"):
            d[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
kern = sorted({k for k, _ in d})
m = lambda k, c: statistics.mean(d[(k, c)]) if (k, c) in d else None
# the build these counters belong to: the pass writes the identity of the tree / library it profiled on the GPU box (tools/pmc_pass.sh -> build_id.json); it is stored
# here together with the commit the working tree stood at, and bench.py attaches the traffic to a run only when the sources of the tree it runs from are the same
sys.path.insert(0, os.path.join(root, "parallel-ddp_amd"))
import pyddp  # noqa: E402
bid_file = os.path.join(src, "build_id.json")
build = json.load(open(bid_file)) if os.path.exists(bid_file) else pyddp.build_id()
here = pyddp.build_id()
if build.get("sources") != here["sources"]:
    print(f"WARNING: the counters were taken on sources {build.get('sources')}, this tree is {here['sources']}: bench.py will report them as stale", file=sys.stderr)
try:
    build["git_head"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    build["git_dirty"] = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "parallel-ddp_amd/csrc", "include", "parallel-ddp_amd/Makefile"], text=True).strip())
except Exception:      # noqa: BLE001
    build["git_head"] = None
out = {"batch": batch, "handle_options": handle_options, "build": build, "source": f"profiles/{sub} (tools/pmc_pass.sh: separate rocprofv3 --pmc passes of `python bench.py --no-cpu-baseline --no-latency --no-convergence --no-lean-row --steps 20`)", "kernels": {}}
lines = ["# Counter passes (rocprofv3 --pmc, kernel-trace only) of the bench sweep, per kernel, averages per launch", "",
         "| kernel | HBM read MB (2 x FETCH_SIZE) | HBM write MB | VALU instr / wave | MFMA instr / wave | VALU-active share of wave time | waiting on memory (s_waitcnt) | issue stalls | MFMA pipe busy share |",
         "|---|---|---|---|---|---|---|---|---|"]
for k in kern:
    f, w = m(k, "FETCH_SIZE"), m(k, "WRITE_SIZE")
    waves, wc = m(k, "SQ_WAVES"), m(k, "SQ_WAVE_CYCLES")
    ent = {}
    if f is not None and w is not None:
        ent["hbm_bytes_per_launch"] = 2 * f * 1024 + w * 1024
    c = {}
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"):
        if m(k, name) is not None:
            c[name] = m(k, name)
    if wc:
        for name in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if m(k, name) is not None:
                c[name + "_share_of_wave_cycles"] = round(m(k, name) / wc, 4)
    busy, mf = m(k, "SQ_BUSY_CYCLES"), m(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    if busy and mf is not None:
        c["mfma_pipe_busy_share"] = round(mf / (busy / 32.0 * 1024), 4)   # SQ_BUSY_CYCLES sums 32 shader-engine counters; MFMA busy cycles sum over the 1024 SIMDs
    ent["counters"] = c
    out["kernels"][k] = ent
    g = lambda v, s=1.0, fmt="%.1f": "-" if v is None else fmt % (v * s)
    lines.append(f"| {k} | {g(f, 2 * 1024 / 1e6)} | {g(w, 1024 / 1e6)} | {g(None if not waves or m(k, 'SQ_INSTS_VALU') is None else m(k, 'SQ_INSTS_VALU') / waves, 1, '%.0f')} | "
                 f"{g(None if not waves or m(k, 'SQ_INSTS_MFMA') is None else m(k, 'SQ_INSTS_MFMA') / waves, 1, '%.0f')} | {g(c.get('SQ_ACTIVE_INST_VALU_share_of_wave_cycles'), 100, '%.0f %%')} | "
                 f"{g(c.get('SQ_WAIT_ANY_share_of_wave_cycles'), 100, '%.0f %%')} | {g(c.get('SQ_WAIT_INST_ANY_share_of_wave_cycles'), 100, '%.0f %%')} | {g(c.get('mfma_pipe_busy_share'), 100, '%.0f %%')} |")
os.makedirs(os.path.join(root, "profiles", sub), exist_ok=True)
open(os.path.join(root, "profiles", sub, "summary.md"), "w").write("\n".join(lines) + "\n")
open(os.path.join(root, "profiles", sub, "counters_raw.txt"), "w").write(open(os.path.join(src, "summary.txt")).read() if os.path.exists(os.path.join(src, "summary.txt")) else "")
json.dump(out, open(os.path.join(root, "profiles", "roofline_traffic.json"), "w"), indent=1)
print("\n".join(lines))
