#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory (gpurun_out/prof_<tag>) into profiles/<name>/:
   kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim), bench_line.json, and summary.md with the per-kernel
   average durations and the HBM traffic per launch from the separate --pmc passes, corrected as
   MI355X_MICROARCH.md ("HBM") prescribes and calibrated on k_hbm_calib_dword's known byte count.

usage: tools/prof_summarize.py gpurun_out/prof_<tag> profiles/<name>
"""
import collections
import csv
import json
import os
import shutil
import statistics
import sys


def short(name):
    n = name.replace("void ", "").replace("pddp::", "")
    return n.split("<")[0].split("(")[0]


def counters(path):
    d = collections.defaultdict(list)
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        d[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return d


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "run_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    line = None
    for ln in open(os.path.join(src, "trace_bench.log")):
        if ln.startswith("{"):
            line = json.loads(ln)
    if line:
        json.dump(line, open(os.path.join(dst, "bench_line_under_rocprof.json"), "w"), indent=1)
    stats = list(csv.DictReader(open(os.path.join(src, "trace", "run_kernel_stats.csv"))))
    fetch = counters(os.path.join(src, "pmc_fetch", "run_counter_collection.csv"))
    write = counters(os.path.join(src, "pmc_write", "run_counter_collection.csv"))
    sq = counters(os.path.join(src, "pmc_sq", "run_counter_collection.csv"))
    calib_bytes = float(1 << 30)
    cf = fetch.get(("k_hbm_calib_dword", "FETCH_SIZE"))
    cw = write.get(("k_hbm_calib_dword", "WRITE_SIZE"))
    f_scale = calib_bytes / (statistics.mean(cf) * 1024) if cf else 2.0     # guide: x2 on gfx950
    w_scale = calib_bytes / (statistics.mean(cw) * 1024) if cw else 1.0
    out = ["# rocprofv3 summary: " + os.path.basename(dst), ""]
    if line:
        out += ["bench line of the profiled command (timings under the profiler):", "```", json.dumps(line), "```", ""]
    out += [f"HBM counter calibration on k_hbm_calib_dword (1 GiB read + 1 GiB written, one dword per lane): FETCH_SIZE[KB] x1024 x **{f_scale:.3f}** = bytes read, "
            f"WRITE_SIZE[KB] x1024 x **{w_scale:.3f}** = bytes written (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of a streaming read on gfx950).", "",
            "| kernel | calls | avg ms (--stats) | % | HBM read / launch | HBM written / launch | traffic / launch |", "|---|---|---|---|---|---|---|"]
    summary = {}
    for r in stats:
        k = short(r["Name"])
        if not k.startswith("k_"):
            continue
        fr = statistics.mean(fetch[(k, "FETCH_SIZE")]) * 1024 * f_scale if (k, "FETCH_SIZE") in fetch else None
        wr = statistics.mean(write[(k, "WRITE_SIZE")]) * 1024 * w_scale if (k, "WRITE_SIZE") in write else None
        tr = (fr or 0) + (wr or 0) if fr is not None or wr is not None else None
        summary[k] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) * 1e-6, "hbm_read_bytes": fr, "hbm_write_bytes": wr, "traffic_bytes": tr}
        fmt = lambda v: "-" if v is None else f"{v / 1e6:.1f} MB"
        out.append(f"| {k} | {r['Calls']} | {float(r['AverageNs']) * 1e-6:.4f} | {float(r['Percentage']):.1f} | {fmt(fr)} | {fmt(wr)} | {fmt(tr)} |")
    if sq:
        out += ["", "SQ counters (separate pass), mean per launch:", "", "| kernel | " + " | ".join(sorted({c for (_, c) in sq})) + " |",
                "|---|" + "---|" * len({c for (_, c) in sq})]
        for k in sorted({k for (k, _) in sq if k.startswith("k_")}):
            out.append(f"| {k} | " + " | ".join(f"{statistics.mean(sq[(k, c)]):.3g}" if (k, c) in sq else "-" for c in sorted({c for (_, c) in sq})) + " |")
    # HBM bytes per launch of each sweep phase, for bench.py's roofline.traffic (only valid for the profiled batch size)
    if line:
        tr = lambda ks: sum((summary.get(k, {}).get("traffic_bytes") or 0) for k in ks) or None
        phase_bytes = {"bp": tr(["k_bp_lg", "k_bp"]), "fp": tr(["k_sweep_lg", "k_fp_lg", "k_fp"]), "ls": tr(["k_ls"]), "nis": tr(["k_nis_lg", "k_nis"])}
        json.dump({"batch": line["config"]["problems_per_gpu"], "source": os.path.basename(dst), "phase_bytes": phase_bytes},
                  open(os.path.join(os.path.dirname(os.path.abspath(dst)), "roofline_traffic.json"), "w"), indent=1)
    open(os.path.join(dst, "summary.md"), "w").write("\n".join(out) + "\n")
    json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
