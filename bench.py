#!/usr/bin/env python3
"""bench.py -- DDP iterations/s of the HIP hot path on BASELINE.json's headline configuration.

Workload (config.workload): BASELINE configs[2] -- KUKA iiwa14 (n=14, m=7), N=128 knots, 8 line-search alphas x 4
multiple-shooting segments, Euler, float (algType, config.cuh:74), joint-space cost, inputs of
examples/WAFR_iLQR_examples.cu:69-121 with a seeded N(0, 0.001) velocity noise per problem.

A "step" is one DDP sweep (backward pass -> forward sweep+rollout+cost for every alpha -> line search + accept/reject
-> next-iteration setup) over the rank's batch of independent problems.  All inputs are resident in HBM before the
timed region starts (pddp_load is untimed); the exit tests are live (TOL_COST 0, MAX_ITER >= warmup+steps like
the example's TOL_COST 0 / MAX_ITER 100), nothing is cached or skipped.

value = problems x sweeps / second summed over all ranks (weak scaling: the per-GPU batch is fixed).  The single-
problem latency figures the reference reports (iterations/s of ONE solve, ms to convergence) are in "latency".

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the batch axis is sharded, the solves are
independent, and the only exchange is one all-gather of the per-problem cost per poll (pyddp.shard).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "parallel-ddp_amd"))

import pyddp  # noqa: E402
from pyddp import shard  # noqa: E402

BENCH_BATCH = 16384       # independent problems per GPU of the headline line (tests/test_fp32_bar.py runs its what-the-bench-runs cases at this size)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 matrix-core peak of MI355X (256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~ 8 TB/s
PHASES = ("bp", "fp", "ls", "nis")                      # the four phases of a sweep, in launch order (per-phase figures of the latency block)
# reference phase whose algorithmic bytes a kernel covers, by kernel-name prefix
KERNEL_BYTES = {"k_bp": ("k_bp", 1.0), "k_sweep": ("k_sweep", 1.0), "k_fp": ("k_sim", 1.0), "k_ls": ("k_ls", 1.0), "k_win": ("k_nis_copies", 1.0), "k_nis": ("k_nis_derivs", 1.0)}


def example_inputs(N, rng, count):
    """examples/WAFR_iLQR_examples.cu:34-53,80-82,93-94,117 (arm, USE_WAFR_URDF): x0,u0,xGoal for `count` problems."""
    PI = 3.14159
    x = np.zeros((count, N, 14), np.float32)
    x[:, :, :7] = np.asarray([-0.5 * PI, 0.25 * PI, 0.167 * PI, -0.167 * PI, 0.125 * PI, 0.167 * PI, 0.5 * PI], np.float32)
    x[:, :, 7:] = rng.normal(0, 0.001, (count, N, 7)).astype(np.float32)
    u = np.zeros((count, N, 7), np.float32)
    u[:] = np.asarray([0.0, -102.9832, 11.1968, 47.0724, 2.5993, -7.0290, -0.0907], np.float32)
    g = np.zeros((count, 14), np.float32)
    g[:, :7] = np.asarray([0, 0, 0, -0.25 * PI, 0, 0.25 * PI, 0.5 * PI], np.float32)
    return x, u, g


def cpu_baseline(budget_s=14.0):
    """The oracle's runiLQR_CPU restatement (reference thread-per-phase structure) timed on this host's cores on a
    bounded sample of the same workload.  The oracle is the CHECKER/baseline only -- never the measured product.
    The reference sizes its thread counts from hardware_concurrency (config.cuh:148-161); on a many-core host that
    over-subscribes 128 knots, so the same code is also timed with CPU_CORES = 8 and the FASTER of the two is reported."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle, default_cfg, example_inputs as ora_inputs
    hw = os.cpu_count() or 1
    results = {}
    for cores in sorted({hw, min(hw, 8)}):
        o = Oracle(default_cfg(4, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=30, cores=cores, spawn_threads=1), np.float32)
        rng = np.random.default_rng(99)
        iters, ms, solves = 0, 0.0, 0
        t0 = time.time()
        while time.time() - t0 < budget_s / 2:
            x0, u0, xg = ora_inputs(4, 128, np.float32, noise=rng.normal(0, 0.001, (128, 14)))
            r = o.run_ilqr_cpu(x0, u0, xg)
            iters += r["iters"]; ms += r["t_total_ms"] - r["t_init_ms"]; solves += 1
        results[cores] = (iters / (ms * 1e-3), solves)
    best = max(results, key=lambda c: results[c][0])
    return {"value": round(results[best][0], 2), "unit": "DDP iterations/s", "cores": best, "kind": "port",
            # both settings as structured fields: CPU_CORES = hardware_concurrency() is the reference's own default (config.cuh:148)
            "by_cpu_cores": {str(c): {"iterations_per_s": round(results[c][0], 2), "solves": results[c][1]} for c in sorted(results)},
            "reference_default_cpu_cores": hw, "host_hardware_threads": hw,
            "sample": "runiLQR_CPU semantics (first-acceptable serial line search, pthreads created per phase like the reference), Kuka N=128 A=8 M=4, "
                      "30 iterations per solve; " + "; ".join(f"CPU_CORES={c}: {results[c][1]} solves, {results[c][0]:.1f} it/s" for c in sorted(results))
                      + f"; host has {hw} hardware threads"}


FP32_LANE_PEAK_TFLOPS = 157.3  # the SIMDs' float32 lanes: vector FMAs (64 lanes x 2 flop per 2 cycles x 1024 SIMDs x 2.4 GHz) and the f32 matrix instruction share them
BP_FLOPS_PER_KNOT = 2 * 16268  # dense products of one backward-pass knot, n = 14, m = 7 (multiply-adds): W = P'[A B] 4116, H += [A B]'W 6174, Huu^-1 343, K | du 735,
                               # T1 686, P+ 2744, A - B K | B du 1470   (bpHelpers.cuh:39-334)


def design_bytes_per_problem(kernel, N, M, A, n, m, keep_ctg=True, elem=4):
    """THIS design's algorithmic HBM bytes of one launch, per problem: every array the kernel has to read or write ONCE, in the layout it is kept in.
    (bytes, accounting) or None for a kernel without a statement.  The ratio of the counter traffic to this figure is the kernel's wasted traffic (VERDICT r5 task 5)."""
    nm = n + m
    if kernel.startswith("k_bp_mfma"):
        per_knot = (n // 2) * nm * elem + nm * elem + n * m * elem + m * elem + ((n * n + n) * elem if keep_ctg else 0)      # compact [A B] 588 + g 84 | K 392 + du 28 | [P | p] 840
        return per_knot * (N - M), f"{per_knot} B per knot (compact [A B] {n // 2 * nm * elem} + g {nm * elem} read; K {n * m * elem} + du {m * elem}" + (f" + [P | p] {(n * n + n) * elem}" if keep_ctg else "") + f" written) x {N - M} knots"
    if kernel.startswith("k_fp_tl"):
        rd = (N - 1) * (n * m + m + m) * elem + N * n * elem + A * (M - 1) * n * elem + n * elem      # gains, feed-forward, current controls (N - 1 knots), current states, the candidates' segment start states, goal
        wr = N * A * (nm + 1) * elem + 2 * A * M * elem + A * (M - 1) * n * elem                     # 22-float records of every candidate and knot, partial cost / defect sums, boundary defects
        return rd + wr, (f"read {rd} B: K {(N - 1) * n * m * elem} + du {(N - 1) * m * elem} + u {(N - 1) * m * elem} + x {N * n * elem} (shared by the {A} candidates) + start states {A * (M - 1) * n * elem} + goal; "
                         f"written {wr} B: {N} x {A} records of {nm + 1} floats {N * A * (nm + 1) * elem} + partial sums + boundary defects")
    if kernel.startswith("k_nis_tl"):
        rd = N * (nm + 1) * elem + (M - 1) * n * elem                                                  # the accepted candidate's records, its boundary defects
        wr = (N - 1) * (n // 2) * nm * elem + N * nm * elem + N * n * elem + N * m * elem + (M - 1) * n * elem   # compact [A B], g, adopted x, u, d
        return rd + wr, (f"read {rd} B: the accepted candidate's {N} records + boundary defects; written {wr} B: compact [A B] {(N - 1) * (n // 2) * nm * elem} + g {N * nm * elem} + adopted x {N * n * elem}, u {N * m * elem}, d")
    if kernel.startswith("k_sweep_maps"):
        b_ = M * 16 * 16 * elem + A * (M - 1) * n * elem + (M - 1) * n * elem + n * elem
        return b_, f"{M} segment maps of 16 x 16 read, {A} x {M - 1} start states written"
    return None


def roofline_record(dom_name, dom_ms, kern, alg_of, B, N, M, traffic, counters, tsrc, sweep_bytes, s_per_step, keep_ctg=True, per_kernel_traffic=None, A=8, n=14, m=7):
    """`roofline` of the bench line, for the kernel with the longest average launch.

    Top level = the roofline that BINDS that kernel.  Every heavy kernel of this sweep is bound by the SIMDs' float32 lanes, not by HBM: the matrix-core
    backward pass needs 26 flop per byte it really moves (ridge of 157.3 TFLOP/s over 8 TB/s: 19.7), the thread-lane kernels hundreds.  So:
      bound "mfma": achieved = ALGORITHMIC flop of the launch / its duration against the dense float32 matrix-core peak (157.3 TFLOP/s) -- k_bp_mfma;
      bound "valu": achieved = vector instructions issued x 128 flop (64 lanes, FMA-equivalent: an UPPER bound of the useful flop) / duration against the
                    float32 vector peak, the same 157.3 TFLOP/s of the same lanes -- the thread-lane kernels (needs the counter pass);
      bound "hbm":  achieved = HBM bytes really moved (counter pass) / duration against 8 TB/s -- anything else.
    `hbm` always carries the counter traffic as a rate; `reference_equivalent` is the rate at which the launch gets through the bytes of the REFERENCE's phase
    accounting -- most of which this design no longer moves, so it exceeds what the memory system does and is NOT a bandwidth."""
    dur = dom_ms * 1e-3
    roof = {"kernel": dom_name, "avg_launch_ms": round(dom_ms, 5), "traffic": traffic, "traffic_source": tsrc, "counters": counters}
    if dom_name.startswith("k_bp_mfma"):
        knots = B * (N - M)                                                    # every block walks N/M - 1 knots
        useful = BP_FLOPS_PER_KNOT * knots
        mfma_rec = {"bound": "mfma", "achieved": round(useful / dur / 1e12, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(useful / dur / 1e12 / MFMA_F32_PEAK_TFLOPS, 5), "algorithmic_flop_per_launch": useful,
                    "accounting": f"{BP_FLOPS_PER_KNOT} flop per knot (the dense products of backPassKern for n=14, m=7) x {N - M} knots x {B} problems"}
        # Which roof binds is a question of arithmetic intensity against the ridge (157.3 TFLOP/s / 8 TB/s = 19.7 flop per byte).  THIS design's algorithmic bytes of
        # a knot: reads compact [A B] 588 + cost gradient 84, writes K 392 + du 28, and -- when every cost-to-go slot is written, the library
        # default and the reference's output set -- [P | p] 784 + 56.  With those stores the kernel moves more than a byte per 15 flop: the HBM roof is the nearer one.
        per_knot = 588 + 84 + 392 + 28 + (784 + 56 if keep_ctg else 0)
        alg_bytes = float(per_knot) * knots
        hbm_rec = {"bound": "hbm", "achieved": round(alg_bytes / dur / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / dur / 1e9 / HBM_PEAK_GBS, 5),
                   "algorithmic_bytes_per_launch": alg_bytes,
                   "accounting": f"{per_knot} bytes per knot of this design's data layout (compact [A B] 588 + g 84 read; K 392 + du 28" + (" + [P | p] 840" if keep_ctg else "") +
                                 f" written) x {N - M} knots x {B} problems; `traffic` = what the counters saw"}
        intensity = useful / (traffic if traffic else alg_bytes)
        roof["arithmetic_intensity_flop_per_byte"] = round(intensity, 2)
        roof["ridge_flop_per_byte"] = round(MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9), 2)
        if intensity < roof["ridge_flop_per_byte"]:
            roof.update(hbm_rec); roof["mfma"] = {k: mfma_rec[k] for k in ("achieved", "peak", "unit", "frac", "accounting")}
        else:
            roof.update(mfma_rec); roof["hbm_algorithmic"] = {k: hbm_rec[k] for k in ("achieved", "peak", "unit", "frac", "accounting")}
        if counters and counters.get("SQ_INSTS_MFMA"):
            n_mx = counters["SQ_INSTS_MFMA"]
            issued = n_mx * 2048.0                                             # v_mfma_f32_16x16x4_f32: 16 x 16 x 4 multiply-adds
            roof["issued"] = {"matrix_instructions_per_knot": round(n_mx / knots, 2), "TFLOPs": round(issued / dur / 1e12, 2), "frac": round(issued / dur / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
            if counters.get("SQ_INSTS_VALU"):
                # lane model: a float32 matrix instruction occupies the SIMD's float32 lanes for 32 cycles and excludes the vector instructions of the other resident waves
                # (tools/probes/mfma_valu_overlap.hip); a vector instruction issues over 2 cycles (MI355X_MICROARCH.md).  SQ_INSTS_VALU counts both kinds.
                n_v = counters["SQ_INSTS_VALU"] - n_mx
                cyc = (32.0 * n_mx + 2.0 * n_v) / 1024.0
                roof["lanes"] = {"model": "cycles per SIMD = 32 x matrix instructions + 2 x other vector instructions, 1024 SIMDs at 2.4 GHz",
                                 "matrix_ms": round(32.0 * n_mx / 1024.0 / 2.4e6, 4), "vector_ms": round(2.0 * n_v / 1024.0 / 2.4e6, 4),
                                 "frac_of_launch": round(cyc / 2.4e6 / dom_ms, 4)}
    elif counters and counters.get("SQ_INSTS_VALU") and (dom_name.startswith("k_fp_tl") or dom_name.startswith("k_nis_tl")):
        issued = counters["SQ_INSTS_VALU"] * 128.0
        roof.update({"bound": "valu", "achieved": round(issued / dur / 1e12, 3), "peak": FP32_LANE_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(issued / dur / 1e12 / FP32_LANE_PEAK_TFLOPS, 5),
                     "accounting": "vector instructions issued (SQ_INSTS_VALU of the counter pass) x 128 flop -- FMA-equivalent issue rate, an upper bound of the useful flop; "
                                   "one thread per rollout / knot: no matrix-core work, ~0.1 byte of HBM traffic per flop"})
    else:
        ach = (traffic / dur / 1e9) if traffic else None
        roof.update({"bound": "hbm", "achieved": None if ach is None else round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 5), "accounting": "HBM bytes moved (counter pass) / launch duration"})
    if traffic:
        roof["hbm"] = {"traffic_GBs": round(traffic / dur / 1e9, 1), "frac_of_peak": round(traffic / dur / 1e9 / HBM_PEAK_GBS, 4), "peak_GBs": HBM_PEAK_GBS}
    ref_bytes = alg_of(dom_name) * B
    roof["reference_equivalent"] = {
        "bytes_per_launch": ref_bytes, "GBs": round(ref_bytes / dur / 1e9, 1), "ratio_to_hbm_peak": round(ref_bytes / dur / 1e9 / HBM_PEAK_GBS, 4),
        "whole_sweep_GBs": round(sweep_bytes / s_per_step / 1e9, 1), "whole_sweep_ratio_to_hbm_peak": round(sweep_bytes / s_per_step / 1e9 / HBM_PEAK_GBS, 4),
        "note": "rate at which the launch gets through the bytes the REFERENCE's phase decomposition moves (SURVEY.md 8(d): every array once per phase and per alpha). "
                "It is not a bandwidth: the design reads shared operands once for all alphas, keeps [A B] compact, never reads the diagonal cost Hessian and never writes "
                "A - B K -- a ratio above 1 means exactly that."}
    # every kernel of the sweep: live duration, this design's algorithmic bytes, the counter traffic (when the counter file belongs to this build) and their ratio
    roof["per_kernel"] = {}
    for nm_, ms in kern:
        e = {"ms": round(ms, 5)}
        db = design_bytes_per_problem(nm_, N, M, A, n, m, keep_ctg)
        if db:
            e["algorithmic_bytes_per_launch"] = float(db[0]) * B
            e["algorithmic_GBs"] = round(db[0] * B / (ms * 1e-3) / 1e9, 1) if ms else None
            e["accounting"] = db[1] + f" x {B} problems"
        tr = (per_kernel_traffic or {}).get(nm_)
        if tr is not None:
            e["traffic"] = tr
            e["traffic_GBs"] = round(tr / (ms * 1e-3) / 1e9, 1) if ms else None
            if db:
                e["traffic_over_algorithmic"] = round(tr / (db[0] * B), 4)
        roof["per_kernel"][nm_] = e
    if traffic and roof.get("algorithmic_bytes_per_launch"):
        roof["traffic_over_algorithmic"] = round(traffic / roof["algorithmic_bytes_per_launch"], 4)
    return roof


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- the same command the driver uses
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>), one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BENCH_BATCH, help="independent problems per GPU")
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--lean-ctg", action="store_true", help="headline handle with pddp_config.boundary_cost_to_go_only = 1 (cost-to-go written at the block boundaries only) instead of the "
                                                            "library default that writes every knot's d_P / d_p like backPassKern (bpHelpers.cuh:253,396)")
    ap.add_argument("--keep-ctg", action="store_true", help="(the default since round 5; kept so that older command lines still parse)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-default-options", "--no-lean-row", dest="no_lean_row", action="store_true",
                    help="skip the side row `lean_cost_to_go` (a second handle with boundary_cost_to_go_only = 1); profiling runs: kernel statistics and counters of the headline handle only")
    ap.add_argument("--no-convergence", action="store_true", help="skip the whole-batch time-to-convergence block (profiling runs: keeps the kernel statistics to the timed sweeps)")
    ap.add_argument("--lib", default=None, help="alternative libpddp build (measurement of build variants only)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass (profiles/)")
    ap.add_argument("--calibrate-hbm", action="store_true", help="also launch the known-byte-count copy kernel (for --pmc passes)")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3"],
                    help="config2 (default, the headline): BASELINE configs[2]; config3: BASELINE configs[3] -- 64 Kuka MPC rollouts with the end-effector cost, "
                         "64 / N per GPU, exchanges through the C ABI's own RCCL collectives (pddp_comm_*)")
    ap.add_argument("--rows", action="store_true", help="only the rows beside the headline (other BASELINE configs, widening): for rocprofv3 passes over THEIR kernels "
                                                        "(tools/pmc_rows.sh); prints them as one JSON object, not the bench line")
    args = ap.parse_args()
    if args.rows:
        print(json.dumps({"other_baseline_configs": other_config_rows(0), "widening": widening_rows(0), "bit_exact_family": bit_exact_family_row(0)}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return relaunch_under_torchrun(args.gpus)
    if args.workload == "config3":
        return config3_sharded(args)

    ctx = shard.init_from_env(args.gpus)              # rank, world, local_rank; torch.distributed if world > 1
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    torch.cuda.set_device(ctx.device)
    K, W, B = args.steps, args.warmup, args.batch
    N, M, A, n, m = 128, 4, 8, 14, 7

    # The headline handle is created with the LIBRARY DEFAULTS -- what allocateMemory_GPU of the facade creates: the backward pass writes every knot's cost-to-go like
    # backPassKern does (d_P / d_p, bpHelpers.cuh:253,396; the MPC warm start shifts the whole arrays, MPCHelpers.cuh:620-622).  --lean-ctg (and the side row
    # `lean_cost_to_go`) = pddp_config.boundary_cost_to_go_only: only the block-boundary slots a later pass reads (same bits in every output of runiLQR_GPU,
    # tests/test_f64_benched_family.py).
    lean = 1 if args.lean_ctg else 0
    cfg = pyddp.default_config(4, N=N, M=M, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=B, boundary_cost_to_go_only=lean,
                               max_iter=max(100, K + W + 1), device=ctx.device, use_graph=args.graph, _lib_path=args.lib)
    s = pyddp.Solver(cfg, _lib_path=args.lib)
    rng = np.random.default_rng(1234 + ctx.rank)      # every rank owns different problems
    x0, u0, xg = example_inputs(N, rng, B)
    s.load(x0, u0, xg)                                # untimed: inputs are in HBM from here on
    if args.calibrate_hbm:
        s.hbm_calibration(1 << 30, 3)

    s.iterate(W); s.sync()
    shard.barrier(ctx); torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.iterate(K)
    s.sync(); torch.cuda.synchronize()
    shard.barrier(ctx); torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    t = shard.max_over_ranks(ctx, t_local)

    # ---- what happened in the timed sweeps (honesty: accepted / rejected iterations, decrease of J)
    out = s.store()
    done, iters = s.status()
    acc = np.mean([(out["alphaOut"][b][W + 1: W + K + 1] >= 0).mean() for b in range(min(B, 64))])
    J_all = shard.allgather_costs(ctx, s.device_array("Jout"), B, cfg.max_iter + 2, int(iters.min()) - 1)   # RCCL exchange

    # ---- per-KERNEL durations with HIP events on the solver's own stream: the SAME sweeps again (reload, same warm-up), launched kernel by kernel
    # with an event after every launch (pddp_time_kernels); the names are the kernels this handle's selection launches
    s.load(x0, u0, xg)
    s.iterate(W); s.sync()
    kern = s.time_kernels(K)
    alg = pyddp.algorithmic_bytes_per_kernel(n, m, N, A, M, 4)
    has_win = any(nm.startswith("k_win") or nm.startswith("k_adopt") for nm, _ in kern)
    def alg_of(name):
        key = next(v for p_, v in KERNEL_BYTES.items() if name.startswith(p_))
        extra = alg["k_nis_copies"] if (name.startswith("k_nis") and not has_win) else 0.0     # the setup kernel adopts the winner itself
        return alg[key[0]] * key[1] + extra
    dom_name, dom_ms = max(kern, key=lambda kv: kv[1])
    bytes_launch = alg_of(dom_name) * B
    achieved = bytes_launch / (dom_ms * 1e-3) / 1e9
    sweep_bytes = sum(alg.values()) * B
    # HBM traffic and issue counters of the dominant kernel: separate rocprofv3 --pmc passes of THIS command (tools/pmc_pass.sh), committed
    # under profiles/ -- not measured inside this run (a counter pass serialises the kernels and cannot share a run with the timing)
    traffic, counters, tsrc = args.traffic_bytes, None, "command line" if args.traffic_bytes else None
    tfile = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    sweep_traffic = None
    per_kernel_traffic = {}
    if os.path.exists(tfile):
        tj = json.load(open(tfile))
        # a counter pass says something about THIS launch only if it was taken at the same batch on the same handle options (files written before round 5: lean handles)
        # AND on the build that is running (VERDICT r5 task 7): the sources of the tree the pass profiled against the sources of the tree this script runs from
        built = (tj.get("build") or {})
        same_build = built.get("sources") is not None and built.get("sources") == pyddp.build_id(args.lib).get("sources") and not args.lib
        if traffic is None and not same_build:
            tsrc = f"stale ({built.get('git_head') or 'no build identity'}: profiles/roofline_traffic.json was taken on other sources -- re-run tools/profile_round.sh)"
        if same_build and tj.get("batch") == B and tj.get("handle_options", "lean") == ("lean" if lean else "library defaults"):
            per_kernel_traffic = {nm: ent.get("hbm_bytes_per_launch") for nm, ent in tj.get("kernels", {}).items() if ent.get("hbm_bytes_per_launch") is not None}
            ent = tj.get("kernels", {}).get(dom_name)
            if traffic is None and ent:
                traffic, counters, tsrc = ent.get("hbm_bytes_per_launch"), ent.get("counters"), f"{tj.get('source')}; build {built.get('git_head')} sources {built.get('sources')}"
            # whole-sweep HBM traffic (the north-star's "achieved HBM-bandwidth fraction"): the counter traffic of EVERY kernel of one sweep
            per = [tj["kernels"].get(nm, {}).get("hbm_bytes_per_launch") for nm, _ in kern]
            if per and all(v is not None for v in per):
                sweep_traffic = float(sum(per))
    roof = roofline_record(dom_name, dom_ms, kern, alg_of, B, N, M, traffic, counters, tsrc, sweep_bytes, t_local / K, keep_ctg=not lean, per_kernel_traffic=per_kernel_traffic, A=A, n=n, m=m)

    line = {"metric": "DDP iterations/sec (Kuka iiwa14 N=128, 8 alphas, 4 shooting segments)", "value": round(ctx.world * B * K / t, 1),
            "unit": "DDP iterations/s", "n_gpus": ctx.world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * t / K, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Kuka iiwa14 RBD (n=14,m=7), N=128, 8 alphas x 4 shooting segments, Euler, fp32, "
                                   "joint cost, T=0.5 s, WAFR example inputs + N(0,1e-3) velocity noise",
                       "problems_per_gpu": B, "problems_total": ctx.world * B, "hipgraph": bool(args.graph), "sharding": "batch axis, no data-path collective",
                       "cost_to_go_slots_written": "block boundaries only (pddp_config.boundary_cost_to_go_only)" if lean else "all", "handle_options": "lean" if lean else "library defaults"},
            "rccl_ranks_seen": ctx.world if ctx.backend == "nccl" else (ctx.world if ctx.world == 1 else 0), "dist_backend": ctx.backend or "none",
            "accepted_fraction_in_timed_sweeps": round(float(acc), 3),
            "J_first_last_mean": [round(float(J_all[:, 0].mean()), 3), round(float(J_all[:, -1].mean()), 3)],
            # top-level copies of the HBM figures (the north-star metric; nested keys do not survive every consumer of this line)
            "hbm_frac_of_peak_dominant_kernel": (roof.get("hbm") or {}).get("frac_of_peak"),
            "hbm_frac_of_peak_whole_sweep": None if sweep_traffic is None else round(sweep_traffic / (t_local / K) / 1e9 / HBM_PEAK_GBS, 4),
            "hbm_bytes_per_sweep": sweep_traffic, "hbm_traffic_source": tsrc,
            "roofline": roof}

    s.close()
    # ---- side row: the same sweeps on a handle with boundary_cost_to_go_only = 1 (interior cost-to-go slots not written: they are no output of runiLQR_GPU and no
    # input of a later phase of the SAME solve).  Until round 4 this was the headline handle; it is NOT reference-equivalent and stays a side figure.
    if not lean and not args.no_lean_row:
        cfg_d = pyddp.default_config(4, N=N, M=M, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=B, boundary_cost_to_go_only=1,
                                     max_iter=max(100, K + W + 1), device=ctx.device, use_graph=args.graph, _lib_path=args.lib)
        sd = pyddp.Solver(cfg_d, _lib_path=args.lib)
        sd.load(x0, u0, xg)
        sd.iterate(W); sd.sync()
        shard.barrier(ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        sd.iterate(K); sd.sync(); torch.cuda.synchronize()
        shard.barrier(ctx); torch.cuda.synchronize()
        td = shard.max_over_ranks(ctx, time.perf_counter() - t0)
        sd.load(x0, u0, xg); sd.iterate(W); sd.sync()
        kd = sd.time_kernels(K)
        line["lean_cost_to_go"] = {"cost_to_go_slots_written": "block boundaries only (pddp_config.boundary_cost_to_go_only = 1; not the reference's output set)",
                                   "value": round(ctx.world * B * K / td, 1), "unit": "DDP iterations/s",
                                   "ms_per_step": round(1e3 * td / K, 4), "per_kernel_ms": {nm: round(ms, 5) for nm, ms in kd}}
        sd.close()
    # ---- what the first run on more than one GPU should show (stated BEFORE it is measured: no node with more than one device has run this yet)
    line["multi_gpu"] = {
        "mode": "batch axis sharded round-robin, every step size of a problem on its rank, no data-path collective (SURVEY 8(e) mode R); one all-reduce(max) of the exit flags per poll",
        "measured_ranks": ctx.world,
        "expected": {"headline_weak_scaling": "value ~ N x the one-GPU value (fixed 16384 problems per GPU, no exchange inside the timed sweeps; the barrier pair around them costs microseconds against "
                                              "~0.1 s): efficiency 0.97-1.0 at N = 2, 4, 8, bounded by the box-to-box spread of the pool (~7 %) since the slowest rank sets the time",
                     "config3_64_rollouts_strong_scaling": "~1.0 x at every N: 64 problems take 0.148 ms per sweep on one GPU and 8 problems 0.13 ms -- the sweep of so few problems is the latency of its "
                                                           "kernels' dependency chains, not throughput (bench.py --workload config3); sharding it buys nothing and is offered for placement only"}}
    if not args.no_convergence:
        line["convergence"] = batch_convergence(ctx, args, torch, x0, u0, xg, B, N, M, A)

    # the secondary figures must never cost the headline line: a failure in one of them is reported in its place
    def guarded(fn, *a):
        try:
            return fn(*a)
        except Exception as e:      # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"}

    if ctx.rank == 0 and ctx.world == 1 and not args.no_latency:
        line["latency"] = guarded(latency_single_problem, ctx.device)
        line["widening"] = guarded(widening_rows, ctx.device)
        line["other_baseline_configs"] = guarded(other_config_rows, ctx.device)
        line["bit_exact_family"] = guarded(bit_exact_family_row, ctx.device)
    if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = guarded(cpu_baseline)
    elif ctx.rank == 0:
        line["cpu_baseline"] = None
    if ctx.rank == 0:
        print(json.dumps(line), flush=True)
    shard.finalize(ctx)


def config3_sharded(args):
    """BASELINE configs[3]: 64 concurrent rollouts x 8 alphas (Kuka, N=64, M=4, MPC_MODE, end-effector cost), rank g owns the rollouts {r : r % N == g}
    (64 / N per GPU), no collective per sweep; the exit poll (all-reduce) and the cost table (all-gather) go through the C ABI's native RCCL
    exchanges (include/pddp.h "multi-GPU").  STRONG scaling: the 64 rollouts are the whole job.  Not the headline line (that is configs[2])."""
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    total, N = 64, 64
    if total % world:
        raise SystemExit("config3: 64 rollouts must split evenly over the ranks")
    B = total // world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    comm = pyddp.Comm(rank, world, local, id_path=f"/tmp/pddp_bench_{os.environ.get('MASTER_PORT', 'solo')}.id" if world > 1 else None)
    K, W = args.steps, args.warmup
    cfg = pyddp.default_config(4, N=N, M=4, A=8, batch=B, max_iter=max(100, K + W + 1), ee_cost=1, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5,
                               ignore_max_rho_exit=0, device=local, use_graph=args.graph)
    s = pyddp.Solver(cfg)
    x_all, u_all, g_all = ee_inputs(N, np.random.default_rng(77), total)
    mine = list(range(rank, total, world))
    s.load(x_all[mine], u_all[mine], g_all[mine])
    s.set_benchmark_mode(1)                     # keep every rollout iterating through the timed sweeps (no early exit changes the work per step)
    s.iterate(W); s.sync(); comm.barrier()
    t0 = time.perf_counter()
    s.iterate(K); s.sync(); comm.barrier()
    t = comm.max_over_ranks(time.perf_counter() - t0)
    s.set_benchmark_mode(0)
    done = comm.all_done(s)
    costs = comm.allgather_costs(s)
    if rank == 0:
        print(json.dumps({"metric": "DDP iterations/sec (Kuka MPC shape N=64, 64 rollouts x 8 alphas x 4 segments, end-effector cost)", "value": round(total * K / t, 1),
                          "unit": "DDP iterations/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * t / K, 4), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "BASELINE configs[3]: 64 concurrent Kuka MPC rollouts (N=64, 8 alphas x 4 segments, MPC_MODE, end-effector cost), "
                                                 f"{B} per GPU, round-robin", "problems_total": total, "sharding": "batch axis; RCCL all-reduce per poll + all-gather of the cost table (pddp_comm_*)"},
                          "rccl_ranks_seen": comm.world, "all_done_after_timed_sweeps": bool(done),
                          "J_first_last_mean": [round(float(costs[:, 0].mean()), 4), round(float(costs[:, 1].mean()), 4)]}), flush=True)
    s.close(); comm.close()


def batch_convergence(ctx, args, torch, x0, u0, xg, B, N, M, A):
    # ---- wall clock to convergence of the whole sharded batch (BASELINE metric, second half): TOL_COST 1e-4 (config.cuh:85-87), MAX_ITER 100;
    # every rank iterates its own problems, the ranks agree on "all done" with one max-reduce per poll (pyddp.shard.all_done)
    cfg2 = pyddp.default_config(4, N=N, M=M, A=A, wafr_urdf=1, tol_cost=1e-4, total_time=0.5, batch=B, max_iter=100, device=ctx.device,
                                boundary_cost_to_go_only=1 if args.lean_ctg else 0, use_graph=args.graph, _lib_path=args.lib)
    s2 = pyddp.Solver(cfg2, _lib_path=args.lib)
    s2.load(x0, u0, xg)
    s2.iterate(1); s2.sync(); s2.load(x0, u0, xg)      # graph instantiation outside the timed region
    shard.barrier(ctx); torch.cuda.synchronize()
    t0 = time.perf_counter()
    polls = 0
    while polls < 64:
        s2.iterate(8); polls += 1
        done2, iters2 = s2.status()
        if shard.all_done(ctx, done2):
            break
    torch.cuda.synchronize()
    t_conv = shard.max_over_ranks(ctx, time.perf_counter() - t0)
    out2 = s2.store()
    conv_it = [convergence_iteration(out2["Jout"][b], out2["alphaOut"][b], int(iters2[b])) for b in range(min(B, 256))]
    res = {"tol_cost": 1e-4, "max_iter": 100, "problems_total": ctx.world * B, "ms_until_every_problem_exited": round(1e3 * t_conv, 3),
                           "sweeps_enqueued": 8 * polls, "median_exit_iteration_rank0": float(np.median(iters2)),
                           "median_iterations_to_convergence_rank0": float(np.median(conv_it)),
                           "exit_reasons_rank0": {str(k): int((done2 == k).sum()) for k in (1, 2, 3)}}
    s2.close()
    return res


def convergence_iteration(J, alpha, iters):
    """Reference-style convergence point of one cost trace (examples/WAFR_iLQR_examples.cu:150-187)."""
    rej = 0
    for i in range(1, iters + 1):
        if alpha[i] < 0:
            rej += 1
            if rej == 3:
                return i - 2
        else:
            rej = 0
            if J[i - 1] > 0 and (J[i - 1] - J[i]) / J[i - 1] < 1e-4:
                return i
    return iters


def latency_single_problem(device):
    """The reference's own figures of merit for ONE problem (examples/WAFR_iLQR_examples.cu:141-187): iterations/s of a
    solve = iter / (tTime - initTime), and ms to convergence with the default TOL_COST 1e-4 (config.cuh:85-87)."""
    res = {}
    rng = np.random.default_rng(4321)
    # the third entry is the shape of the reference's only published timing (test/WAFR_fig8.py:5-12: Kuka MPC, N=64, A=16, M=4, about 1.36 ms
    # per iteration on a Pascal-class GPU, end-effector cost) -- here with the joint-space cost, so an indication, not a like-for-like number
    for name, tol, max_iter, Nk, Ak in (("fixed_100_iterations", 0.0, 100, 128, 8), ("to_convergence_tol_1e-4", 1e-4, 100, 128, 8),
                                        ("published_shape_N64_A16_M4_fixed_100_iterations", 0.0, 100, 64, 16)):
        cfg = pyddp.default_config(4, N=Nk, M=4, A=Ak, wafr_urdf=1, tol_cost=tol, total_time=0.5, batch=1, max_iter=max_iter,
                                   device=device, use_graph=1)
        s = pyddp.Solver(cfg)
        its, mss, Js, convs = [], [], [], []
        for rep in range(21):
            x0, u0, xg = example_inputs(Nk, rng, 1)
            r = s.solve_timed(x0, u0, xg)
            if rep == 0:
                continue                           # first solve instantiates the graph
            its.append(r["iters"]); mss.append(r["ms_loop"]); Js.append(r["J_final"])
            convs.append(convergence_iteration(r["Jout"][0], r["alphaOut"][0], r["iters"]))
        res[name] = {"median_iterations": float(np.median(its)), "median_ms": round(float(np.median(mss)), 3),
                     "iterations_per_s": round(float(np.median(np.asarray(its) / (np.asarray(mss) * 1e-3))), 1),
                     "median_J_final": round(float(np.median(Js)), 3), "solves": len(its),
                     # first iteration with relative decrease < 1e-4, or the first of 3 consecutive rejections
                     "median_iterations_to_convergence": float(np.median(convs)),
                     "median_ms_to_convergence": round(float(np.median(np.asarray(convs) * np.asarray(mss) / np.asarray(its))), 3),
                     "ms_per_iteration": round(float(np.median(np.asarray(mss) / np.asarray(its))), 4)}
        # per-phase medians like the reference prints them (BP / sweep+sim / line search / NIS; DDPWrappers.cuh:54-105) and the (time, J) trace of one
        # solve, so that another convergence threshold can be applied: kernel durations from HIP events, launched kernel by kernel
        x0, u0, xg = example_inputs(Nk, np.random.default_rng(4321), 1)
        pt = s.solve_phase_timed(x0, u0, xg)
        n_it = int(pt["iters"][0])
        ph = pt["phase_ms"][:4, :n_it]                                    # (row 4 = the linear sweep's kernel alone, a part of row 1)
        res[name]["per_phase_median_us"] = {k: round(float(np.median(ph[i]) * 1e3), 1) for i, k in enumerate(PHASES)}
        if name == "to_convergence_tol_1e-4":
            cum = np.cumsum(ph.sum(axis=0))
            res[name]["trace"] = {"J": [round(float(v), 4) for v in pt["Jout"][0][: n_it + 1]], "alpha": [int(v) for v in pt["alphaOut"][0][: n_it + 1]],
                                  "cumulative_kernel_ms": [0.0] + [round(float(v), 4) for v in cum]}
        s.close()
    return res


def closed_form_inputs(plant, N, rng, count):
    """examples/WAFR_iLQR_examples.cu:19-33,72-78,88-90,111-115: start, nominal control and goal of the pendulum / cart-pole / quadrotor + N(0, 1e-3) noise on the velocities"""
    n, m = {1: (2, 1), 2: (4, 1), 3: (12, 4)}[plant]
    x = np.zeros((count, N, n), np.float32)
    x[:, :, n // 2:] = rng.normal(0, 0.001, (count, N, n // 2)).astype(np.float32)
    if plant == 3:
        x[:, :, 2] = 0.5
    u = np.full((count, N, m), 1.22625 if plant == 3 else 0.01, np.float32)
    g = np.zeros((count, n), np.float32)
    g[:] = {1: [3.1416, 0.0], 2: [0.0, 3.1416, 0.0, 0.0], 3: [7.0, 10.0, 0.5] + [0.0] * 9}[plant]
    return x, u, g


F64_LANE_PEAK_TFLOPS = 78.6   # double-precision vector FMAs: 4 cycles per wave64 instruction (tools/probes/valu_ops.hip)


def row_roofline(per_kernel_ms, plant, dtype):
    """`roofline` of a row beside the headline, for its kernel with the longest launch: live HIP-event duration from this run, HBM bytes and instruction counts per launch
    from the committed counter pass over the same command (profiles/rows_traffic.json <- tools/pmc_rows.sh, tools/make_rows_traffic.py).  None of these kernels is bound
    by HBM or uses the matrix cores except k_bp_mfma: the closed-form plants' kernels are scalar code on the vector ALU, so the top level is the vector-issue roofline
    (instructions issued x 128 flop, FMA-equivalent: an UPPER bound of the useful flop) and `hbm` carries the measured traffic as a rate."""
    tfile = os.path.join(ROOT, "profiles", "rows_traffic.json")
    if not os.path.exists(tfile):
        return None
    tab = json.load(open(tfile))
    built = tab.get("build") or {}
    if built.get("sources") is None or built.get("sources") != pyddp.build_id().get("sources"):      # counters of another build say nothing about these launches (VERDICT r5 task 7)
        return {"traffic": None, "traffic_source": f"stale ({built.get('git_head') or 'no build identity'}: profiles/rows_traffic.json was taken on other sources -- re-run tools/pmc_rows.sh)",
                "per_kernel": {nm: {"ms": round(ms, 5)} for nm, ms in per_kernel_ms.items()}}
    per = {}
    for nm, ms in per_kernel_ms.items():
        recs = tab["kernels"].get(f"{nm}|{plant}|{dtype}") or tab["kernels"].get(f"{nm}_cf|{plant}|{dtype}")      # (pddp_time_kernels reports the closed-form plants' k_sweep_maps_cf as k_sweep_maps)
        if not recs or not ms:
            continue
        r = max(recs, key=lambda q: q["grid"])                          # the row with the device full is the largest launch of that kernel in the pass
        e = {"ms": round(ms, 5)}
        if "hbm_read_bytes" in r:
            tb = r["hbm_read_bytes"] + r["hbm_write_bytes"]
            e.update({"hbm_bytes_per_launch": tb, "hbm_GBs": round(tb / (ms * 1e-3) / 1e9, 1), "hbm_frac_of_peak": round(tb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        if "SQ_INSTS_VALU" in r:
            peak = F64_LANE_PEAK_TFLOPS if dtype == "f64" else FP32_LANE_PEAK_TFLOPS
            e.update({"vector_instructions_per_launch": r["SQ_INSTS_VALU"], "matrix_instructions_per_launch": r.get("SQ_INSTS_MFMA", 0.0),
                      "issued_TFLOPs": round(r["SQ_INSTS_VALU"] * 128.0 / (ms * 1e-3) / 1e12, 3), "issued_frac_of_vector_peak": round(r["SQ_INSTS_VALU"] * 128.0 / (ms * 1e-3) / 1e12 / peak, 4)})
        per[nm] = e
    if not per:
        return None
    dom = max(per, key=lambda k: per[k]["ms"])
    d = per[dom]
    peak = F64_LANE_PEAK_TFLOPS if dtype == "f64" else FP32_LANE_PEAK_TFLOPS
    roof = {"kernel": dom, "avg_launch_ms": d["ms"], "traffic": d.get("hbm_bytes_per_launch"), "traffic_source": f"{tab['source']}; build {built.get('git_head')} sources {built.get('sources')}", "per_kernel": per}
    if d.get("matrix_instructions_per_launch"):
        issued = d["matrix_instructions_per_launch"] * 2048.0
        roof.update({"bound": "mfma", "achieved": round(issued / (d["ms"] * 1e-3) / 1e12, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(issued / (d["ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 5), "accounting": "matrix instructions ISSUED (counter pass) x 2048 flop / live launch duration"})
    elif "issued_TFLOPs" in d:
        roof.update({"bound": "valu", "achieved": d["issued_TFLOPs"], "peak": peak, "unit": "TFLOP/s", "frac": d["issued_frac_of_vector_peak"],
                     "accounting": "vector instructions issued (counter pass) x 128 flop / live launch duration against the vector FMA peak of the element type -- an upper bound of the useful flop"})
    else:
        roof.update({"bound": "hbm", "achieved": d.get("hbm_GBs"), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d.get("hbm_frac_of_peak")})
    if d.get("hbm_GBs") is not None:
        roof["hbm"] = {"traffic_GBs": d["hbm_GBs"], "frac_of_peak": d["hbm_frac_of_peak"], "peak_GBs": HBM_PEAK_GBS}
    return roof


def other_config_rows(device):
    """BASELINE configs[1] (cart-pole N=128, 8 alphas, M=4) and configs[4] (quadrotor N=256, RK3, 16 alphas, float32 and float64) with the device full: whole-batch sweeps/s,
    per-kernel HIP-event times, and the rate at which the sweep gets through the reference's byte accounting (NOT a bandwidth -- see roofline.reference_equivalent).
    configs[0] (pendulum, CPU path) is the reference's CPU-only case: libpddp_cpu's runiLQR_CPU is timed in its place.  Not part of `value`."""
    res = {}
    rng = np.random.default_rng(99)
    for name, plant, B, kw, dtype in (("config1_cartpole_N128_A8_M4_rk3_f32", 2, 16384, dict(N=128, M=4, A=8, integrator=3, total_time=4.0), 0),
                                      ("config4_quadrotor_N256_A16_M4_rk3_f32", 3, 16384, dict(N=256, M=4, A=16, integrator=3, total_time=4.0), 0),
                                      ("config4_quadrotor_N256_A16_M4_rk3_f64", 3, 8192, dict(N=256, M=4, A=16, integrator=3, total_time=4.0), 1)):
        n, m = {2: (4, 1), 3: (12, 4)}[plant]
        s = pyddp.Solver(pyddp.default_config(plant, batch=B, max_iter=100, tol_cost=0.0, dtype=dtype, device=device, use_graph=1, **kw))
        x0, u0, xg = closed_form_inputs(plant, kw["N"], rng, B)
        s.load(x0, u0, xg)
        s.iterate(3); s.sync()
        ms_plain, _ = s.time_sweeps(10, phases=False)
        s.load(x0, u0, xg); s.iterate(3); s.sync()
        kern = s.time_kernels(10)
        out = s.store()
        acc = float(np.mean([(out["alphaOut"][b][1:14] >= 0).mean() for b in range(min(B, 64))]))
        alg = pyddp.algorithmic_bytes(n, m, kw["N"], kw["A"], kw["M"], 8 if dtype else 4)
        res[name] = {"problems": B, "iterations_per_s": round(B * 10 / (ms_plain * 1e-3), 1), "ms_per_sweep": round(ms_plain / 10, 4),
                     "per_kernel_ms": {nm: round(ms, 5) for nm, ms in kern}, "accepted_fraction": round(acc, 3),
                     "reference_equivalent": {"bytes_per_sweep_per_problem": sum(alg.values()), "GBs": round(sum(alg.values()) * B / (ms_plain / 10 * 1e-3) / 1e9, 1),
                                              "ratio_to_hbm_peak": round(sum(alg.values()) * B / (ms_plain / 10 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        res[name]["roofline"] = row_roofline({nm: ms for nm, ms in kern if nm}, {2: "cart", 3: "quad"}[plant], "f64" if dtype else "f32")
        s.close()
    return res


def bit_exact_family_row(device, B=BENCH_BATCH):
    """The price of bit-identity with the float32 oracle: the headline workload on the kernel family that executes the reference host path's IEEE
    operations one for one -- lane groups, built with -ffp-contract=off, float32 backward pass BIT-IDENTICAL to oracle32 (tests/test_lanegroup.py) -- selected through
    pddp_config.kernels (bp = lg, fp = lg).  The headline runs the matrix-core / thread-lane family instead, which lives under the float32 bar (tests/test_fp32_bar.py)."""
    N, M, A = 128, 4, 8
    cfg = pyddp.default_config(4, N=N, M=M, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5, batch=B, max_iter=100, device=device, use_graph=1, kernels=dict(bp="lg", fp="lg"))
    s = pyddp.Solver(cfg)
    x0, u0, xg = example_inputs(N, np.random.default_rng(1234), B)
    s.load(x0, u0, xg)
    s.iterate(3); s.sync()
    ms_plain, _ = s.time_sweeps(10, phases=False)
    s.load(x0, u0, xg); s.iterate(3); s.sync()
    kern = s.time_kernels(10)
    s.close()
    return {"problems": B, "kernels": "lane groups, -ffp-contract=off (pddp_config.kernels: bp = lg, fp = lg)", "iterations_per_s": round(B * 10 / (ms_plain * 1e-3), 1),
            "ms_per_sweep": round(ms_plain / 10, 4), "per_kernel_ms": {nm: round(ms, 5) for nm, ms in kern if nm},
            "parity": "float32 backward pass bit-identical to the float32 oracle (tests/test_lanegroup.py); rollouts / setup the reference's operation order"}


def ee_inputs(N, rng, count):
    """The MPC example's start (utils/exampleUtils.cuh:40-58: constant pose, u = 0.01, K = 0) with per-problem tool-point goals on a lemniscate --
    SURVEY.md section 8(d) config 4: "64 rollouts" = 64 independent problems, goal r at phase r/64 of the figure."""
    x = np.zeros((count, N, 14), np.float32); x[:, :, 1] = 0.7; x[:, :, 3] = -0.8; x[:, :, 5] = 0.75
    x[:, :, :7] += rng.normal(0, 0.01, (count, 1, 7)).astype(np.float32)
    u = np.full((count, N, 7), 0.01, np.float32)
    ph = 2 * np.pi * np.arange(count) / count
    g = np.zeros((count, 14), np.float32); g[:, 0] = 0.55; g[:, 1] = 0.20 * np.sin(ph); g[:, 2] = 0.45 + 0.12 * np.sin(2 * ph)
    return x, u, g


def widening_rows(device):
    """SURVEY.md section 8(f) rows N1 / N2, measured: the receding-horizon wrapper and the end-effector cost family (EE_COST 1, MPC_MODE 1,
    the configuration of examples/WAFR_MPC_examples.cu:4-37).  Not part of `value`."""
    res = {}
    rng = np.random.default_rng(77)
    kw = dict(wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, ignore_max_rho_exit=0, device=device, use_graph=1)
    # (1) BASELINE configs[3] stage 4b: 64 independent rollouts x 8 alphas, N=64, M=4, end-effector cost -- whole-batch sweeps/s
    B, N = 64, 64
    s = pyddp.Solver(pyddp.default_config(4, N=N, M=4, A=8, batch=B, max_iter=100, ee_cost=1, **kw))
    x0, u0, xg = ee_inputs(N, rng, B)
    s.load(x0, u0, xg)
    s.set_benchmark_mode(1)
    s.iterate(5); s.sync()
    ms_tot, ms_phase = s.time_sweeps(30, phases=True)
    ms_plain, _ = s.time_sweeps(30, phases=False)
    s.set_benchmark_mode(0)
    res["config4b_64_rollouts_ee_cost_N64_A8_M4"] = {"problems": B, "iterations_per_s": round(B * 30 / (ms_plain * 1e-3), 1), "ms_per_sweep": round(ms_plain / 30, 4),
                                                      "per_phase_ms": {k: round(v / 30, 5) for k, v in zip(PHASES, ms_phase)}}
    s.close()
    # (1b) the same problem shape with the GPU full: 4096 problems in flight, per-phase HIP-event times and the dominant phase's roofline fraction
    B2 = 4096
    s = pyddp.Solver(pyddp.default_config(4, N=N, M=4, A=8, batch=B2, max_iter=100, ee_cost=1, **kw))
    x0, u0, xg = ee_inputs(N, rng, B2)
    s.load(x0, u0, xg)
    s.set_benchmark_mode(1)
    s.iterate(5); s.sync()
    ms_tot, ms_phase = s.time_sweeps(20, phases=True)
    ms_plain, _ = s.time_sweeps(20, phases=False)
    s.set_benchmark_mode(0)
    alg = pyddp.algorithmic_bytes(14, 7, N, 8, 4, 4, ee_cost=True)
    per = [v / 20 for v in ms_phase]
    s.load(x0, u0, xg); s.set_benchmark_mode(1); s.iterate(5); s.sync()
    kern = s.time_kernels(10)
    s.set_benchmark_mode(0)
    res["ee_cost_4096_problems_N64_A8_M4"] = {"problems": B2, "iterations_per_s": round(B2 * 20 / (ms_plain * 1e-3), 1), "ms_per_sweep": round(ms_plain / 20, 4),
                                              "per_phase_ms": {k: round(v, 5) for k, v in zip(PHASES, per)}, "per_kernel_ms": {nm: round(ms, 5) for nm, ms in kern},
                                              "reference_equivalent": {"bytes_per_sweep_per_problem": sum(alg.values()), "GBs": round(sum(alg.values()) * B2 / (ms_plain / 20 * 1e-3) / 1e9, 1),
                                                                       "ratio_to_hbm_peak": round(sum(alg.values()) * B2 / (ms_plain / 20 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    res["ee_cost_4096_problems_N64_A8_M4"]["roofline"] = row_roofline({nm: ms for nm, ms in kern if nm}, "arm", "f32")
    s.close()
    # (2) the published shape with its own cost (test/WAFR_fig8.py:5-12: Kuka MPC, N=64, A=16, M=4, EE cost, ~1.36 ms per iteration published):
    #     one problem, a warm start to convergence, then control cycles of 4 iterations shifted by one knot (runiLQR_MPC_GPU)
    for name, ee in (("mpc_cycle_published_shape_ee_cost", 1), ("mpc_cycle_published_shape_joint_cost", 0)):
        s = pyddp.Solver(pyddp.default_config(4, N=N, M=4, A=16, batch=1, max_iter=100, ee_cost=ee, **kw))
        x0, u0, xg = ee_inputs(N, rng, 1)
        if not ee:
            xg[0, :7] = [0.5, 0.6, -0.3, -0.9, 0.2, 0.7, 0.1]
        s.load(x0, u0, xg)
        t0 = time.perf_counter()
        first = s.mpc_solve(x0[0, 0], xg, 0, clear_vars=1, max_iter=100)
        t_first = (time.perf_counter() - t0) * 1e3
        cyc_ms, cyc_it, succ = [], [], 0
        xa = first["x"][0][1]
        for c in range(24):
            t0 = time.perf_counter()
            r = s.mpc_solve(xa + rng.normal(0, 0.0005, 14).astype(np.float32), xg, 1, max_iter=4)
            cyc_ms.append((time.perf_counter() - t0) * 1e3); cyc_it.append(int(r["iters"][0])); succ += int(r["success"][0])
            xa = r["x"][0][1]
        res[name] = {"warm_start_iterations": int(first["iters"][0]), "warm_start_ms": round(t_first, 3), "cycles": len(cyc_ms), "iterations_per_cycle": 4,
                     "median_cycle_ms": round(float(np.median(cyc_ms[2:])), 3), "ms_per_iteration": round(float(np.median(np.asarray(cyc_ms[2:]) / np.asarray(cyc_it[2:]))), 4),
                     "successful_cycles": succ, "note": "wall clock around pddp_mpc_solve: includes H2D of the measured state and D2H of x, u, K every cycle"}
        s.close()
    return res


if __name__ == "__main__":
    main()
