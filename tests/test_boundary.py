"""The drop-in boundary: the C ABI (include/pddp.h -> libpddp.so) and the source-level facade (hostapi/) that carries the
reference's own entry-point names (allocateMemory_GPU / runiLQR_GPU / freeMemory_GPU, DDPWrappers.cuh:10-21,
nisInitHelpers.cuh:768-772,865-868).  CPU part: the library loads, exports every declared symbol, the facade compiles with
plain g++, and the product fails loudly without a HIP device.  GPU part: the reference example's shape runs through the
facade and reproduces the oracle's J trace."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import pyddp
from oracle_binding import Oracle, default_cfg, example_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parallel-ddp_amd")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pddp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pddp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(pyddp.library_path())
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s


def test_header_is_plain_c99_and_links_from_c():
    """include/pddp.h is the FFI surface: it must compile with a C compiler (-std=c99 -pedantic) and the library must link from C."""
    exe = os.path.join(ROOT, "tests", "cabi", "cabi_min")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "cabi_min.c"),
                           "-L" + os.path.join(PKG, "lib"), "-lpddp", "-Wl,-rpath," + os.path.join(PKG, "lib"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    import torch
    assert r.returncode == 0
    assert ("create rc 0" in r.stdout) if torch.cuda.is_available() else ("create rc -2" in r.stdout and "no HIP device" in r.stdout)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(pyddp.PddpError, match="no HIP device"):
        pyddp.Solver(pyddp.default_config(4, N=16, M=1, A=1))


def build_examples():
    subprocess.check_call(["make", "-C", PKG, "-s", "examples"])
    return os.path.join(PKG, "examples", "iLQR_examples")


def test_facade_compiles_with_plain_gxx_and_fails_loudly_without_gpu():
    exe = build_examples()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    r = subprocess.run([exe, "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr


@pytest.mark.parametrize("arg,banner", [("CS", "<<<TESTING CPU 1/1>>>"), ("C", "<<<TESTING CPU-P 1/1>>>")])
def test_reference_example_cpu_branches_through_the_facade(arg, banner):
    """testCPU of examples/WAFR_iLQR_examples.cu:231-299 against hostapi/, both branches of its `serialAlphas` switch as the reference's main selects them
    (:434: 'CS' serial line search = allocateMemory_CPU / runiLQR_CPU / freeMemory_CPU, 'C' parallel = the ..._CPU2 entry points) over libpddp_cpu.so,
    Kuka N=128, A=8, M=4, TOL_COST 0, 1 solve.  Runs without a GPU (it is the reference's CPU path, not a fallback of the GPU one)."""
    exe = build_examples()
    r = subprocess.run([exe, arg, "1", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("CPU Parallel blocks:[4]") == 1 and banner in r.stdout
    m = re.search(r"solve 0: (\d+) iterations, J ([0-9.]+) -> ([0-9.]+)", r.stdout)
    assert m and int(m.group(1)) == 100 and float(m.group(3)) < 0.5 * float(m.group(2))      # the first-acceptable line search stalls around 600-750 on this problem (SURVEY.md 8c, G3 trace)


@pytest.mark.gpu
def test_reference_example_shape_through_the_facade():
    """testGPU of examples/WAFR_iLQR_examples.cu:303-361 against hostapi/: Kuka N=128, A=8, M=4, TOL_COST 0, 2 solves."""
    exe = build_examples()
    r = subprocess.run([exe, "2", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("GPU (MI355X) Parallel blocks:[4]") == 2 and "iters:[100]" in r.stdout
    J = [float(m.group(1)) for m in re.finditer(r"iter\s+\d+\s+J\s+([0-9.]+)", r.stdout)]
    # zero-noise oracle trace: the example's velocity noise (1e-3) moves J[0] by < 1e-3 relative
    o = Oracle(default_cfg(4, N=128, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=100, cores=1, spawn_threads=0), np.float32)
    ref = o.run_ilqr_gpusem(*example_inputs(4, 128, np.float32))
    assert abs(J[0] - ref["Jout"][0]) < 2e-3 * ref["Jout"][0]
    assert J[1] < J[0] and min(J) < 0.15 * J[0]
    phases = re.search(r"BP ([0-9.]+) FP ([0-9.]+) NIS ([0-9.]+) ms", r.stdout)
    assert phases and all(float(v) > 0 for v in phases.groups())


def test_mpc_facade_compiles_with_plain_gxx_and_fails_loudly_without_gpu():
    build_examples()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    r = subprocess.run([os.path.join(PKG, "examples", "MPC_examples"), "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_mpc_lockstep_loop_through_the_struct_facade():
    """testMPC_lockstep's shape (examples/WAFR_MPC_examples.cu:160-238) against hostapi/MPCHelpers.hpp: warm start, then
    receding-horizon cycles of one knot with a 10-iteration cap; the bookkeeping of storeVarsGPU_MPC (MPCHelpers.cuh:755-774)
    shows in last_successful_solve (1 after a successful solve, counting up otherwise)."""
    build_examples()
    r = subprocess.run([os.path.join(PKG, "examples", "MPC_examples"), "12", "10", "1000", "1.0", "0.001"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    warm = re.search(r"warm start: (\d+) iterations, J ([0-9.]+) -> ([0-9.]+)", r.stdout)
    assert warm and int(warm.group(1)) >= 5 and float(warm.group(3)) < 0.2 * float(warm.group(2))
    cyc = [(int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), int(m.group(5)))
           for m in re.finditer(r"cycle\s+\d+\s+shift (\d+)\s+iterations\s+(\d+)\s+J ([0-9.]+) -> ([0-9.]+)\s+last_successful_solve (\d+)", r.stdout)]
    assert len(cyc) == 12
    assert all(c[0] == 1 and 1 <= c[1] <= 10 and c[3] <= c[2] * (1 + 1e-6) for c in cyc)
    assert sum(c[4] == 1 for c in cyc) >= 6                 # most cycles take a step
    streak = 0
    for c in cyc:                                           # the failure counter counts up by one per unsuccessful solve
        streak = 1 if c[4] == 1 else streak + 1
        assert c[4] == streak or c[4] == 1
    # every cycle re-converges from the open-loop rollout of the shifted controls to the neighbourhood of the warm-start optimum
    assert all(c[3] < 1.5 * float(warm.group(3)) for c in cyc)


@pytest.mark.gpu
def test_lockstep_figure_eight_experiment_end_to_end():
    """testMPC_lockstep (examples/WAFR_MPC_examples.cu:185-238) against hostapi/: solve -> simulated robot (pddp_simulate) -> goal moves along the
    reference's 200-point figure (tests/golden/fig8_goals.csv) -> solve ...; fixed 8 ms control cycles and a 4 s figure so that the run is
    reproducible.  The reference's own run of this experiment reports an average tracking error of 0.0878 m (test/WAFR_fig8.py:5-6, 10 s figure)."""
    build_examples()
    goals = os.path.join(ROOT, "tests", "golden", "fig8_goals.csv")
    res = {}
    for exe in ("MPC_lockstep", "MPC_lockstep_fix"):
        r = subprocess.run([os.path.join(PKG, "examples", exe), "4", "10", "4", goals, "8000"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        m = re.search(r"cycles: (\d+)  figure completed: (\d)  reached the start of the figure: (\d)", r.stdout)
        e = re.search(r"Average tracking error: \[([0-9.]+)\]", r.stdout)
        assert m and e and m.group(2) == "1" and m.group(3) == "1", r.stdout[-400:]
        assert "CRITICAL FAILURE" not in r.stdout
        res[exe] = (int(m.group(1)), float(e.group(1)))
        assert 0.0 < res[exe][1] < 0.2, res
        assert 4.0e6 / 8000 <= res[exe][0] < 3 * 4.0e6 / 8000        # the figure itself takes 500 cycles; reaching its start takes some more


@pytest.mark.gpu
def test_two_handles_on_two_host_threads_do_not_disturb_each_other():
    """The reference's MPC set-up runs the solver and the trajectory runner on different host threads (MPCHelpers.cuh:62): two handles (own streams, own
    buffers, thread-local error string) driven concurrently from two threads give exactly what each gives alone."""
    import threading
    kw = dict(N=64, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=25)
    inputs = [example_inputs(4, 64, np.float32, noise=np.random.default_rng(s).normal(0, 0.002, (64, 14))) for s in (1, 2)]
    alone = []
    for x0, u0, xg in inputs:
        alone.append(pyddp.Solver(pyddp.default_config(4, **kw)).solve(x0, u0, xg))
    solvers = [pyddp.Solver(pyddp.default_config(4, **kw)) for _ in inputs]
    out = [None, None]

    def work(i):
        for _ in range(3):                      # several solves each, interleaved by the scheduler
            out[i] = solvers[i].solve(*inputs[i])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for i in range(2):
        assert np.array_equal(out[i]["Jout"], alone[i]["Jout"]) and np.array_equal(out[i]["x"], alone[i]["x"]) and np.array_equal(out[i]["KT"], alone[i]["KT"])
    assert not np.array_equal(out[0]["Jout"], out[1]["Jout"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_reference_views_after_a_solve(dtype):
    """What the reference leaves in d_ApBK / d_Bdu (computeFSVars, bpHelpers.cuh:281-312) and in EVERY alpha slot of d_x / d_u / d_d (memcpyCurrAKern x 3,
    nisInitHelpers.cuh:24-32,270-272) after runiLQR_GPU -- production sweeps write neither; pddp_refresh_reference_views (called by the facade's runiLQR_GPU)
    and pddp_get_array("ApBK" / "Bdu") rebuild them.  Checked against the arithmetic itself and against the oracle's backward pass on the solve's final state."""
    N, M, A, n, m = 32, 4, 8, 14, 7
    x0, u0, xg = example_inputs(4, N, dtype, noise=np.random.default_rng(3).normal(0, 0.002, (N, 14)))
    s = pyddp.Solver(pyddp.default_config(4, N=N, M=M, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=12, dtype=1 if dtype == np.float64 else 0))
    out = s.solve(x0, u0, xg)
    AB = s.get("AB").reshape(N, n + m, n); KT = s.get("KT").reshape(N, m, n); du = s.get("du").reshape(N, m)        # column-major blocks: [col][row]
    F = s.get("ApBK").reshape(N, n, n); Bd = s.get("Bdu").reshape(N, n)                                             # materialised by pddp_get_array
    tol = 1e-12 if dtype == np.float64 else 2e-6
    for k in range(N - 1):
        Amat, Bmat, K = AB[k, :n, :].T.astype(np.float64), AB[k, n:, :].T.astype(np.float64), KT[k].astype(np.float64)     # A (n x n), B (n x m), K (m x n)
        ref = Amat - Bmat @ K
        assert np.abs(F[k].T - ref).max() <= tol * max(1.0, np.abs(ref).max()), k
        assert np.abs(Bd[k] - Bmat @ du[k]).max() <= tol * max(1.0, np.abs(Bmat @ du[k]).max()), k
    assert np.abs(F[: N - 1]).max() > 0.5
    if dtype == np.float64:
        # ... and against the ORACLE at the exit: the reference's d_ApBK / d_Bdu hold the LAST backward pass's computeFSVars output -- its loop breaks before
        # nextIterationSetupGPU (DDPWrappers.cuh:104-113) -- which is what the oracle's stepped GPU-semantics loop recorded in its last iteration
        from gpusem_steps import gpusem_iterations
        from oracle_binding import Oracle, default_cfg
        o = Oracle(default_cfg(4, cores=1, spawn_threads=0, N=N, M=M, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=12), np.float64)
        with np.errstate(all="ignore"):
            last = list(gpusem_iterations(o, x0, u0, xg, 12))[-1]
        assert last.iter == int(out["iters"][0])
        cnt = (N - 1) * n * n
        assert np.abs(F.ravel()[:cnt] - last.ApBK[:cnt]).max() <= 1e-8 * np.abs(last.ApBK[:cnt]).max()
        assert np.abs(Bd.ravel()[: (N - 1) * n] - last.Bdu[: (N - 1) * n]).max() <= 1e-8 * max(np.abs(last.Bdu[: (N - 1) * n]).max(), 1e-300)
    s.refresh_reference_views()
    xs = s.get("xs").reshape(A, N, n); us = s.get("us").reshape(A, N, m); ds = s.get("ds").reshape(A, N, n)
    for a in range(A):
        assert np.array_equal(xs[a], out["x"][0]) and np.array_equal(us[a][: N - 1], out["u"][0][: N - 1]), a
        assert np.array_equal(ds[a], ds[0])
    assert np.array_equal(ds[0], s.get("dcur").reshape(N, n))
    s.close()


@pytest.mark.gpu
def test_set_cost_equals_creating_with_those_weights():
    x0, u0, xg = example_inputs(4, 32, np.float32)
    kw = dict(N=32, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=15)
    a = pyddp.Solver(pyddp.default_config(4, Q1=0.3, Q2=0.002, R=0.0005, QF1=500.0, QF2=200.0, **kw))
    ra = a.solve(x0, u0, xg)
    b = pyddp.Solver(pyddp.default_config(4, **kw))
    rb0 = b.solve(x0, u0, xg)
    b.set_cost(0.3, 0.002, 0.0005, 500.0, 200.0)
    rb = b.solve(x0, u0, xg)
    assert not np.array_equal(rb0["Jout"], rb["Jout"])
    for k in ("Jout", "alphaOut", "x", "u"):
        assert np.array_equal(ra[k], rb[k]), k


@pytest.mark.parametrize("kw,msg", [
    (dict(N=100), "power of two"),
    (dict(N=128, M=3), "M must divide N"),
    (dict(N=128, M=128), "N/M >= 2"),
    (dict(A=0), "A in"),
    (dict(A=65), "A in"),
    (dict(batch=0), "batch"),
    (dict(N=128, batch=80000), "32-bit element offsets"),  # maximum size: 80000 * 128 * 441 elements of H do not fit the lane-group kernels' offsets
    (dict(integrator=3), "Euler"),                       # the arm is Euler-only, like config.cuh:58
    (dict(dtype=2), "unsupported"),
    (dict(N=128, M=16, A=12), "must not exceed 128"),    # lane-group forward pass: a workgroup rolls out all 12 candidates x 16 segments (A not a multiple of 8)
])
def test_create_rejects_bad_configurations_with_a_code_and_a_message(kw, msg):
    """Errors are codes + pddp_last_error(), never exit() (the reference's gpuAssert exits, utils/cudaUtils.cu:31-37).
    Validation comes before the device check, so this runs without a GPU on the product library."""
    lib = ctypes.CDLL(pyddp.library_path())
    lib.pddp_last_error.restype = ctypes.c_char_p
    cfg = pyddp.default_config(4, **{**dict(N=64, M=4, A=8), **kw})
    h = ctypes.c_void_p()
    rc = lib.pddp_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -1, rc                                  # PDDP_EINVAL
    assert msg in lib.pddp_last_error().decode(), lib.pddp_last_error().decode()
    assert not h.value


def test_default_config_matches_config_cuh():
    """pddp_default_config = the per-plant blocks of config.cuh:24-61 and the #ifndef defaults below them."""
    exp = {1: dict(N=128, A=32, integrator=3, alpha_base=0.75, rho_init=10.0, max_defect=1.0, total_time=4.0),
           2: dict(N=128, A=32, integrator=3, alpha_base=0.75, rho_init=10.0, max_defect=0.75, total_time=4.0),
           3: dict(N=128, A=16, integrator=3, alpha_base=0.5, rho_init=1.0, max_defect=1.0, total_time=4.0),
           4: dict(N=64, A=16, integrator=1, alpha_base=0.5, rho_init=12.5, max_defect=1.0, total_time=0.5)}
    for plant, e in exp.items():
        c = pyddp.default_config(plant)
        for k, v in e.items():
            assert getattr(c, k) == v, (plant, k)
        assert c.M == 4 and c.max_iter == 100 and c.tol_cost == 0.0001 and c.exp_red_min == 0.05 and c.exp_red_max == 1.25
        assert (c.Q1, c.Q2, c.R, c.QF1, c.QF2) == (0.1, 0.001, 0.0001, 1000.0, 1000.0)


def test_facade_carries_the_fixed_switches_and_refuses_other_values(tmp_path):
    """config.cuh:81-82,95,98,102-104,116,123 fix LINEAR_TRANSFORM_SWITCH, ALPHA_BEST_SWITCH, FORCE_PARALLEL, STATE_REG, RHO_MAX / RHO_MIN / RHO_FACTOR, USE_EXP_RED and
    USE_MAX_DEFECT with unconditional #defines; the facade defines the same names with the same values and refuses a translation unit that pre-defines another one (the
    kernels are built for these values -- a changed switch must not be ignored silently)."""
    hostapi = os.path.join(PKG, "hostapi")
    src = tmp_path / "sw.cpp"
    src.write_text('#define PLANT 4\n#include "config.hpp"\n'
                   "static_assert(STATE_REG == 1 && ALPHA_BEST_SWITCH == 1 && LINEAR_TRANSFORM_SWITCH == 1 && FORCE_PARALLEL == 1 && USE_EXP_RED == 1 && USE_MAX_DEFECT == 1, \"\");\n"
                   "static_assert(RHO_MAX == 10000000.0 && RHO_MIN == 0.01 && RHO_FACTOR == 1.25, \"\");\nint main() { return 0; }\n")
    base = ["g++", "-std=c++17", "-fsyntax-only", "-I", hostapi, "-I", os.path.join(ROOT, "include"), str(src)]
    assert subprocess.run(base, capture_output=True, text=True).returncode == 0
    for flag in ("-DSTATE_REG=0", "-DALPHA_BEST_SWITCH=0", "-DRHO_FACTOR=1.6", "-DUSE_EXP_RED=0"):
        r = subprocess.run(base + [flag], capture_output=True, text=True)
        assert r.returncode != 0 and "fixed" in r.stderr, flag
