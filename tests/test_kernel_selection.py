"""Which kernels a handle's sweep launches (pddp_time_kernels reports them in launch order), as a function of the configuration alone: the selection is DATA
(pddp_config.kernels, include/pddp.h; all zero = the library's documented choice) and the library reads no environment variable for it.  The parity suites run with the library's automatic selection; this file
states what that selection IS for the shapes they use, so that a handle silently falling back to another family (the lane-group kernels compute the same functions) cannot
pass unnoticed as coverage of the family a test was written for.  DESIGN.md section 4: matrix-core backward pass for the arm in float at every batch size; rollouts / setup on
thread lanes from 512 problems (k_fp_tl, k_nis_tl), as a four-wave pipeline and one thread per (knot, joint) for few problems in flight (k_fp_tl4, k_nis_tl7), for the
joint-space AND the end-effector cost; closed-form plants thread-serial from 256 (problem, segment) units."""
import ctypes
import os

import numpy as np
import pytest

from backends import make_solver
from oracle_binding import example_inputs

pytestmark = pytest.mark.gpu
KUKA = dict(N=64, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=20)


def kernels(plant, batch, sel=None, dtype=0, **kw):
    """names of the kernels one sweep of such a handle launches; sel: pddp_config.kernels by name (pyddp.set_kernels), None = the library's own choice"""
    s = make_solver("hip", plant, dtype=dtype, batch=batch, use_graph=0, kernels=sel, **kw)
    N = kw["N"]
    x0, u0, xg = example_inputs(plant, N, np.float64 if dtype else np.float32, noise=np.random.default_rng(3).normal(0, 0.001, (N, s.n)))
    if kw.get("ee_cost"):
        xg = np.asarray([0.5, 0.1, 0.6, 0, 0, 0] + [0] * 8, x0.dtype)
    s.load(np.tile(x0, batch), np.tile(u0, batch), np.tile(xg, batch))
    s.iterate(2); s.sync()
    names = [n for n, _ in s.time_kernels(2)]
    s.close()
    return names


@pytest.mark.parametrize("ee", [0, 1])
def test_few_problems_in_flight_run_the_pipeline_and_the_per_joint_setup(ee):
    kw = dict(KUKA, ee_cost=ee, **(dict(mpc_mode=1, ignore_max_rho_exit=0) if ee else {}))
    # (no k_ls: with a problem's M x A rollouts inside one wavefront the rollout pipeline ends with the line search itself, round 5; A = 16 x M = 8 does not fit and keeps it)
    # (no k_sweep_maps either: the same rollout kernel begins with the linear forward sweep -- the maps the backward pass composed; kernels.sweep = maps keeps the kernel)
    assert kernels(4, 1, **kw) == ["k_bp_mfma", "k_fp_tl4", "k_nis_tl7"]
    assert kernels(4, 3, **kw) == ["k_bp_mfma", "k_fp_tl4", "k_nis_tl7"]
    assert kernels(4, 1, dict(sweep="maps"), **kw) == ["k_bp_mfma", "k_sweep_maps", "k_fp_tl4", "k_nis_tl7"]
    assert kernels(4, 1, dict(ls="wg"), **kw) == ["k_bp_mfma", "k_fp_tl4", "k_ls", "k_nis_tl7"]
    assert kernels(4, 1, **dict(kw, N=128, M=8, A=16)) == ["k_bp_mfma", "k_sweep_maps", "k_fp_tl4", "k_ls", "k_nis_tl7"]


def test_the_library_does_not_read_the_environment_for_its_selection():
    """Kernel families were environment switches of the library until round 4; the selection is data now (pddp_config.kernels, all zero from pddp_default_config = the
    library's own choice) and the library's source asks the environment for nothing but its two debugging aids."""
    import pyddp
    lib = ctypes.CDLL(pyddp.library_path())
    c = pyddp.PddpConfig()
    assert lib.pddp_default_config(ctypes.byref(c), 4) == 0
    for k, v in dict(KUKA, batch=1, use_graph=0).items():
        setattr(c, k, v)
    assert all(getattr(c.kernels, f) == 0 for f in pyddp.KERNEL_NAMES)
    s = pyddp.Solver(c)
    x0, u0, xg = example_inputs(4, 64, np.float32)
    s.load(x0, u0, xg); s.iterate(2); s.sync()
    assert [n for n, _ in s.time_kernels(2)] == ["k_bp_mfma", "k_fp_tl4", "k_nis_tl7"]
    s.close()
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parallel-ddp_amd", "csrc")
    src = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp"))))     # every translation unit of the library and what they include
    assert sorted(set(re.findall(r'getenv\("(\w+)"\)', src))) == ["PDDP_EVAL_GRID", "PDDP_POISON_LDS"]      # debugging / micro-benchmark aids only


def test_selection_overrides_reach_the_older_kernels():
    assert kernels(4, 1, dict(fp="tl2"), **KUKA)[2] == "k_fp_tl2"
    assert kernels(4, 1, dict(fp="lg"), **KUKA)[2:] == ["k_fp_lg", "k_ls", "k_nis_lg"]
    assert kernels(4, 64, dict(bp="lg", fp="lg"), **KUKA)[0] != "k_bp_mfma"


@pytest.mark.parametrize("ee", [0, 1])
def test_large_batches_run_one_thread_per_rollout_and_per_knot(ee):
    kw = dict(KUKA, ee_cost=ee, **(dict(mpc_mode=1, ignore_max_rho_exit=0) if ee else {}))
    assert kernels(4, 512, **kw) == ["k_bp_mfma", "k_sweep_maps", "k_fp_tl", "k_ls", "k_nis_tl"]
    if not ee:
        assert kernels(4, 2048, **kw) == ["k_bp_mfma", "k_sweep_maps", "k_fp_tl", "k_ls_many", "k_nis_tl"]


def test_float64_handles_default_to_lane_groups_and_reach_the_benched_family_on_request():
    assert kernels(4, 2, dtype=1, **KUKA)[-3:] == ["k_fp_lg", "k_ls", "k_nis_lg"]
    got = kernels(4, 2, dict(bp="mx", fp="tl"), dtype=1, **KUKA)
    assert got[0].startswith("k_bp_mfma") and got[-3:] == ["k_fp_tl", "k_ls", "k_nis_tl"]


def test_closed_form_plants_switch_to_thread_serial_kernels_with_the_device_full():
    cart = dict(N=64, M=4, A=8, integrator=3, total_time=4.0, tol_cost=0.0, max_iter=20)
    assert kernels(2, 2, **cart) == ["k_bp", "k_fp", "k_ls", "k_nis"]
    assert kernels(2, 4096, **cart) == ["k_bp_ts", "k_sweep_cf", "k_fp_cf", "k_ls_many", "k_nis_ts"]      # line search: one thread per problem from 2048 problems in flight; rollouts: thread per rollout with the knot's operands staged per wavefront (8 or 16 step sizes)
    assert kernels(2, 4096, **dict(cart, A=4)) == ["k_bp_ts", "k_fp_ts", "k_ls_many", "k_nis_ts"]
    assert kernels(2, 4096, dict(cf_fp="ts"), **cart)[1] == "k_fp_ts"
    quad = dict(N=64, M=4, A=8, integrator=3, total_time=4.0, tol_cost=0.0, max_iter=20)      # 12 states: rollouts thread-serial, setup on 16-lane groups, backward pass cooperative (32-lane groups from 8192 blocks of knots)
    assert kernels(3, 1024, **quad) == ["k_bp", "k_fp_ts", "k_ls", "k_nis_kb"]               # RK3: the knot-batched setup (lane = knot for the scalar gradients, lane = column of [A B] after)
    assert kernels(3, 2048, **quad) == ["k_bp_mq", "k_fp_ts", "k_ls_many", "k_nis_kb"]          # backward pass: the matrix cores, one wavefront per block of knots (round 5; until then k_bp_cl: 16 lanes per block of knots, lane = column (8 step sizes of a 12-state plant: 8 problems x 12 states do not fit one fetch per lane, the staged rollouts take 16)
    assert kernels(3, 2048, **dict(quad, A=16)) == ["k_bp_mq", "k_sweep_maps", "k_fp_cf", "k_ls_many", "k_nis_kb"]      # (round 6: the matrix-core backward pass composes the segments' sweep maps, k_sweep_maps_cf finishes)
    assert kernels(3, 2048, dict(sweep="st"), **dict(quad, A=16)) == ["k_bp_mq", "k_sweep_cf", "k_fp_cf", "k_ls_many", "k_nis_kb"]      # kernels.sweep = st: A - B K | B du of every knot + the per-knot sweep
    assert kernels(3, 2048, dict(cf_bp="cl"), **dict(quad, A=16))[:2] == ["k_bp_cl", "k_sweep_cf"]                         # (only the matrix-core kernel composes maps)
    assert kernels(3, 2048, **dict(quad, integrator=1)) == ["k_bp_mq", "k_fp_ts", "k_ls_many", "k_nis_gl"]
    assert kernels(3, 2048, dict(cf_bp="gl32"), **quad)[0] == "k_bp_gl"
    assert kernels(3, 2048, dict(cf_nis="gl"), **quad)[-1] == "k_nis_gl"


def _solve_with(sel, batch, N, M, iters=6):
    s = make_solver("hip", 4, dtype=0, batch=batch, use_graph=0, N=N, M=M, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=iters, kernels=sel)
    rng = np.random.default_rng(11)
    xs, us, gs = [], [], []
    for _ in range(batch):                                            # every problem its own start: a wrong knot / problem offset cannot hide behind identical data
        x0, u0, xg = example_inputs(4, N, np.float32, noise=rng.normal(0, 0.002, (N, 14)))
        xs.append(x0); us.append(u0); gs.append(xg)
    out = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    names = [n for n, _ in s.time_kernels(1)]
    res = {k: np.array(out[k]) for k in ("x", "u", "KT", "Jout", "alphaOut")}
    s.close()
    return names, res


@pytest.mark.parametrize("N,M", [(32, 4), (32, 1), (64, 4), (128, 1), (128, 8), (256, 4)])
def test_compact_operands_through_the_lds_prefetch_follow_the_reference_layout(N, M):
    """The float matrix-core backward pass reads the compact [A B] and the cost gradient through LDS-direct buffer loads one knot ahead (bp_mfma.hpp mx_dma_knot), with
    scalar chunk / knot offsets: chunks of 64 knots shared by two problems (N = 32), several chunks per problem (N = 256), blocks of knots of every length, one block (M = 1).
    The same first iteration with kernels.ab = full (reference-layout [A B], plain loads, no prefetch) must land on the same trajectory up to float32 rounding -- a wrong knot,
    chunk or problem offset would be off by the size of the data, not by 1e-6.  (Across BUILDS the compact path is held bit for bit: tools/cmp_compact_vs_full.py builds,
    profiles/r04_bp_mfma.md.)"""
    sel = dict(bp="mx", fp="tl")
    names_c, c = _solve_with(sel, 5, N, M, iters=1)
    names_f, f = _solve_with(dict(sel, ab="full"), 5, N, M, iters=1)
    assert names_c[0] == "k_bp_mfma" and names_f[0] == "k_bp_mfma" and "k_nis_tl" in names_c
    np.testing.assert_array_equal(c["alphaOut"], f["alphaOut"])
    np.testing.assert_allclose(c["x"], f["x"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(c["u"], f["u"], atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(c["KT"], f["KT"], atol=5e-2, rtol=5e-3)            # (the float32 Riccati recursion amplifies a rounding difference of the operands)
    np.testing.assert_allclose(c["Jout"][:, :2], f["Jout"][:, :2], rtol=1e-5)


@pytest.mark.parametrize("dtype,ee,M,A,batch", [(0, 0, 4, 8, 1), (0, 0, 4, 8, 3), (0, 0, 4, 16, 2), (0, 0, 2, 8, 5), (0, 0, 8, 8, 2), (0, 1, 4, 8, 3), (1, 0, 4, 8, 3), (1, 0, 2, 16, 1)])
def test_the_sweep_inside_the_rollout_kernel_is_the_sweep_kernel_bit_for_bit(dtype, ee, M, A, batch):
    """Few problems in flight: k_fp_tl4 begins with the linear forward sweep (the segment maps of the matrix-core backward pass walked by 14 lanes of the problem's M x A,
    every rollout's start state formed in its own lane) instead of a k_sweep_maps launch in front of it.  Same operations in the same order: a whole solve -- graph replay,
    the production path -- ends on the same bits as with kernels.sweep = maps, for 16 / 32 / 64 lanes per problem, partial wavefronts, both cost families, both
    element types."""
    kw = dict(N=64, M=M, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=12, ee_cost=ee, **(dict(mpc_mode=1, ignore_max_rho_exit=0) if ee else {}))
    fam = dict(bp="mx", fp="tl4") if dtype else {}
    res = []
    for sel in (dict(fam), dict(fam, sweep="maps")):
        s = make_solver("hip", 4, dtype=dtype, batch=batch, use_graph=1, kernels=sel or None, **kw)
        rng = np.random.default_rng(5)
        xs, us, gs = [], [], []
        for _ in range(batch):
            x0, u0, xg = example_inputs(4, 64, np.float64 if dtype else np.float32, noise=rng.normal(0, 0.002, (64, 14)))
            if ee:                                                                  # (the start of tests/test_ee_thread_lanes.py, every problem its own goal)
                x0 = np.zeros((64, 14), x0.dtype); x0[:, 1] = 0.7; x0[:, 3] = -0.8; x0[:, 5] = 0.75; x0 = x0.ravel()
                u0 = np.full(64 * 7, 0.01, x0.dtype)
                xg = np.asarray([0.45, 0.15 + 0.05 * len(xs), 0.75] + [0] * 11, x0.dtype)
            xs.append(x0); us.append(u0); gs.append(xg)
        out = s.solve(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
        res.append({k: np.array(out[k]) for k in ("x", "u", "KT", "dmax", "Jout", "alphaOut")})
        names = [n for n, _ in s.time_kernels(1)]
        assert ("k_sweep_maps" in names) == ("sweep" in sel) and "k_fp_tl4" in names
        s.close()
    assert (res[0]["alphaOut"][:, 1:6] >= 0).any()                                  # steps were taken
    for k in res[0]:
        np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)
