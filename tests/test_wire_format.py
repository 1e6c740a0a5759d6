"""Trajectory wire format and the robot-side trajectory runner (lcmtypes/lcmt_trajectory_{f,d}.lcm, lcmt_solver_params.lcm,
lcmt_cost_params.lcm, DDPHelpers/LCMHelpers.cuh:98-153, 203-262) -- hostapi/LCMHelpers.hpp, exercised by the C++ test tool tests/wire/wire_driver.cpp.

Fingerprints are PINNED to the reference's own lcm-gen output: tests/golden/lcm_hashes.json holds the four base hash constants of
lcmtypes/{drake,kuka}/*.hpp (extracted by the committed tests/golden/make_lcm_hashes.py); getHash() rotates them left by one bit.  Also checked: a
second, independent implementation of the generator's hash (below), the byte layout (big-endian scalars in declaration order), the reference's
byte-count size quirk, round trips, rejection of foreign / truncated buffers, and the trajectory runner's command against numpy."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parallel-ddp_amd")


def lcm_hash(members):
    """lcm-gen: v = 0x12345678; per member: name, primitive type name, number of dimensions, per dimension (mode, size name);
    update(v, c) = ((v << 8) ^ (v >> 55)) + c on a signed 64-bit value; strings are prefixed by their length; final rotate-left by one."""
    M = (1 << 64) - 1

    def s64(v):
        v &= M
        return v - (1 << 64) if v >> 63 else v

    def upd(v, c):
        return s64(((v << 8) & M) ^ ((v >> 55) & M)) + c          # v >> 55 on a Python int is arithmetic, like int64_t

    def upds(v, s):
        v = s64(upd(v, len(s)))
        for ch in s.encode():
            v = s64(upd(v, ch))
        return v

    v = 0x12345678
    for name, typ, dim in members:
        v = upds(v, name); v = upds(v, typ)
        v = s64(upd(v, 1 if dim else 0))
        if dim:
            v = s64(upd(v, 1)); v = upds(v, dim)
    h = v & M
    return s64(((h << 1) & M) + (h >> 63))


TRAJ = lambda t: [("utime", "int64_t", None), ("x_size", "int32_t", None), ("u_size", "int32_t", None), ("KT_size", "int32_t", None),
                  ("x", t, "x_size"), ("u", t, "u_size"), ("KT", t, "KT_size")]
SOLVER = [("utime", "int64_t", None)] + [(n, "int32_t", None) for n in ("iterLimit", "timeLimit", "clearVars", "useCostShift")]
COST = [("utime", "int64_t", None)] + [(n, "float", None) for n in ("q_ee1 q_ee2 qf_ee1 qf_ee2 q_eev1 q_eev2 qf_eev1 qf_eev2 q_xdee qf_xdee q_xee qf_xee r_ee "
                                                                       "q1 q2 qf1 qf2 r").split()]


@pytest.fixture(scope="module")
def out():
    if not os.path.exists(os.path.join(PKG, "lib", "libpddp.so")):
        subprocess.check_call(["make", "-C", PKG, "-s"])
    exe = os.path.join(ROOT, "tests", "wire", "wire_driver")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-Wall", os.path.join(ROOT, "tests", "wire", "wire_driver.cpp"), "-L" + os.path.join(PKG, "lib"), "-lpddp",
                           "-Wl,-rpath," + os.path.join(PKG, "lib"), "-o", exe])
    text = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    d = {}
    for line in text.splitlines():
        if line.strip():
            k, _, v = line.partition(" ")
            d[k] = v
    return d


def test_fingerprints_equal_the_reference_lcm_gen_constants(out):
    """The four fingerprints on the wire == rotl1(base hash in the reference's generated headers)."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "lcm_hashes.json")))
    M = (1 << 64) - 1
    for key, okey in (("traj_f", "hash_traj_f"), ("traj_d", "hash_traj_d"), ("solver", "hash_solver"), ("cost", "hash_cost")):
        base = int(gold[key]["base_hash"], 16)
        fp = ((base << 1) & M) + (base >> 63)
        assert int(out[okey]) & M == fp, (key, gold[key]["source"])


def test_fingerprints_agree_with_an_independent_implementation_of_the_generator_hash(out):
    assert int(out["hash_traj_f"]) == lcm_hash(TRAJ("float"))
    assert int(out["hash_traj_d"]) == lcm_hash(TRAJ("double"))
    assert int(out["hash_solver"]) == lcm_hash(SOLVER)
    assert int(out["hash_cost"]) == lcm_hash(COST)
    assert len({out[k] for k in ("hash_traj_f", "hash_traj_d", "hash_solver", "hash_cost")}) == 4


def test_trajectory_message_layout_and_the_byte_count_quirk(out):
    N, n, m = 8, 14, 7
    xs, us, ks, ex, eu, ek = (int(v) for v in out["sizes"].replace("elems", "").split())
    # LCM_MPCLoop_Handler::handleStatus (LCMHelpers.cuh:241-246): sizes are ld * TRAJ_RUNNER_TIME_STEPS * sizeof(float) -- BYTES -- and the arrays have that many ELEMENTS
    assert (xs, us, ks) == (n * N * 4, m * N * 4, n * m * N * 4) and (ex, eu, ek) == (xs, us, ks)
    assert int(out["wire_len"]) == 8 + 8 + 12 + 4 * (xs + us + ks)
    head = bytes.fromhex(out["wire_head"])
    fp, utime, a, b, c = struct.unpack(">qqiii", head[:28])
    assert fp == lcm_hash(TRAJ("float")) and utime == 123456789012345 and (a, b, c) == (xs, us, ks)
    x0, x1 = struct.unpack(">ff", head[28:36])           # x[0][0] = 0, x[0][1] = 0.01: big-endian IEEE floats right after the header
    assert x0 == 0.0 and x1 == np.float32(0.01)
    assert out["decode_ok"] == "1" and out["roundtrip"] == "1" and out["bad_fingerprint_rejected"] == "1" and out["truncated_rejected"] == "1"


def test_parameter_messages(out):
    w = bytes.fromhex(out["solver_wire"])
    fp, utime, it, tl, cv, cs = struct.unpack(">qqiiii", w)
    assert fp == lcm_hash(SOLVER) and (utime, it, tl, cv, cs) == (42, 4, 10, 0, 1) and out["solver_roundtrip"] == "1"
    ln, rt = out["cost_len"].split(" cost_roundtrip ")
    assert int(ln) == 8 + 8 + 18 * 4 and rt == "1"
    assert [float(v) for v in out["cost_applied"].split()] == [0.5, 12.5, 13.5, 17.5]      # q_ee1, r_ee, q1, r in the .lcm member order


def test_trajectory_runner_command(out):
    """LCM_TrajRunner::statusCallback = getHardwareControls (MPCHelpers.cuh:819-858) on the stored message: zero-order hold on u and K, first-order hold on x."""
    N, n, m = 8, 14, 7
    x = np.array([[k + i / 100.0 for i in range(n)] for k in range(N)], np.float32)
    u = np.array([[-(k + i / 10.0) for i in range(m)] for k in range(N)], np.float32)
    KT = (np.float32(0.01) * (np.arange(N * n * m) % 97).astype(np.float32)).reshape(N, m, n)
    q = np.array([2.0 + i / 100.0 + 0.001 for i in range(7)]); qd = np.array([2.0 + (i + 7) / 100.0 for i in range(7)])
    k, frac = 2, 0.25
    # t = t0 + int64(2.25 * step): the truncation to whole microseconds moves the fraction slightly
    step_us = 0.5 / (N - 1) * 1e6
    steps = float(int(2.25 * step_us)) / step_us
    k, frac = int(steps), steps - int(steps)
    nominal = np.float32(1.0 - frac) * x[k] + np.float32(frac) * x[k + 1]
    dx = np.concatenate([q, qd]).astype(np.float32) - nominal
    tau = u[k].copy()
    for r in range(m):
        val = u[k][r]
        for c in range(n):
            val = np.float32(val - np.float32(KT[k, r, c] * dx[c]))
        tau[r] = val
    assert out["not_ready"] == "1" and out["runner_beyond"] == "1"
    got = [float(v) for v in out["runner_err"].split("tau")[1].split("[!]")[0].split()]
    assert out["runner_err"].startswith("0")
    np.testing.assert_allclose(got, tau, rtol=2e-6)


@pytest.mark.gpu
def test_a_trajectory_solved_on_the_gpu_travels_through_the_wire_format_bit_for_bit(tmp_path):
    """End to end across the N4 boundary: runiLQR_MPC_GPU (struct facade, HIP kernels) -> trajVars -> trajectoryMessage -> LCM bytes (tests/wire/wire_solve.cpp); the bytes
    are decoded HERE, independently of hostapi/LCMHelpers.hpp (big-endian members in declaration order behind the 8-byte fingerprint), and must carry exactly the plan
    the solver left -- every float bit for bit, the reference's byte-count sizes, the plan's time stamp (LCM_MPCLoop_Handler::handleStatus, LCMHelpers.cuh:239-262)."""
    exe = os.path.join(ROOT, "tests", "wire", "wire_solve")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-Wall", os.path.join(ROOT, "tests", "wire", "wire_solve.cpp"), "-L" + os.path.join(PKG, "lib"), "-lpddp",
                           "-Wl,-rpath," + os.path.join(PKG, "lib"), "-o", exe])
    wire_file, raw_file = str(tmp_path / "traj.lcm"), str(tmp_path / "traj.raw")
    text = subprocess.run([exe, wire_file, raw_file], capture_output=True, text=True, check=True).stdout
    info = dict(l.split(" ", 1) for l in text.splitlines() if l.startswith(("iterations", "wire_len")))
    its = info["iterations"].split()
    assert int(its[0]) >= 4 and float(its[4]) < 0.5 * float(its[2])                      # a real solve: J fell
    N, n, m = 64, 14, 7                                                                     # the arm's default horizon (config.cuh:51-53)
    raw = open(raw_file, "rb").read()
    utime = struct.unpack("<q", raw[:8])[0]
    plan = np.frombuffer(raw[8:], "<f4")
    assert plan.size == N * (n + m + n * m)
    x, u, KT = plan[: N * n], plan[N * n: N * (n + m)], plan[N * (n + m):]
    assert np.abs(KT).max() > 1e-3 and np.abs(np.diff(x.reshape(N, n), axis=0)).max() > 1e-4   # gains and a moving state: not an empty plan
    wire = open(wire_file, "rb").read()
    fp, ut, xs, us, ks = struct.unpack(">qqiii", wire[:28])
    assert fp == lcm_hash(TRAJ("float")) and ut == utime
    assert (xs, us, ks) == (n * N * 4, m * N * 4, n * m * N * 4)                             # BYTE counts (the quirk that is the contract)
    assert len(wire) == 28 + 4 * (xs + us + ks)
    body = np.frombuffer(wire[28:], ">f4")
    wx, wu, wk = body[:xs], body[xs: xs + us], body[xs + us:]
    for sent, kept in ((wx, x), (wu, u), (wk, KT)):
        assert np.array_equal(sent[: kept.size].astype("<f4").view("<u4"), kept.view("<u4"))   # bit for bit
        assert not sent[kept.size:].any()                                                   # the elements beyond the copied bytes are zeros
