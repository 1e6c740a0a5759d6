"""The oracle's restatement of the MPC wrapper (ora_gs_*: loadVarsGPU_MPC, runiLQR_MPC_GPU, storeVarsGPU_MPC -- DDPHelpers/MPCHelpers.cuh:602-655, 864-1045, 755-774) pinned by the
reference's OWN statements, executed.

tests/golden/mpc_fixtures.{npz,json} hold receding-horizon sequences produced at fixture-generation time (tests/golden/make_phase_fixtures.py --mpc, in the build container,
where /root/reference exists): the reference's driver and every function it calls translated statement by statement (tests/golden/refc2py.py), its kernels (shiftAndCopyKern,
rolloutMPCKern / rolloutMPCKern2, and the whole solver loop's) run under the SIMT emulation with the reference's launch geometry, on ONE persistent GPUVars / trajVars pair laid
out as allocateMemory_GPU_MPC / loadTraj do.  Data only: per control cycle the measured state that went in, the driver's own trace of J and step-size indices
(USE_ALG_TRACE), the device arrays the next cycle starts from and the host trajectory the driver handed out.

The oracle replays the same cycles.  Asserted: identical step-size indices and iteration counts, the success flag against the reference's `last_successful_solve`
bookkeeping (1 = this cycle took a step with a step-size index > 0, sic), J to 1e-10 and the device-side x, u, K, d to 1e-10 (measured: bit-identical, INCLUDING the
fall-back of a cycle without such a step, where the reference restores its d_x_old / d_u_old / d_KT_old copies).  Not asserted: the reference's HOST copy after a failed cycle
(it keeps the trajectory of the last successful cycle while moving its time stamp; the C ABI hands out the device-side fall-back instead -- DESIGN.md section 7)."""
import json
import os

import numpy as np
import pytest

from oracle_binding import OracleMpc, default_cfg

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(HERE, "mpc_fixtures.json")))
DATA = np.load(os.path.join(HERE, "mpc_fixtures.npz"))
CASES = {c["name"]: c for c in MAN["cases"]}


def rel(a, ref):
    a, ref = np.asarray(a, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300))


@pytest.mark.parametrize("name", sorted(CASES))
def test_receding_horizon_sequences_of_the_reference_runiLQR_MPC_GPU(name):
    case = CASES[name]
    c = case["cfg"]
    o = OracleMpc(default_cfg(c["plant"], cores=8, spawn_threads=0, N=c["N"], M=c["M"], A=c["A"], integrator=c["integrator"], total_time=c["total_time"], tol_cost=c["tol_cost"],
                              max_iter=c["max_iter"], wafr_urdf=c["wafr_urdf"], mpc_mode=1), np.float64)
    o.set_traj(DATA[name + ".in.x0"], DATA[name + ".in.u0"])
    xg = DATA[name + ".in.xg"]
    successes = failures = 0
    for cyc, (shift, clear, mi) in enumerate(case["plan"]):
        ref = {k: DATA["%s.c%d.%s" % (name, cyc, k)] for k in ("xActual", "J", "alpha", "x", "u", "KT", "x_dev", "u_dev", "KT_dev", "d_dev", "last_successful_solve")}
        r = o.mpc_solve(ref["xActual"], xg, shift, clear_vars=clear, full_rollout=c["full_rollout"], max_iter=mi)
        it = r["iters"]
        assert it + 1 == len(ref["alpha"]), (name, cyc, it, ref["alpha"])
        assert list(r["alphaOut"][: it + 1]) == list(ref["alpha"]), (name, cyc)
        assert rel(r["Jout"][: it + 1], ref["J"]) <= 1e-10, (name, cyc)
        assert r["success"] == int(ref["last_successful_solve"] == 1), (name, cyc, r["success"], ref["last_successful_solve"])
        for k in ("x", "u", "KT", "d"):
            assert rel(r[k], ref[k + "_dev"]) <= 1e-10, (name, cyc, k)
        if r["success"]:                                          # a successful cycle also hands the device trajectory out to the host copy
            successes += 1
            for k in ("x", "u", "KT"):
                assert rel(r[k], ref[k]) <= 1e-10, (name, cyc, k, "host copy")
        else:
            failures += 1
    assert successes >= 2, "a sequence must contain warm-started successful cycles"
    if name.endswith("N16_M2_A4_full"):
        assert failures >= 1, "this sequence pins the fall-back of a cycle without an accepted step of index > 0"


def test_fixture_is_data_only():
    assert set(MAN) == {"_provenance", "cases"}
    for c in MAN["cases"]:
        assert set(c) == {"name", "cfg", "plan", "cycles"}
    assert all(DATA[k].dtype.kind in "fi" for k in DATA.files)
