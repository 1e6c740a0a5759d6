"""The trajectory a production sweep adopts against the candidates the teacher-forcing hook keeps (thread-lane kernels of the KUKA arm, float32).

k_fp_tl (the sweep) leaves every candidate's (state | control | pad) record of every knot in the knot-major array xw -- staged through the wave's LDS area and written as
16-byte pieces -- and k_nis_tl adopts the accepted candidate's records: states AND controls are read, not recomputed.  The hook's rollouts (pddp_run_phase(FP), the
`ALL` instantiation of the same kernel) store the candidates in the reference's arrays xs / us (forwardSimKern's outputs, fpHelpers.cuh:279-301).  Both run the same
arithmetic, so what the sweep adopted must be the hook's candidate of the accepted index BIT FOR BIT -- for candidate counts that take the staged store path with one
(8) or several (16) candidates' records per 16-byte piece row, and the unstaged path (4)."""
import os

import numpy as np
import pytest

import pyddp
from backends import make_solver
from oracle_binding import example_inputs

pytestmark = pytest.mark.gpu
ENV = dict(fp="tl", bp="mx", sweep="st")      # thread lanes at any batch size; the sweep kernel the hook also uses (M > 1)


def handle(B, **kw):
    return make_solver("hip", 4, dtype=0, batch=B, use_graph=0, **kw, kernels=dict(ENV))


@pytest.mark.parametrize("A,M", [(8, 1), (16, 1), (4, 1), (8, 4), (16, 4)])
def test_sweep_adopts_the_hooks_candidate_bit_for_bit(A, M):
    B, N, n, m = 24, 64, 14, 7
    kw = dict(N=N, M=M, A=A, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=10)
    rng = np.random.default_rng(11 * A + M)
    X, U = [], []
    for _ in range(B):
        x0, u0, xg = example_inputs(4, N, np.float32, noise=rng.normal(0, 0.01, (N, n)))
        X.append(x0); U.append(u0)
    X, U, G = np.concatenate(X), np.concatenate(U), np.tile(xg, B)

    prod = handle(B, **kw)
    prod.load(X, U, G); prod.iterate(1); prod.sync()
    sp = prod.get_state()
    xb_p, uc_p, gp = prod.get("xb").reshape(B, 2, N, n), prod.get("ucur").reshape(B, N, m), prod.get("g").reshape(B, N, n + m)
    names = [k for k, _ in prod.time_kernels(1)]
    assert "k_fp_tl" in names and "k_nis_tl" in names, names

    hook = handle(B, **kw)
    hook.load(X, U, G)
    for ph in (pyddp.PHASE_BP, pyddp.PHASE_FP, pyddp.PHASE_LS, pyddp.PHASE_NIS):
        hook.run_phase(ph)
    sh = hook.get_state()
    xs, us = hook.get("xs").reshape(B, A, N, n), hook.get("us").reshape(B, A, N, m)
    xb_h, uc_h = hook.get("xb").reshape(B, 2, N, n), hook.get("ucur").reshape(B, N, m)

    accepted = 0
    for b in range(B):
        assert (sp[b].accepted, sp[b].alphaIndex, sp[b].cur) == (sh[b].accepted, sh[b].alphaIndex, sh[b].cur), b
        if sp[b].accepted != 1:
            continue
        accepted += 1
        w, cur = sp[b].alphaIndex, sp[b].cur
        assert np.array_equal(xb_p[b, cur], xs[b, w]), ("states", b)
        assert np.array_equal(uc_p[b, : N - 1], us[b, w, : N - 1]), ("controls", b)
        assert np.array_equal(uc_p[b, N - 1], U.reshape(B, N, m)[b, N - 1]), ("the terminal knot keeps its nominal control", b)
        assert np.array_equal(xb_h[b, cur], xs[b, w]) and np.array_equal(uc_h[b, : N - 1], us[b, w, : N - 1]), ("hook", b)
    assert accepted >= B // 2, accepted
    gh = hook.get("g").reshape(B, N, n + m)                        # ... and the cost gradient taken there
    for b in range(B):
        if sp[b].accepted == 1:
            assert np.array_equal(gp[b], gh[b]), ("g", b, np.abs(gp[b] - gh[b]).max(), np.argwhere(gp[b] != gh[b])[:4])
    prod.close(); hook.close()
