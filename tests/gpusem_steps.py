"""TEST INFRASTRUCTURE -- the oracle's GPU-semantics DDP loop (oracle/ora_core.inc gs_init_and_loop, a restatement of runiLQR_GPU,
DDPHelpers/DDPWrappers.cuh:10-138) stepped from Python so that the state at the top of EVERY iteration and every phase's inputs and
outputs are visible to the teacher-forced tests.  All arithmetic is done by the oracle's own C phase functions; only the control flow of
SURVEY.md Appendix D lives here, and test_fp32_bar.py checks that the stepped loop reproduces ora_run_ilqr_gpusem bit for bit.
"""
import numpy as np

RHO_FACTOR, RHO_MAX, RHO_MIN = 1.25, 1e7, 0.01          # config.cuh:102-104


def rho_increase(rho, drho, dtype):                      # bpHelpers.cuh:500-501 / nisInitHelpers.cuh:494
    f = dtype.type
    drho = max(f(drho * f(RHO_FACTOR)), f(RHO_FACTOR))
    return min(f(rho * drho), f(RHO_MAX)), drho


def rho_decrease(rho, drho, dtype):                      # nisInitHelpers.cuh:508
    f = dtype.type
    drho = min(f(drho / f(RHO_FACTOR)), f(1.0 / RHO_FACTOR))
    return max(f(rho * drho), f(RHO_MIN)), drho


class Iteration(dict):
    """Inputs and outputs of the four phases of one DDP iteration (flat arrays, the oracle's layouts)."""
    __getattr__ = dict.__getitem__


def gpusem_iterations(o, x0, u0, xg, max_iter, ignore_first_defect=1):
    """Generator over the iterations of one solve with GPU semantics on oracle `o` (float64 or float32).  Yields an Iteration holding
    bp_in / bp_out / fp_out (per alpha) / ls / nis_out, then advances the state exactly as gs_init_and_loop does."""
    c, dt = o.c, o.dtype
    n, m, N, M, A = o.n, o.m, c.N, c.M, c.A
    f = dt.type
    z = lambda *s: np.zeros(s, dt)
    x, u, d = o.arr(x0).copy(), o.arr(u0).copy(), z(N * n)
    xg = o.arr(xg)
    P, p, Pp, pp = z(N * n * n), z(N * n), z(N * n * n), z(N * n)
    xp2 = x.copy()
    AB, H, g = o.next_iteration_setup(x, u, xg)
    alphas = np.asarray([c.alpha_base ** i for i in range(A)], dt)
    prevJ = f(f(o.total_cost(1, x, u, xg)) + f(2 * c.tol_cost))
    rho, drho, it, ign, alphaIndex = f(c.rho_init), f(1.0), 1, int(ignore_first_defect), 0
    Jout, alphaOut = [f(prevJ - f(2 * c.tol_cost))], [-1]
    while True:
        rec = Iteration(iter=it, x=x.copy(), u=u.copy(), d=d.copy(), xp2=xp2.copy(), AB=AB.copy(), H=H.copy(), g=g.copy(), Pp=Pp.copy(), pp=pp.copy(),
                        rho=float(rho), drho=float(drho), prevJ=float(prevJ), ignore_defect=ign, alphas=alphas)
        # ---- backward pass (the arm's generic inversion never fails; the closed-form plants may: retry like backwardPassGPU)
        while True:
            KT, du, ApBK, Bdu = z(N * n * m), z(N * m), z(N * n * n), z(N * n)
            fail, dJexp, err = o.backward_pass(1, AB, P, p, Pp.copy(), pp.copy(), H.copy(), g.copy(), KT, du, d, ApBK, Bdu, x, xp2, rho)
            if not fail:
                break
            rho, drho = rho_increase(rho, drho, dt)
            P[:], p[:] = Pp, pp
            rec["rho"] = float(rho)
        rec.update(KT=KT, du=du, ApBK=ApBK, Bdu=Bdu, P=P.copy(), p=p.copy(), dJexp=dJexp.copy(), err=err.copy())
        # ---- forward pass for every alpha from the same base
        xs, us, ds, J, dmax = [], [], [], [], []
        for a in range(A):
            xa, ua, da = x.copy(), u.copy(), d.copy()
            if M > 1:
                o.forward_sweep(xa, ApBK, Bdu, d, x, alphas[a])
            o.forward_sim(xa, ua, KT, du, da, alphas[a], x)
            xs.append(xa); us.append(ua); ds.append(da)
            with np.errstate(all="ignore"):
                J.append(f(o.total_cost(1, xa, ua, xg))); dmax.append(f(o.max_defect(1, da)) if M > 1 else f(0))
        xp2 = x.copy()
        dsum = dJexp.copy()
        for i in range(1, M):
            dsum[0] += dJexp[2 * i]; dsum[1] += dJexp[2 * i + 1]
        Jt = np.asarray([j if np.isfinite(j) else f(np.finfo(dt).max) for j in J], dt)
        dm_ = np.asarray([v if np.isfinite(v) else f(np.finfo(dt).max) for v in dmax], dt)
        ai, ign_new, dJ, zz = o.line_search_gpu(Jt, dm_, dsum, prevJ, ign, alphaIndex)
        rec.update(xs=xs, us=us, ds=ds, J=Jt, dmax=dm_, dJexp_sum=dsum, ls_alpha=ai, ls_ignore_defect=ign_new, dJ=float(dJ), z=float(zz))
        ign = ign_new
        # ---- accept / reject (acceptRejectTrajGPU)
        done = False
        if dJ < 0:
            rho, drho = rho_increase(rho, drho, dt)
            alphaIndex = 0; alphaOut.append(-1); Jout.append(prevJ)
            rec["accepted"] = 0
            done = (rho == f(RHO_MAX) and not c.ignore_max_rho_exit)
        else:
            rho, drho = rho_decrease(rho, drho, dt)
            alphaIndex = ai
            x, u, d = xs[ai].copy(), us[ai].copy(), ds[ai].copy()
            rel = f(f(dJ) / prevJ); prevJ = Jt[ai]; alphaOut.append(ai); Jout.append(Jt[ai])
            rec["accepted"] = 1
            done = bool(rel < f(c.tol_cost))
        rec.update(rho_next=float(rho), drho_next=float(drho))
        if not done and it != max_iter:
            AB, H, g = o.next_iteration_setup(x, u, xg)
            rec.update(AB_next=AB.copy(), g_next=g.copy(), x_next=x.copy(), u_next=u.copy())
        Pp, pp = P.copy(), p.copy()
        yield rec
        if done or it == max_iter:
            break
        it += 1
    gpusem_iterations.last = dict(Jout=np.asarray(Jout, dt), alphaOut=np.asarray(alphaOut, np.int32), iters=it, x=x, u=u, KT=KT)
