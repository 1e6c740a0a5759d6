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


def _ee_setup(o, x, u, goal):
    """nextIterationSetupGPU with EE_COST 1 from the oracle's per-knot functions: [A B] of the first N - 1 knots, H_k / g_k of every knot (costGradientHessianKern's
    end-effector branch, nisInitHelpers.cuh:52-84) and the per-knot costs it leaves in d_JT"""
    c, dt = o.c, o.dtype
    n, m, N, nm = o.n, o.m, c.N, o.n + o.m
    AB, H, g, ck = np.zeros(N * n * nm, dt), np.zeros(N * nm * nm, dt), np.zeros(N * nm, dt), np.zeros(N, dt)
    X, U = x.reshape(N, n), u.reshape(N, m)
    for k in range(N):
        if k < N - 1:
            AB[k * n * nm:(k + 1) * n * nm] = o.integrator_gradient(X[k], U[k])
        Hk, gk = o.ee_cost_grad(X[k], U[k], goal, k)
        H[k * nm * nm:(k + 1) * nm * nm] = Hk.ravel(); g[k * nm:(k + 1) * nm] = gk
        ck[k] = o.ee_cost(X[k], U[k], goal, k)
    return AB, H, g, ck


def _tree_sum(v, dt):
    """costKern<T,1> (fpHelpers.cuh:179-190) with blockDim.x = N: the pairwise tree of reduceSum over the knots"""
    s = np.asarray(v, dt).copy()
    h = len(s) // 2
    while h >= 1:
        s[:h] = s[:h] + s[h:2 * h]
        h //= 2
    return s[0]


def gpusem_iterations(o, x0, u0, xg, max_iter, ignore_first_defect=1):
    if o.c.ee_cost:
        yield from _gpusem_iterations_ee(o, x0, u0, xg, max_iter, ignore_first_defect)
        return
    yield from _gpusem_iterations_joint(o, x0, u0, xg, max_iter, ignore_first_defect)


def _gpusem_iterations_ee(o, x0, u0, xg, max_iter, ignore_first_defect=1):
    """The same loop with the end-effector cost family (EE_COST 1): the cost of a candidate is accumulated INSIDE its rollout, per joint and per shooting segment
    (forwardSimKern fpHelpers.cuh:259-265,298-300; costKern<T,0> adds the M segment sums in order), the initial cost is the tree sum of the per-knot costs the setup leaves
    (costKern<T,1>), and H_k moves with the trajectory.  xg[0:6] = the tool-point goal."""
    c, dt = o.c, o.dtype
    n, m, N, M, A = o.n, o.m, c.N, c.M, c.A
    f = dt.type
    z = lambda *s: np.zeros(s, dt)
    x, u, d = o.arr(x0).copy(), o.arr(u0).copy(), z(N * n)
    goal = o.arr(xg)[:6].copy()
    P, p, Pp, pp = z(N * n * n), z(N * n), z(N * n * n), z(N * n)
    xp2 = x.copy()
    AB, H, g, ck = _ee_setup(o, x, u, goal)
    alphas = np.asarray([c.alpha_base ** i for i in range(A)], dt)
    prevJ = f(f(_tree_sum(ck, dt)) + f(2 * c.tol_cost))
    rho, drho, it, ign, alphaIndex = f(c.rho_init), f(1.0), 1, int(ignore_first_defect), 0
    Jout, alphaOut = [f(prevJ - f(2 * c.tol_cost))], [-1]
    while True:
        rec = Iteration(iter=it, x=x.copy(), u=u.copy(), d=d.copy(), xp2=xp2.copy(), AB=AB.copy(), H=H.copy(), g=g.copy(), Pp=Pp.copy(), pp=pp.copy(),
                        rho=float(rho), drho=float(drho), prevJ=float(prevJ), ignore_defect=ign, alphas=alphas, costk=ck.copy())
        KT, du, ApBK, Bdu = z(N * n * m), z(N * m), z(N * n * n), z(N * n)
        fail, dJexp, err = o.backward_pass(1, AB, P, p, Pp.copy(), pp.copy(), H.copy(), g.copy(), KT, du, d, ApBK, Bdu, x, xp2, rho)
        assert not fail                                              # the arm's generic inversion never reports failure
        rec.update(KT=KT, du=du, ApBK=ApBK, Bdu=Bdu, P=P.copy(), p=p.copy(), dJexp=dJexp.copy(), err=err.copy())
        xs, us, ds, J, dmax = [], [], [], [], []
        for a in range(A):
            xa, ua, da = x.copy(), u.copy(), d.copy()
            with np.errstate(all="ignore"):
                if M > 1:
                    o.forward_sweep(xa, ApBK, Bdu, d, x, alphas[a])
                JT = o.forward_sim_ee(xa, ua, KT, du, da, alphas[a], x, goal)
                Js = f(0)
                for b in range(M):
                    Js = f(Js + JT[b])
            xs.append(xa); us.append(ua); ds.append(da); J.append(Js)
            with np.errstate(all="ignore"):
                dmax.append(f(o.max_defect(1, da)) if M > 1 else f(0))
        xp2 = x.copy()
        dsum = dJexp.copy()
        for i in range(1, M):
            dsum[0] += dJexp[2 * i]; dsum[1] += dJexp[2 * i + 1]
        Jt = np.asarray([j if np.isfinite(j) else f(np.finfo(dt).max) for j in J], dt)
        dm_ = np.asarray([v if np.isfinite(v) else f(np.finfo(dt).max) for v in dmax], dt)
        ai, ign_new, dJ, zz = o.line_search_gpu(Jt, dm_, dsum, prevJ, ign, alphaIndex)
        rec.update(xs=xs, us=us, ds=ds, J=Jt, dmax=dm_, dJexp_sum=dsum, ls_alpha=ai, ls_ignore_defect=ign_new, dJ=float(dJ), z=float(zz))
        ign = ign_new
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
            AB, H, g, ck = _ee_setup(o, x, u, goal)
        Pp, pp = P.copy(), p.copy()
        yield rec
        if done or it == max_iter:
            break
        it += 1
    gpusem_iterations.last = dict(Jout=np.asarray(Jout, dt), alphaOut=np.asarray(alphaOut, np.int32), iters=it, x=x, u=u, KT=KT)


def _gpusem_iterations_joint(o, x0, u0, xg, max_iter, ignore_first_defect=1):
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
