"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (parallel-ddp_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class OraCfg(C.Structure):
    _fields_ = [
        ("plant", C.c_int), ("N", C.c_int), ("M", C.c_int), ("A", C.c_int), ("integrator", C.c_int),
        ("wafr_urdf", C.c_int), ("mpc_mode", C.c_int), ("max_iter", C.c_int), ("ignore_max_rho_exit", C.c_int),
        ("cores", C.c_int), ("spawn_threads", C.c_int), ("survey_int_minmax", C.c_int), ("survey_double_trig", C.c_int),
        ("total_time", C.c_double), ("alpha_base", C.c_double), ("rho_init", C.c_double), ("max_defect", C.c_double),
        ("tol_cost", C.c_double), ("exp_red_min", C.c_double), ("exp_red_max", C.c_double),
        ("Q1", C.c_double), ("Q2", C.c_double), ("R", C.c_double), ("QF1", C.c_double), ("QF2", C.c_double),
        ("ee_cost", C.c_int), ("ee_cost_shift", C.c_int),
        ("Q_EE1", C.c_double), ("Q_EE2", C.c_double), ("QF_EE1", C.c_double), ("QF_EE2", C.c_double), ("R_EE", C.c_double),
        ("Q_xEE", C.c_double), ("QF_xEE", C.c_double), ("Q_xdEE", C.c_double), ("QF_xdEE", C.c_double), ("ee_on_link_z", C.c_double),
        ("xTarget", C.c_double * 14), ("ee_type", C.c_int), ("use_finite_diff", C.c_int), ("finite_diff_epsilon", C.c_double), ("use_smooth_abs", C.c_int), ("smooth_abs_alpha", C.c_double), ("use_limits", C.c_int),
    ]


class OraResult(C.Structure):
    _fields_ = [("iters", C.c_int), ("t_total_ms", C.c_double), ("t_init_ms", C.c_double)]


def build(force=False, variant="strict"):
    # PDDP_ORACLE_SAN=1: the strict oracle from its AddressSanitizer / UBSan build (make -C oracle SAN=1 -> liboracle_san.so; the suite then runs with libasan preloaded:
    # tools/sanitizers.sh, profiles/r06_sanitizers.log)
    san = variant == "strict" and os.environ.get("PDDP_ORACLE_SAN") == "1"
    so = os.path.join(ORACLE_DIR, "liboracle_san.so" if san else "liboracle.so" if variant == "strict" else "liboracle_fma.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "all"] + (["SAN=1"] if san else []))
    return so


_LIBS = {}


def lib(variant="strict"):
    """variant "strict": IEEE operation by operation (the default everywhere); "fma": the same restatement with contracted multiply-adds (only the
    float32 noise-floor ensemble of test_fp32_bar.py uses it)."""
    global _LIB
    if variant != "strict":
        if variant not in _LIBS:
            L = C.CDLL(build(variant=variant))
            L.ora_default_cfg.argtypes = [C.POINTER(OraCfg), C.c_int]
            L.ora_total_cost_f32.restype = C.c_float
            L.ora_total_cost_f64.restype = C.c_double
            L.ora_max_defect_f32.restype = C.c_float
            L.ora_max_defect_f64.restype = C.c_double
            _LIBS[variant] = L
        return _LIBS[variant]
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ora_default_cfg.argtypes = [C.POINTER(OraCfg), C.c_int]
        _LIB.ora_cost_func_f32.restype = C.c_float
        _LIB.ora_cost_func_f64.restype = C.c_double
        _LIB.ora_total_cost_f32.restype = C.c_float
        _LIB.ora_total_cost_f64.restype = C.c_double
        _LIB.ora_max_defect_f32.restype = C.c_float
        _LIB.ora_max_defect_f64.restype = C.c_double
        _LIB.ora_ee_cost_f32.restype = C.c_float
        _LIB.ora_ee_cost_f64.restype = C.c_double
        _LIB.ora_simulate_f32.restype = C.c_float
        _LIB.ora_simulate_f64.restype = C.c_double
    return _LIB


PLANT_DIMS = {1: (1, 2, 1), 2: (2, 4, 1), 3: (6, 12, 4), 4: (7, 14, 7)}  # npos, n, m


_PLUGIN_SHIMS = []


def register_plugin(shim_path):
    """Plant 5 of the oracle = a user's plant + cost files compiled for the host by tests/plugin/plugin_shim.cpp (the plug-in is an INPUT of the solver the oracle restates).
    Registers the shim's callbacks with both precisions of every loaded variant and teaches PLANT_DIMS the plug-in's sizes."""
    shim = C.CDLL(shim_path)
    _PLUGIN_SHIMS.append(shim)                      # the callbacks live in the shim: keep it loaded
    shim.plugin_f32.restype = C.c_void_p; shim.plugin_f64.restype = C.c_void_p
    p32, p64 = shim.plugin_f32(), shim.plugin_f64()
    for variant in ("strict", "fma"):
        L = lib(variant)
        L.ora_set_plugin_f32.argtypes = [C.c_void_p]; L.ora_set_plugin_f64.argtypes = [C.c_void_p]
        L.ora_set_plugin_f32(p32); L.ora_set_plugin_f64(p64)
    npos, m = C.cast(p64, C.POINTER(C.c_int))[0], C.cast(p64, C.POINTER(C.c_int))[1]
    PLANT_DIMS[5] = (npos, 2 * npos, m)
    return PLANT_DIMS[5]


def default_cfg(plant, **kw):
    c = OraCfg()
    lib().ora_default_cfg(C.byref(c), plant)
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def _suf(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _r(dtype, v):
    return C.c_float(v) if np.dtype(dtype) == np.float32 else C.c_double(v)


class Oracle:
    """Thin, explicit wrapper: every array is a contiguous numpy array of `dtype`."""

    def __init__(self, cfg, dtype=np.float32, variant="strict"):
        self.variant = variant
        self.c = cfg
        self.dtype = np.dtype(dtype)
        self.suf = _suf(dtype)
        self.npos, self.n, self.m = PLANT_DIMS[cfg.plant]

    def _f(self, name):
        return getattr(lib(self.variant), f"ora_{name}_{self.suf}")

    def arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    # ---- plant level
    def dynamics(self, x, u):
        x, u = self.arr(x), self.arr(u)
        qdd = np.zeros(self.npos, self.dtype)
        self._f("dynamics")(C.byref(self.c), _p(qdd), _p(x), _p(u))
        return qdd

    def dynamics_gradient(self, x, u):
        x, u = self.arr(x), self.arr(u)
        qdd = np.zeros(self.npos, self.dtype)
        dqdd = np.zeros(self.npos * (self.n + self.m), self.dtype)
        self._f("dynamics_gradient")(C.byref(self.c), _p(dqdd), _p(qdd), _p(x), _p(u))
        return dqdd, qdd

    def integrator(self, x, u):
        x, u = self.arr(x), self.arr(u)
        xn = np.zeros(self.n, self.dtype)
        self._f("integrator")(C.byref(self.c), _p(xn), _p(x), _p(u))
        return xn

    def integrator_gradient(self, x, u):
        x, u = self.arr(x), self.arr(u)
        AB = np.zeros(self.n * (self.n + self.m), self.dtype)
        self._f("integrator_gradient")(C.byref(self.c), _p(AB), _p(x), _p(u))
        return AB

    def cost_func(self, x, u, xg, k):
        x, u, xg = self.arr(x), self.arr(u), self.arr(xg)
        return float(self._f("cost_func")(C.byref(self.c), _p(x), _p(u), _p(xg), int(k)))

    def cost_grad(self, x, u, xg, k):
        x, u, xg = self.arr(x), self.arr(u), self.arr(xg)
        nm = self.n + self.m
        H = np.zeros(nm * nm, self.dtype)
        g = np.zeros(nm, self.dtype)
        self._f("cost_grad")(C.byref(self.c), _p(H), _p(g), _p(x), _p(u), _p(xg), int(k))
        return H, g

    # ---- end-effector cost family (arm)
    def ee_pos(self, x, jac=True):
        x = self.arr(x)
        pos, dpos = np.zeros(6, self.dtype), np.zeros(42, self.dtype)
        self._f("ee_pos")(C.byref(self.c), _p(x), _p(pos), _p(dpos) if jac else None)
        return pos, dpos.reshape(7, 6)

    def ee_cost(self, x, u, goal, k, tshift=0):
        x, u, goal = self.arr(x), self.arr(u), self.arr(goal)
        return float(self._f("ee_cost")(C.byref(self.c), _p(x), _p(u), _p(goal), int(k), int(tshift)))

    def ee_cost_grad(self, x, u, goal, k, tshift=0):
        x, u, goal = self.arr(x), self.arr(u), self.arr(goal)
        H, g = np.zeros(21 * 21, self.dtype), np.zeros(21, self.dtype)
        self._f("ee_cost_grad")(C.byref(self.c), _p(H), _p(g), _p(x), _p(u), _p(goal), int(k), int(tshift))
        return H.reshape(21, 21), g

    def forward_sim_ee(self, x, u, KT, du, d, alpha, xp, goal, tshift=0):
        """forwardSimKern with the end-effector cost for one candidate (in place on x, u, d); returns the per-segment in-sim costs"""
        JT = np.zeros(self.c.M, self.dtype)
        goal = self.arr(goal)
        self._f("forward_sim_ee")(C.byref(self.c), _p(x), _p(u), _p(KT), _p(du), _p(d), _r(self.dtype, alpha), _p(xp), _p(goal), int(tshift), _p(JT))
        return JT

    def simulate(self, x, u, KT, t0_us, elapsed_us, substeps=150, goal_xyz=None, xActual=None):
        x, u, KT, xa = self.arr(x), self.arr(u), self.arr(KT), self.arr(xActual).copy()
        g = None if goal_xyz is None else self.arr(goal_xyz)
        failed = C.c_int(0)
        f = self._f("simulate")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        err = f(C.byref(self.c), _p(x), _p(u), _p(KT), float(t0_us), float(elapsed_us), int(substeps), None if g is None else _p(g), _p(xa), C.byref(failed))
        return xa, float(err), failed.value

    # ---- phase level (arrays are modified in place, like the reference)
    def backward_pass(self, sem_gpu, AB, P, p, Pp, pp, H, g, KT, du, d, ApBK, Bdu, x, xp2, rho):
        M = self.c.M
        dJexp = np.zeros(2 * M, self.dtype)
        err = np.zeros(M, np.int32)
        fail = self._f("backward_pass")(C.byref(self.c), int(sem_gpu), _p(AB), _p(P), _p(p), _p(Pp), _p(pp), _p(H), _p(g),
                                        _p(KT), _p(du), _p(d), _p(ApBK), _p(Bdu), _p(x), _p(xp2), _r(self.dtype, rho),
                                        _p(dJexp), _p(err))
        return int(fail), dJexp, err

    def forward_sweep(self, x, ApBK, Bdu, d, xp, alpha):
        self._f("forward_sweep")(C.byref(self.c), _p(x), _p(ApBK), _p(Bdu), _p(d), _p(xp), _r(self.dtype, alpha))

    def forward_sim(self, x, u, KT, du, d, alpha, xp):
        self._f("forward_sim")(C.byref(self.c), _p(x), _p(u), _p(KT), _p(du), _p(d), _r(self.dtype, alpha), _p(xp))

    def total_cost(self, sem_gpu, x, u, xg):
        return float(self._f("total_cost")(C.byref(self.c), int(sem_gpu), _p(x), _p(u), _p(xg)))

    def max_defect(self, sem_gpu, d):
        return float(self._f("max_defect")(C.byref(self.c), int(sem_gpu), _p(d)))

    def next_iteration_setup(self, x, u, xg):
        n, m, N = self.n, self.m, self.c.N
        nm = n + m
        AB = np.zeros(n * nm * N, self.dtype)
        H = np.zeros(nm * nm * N, self.dtype)
        g = np.zeros(nm * N, self.dtype)
        self._f("next_iteration_setup")(C.byref(self.c), _p(x), _p(u), _p(xg), _p(AB), _p(H), _p(g))
        return AB, H, g

    def line_search_gpu(self, J, dmax, dJexp, prevJ, ignore_defect, alphaIndex):
        J, dmax, dJexp = self.arr(J), self.arr(dmax), self.arr(dJexp)
        ign, ai = C.c_int(ignore_defect), C.c_int(alphaIndex)
        dJ = (C.c_float if self.suf == "f32" else C.c_double)(0)
        z = (C.c_float if self.suf == "f32" else C.c_double)(0)
        self._f("line_search_gpu")(C.byref(self.c), _p(J), _p(dmax), _p(dJexp), _r(self.dtype, prevJ), C.byref(ign),
                                   C.byref(ai), C.byref(dJ), C.byref(z))
        return ai.value, ign.value, dJ.value, z.value

    # ---- solver level
    def _run(self, which, x0, u0, xGoal, rollout=0, ignore_first_defect=1):
        x0, u0, xGoal = self.arr(x0).copy(), self.arr(u0).copy(), self.arr(xGoal)
        Jout = np.zeros(self.c.max_iter + 2, self.dtype)
        aout = np.full(self.c.max_iter + 2, -99, np.int32)
        KT = np.zeros(self.n * self.m * self.c.N, self.dtype)
        res = OraResult()
        self._f(which)(C.byref(self.c), _p(x0), _p(u0), _p(xGoal), _p(Jout), _p(aout), int(rollout),
                       int(ignore_first_defect), _p(KT), C.byref(res))
        return dict(x=x0, u=u0, KT=KT, Jout=Jout, alphaOut=aout, iters=res.iters, t_total_ms=res.t_total_ms,
                    t_init_ms=res.t_init_ms)

    def run_ilqr_cpu(self, x0, u0, xGoal, **kw):
        return self._run("run_ilqr_cpu", x0, u0, xGoal, **kw)

    def run_ilqr_cpu2(self, x0, u0, xGoal, **kw):
        """runiLQR_CPU2: the CPU path with the parallel line search (chunks of max(cores / M, 1) step sizes)"""
        return self._run("run_ilqr_cpu2", x0, u0, xGoal, **kw)

    def run_ilqr_gpusem(self, x0, u0, xGoal, **kw):
        return self._run("run_ilqr_gpusem", x0, u0, xGoal, **kw)


class OracleMpc:
    """Persistent GPU-semantics solver state of the oracle for the MPC wrapper (ora_gs_*)."""

    def __init__(self, cfg, dtype=np.float64):
        self.c, self.dtype, self.suf = cfg, np.dtype(dtype), _suf(dtype)
        self.npos, self.n, self.m = PLANT_DIMS[cfg.plant]
        f = getattr(lib(), f"ora_gs_create_{self.suf}"); f.restype = C.c_void_p
        self.h = C.c_void_p(f(C.byref(cfg)))

    def _f(self, name):
        return getattr(lib(), f"ora_gs_{name}_{self.suf}")

    def set_traj(self, x, u):
        x, u = np.ascontiguousarray(x, self.dtype), np.ascontiguousarray(u, self.dtype)
        self._f("set_traj")(self.h, _p(x), _p(u))

    def get_traj(self):
        N = self.c.N
        x, u, KT, d = (np.zeros(N * self.n, self.dtype), np.zeros(N * self.m, self.dtype), np.zeros(N * self.n * self.m, self.dtype), np.zeros(N * self.n, self.dtype))
        self._f("get_traj")(self.h, _p(x), _p(u), _p(KT), _p(d))
        return x, u, KT, d

    def mpc_solve(self, xActual, xGoal, shift, clear_vars=0, full_rollout=1, ignore_first_defect=1, max_iter=None):
        xActual, xGoal = np.ascontiguousarray(xActual, self.dtype), np.ascontiguousarray(xGoal, self.dtype)
        mi = self.c.max_iter
        Jout, aout, succ = np.zeros(mi + 2, self.dtype), np.full(mi + 2, -99, np.int32), C.c_int(0)
        it = self._f("mpc_solve")(self.h, _p(xActual), _p(xGoal), int(shift), int(clear_vars), int(full_rollout), int(ignore_first_defect),
                                  int(max_iter if max_iter is not None else mi), _p(Jout), _p(aout), C.byref(succ))
        x, u, KT, d = self.get_traj()
        return dict(iters=it, Jout=Jout, alphaOut=aout, success=succ.value, x=x, u=u, KT=KT, d=d)

    def __del__(self):
        try:
            self._f("destroy")(self.h)
        except Exception:
            pass


# ---- the reference example's inputs (examples/WAFR_iLQR_examples.cu:69-121), noise supplied by the caller
def example_inputs(plant, N, dtype=np.float32, noise=None, wafr_urdf=1):
    npos, n, m = PLANT_DIMS[plant]
    x = np.zeros((N, n), np.float64)
    u = np.zeros((N, m), np.float64)
    if noise is None:
        noise = np.zeros((N, n))
    PI = 3.14159
    if plant == 1:
        x[:, 1] = noise[:, 1]; u[:] = 0.01; goal = [3.1416, 0.0]
    elif plant == 2:
        x[:, 2] = noise[:, 2]; x[:, 3] = noise[:, 3]; u[:] = 0.01; goal = [0.0, 3.1416, 0.0, 0.0]
    elif plant == 3:
        x[:, 2] = 0.5; x[:, 6:] = noise[:, 6:]; u[:] = 1.22625; goal = [7.0, 10.0, 0.5] + [0.0] * 9
    else:
        x[:, :7] = [-0.5 * PI, 0.25 * PI, 0.167 * PI, -0.167 * PI, 0.125 * PI, 0.167 * PI, 0.5 * PI]
        x[:, 7:] = noise[:, 7:]
        u[:] = ([0.0, -102.9832, 11.1968, 47.0724, 2.5993, -7.0290, -0.0907] if wafr_urdf else
                [-0.0000000001, -62.282937, 4.172921, 21.513797, -0.088674, -0.890626, 0.0000000001])
        goal = [0, 0, 0, -0.25 * PI, 0, 0.25 * PI, 0.5 * PI] + [0.0] * 7
    return x.astype(dtype).ravel(), u.astype(dtype).ravel(), np.asarray(goal, dtype)
